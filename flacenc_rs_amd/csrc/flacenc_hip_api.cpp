// flacenc_hip_api.cpp -- the C ABI declared in include/flacenc_hip.h.
//
// Owns the per-handle state the reference keeps in thread-locals (`reusable!`,
// src/lib.rs:92-116): the window cache (WINDOW_CACHE, src/lpc.rs:219-231) and
// the scratch buffers, here as device memory.  No allocation happens on the
// device-pointer path once the window for a block size is cached.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "comm.h"
#include "direct_mse.h"
#include "flacenc_hip.h"
#include "flacenc_hip_debug.h"
#include "frame_decide.h"
#include "frame_pack.h"
#include "qlpc_kernel.h"

namespace {

struct WindowEntry {
  uint32_t n;
  uint32_t type;
  uint32_t alpha_bits;
  float* dev;  // 32 pad + rows*16 floats
  int32_t flat_lo, flat_hi;
};

struct DeviceBuffer {
  void* ptr = nullptr;
  size_t cap = 0;
};

// Staging copies between ordinary (pageable) caller memory and the pinned slots of the streaming path: one
// memcpy stream moves ~20 GB/s on the host, well under what PCIe takes in both directions at once, so a
// copy is cut into slices for a few helper threads (the caller's thread takes one slice itself).
class CopyPool {
 public:
  explicit CopyPool(unsigned workers) : tasks_(workers) {
    try {
      threads_.reserve(workers);
      for (unsigned i = 0; i < workers; ++i) threads_.emplace_back([this, i] { run(i); });
    } catch (...) {  // a thread that did start must be joined before its std::thread is destroyed
      {
        std::lock_guard<std::mutex> g(m_);
        stop_ = true;
      }
      cv_.notify_all();
      for (std::thread& t : threads_) t.join();
      throw;
    }
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  unsigned workers() const { return static_cast<unsigned>(threads_.size()); }
  void copy(void* dst, const void* src, size_t n) {
    const size_t parts = threads_.size() + 1;
    if (threads_.empty() || n < (size_t(1) << 20)) {
      std::memcpy(dst, src, n);
      return;
    }
    const size_t slice = ((n + parts - 1) / parts + 4095) & ~size_t(4095);
    char* d = static_cast<char*>(dst);
    const char* sp = static_cast<const char*>(src);
    {
      std::lock_guard<std::mutex> g(m_);
      for (size_t i = 0; i < threads_.size(); ++i) {
        const size_t lo = (i + 1) * slice;
        const size_t len = lo >= n ? 0 : (n - lo < slice ? n - lo : slice);
        tasks_[i] = Task{d + lo, sp + lo, len};
      }
      pending_ = static_cast<unsigned>(threads_.size());
      ++generation_;
    }
    cv_.notify_all();
    std::memcpy(d, sp, slice < n ? slice : n);
    std::unique_lock<std::mutex> g(m_);
    done_cv_.wait(g, [this] { return pending_ == 0; });
  }

 private:
  struct Task {
    char* d = nullptr;
    const char* s = nullptr;
    size_t n = 0;
  };
  void run(unsigned idx) {
    unsigned long long seen = 0;
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
        t = tasks_[idx];
      }
      if (t.n) std::memcpy(t.d, t.s, t.n);
      {
        std::lock_guard<std::mutex> g(m_);
        --pending_;
      }
      done_cv_.notify_one();
    }
  }
  std::vector<std::thread> threads_;
  std::vector<Task> tasks_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  unsigned long long generation_ = 0;
  unsigned pending_ = 0;
  bool stop_ = false;
};

}  // namespace

struct flacenc_hip_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  std::vector<WindowEntry> windows;
  DeviceBuffer d_samples, d_residual, d_params, d_bps, d_autocorr, d_lpc, d_tables, d_keys, d_sel, d_results, d_out, d_outlen, d_cparams, d_cresid, d_fparams, d_fresid, d_fkeys, d_split, d_presid, d_sumabs, d_minmax, d_marked, d_irlsw, d_gram;
  // streaming host path (flacenc_hip_encode_pcm_stereo): copy-in / copy-out streams, two slots of pinned
  // staging and device buffers, the events that order them
  uint32_t marked_parity = 0;  // which of d_marked's two counters the current pipeline counts into
  hipStream_t s_in = nullptr, s_out = nullptr;
  hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_fill[2] = {nullptr, nullptr}, ev_pack[2] = {nullptr, nullptr},
             ev_d2h[2] = {nullptr, nullptr};
  DeviceBuffer d_pcm[2], d_pack[2], d_plen[2], d_poff[2], d_cont[2];
  void* pin_in[2] = {nullptr, nullptr};
  void* pin_out[2] = {nullptr, nullptr};
  void* pin_meta[2] = {nullptr, nullptr};
  size_t pin_in_cap = 0, pin_out_cap = 0, pin_meta_cap = 0;
  int host_threads = -1;  // staging-copy threads of the streaming path: -1 = default, see flacenc_hip_set_host_threads
  std::unique_ptr<CopyPool> copy_pool;
  unsigned long long* stamps = nullptr;  // profiling hook, see flacenc_hip_debug_set_stamps
  unsigned long long* fixed_keys = nullptr;  // test hook, see flacenc_hip_debug_set_fixed_keys
  uint32_t* cert_stats = nullptr;  // statistics hook, see flacenc_hip_debug_set_cert_stats
  flacenc_hip::CommState* comm = nullptr;  // RCCL communicator of the ordered gather (comm.cpp)
  // Order mode of the certified shapes by material (launch_adaptive): the certificate's own counters of the last
  // launches, cumulative on the device and mirrored into one pinned word by a one-thread kernel behind each such launch
  uint32_t* d_cert_fb = nullptr;               // device: the three counters of QlpcKernelArgs::cert_stats
  unsigned long long* h_cert_fb = nullptr;     // pinned, device-visible: the latest verdict (cert_feedback_kernel)
  uint32_t fb_seq = 0, fb_seen_seq = 0, fb_probe_seq = 0;  // sequence numbers of the launches that carried the counters
  uint32_t fb_pending = 0;  // subframes counted on the device since the last verdict went out
  bool fb_probe_out = false;
  int two_pass_left = 0, two_pass_span = 0;
  int adaptive_order = 1;                      // flacenc_hip_debug_set_adaptive_order(h, 0) pins the certified kernel
};

namespace flacenc_hip {
CommState*& handle_comm_slot(flacenc_hip_handle* h) { return h->comm; }
int handle_device(const flacenc_hip_handle* h) { return h->device; }
void handle_set_error(flacenc_hip_handle* h, const std::string& what) { h->last_error = what; }
}  // namespace flacenc_hip

namespace {

bool set_error(flacenc_hip_handle* h, const char* what, hipError_t err) {
  if (h) {
    char buf[512];
    std::snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(err));
    h->last_error = buf;
  }
  return false;
}

#define HIP_TRY(h, expr)                          \
  do {                                            \
    hipError_t err__ = (expr);                    \
    if (err__ != hipSuccess) {                    \
      set_error((h), #expr, err__);               \
      return FLACENC_HIP_ERR_DEVICE;              \
    }                                             \
  } while (0)

int ensure(flacenc_hip_handle* h, DeviceBuffer& b, size_t bytes) {
  if (bytes <= b.cap) return FLACENC_HIP_OK;
  if (b.ptr) HIP_TRY(h, hipFree(b.ptr));
  b.ptr = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 4 + 256;
  HIP_TRY(h, hipMalloc(&b.ptr, want));
  b.cap = want;
  return FLACENC_HIP_OK;
}

// lpc::window_weights, src/lpc.rs:96-120: f32 arithmetic in exactly this order,
// libm cosf (what f32::cos lowers to on Linux).  Built with -ffp-contract=off.
void window_weights(uint32_t type, float alpha, size_t len, float* out) {
  if (type == FLACENC_HIP_WINDOW_RECTANGLE || alpha == 0.0f) {
    for (size_t t = 0; t < len; ++t) out[t] = 1.0f;
    return;
  }
  const float pi = 3.14159265358979323846f;
  const float max_t = static_cast<float>(len) - 1.0f;
  const float alpha_len = alpha * max_t;
  for (size_t ti = 0; ti < len; ++ti) {
    const float t = static_cast<float>(ti);
    float w;
    if (t < alpha_len / 2.0f) {
      const float arg = 2.0f * pi * t / alpha_len;
      w = 0.5f * (1.0f - cosf(arg));
    } else if (t < max_t - alpha_len / 2.0f) {
      w = 1.0f;
    } else {
      const float arg = 2.0f * pi * (max_t - t) / alpha_len;
      w = 0.5f * (1.0f - cosf(arg));
    }
    out[ti] = w;
  }
}

// get_window, src/lpc.rs:222-231.  The reference keys its cache by
// (size, fingerprint) where the fingerprint quantises alpha to 16 bits
// (src/lpc.rs:123-132), so two alphas closer than 1/65535 share the first
// one's table; this cache keys by the exact alpha bits instead.
int get_window(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg, uint32_t n,
               const WindowEntry** out) {
  uint32_t alpha_bits;
  std::memcpy(&alpha_bits, &cfg->tukey_alpha, 4);
  uint32_t type = cfg->window_type;
  if (type == FLACENC_HIP_WINDOW_TUKEY && cfg->tukey_alpha == 0.0f) type = FLACENC_HIP_WINDOW_RECTANGLE;
  if (type == FLACENC_HIP_WINDOW_RECTANGLE) alpha_bits = 0;
  for (const WindowEntry& e : h->windows) {
    if (e.n == n && e.type == type && e.alpha_bits == alpha_bits) {
      *out = &e;
      return FLACENC_HIP_OK;
    }
  }
  WindowEntry e;
  e.n = n;
  e.type = type;
  e.alpha_bits = alpha_bits;
  e.dev = nullptr;
  e.flat_lo = -64;
  e.flat_hi = 0x7FFFFFFF;
  if (type != FLACENC_HIP_WINDOW_RECTANGLE) {
    const size_t rows = (n + 15) / 16;
    const size_t total = 32 + rows * 16 + 16;
    std::vector<float> host(total, 0.0f);
    window_weights(type, cfg->tukey_alpha, n, host.data() + 32);
    // longest run of exactly-1.0 weights: chunks inside it skip the table
    int best_lo = 0, best_hi = 0, run_lo = -1;
    for (int t = 0; t <= static_cast<int>(n); ++t) {
      const bool one = t < static_cast<int>(n) && host[32 + t] == 1.0f;
      if (one && run_lo < 0) run_lo = t;
      if (!one && run_lo >= 0) {
        if (t - run_lo > best_hi - best_lo) {
          best_lo = run_lo;
          best_hi = t;
        }
        run_lo = -1;
      }
    }
    e.flat_lo = best_lo;
    e.flat_hi = best_hi;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&e.dev), total * sizeof(float)));
    HIP_TRY(h, hipMemcpy(e.dev, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
  }
  h->windows.push_back(e);
  *out = &h->windows.back();
  return FLACENC_HIP_OK;
}

int check_batch_args(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg, const int32_t* samples,
                     size_t n_subframes, uint32_t block_size, size_t stride,
                     flacenc_hip_subframe_params* params, int32_t* residual, size_t residual_stride,
                     uint32_t min_block = FLACENC_HIP_MIN_BLOCK_SIZE) {
  if (!h || !cfg) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  int rc = flacenc_hip_verify_config(cfg);
  if (rc != FLACENC_HIP_OK) {
    h->last_error = "config::Qlpc / config::Prc verification failed";
    return rc;
  }
  // (the frame-level calls pass min_block = 1: a stream's last block may be shorter than
  // MIN_BLOCK_SIZE_FOR_PREDICTION; encode_subframe then skips both predictors, coding.rs:396)
  if (block_size < min_block || block_size > FLACENC_HIP_MAX_BLOCK_SIZE) {
    h->last_error = "block_size must be in 64..=32767 (1..=32767 for the frame-level calls)";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  if (n_subframes == 0) return FLACENC_HIP_OK;
  if (!samples || !params || !residual || stride < block_size || residual_stride < block_size ||
      n_subframes > 0x7FFFFFFFull) {
    h->last_error = "null pointer, stride < block_size, or too many subframes";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  return FLACENC_HIP_OK;
}

// QlpcKernelArgs::reference_order: 0 = the kernels' canonical sums, 1 = the stable build's orders
// (FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER), 2 = the simd-nightly build's (FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER)
uint32_t sum_order_mode(uint32_t flags) {
  if (flags & FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER) return 2u;
  return (flags & FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER) ? 1u : 0u;
}

// The unflagged order on the fused kernel's shapes is certified (QlpcKernelArgs::certify; launch_qlpc decides where it
// applies).  Launches that cannot run it inside the fused kernel -- unaligned rows, FLACENC_HIP_FLAG_GENERIC_KERNEL, the
// fused bit writer -- take the reference's R[] from acorr_reference_kernel through the split scratch.
void set_certify(flacenc_hip_handle* h, flacenc_hip::QlpcKernelArgs& a, uint32_t flags) {
  a.certify = (flags & FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER) ? 0u : 1u;
  // REFERENCE_SUM_ORDER | INTEGER_PARITY_ONLY: the certified shapes keep their own order (launch_qlpc)
  a.integer_parity_only = ((flags & FLACENC_HIP_FLAG_INTEGER_PARITY_ONLY) && (flags & FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER)) ? 1u : 0u;
  a.cert_stats = h->cert_stats;
}
bool certify_needs_scratch(const flacenc_hip::QlpcKernelArgs& a) {
  return a.certify != 0u && flacenc_hip::cert_shape(a) && (a.reference_order == 0u || a.integer_parity_only) &&
         !a.direct_mse && a.fixed_mode == 0 && (!flacenc_hip::wave_kernel_eligible(a) || a.pack_out != nullptr);
}

// FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER with the ApproxEnt selector: room for sumabs_reference_kernel's
// per-partition f32 sums (launch_qlpc runs it when `sumabs_scratch` is set)
int attach_sumabs_scratch(flacenc_hip_handle* h, flacenc_hip::QlpcKernelArgs& a, bool approx_ent) {
  if (!a.reference_order || !approx_ent) return FLACENC_HIP_OK;
  int rc = ensure(h, h->d_sumabs, static_cast<size_t>(a.n_subframes) * 5 * 64 * sizeof(float));
  if (rc != FLACENC_HIP_OK) return rc;
  a.sumabs_scratch = static_cast<float*>(h->d_sumabs.ptr);
  return FLACENC_HIP_OK;
}

// R[] and the predictor records between the launches of the split pipelines, and the counter through which
// bigblock_residual_kernel tells the clean-up launch whether it marked anything (QlpcKernelArgs::marked_count)
int attach_split_scratch(flacenc_hip_handle* h, flacenc_hip::QlpcKernelArgs& a) {
  int rc = ensure(h, h->d_split, static_cast<size_t>(a.n_subframes) * (33 * 8 + 36 * 4));
  if (rc != FLACENC_HIP_OK) return rc;
  a.split_scratch = h->d_split.ptr;
  // two counters (16 words reserved) + a list of the first kMarkedCap marks behind each
  constexpr uint32_t kMarkedCap = 1024;
  constexpr size_t kMarkedBytes = 64 + 2 * static_cast<size_t>(kMarkedCap) * 4;
  if (h->d_marked.ptr == nullptr) {
    if ((rc = ensure(h, h->d_marked, kMarkedBytes)) != FLACENC_HIP_OK) return rc;
    HIP_TRY(h, hipMemset(h->d_marked.ptr, 0, kMarkedBytes));
  }
  h->marked_parity ^= 1u;
  a.marked_count = static_cast<uint32_t*>(h->d_marked.ptr) + h->marked_parity;
  a.marked_next = static_cast<uint32_t*>(h->d_marked.ptr) + (h->marked_parity ^ 1u);
  a.marked_list = static_cast<uint32_t*>(h->d_marked.ptr) + 16 + h->marked_parity * kMarkedCap;
  a.marked_cap = kMarkedCap;
  a.marked_unit = 1;
  return FLACENC_HIP_OK;
}

// The certified shapes (blocks of 4096 / 4608 samples at orders up to 12) by material.  The fused kernel settles a
// subframe's order certificate in its first tier for next to nothing; the second tier and the recomputation from the
// reference's chains are serial work of one wave while its workgroup waits -- rare on noise-like material (2 subframes in
// 393 216 of the bench signal), the rule on music (the reference's real-audio fixtures: 26 % / 82 % / 100 % of the
// subframes at orders 8 / 10 / 12 count as unsettled, 220 / 146 / 100 G samples/s where the bench signal runs at 350).  Two passes -- the
// reference's chains on the matrix cores for every subframe, then the fused kernel on their R[] -- give the SAME integers
// at a flat 1.4 x the certified kernel's best time.  So launches that return integers only (no R[], no coefficients: their
// bits depend on the pass that produced them) watch the certificate's counters and take the two-pass form while the
// material they were last given was hard: above kHardShare of the subframes unsettled by the first tier, for a span of
// launches that doubles (8 .. 64) while the probes between the spans keep finding it so.
// One thread behind a launch that carried the counters: its verdict -- (sequence number, hard subframes, analysed subframes)
// in one 64-bit store to the pinned word -- and the counters cleared for the next one.
__global__ void cert_feedback_kernel(uint32_t* counters, unsigned long long* out, uint32_t seq) {
  const unsigned long long analysed = counters[0], hard = counters[1] + counters[2];
  counters[0] = counters[1] = counters[2] = 0u;
  *out = ((unsigned long long)(seq & 0xFFFu) << 52) | ((hard & 0x3FFFFFFull) << 26) | (analysed & 0x3FFFFFFull);
}
constexpr double kHardShare = 0.10;
// a verdict is taken from at least this many subframes (counters of smaller launches add up until they are), a probe's from
// at least kProbeMinSubframes; launches of 2^25 subframes and more are not watched (the verdict's fields are 26 bits wide)
constexpr uint32_t kFeedbackMinSubframes = 4096, kProbeMinSubframes = 1024, kFeedbackMaxSubframes = 1u << 25;
inline uint32_t next_seq(uint32_t seq) { return ((seq + 1u) & 0xFFFu) ? ((seq + 1u) & 0xFFFu) : 1u; }  // (0: the word's initial state)

int launch_adaptive(flacenc_hip_handle* h, flacenc_hip::QlpcKernelArgs& a, const flacenc_hip::QlpcLaunchPlan& plan,
                    hipStream_t stream) {
  const bool order_ok = a.reference_order == 0u || (a.reference_order == 1u && a.integer_parity_only != 0u);
  const bool fused_certified = a.certify != 0u && flacenc_hip::cert_shape(a) && order_ok && !a.direct_mse && a.fixed_mode == 0 &&
                               a.lpc_stage == 0 && a.acorr_in == nullptr && !a.only_marked && a.pack_out == nullptr &&
                               flacenc_hip::wave_kernel_eligible(a);
  const bool watch = fused_certified && h->adaptive_order != 0 && h->cert_stats == nullptr && a.autocorr == nullptr &&
                     a.lpc_coefs == nullptr && a.n_subframes < kFeedbackMaxSubframes;
  if (!watch) {
    HIP_TRY(h, flacenc_hip::launch_qlpc(a, plan, stream));
    return FLACENC_HIP_OK;
  }
  if (h->d_cert_fb == nullptr) {
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_cert_fb), 16));
    HIP_TRY(h, hipMemset(h->d_cert_fb, 0, 16));
    HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_cert_fb), 8, hipHostMallocMapped));
    *h->h_cert_fb = 0ull;
  }
  // The latest verdict that has landed (a plain read of the pinned word: late is fine, torn it cannot be).  The host may
  // run many launches ahead of the device, so nothing is concluded from launches whose counters are not in: on easy
  // material every launch carries the counters and any new verdict counts; on hard material a span of two-pass launches
  // is followed by ONE probe (a certified launch), and the launches behind the probe stay two-pass until ITS verdict is in.
  const unsigned long long word = *reinterpret_cast<volatile unsigned long long*>(h->h_cert_fb);
  const uint32_t l_seq = static_cast<uint32_t>(word >> 52), l_hard = static_cast<uint32_t>(word >> 26) & 0x3FFFFFFu,
                 l_an = static_cast<uint32_t>(word) & 0x3FFFFFFu;
  const bool is_hard = l_an != 0u && static_cast<double>(l_hard) > kHardShare * static_cast<double>(l_an);
  bool two_pass;
  if (h->two_pass_span == 0) {  // easy so far
    if (l_seq != h->fb_seen_seq) {
      h->fb_seen_seq = l_seq;
      if (is_hard) h->two_pass_left = h->two_pass_span = 8;
    }
    two_pass = h->two_pass_span != 0;
  } else if (h->two_pass_left > 0) {
    two_pass = true;
  } else if (!h->fb_probe_out) {
    two_pass = false;  // the probe: certified launches until kProbeMinSubframes have been counted
    h->fb_probe_out = true;
    h->fb_probe_seq = 0u;  // (assigned when its feedback goes out)
  } else if (h->fb_probe_seq == 0u) {
    two_pass = false;  // the probe is still collecting
  } else if (l_seq != h->fb_probe_seq) {
    two_pass = true;  // the probe's verdict is not in yet
  } else {
    h->fb_probe_out = false;
    h->fb_seen_seq = l_seq;
    if (is_hard) {
      h->two_pass_span = h->two_pass_span >= 64 ? 64 : 2 * h->two_pass_span;
      h->two_pass_left = h->two_pass_span;
      two_pass = true;
    } else {
      h->two_pass_span = 0;
      two_pass = false;
    }
  }
  if (two_pass) {
    if (h->two_pass_left > 0) --h->two_pass_left;
    flacenc_hip::QlpcKernelArgs b = a;
    b.certify = 0;
    b.reference_order = 1u;  // (the autocorrelation alone: the selector's sums follow sumabs_scratch, which the flags decide)
    if (b.split_scratch == nullptr) {
      int rc = attach_split_scratch(h, b);
      if (rc != FLACENC_HIP_OK) return rc;
    }
    HIP_TRY(h, flacenc_hip::launch_qlpc(b, plan, stream));
    return FLACENC_HIP_OK;
  }
  a.cert_stats = h->d_cert_fb;
  HIP_TRY(h, flacenc_hip::launch_qlpc(a, plan, stream));
  h->fb_pending += a.n_subframes;
  const bool probing = h->fb_probe_out && h->fb_probe_seq == 0u;
  if (h->fb_pending >= (probing ? kProbeMinSubframes : kFeedbackMinSubframes)) {
    h->fb_pending = 0;
    h->fb_seq = next_seq(h->fb_seq);
    if (probing) h->fb_probe_seq = h->fb_seq;
    unsigned long long* out = nullptr;
    HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&out), h->h_cert_fb, 0));
    hipLaunchKernelGGL(cert_feedback_kernel, dim3(1), dim3(1), 0, stream, h->d_cert_fb, out, h->fb_seq);
    HIP_TRY(h, hipGetLastError());
  }
  return FLACENC_HIP_OK;
}

int enqueue(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg, const int32_t* samples,
            size_t n_subframes, uint32_t block_size, size_t stride, const uint8_t* bps,
            flacenc_hip_subframe_params* params, int32_t* residual, size_t residual_stride,
            double* autocorr, double* lpc_coefs, hipStream_t stream, bool stereo = false,
            uint32_t bps_uniform = 16, int32_t* residual_lr = nullptr, size_t residual_lr_stride = 0,
            int32_t* minmax_out = nullptr, bool* placed = nullptr, uint32_t residual_mode = 0) {
  if (placed) *placed = false;
  const WindowEntry* win = nullptr;
  int rc = get_window(h, cfg, block_size, &win);
  if (rc != FLACENC_HIP_OK) return rc;
  flacenc_hip::QlpcLaunchPlan plan = flacenc_hip::plan_qlpc_launch(block_size, cfg->lpc_order);
  if (plan.smem_bytes > 160 * 1024) {
    h->last_error = "internal: LDS plan exceeds 160 KiB";
    return FLACENC_HIP_ERR_UNSUPPORTED;
  }
  flacenc_hip::QlpcKernelArgs a;
  a.samples = samples;
  a.stride = stride;
  a.block_size = block_size;
  a.n_subframes = static_cast<uint32_t>(n_subframes);
  a.bps = bps;
  a.bps_uniform = bps_uniform;
  a.stereo = stereo ? 1u : 0u;
  a.window = win->dev;
  a.flat_lo = win->flat_lo;
  a.flat_hi = win->flat_hi;
  a.lpc_order = cfg->lpc_order;
  a.precision = cfg->quant_precision;
  a.max_rice_parameter = cfg->max_rice_parameter;
  a.rice_finest_only = (cfg->flags & FLACENC_HIP_FLAG_FINEST_RICE_ORDER) ? 1u : 0u;
  a.force_generic = (cfg->flags & FLACENC_HIP_FLAG_GENERIC_KERNEL) ? 1u : 0u;
  a.reference_order = sum_order_mode(cfg->flags);
  set_certify(h, a, cfg->flags);
  a.acorr_in = nullptr;
  a.only_marked = 0;
  a.params = params;
  a.residual = residual;
  a.residual_stride = residual_stride;
  a.autocorr = autocorr;
  a.lpc_coefs = lpc_coefs;
  a.table_scratch = nullptr;
  a.stamps = h->stamps;
  a.frame_results = nullptr;
  a.chan_results = nullptr;
  a.use_constant = a.use_lpc = a.use_leftside = a.use_rightside = a.use_midside = 1;
  a.use_fixed = a.fixed_max_order = a.fixed_order_sel = a.fixed_group_log2 = 0;
  a.fixed_keys = nullptr;
  a.fixed_mode = a.fixed_partitions = a.forced_uniform = 0;
  a.forced_orders = nullptr;
  a.selector_keys = nullptr;
  a.lpc_stage = 0;
  a.pack_out = nullptr;
  a.pred = nullptr;
  a.pred_out = nullptr;
  a.split_scratch = nullptr;
  a.direct_mse = cfg->use_direct_mse ? 1u : 0u;
  a.mae_steps = cfg->use_direct_mse ? cfg->mae_optimization_steps : 0u;  // (ignored without it, coding.rs:337-347)
  if (a.direct_mse && flacenc_hip::direct_mse_lds_bytes(block_size, a.mae_steps > 0, cfg->lpc_order) > 160 * 1024) {
    h->last_error = "use_direct_mse: the block does not fit the LDS";  // (not for any block up to 32767 samples)
    return FLACENC_HIP_ERR_UNSUPPORTED;
  }
  if (a.direct_mse) {
    // (R[] and the matrix between the chains and the lane-per-subframe solve; with IRLS steps also the iteration's state)
    rc = ensure(h, h->d_gram, n_subframes * (flacenc_hip::direct_mse_gram_stride(cfg->lpc_order) +
                                             (a.mae_steps ? flacenc_hip::kIrlsStateDoubles : 0)) * sizeof(double));
    if (rc != FLACENC_HIP_OK) return rc;
    a.direct_mse_scratch = static_cast<double*>(h->d_gram.ptr);
  }
  if (a.direct_mse && a.mae_steps > 0) {
    // the IRLS weights between the steps' kernels (orders up to 11); lpc_with_irls_mae (lpc.rs:814-850) has no block
    // limit: above 16384 samples the one-kernel form keeps them here too
    rc = ensure(h, h->d_irlsw, n_subframes * ((static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3)) * sizeof(float));
    if (rc != FLACENC_HIP_OK) return rc;
    a.irls_weight_scratch = static_cast<float*>(h->d_irlsw.ptr);
  }
  // (R[] and the predictor records between the launches of the split pipelines: orders from 13, and blocks of
  // 8192 / 16384 at any order -- the big-block kernels)
  // (... and, round 6, every unflagged launch: the reference's chains go in front of whatever kernel takes the shape)
  if (cfg->lpc_order >= 13 || a.reference_order || a.direct_mse || block_size == 8192 || block_size == 16384 ||
      flacenc_hip::subwave_shape(block_size) || a.certify != 0u) {
    if ((rc = attach_split_scratch(h, a)) != FLACENC_HIP_OK) return rc;
  }
  if (plan.table_scratch_bytes_per_subframe) {
    rc = ensure(h, h->d_tables, plan.table_scratch_bytes_per_subframe * n_subframes);
    if (rc != FLACENC_HIP_OK) return rc;
    a.table_scratch = static_cast<uint32_t*>(h->d_tables.ptr);
  }
  // frame-level callers on the big-block shapes: L / R candidates straight into the output rows, role min / max
  // from the residual kernel (only bigblock_residual_kernel knows how; see QlpcKernelArgs::residual_lr)
  if (stereo && residual_lr != nullptr && minmax_out != nullptr &&
      (reinterpret_cast<uintptr_t>(residual_lr) & 15) == 0 && (residual_lr_stride & 3) == 0 &&
      (flacenc_hip::bigblock_eligible(a) || (a.direct_mse && flacenc_hip::bigblock_shape_eligible(a)))) {
    a.residual_lr = residual_lr;
    a.residual_lr_stride = residual_lr_stride;
    a.minmax_out = minmax_out;
    if (placed) *placed = true;
  }
  // ... or, with residual_mode 1, no rows at all: records and the roles' min / max only (the deciding store pass,
  // bigblock_residual_kernel's mode 2, produces the two rows the frame keeps)
  if (stereo && residual_mode == 1u && minmax_out != nullptr &&
      (flacenc_hip::bigblock_eligible(a) || (a.direct_mse && flacenc_hip::bigblock_shape_eligible(a)))) {
    a.residual_mode = 1u;
    a.minmax_out = minmax_out;
    if (placed) *placed = true;
  }
  return launch_adaptive(h, a, plan, stream);
}

// y = x^(8 per) mod P and its powers for the CRC-16 slice combination (see frame_pack.h)
void fill_crc_powers(uint32_t lds_words, uint32_t* crc_per, uint16_t crc_pow[32]) {
  auto mulmod = [](uint32_t x, uint32_t y) {
    uint32_t r = 0;
    for (int i = 15; i >= 0; --i) {
      r <<= 1;
      if (r & 0x10000u) r ^= 0x18005u;
      if ((y >> i) & 1u) r ^= x;
    }
    return r & 0xFFFFu;
  };
  *crc_per = (lds_words * 4 + 255) / 256;
  uint32_t y = 1;
  for (uint32_t i = 0; i < 8 * *crc_per; ++i) y = mulmod(y, 2);  // times x
  uint32_t acc = 1;
  for (int i = 0; i < 16; ++i) {
    crc_pow[i] = static_cast<uint16_t>(acc);
    acc = mulmod(acc, y);
  }
  const uint32_t y16 = acc;  // y^16
  acc = 1;
  for (int i = 0; i < 16; ++i) {
    crc_pow[16 + i] = static_cast<uint16_t>(acc);
    acc = mulmod(acc, y16);
  }
}

// FrameHeader's constant part for a launch (bitrepr.rs:373-419 as encode_frame_impl fills it,
// coding.rs:431-436): spec tags and the extra bytes that follow the frame number
void fill_header_specs(flacenc_hip::FramePackArgs& a, uint32_t block_size, uint32_t sample_rate,
                       uint32_t bits_per_sample) {
  // BlockSizeSpec::from_size / tag / extra bits, datatype.rs:1239-1294
  uint32_t bs_tag = 0, extra_len = 0;
  a.extra[0] = a.extra[1] = a.extra[2] = a.extra[3] = 0;
  if (block_size == 192) bs_tag = 1;
  for (uint32_t x = 0; x < 4 && !bs_tag; ++x)
    if (block_size == (576u << x)) bs_tag = 2 + x;
  for (uint32_t x = 0; x < 8 && !bs_tag; ++x)
    if (block_size == (256u << x)) bs_tag = 8 + x;
  if (!bs_tag) {
    if (block_size <= 256) {
      bs_tag = 6;
      a.extra[extra_len++] = static_cast<uint8_t>(block_size - 1);
    } else {
      bs_tag = 7;
      a.extra[extra_len++] = static_cast<uint8_t>((block_size - 1) >> 8);
      a.extra[extra_len++] = static_cast<uint8_t>(block_size - 1);
    }
  }
  // SampleRateSpec::from_freq / tag / extra bits, datatype.rs:1427-1453, 1503-1543 (Unspecified if
  // not representable, coding.rs:434-435)
  static const uint32_t known[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
  uint32_t sr_tag = 0;
  for (uint32_t t = 1; t < 12; ++t)
    if (sample_rate == known[t]) sr_tag = t;
  if (!sr_tag && sample_rate) {
    if (sample_rate % 1000 == 0 && sample_rate / 1000 <= 255) {
      sr_tag = 12;
      a.extra[extra_len++] = static_cast<uint8_t>(sample_rate / 1000);
    } else if (sample_rate % 10 == 0 && sample_rate / 10 <= 65535) {
      sr_tag = 14;
      a.extra[extra_len++] = static_cast<uint8_t>((sample_rate / 10) >> 8);
      a.extra[extra_len++] = static_cast<uint8_t>(sample_rate / 10);
    } else if (sample_rate <= 65535) {
      sr_tag = 13;
      a.extra[extra_len++] = static_cast<uint8_t>(sample_rate >> 8);
      a.extra[extra_len++] = static_cast<uint8_t>(sample_rate);
    }
  }
  // SampleSizeSpec::from_bits, datatype.rs:1350-1360
  uint32_t ss_tag = 0;
  switch (bits_per_sample) {
    case 8: ss_tag = 1; break;
    case 12: ss_tag = 2; break;
    case 16: ss_tag = 4; break;
    case 20: ss_tag = 5; break;
    case 24: ss_tag = 6; break;
    default: ss_tag = 0; break;
  }
  a.header_mid = (bs_tag << 12) | (sr_tag << 8) | (ss_tag << 1);
  a.extra_len = extra_len;
}

// config::Fixed::verify (config.rs:246-255) + OrderSel::verify (:419-431)
int verify_fixed(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg) {
  if (cfg->fixed_max_order > FLACENC_HIP_MAX_FIXED_LPC_ORDER ||
      (cfg->fixed_order_sel != FLACENC_HIP_ORDERSEL_BITCOUNT &&
       cfg->fixed_order_sel != FLACENC_HIP_ORDERSEL_APPROXENT) ||
      (cfg->fixed_order_sel == FLACENC_HIP_ORDERSEL_APPROXENT &&
       (cfg->fixed_partitions < 1 || cfg->fixed_partitions > 64))) {
    h->last_error = "fixed: max_order must be ..=4, order_sel BitCount / ApproxEnt, ApproxEnt.partitions 1..=64";
    return FLACENC_HIP_ERR_BAD_CONFIG;
  }
  return FLACENC_HIP_OK;
}

// OrderSel::BitCount (coding.rs:243-264): first minimum of the per-order keys
__global__ void bitcount_pick_kernel(const unsigned long long* keys, uint32_t n, uint32_t n_orders,
                                     uint8_t* orders, unsigned long long* best_keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long best = keys[i];
  uint32_t bk = 0;
  for (uint32_t k = 1; k < n_orders; ++k) {
    const unsigned long long v = keys[(size_t)k * n + i];
    if (v < best) {
      best = v;
      bk = k;
    }
  }
  orders[i] = (uint8_t)bk;
  if (best_keys) best_keys[i] = best;
}

// `fixed_lpc` (coding.rs:298-331) for a batch, device pointers, on `stream`
int enqueue_fixed(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg, const int32_t* samples,
                  size_t n_subframes, uint32_t block_size, size_t stride, const uint8_t* bps,
                  uint32_t bps_uniform, bool stereo, flacenc_hip_subframe_params* params, int32_t* residual,
                  size_t residual_stride, unsigned long long* selector_keys, hipStream_t stream,
                  uint32_t residual_mode = 0) {
  int rc = verify_fixed(h, cfg);
  if (rc != FLACENC_HIP_OK) return rc;
  flacenc_hip::QlpcLaunchPlan plan = flacenc_hip::plan_qlpc_launch(block_size, 4);
  if (plan.smem_bytes > 160 * 1024) {
    h->last_error = "internal: LDS plan exceeds 160 KiB";
    return FLACENC_HIP_ERR_UNSUPPORTED;
  }
  flacenc_hip::QlpcKernelArgs a;
  a.samples = samples;
  a.stride = stride;
  a.block_size = block_size;
  a.n_subframes = static_cast<uint32_t>(n_subframes);
  a.bps = bps;
  a.bps_uniform = bps_uniform;
  a.stereo = stereo ? 1u : 0u;
  a.window = nullptr;
  a.flat_lo = 0;
  a.flat_hi = 0;
  a.lpc_order = 4;
  a.precision = 0;
  a.max_rice_parameter = cfg->qlpc.max_rice_parameter;
  a.rice_finest_only = (cfg->qlpc.flags & FLACENC_HIP_FLAG_FINEST_RICE_ORDER) ? 1u : 0u;
  a.force_generic = (cfg->qlpc.flags & FLACENC_HIP_FLAG_GENERIC_KERNEL) ? 1u : 0u;
  a.reference_order = sum_order_mode(cfg->qlpc.flags);
  set_certify(h, a, cfg->qlpc.flags);
  a.direct_mse = cfg->qlpc.use_direct_mse ? 1u : 0u;  // (keeps the launch off the fused wave kernel)
  a.acorr_in = nullptr;
  a.only_marked = 0;
  a.params = params;
  a.residual = residual;
  a.residual_stride = residual_stride;
  a.autocorr = nullptr;
  a.lpc_coefs = nullptr;
  a.table_scratch = nullptr;
  a.stamps = nullptr;
  a.frame_results = nullptr;
  a.chan_results = nullptr;
  a.use_constant = a.use_lpc = a.use_leftside = a.use_rightside = a.use_midside = 1;
  a.use_fixed = 1;
  a.fixed_max_order = cfg->fixed_max_order;
  a.fixed_order_sel = cfg->fixed_order_sel;
  a.fixed_group_log2 = 0;
  a.fixed_keys = h->fixed_keys;
  a.fixed_partitions = cfg->fixed_partitions;
  a.forced_uniform = 0;
  a.forced_orders = nullptr;
  a.selector_keys = selector_keys;
  a.lpc_stage = 0;
  a.pack_out = nullptr;
  a.pred = nullptr;
  a.pred_out = nullptr;
  a.split_scratch = nullptr;
  if (plan.table_scratch_bytes_per_subframe) {
    rc = ensure(h, h->d_tables, plan.table_scratch_bytes_per_subframe * n_subframes);
    if (rc != FLACENC_HIP_OK) return rc;
    a.table_scratch = static_cast<uint32_t*>(h->d_tables.ptr);
  }
  if (cfg->fixed_order_sel == FLACENC_HIP_ORDERSEL_APPROXENT) {
    a.fixed_mode = 1;
    if ((rc = attach_sumabs_scratch(h, a, true)) != FLACENC_HIP_OK) return rc;
    if (block_size == 4096 || block_size == 8192 || block_size == 16384 ||  // the big-block kernels' predictor records
        flacenc_hip::subwave_shape(block_size)) {                            // ... the sub-wave kernel's clean-up counter
      if ((rc = attach_split_scratch(h, a)) != FLACENC_HIP_OK) return rc;
    }
    if (residual_mode == 1u) {  // (the caller checked that this launch takes the big-block kernels)
      if (!flacenc_hip::bigblock_fixed_eligible(a)) {
        h->last_error = "internal: analyse-only fixed_lpc batch on a shape the big-block kernels do not take";
        return FLACENC_HIP_ERR_UNSUPPORTED;
      }
      a.residual_mode = 1u;
    }
    HIP_TRY(h, flacenc_hip::launch_qlpc(a, plan, stream));
    return FLACENC_HIP_OK;
  }
  // BitCount: code every order, keep the first minimum of bps*order + code_bits, code it again
  const uint32_t n_orders = cfg->fixed_max_order + 1;
  if ((rc = ensure(h, h->d_keys, n_subframes * n_orders * 8)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_sel, n_subframes)) != FLACENC_HIP_OK) return rc;
  unsigned long long* keys = static_cast<unsigned long long*>(h->d_keys.ptr);
  for (uint32_t k = 0; k < n_orders; ++k) {
    a.fixed_mode = 2;
    a.forced_uniform = k;
    a.selector_keys = keys + static_cast<size_t>(k) * n_subframes;
    HIP_TRY(h, flacenc_hip::launch_qlpc(a, plan, stream));
    if (h->fixed_keys) {  // test hook: keys[sf*8 + k]
      HIP_TRY(h, hipMemcpy2DAsync(h->fixed_keys + k, 8 * 8, a.selector_keys, 8, 8, n_subframes,
                                  hipMemcpyDeviceToDevice, stream));
    }
  }
  const uint32_t n32 = static_cast<uint32_t>(n_subframes);
  hipLaunchKernelGGL(bitcount_pick_kernel, dim3((n32 + 255) / 256), dim3(256), 0, stream, keys, n32, n_orders,
                     static_cast<uint8_t*>(h->d_sel.ptr), selector_keys);
  HIP_TRY(h, hipGetLastError());
  a.fixed_mode = 3;
  a.forced_orders = static_cast<const uint8_t*>(h->d_sel.ptr);
  a.selector_keys = nullptr;
  HIP_TRY(h, flacenc_hip::launch_qlpc(a, plan, stream));
  return FLACENC_HIP_OK;
}

}  // namespace

struct PackTarget {  // optional: where the fused kernel puts the packed frames
  uint8_t* out = nullptr;
  size_t out_stride = 0;
  uint32_t* out_len = nullptr;
  uint32_t sample_rate = 0, first_frame_number = 0, frame_number_step = 1;
};
static int encode_stereo_frames_impl(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                     const int32_t* frames, size_t n_frames, uint32_t block_size, size_t stride,
                                     uint32_t bits_per_sample, flacenc_hip_stereo_frame_result* results,
                                     int32_t* residual, size_t residual_stride, void* stream,
                                     const PackTarget* pack, bool* packed);

extern "C" {

int flacenc_hip_abi_version(void) { return FLACENC_HIP_ABI_VERSION; }

int flacenc_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int flacenc_hip_create(flacenc_hip_handle** out, int device_id) {
  if (!out) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FLACENC_HIP_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= n) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  flacenc_hip_handle* h = new (std::nothrow) flacenc_hip_handle();
  if (!h) return FLACENC_HIP_ERR_DEVICE;
  h->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    if (h->d_cert_fb) (void)hipFree(h->d_cert_fb);
  if (h->h_cert_fb) (void)hipHostFree(h->h_cert_fb);
  delete h;
    return FLACENC_HIP_ERR_DEVICE;
  }
  *out = h;
  return FLACENC_HIP_OK;
}

void flacenc_hip_destroy(flacenc_hip_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  flacenc_hip::comm_release(h->comm);
  h->comm = nullptr;
  for (WindowEntry& e : h->windows)
    if (e.dev) (void)hipFree(e.dev);
  for (DeviceBuffer* b : {&h->d_samples, &h->d_residual, &h->d_params, &h->d_bps, &h->d_autocorr,
                          &h->d_lpc, &h->d_tables, &h->d_keys, &h->d_sel, &h->d_results, &h->d_out, &h->d_outlen, &h->d_cparams, &h->d_cresid,
                          &h->d_fparams, &h->d_fresid, &h->d_fkeys, &h->d_split, &h->d_presid, &h->d_sumabs, &h->d_minmax, &h->d_marked, &h->d_irlsw, &h->d_gram})
    if (b->ptr) (void)hipFree(b->ptr);
  for (int i = 0; i < 2; ++i) {
    for (DeviceBuffer* b : {&h->d_pcm[i], &h->d_pack[i], &h->d_plen[i], &h->d_poff[i], &h->d_cont[i]})
      if (b->ptr) (void)hipFree(b->ptr);
    for (void* p : {h->pin_in[i], h->pin_out[i], h->pin_meta[i]})
      if (p) (void)hipHostFree(p);
    for (hipEvent_t e : {h->ev_h2d[i], h->ev_fill[i], h->ev_pack[i], h->ev_d2h[i]})
      if (e) (void)hipEventDestroy(e);
  }
  if (h->s_in) (void)hipStreamDestroy(h->s_in);
  if (h->s_out) (void)hipStreamDestroy(h->s_out);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* flacenc_hip_last_error(const flacenc_hip_handle* h) {
  return h ? h->last_error.c_str() : "null handle";
}

int flacenc_hip_verify_config(const flacenc_hip_qlpc_config* cfg) {
  if (!cfg) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const uint32_t max_order = (cfg->flags & FLACENC_HIP_FLAG_ALLOW_ORDER_32)
                                 ? FLACENC_HIP_MAX_LPC_ORDER
                                 : FLACENC_HIP_REF_MAX_LPC_ORDER;
  // config::Qlpc::verify, src/config.rs:302-326
  if (cfg->lpc_order < 1 || cfg->lpc_order > max_order) return FLACENC_HIP_ERR_BAD_CONFIG;
  if (cfg->quant_precision < 1 || cfg->quant_precision > FLACENC_HIP_MAX_PRECISION)
    return FLACENC_HIP_ERR_BAD_CONFIG;
  // config::Window::verify, src/config.rs:371-387
  if (cfg->window_type == FLACENC_HIP_WINDOW_TUKEY) {
    if (!(cfg->tukey_alpha >= 0.0f && cfg->tukey_alpha <= 1.0f)) return FLACENC_HIP_ERR_BAD_CONFIG;
  } else if (cfg->window_type != FLACENC_HIP_WINDOW_RECTANGLE) {
    return FLACENC_HIP_ERR_BAD_CONFIG;
  }
  // config::Prc::verify, src/config.rs:224-229
  if (cfg->max_rice_parameter > FLACENC_HIP_MAX_RICE_PARAMETER) return FLACENC_HIP_ERR_BAD_CONFIG;
  // config::Qlpc::use_direct_mse / mae_optimization_steps (src/config.rs:280-285): accepted as in the
  // reference's `experimental` build
  if (cfg->use_direct_mse > 1 || cfg->mae_optimization_steps > FLACENC_HIP_MAX_MAE_STEPS) return FLACENC_HIP_ERR_BAD_CONFIG;
  // one summation order at a time; the simd-nightly order is only defined up to lag 15 (beyond, `as_simd`
  // splits 128- and 256-byte vectors at addresses the allocator picks, src/lpc.rs:459, :519-523)
  if ((cfg->flags & FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER) && (cfg->flags & FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER))
    return FLACENC_HIP_ERR_BAD_CONFIG;
  if ((cfg->flags & FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER) && cfg->lpc_order > 15) return FLACENC_HIP_ERR_UNSUPPORTED;
  return FLACENC_HIP_OK;
}

int flacenc_hip_window_weights(const flacenc_hip_qlpc_config* cfg, uint32_t block_size, float* out) {
  if (!cfg || !out) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  int rc = flacenc_hip_verify_config(cfg);
  if (rc != FLACENC_HIP_OK) return rc;
  window_weights(cfg->window_type, cfg->tukey_alpha, block_size, out);
  return FLACENC_HIP_OK;
}

#ifdef FLACENC_HIP_DEBUG_HOOKS  // (flacenc_hip_debug.h: not part of the public ABI)
int flacenc_hip_debug_set_fixed_keys(flacenc_hip_handle* h, unsigned long long* device_keys) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  h->fixed_keys = device_keys;
  return FLACENC_HIP_OK;
}

int flacenc_hip_debug_set_stamps(flacenc_hip_handle* h, unsigned long long* device_stamps) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  h->stamps = device_stamps;
  return FLACENC_HIP_OK;
}

int flacenc_hip_debug_set_adaptive_order(flacenc_hip_handle* h, int on) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  h->adaptive_order = on ? 1 : 0;
  h->two_pass_left = h->two_pass_span = 0;
  h->fb_probe_out = false;
  return FLACENC_HIP_OK;
}

int flacenc_hip_debug_adaptive_state(flacenc_hip_handle* h, int* span, int* left) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (span) *span = h->two_pass_span;
  if (left) *left = h->two_pass_left;
  return FLACENC_HIP_OK;
}

int flacenc_hip_debug_set_cert_stats(flacenc_hip_handle* h, uint32_t* device_counters) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  h->cert_stats = device_counters;
  return FLACENC_HIP_OK;
}
#endif

int flacenc_hip_fixed_lpc_batch_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                      const int32_t* samples, size_t n_units, uint32_t block_size,
                                      size_t stride, const uint8_t* bps, uint32_t bits_per_sample, int layout,
                                      flacenc_hip_subframe_params* params, int32_t* residual,
                                      size_t residual_stride, uint64_t* selector_keys, void* stream) {
  if (!h || !cfg) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const bool stereo = layout == FLACENC_HIP_LAYOUT_STEREO_FRAMES;
  if (!stereo && layout != FLACENC_HIP_LAYOUT_SUBFRAMES) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const size_t n_sub = stereo ? n_units * 4 : n_units;
  int rc = check_batch_args(h, &cfg->qlpc, samples, n_sub, block_size, stride, params, residual, residual_stride);
  if (rc != FLACENC_HIP_OK || n_units == 0) return rc;
  if (!bps && (bits_per_sample < 8 || bits_per_sample > 25)) {
    h->last_error = "bits_per_sample must be in 8..=25 when no per-subframe bps array is given";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  return enqueue_fixed(h, cfg, samples, n_sub, block_size, stride, stereo ? nullptr : bps, bits_per_sample, stereo,
                       params, residual, residual_stride, reinterpret_cast<unsigned long long*>(selector_keys),
                       static_cast<hipStream_t>(stream));
}

int flacenc_hip_fixed_lpc_batch(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                const int32_t* samples, size_t n_units, uint32_t block_size, size_t stride,
                                const uint8_t* bps, uint32_t bits_per_sample, int layout,
                                flacenc_hip_subframe_params* params, int32_t* residual, size_t residual_stride,
                                uint64_t* selector_keys, int memory_kind) {
  if (!h || !cfg) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (memory_kind == FLACENC_HIP_MEM_DEVICE) {
    int rc = flacenc_hip_fixed_lpc_batch_async(h, cfg, samples, n_units, block_size, stride, bps, bits_per_sample,
                                               layout, params, residual, residual_stride, selector_keys, h->stream);
    if (rc != FLACENC_HIP_OK || n_units == 0) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FLACENC_HIP_OK;
  }
  if (memory_kind != FLACENC_HIP_MEM_HOST) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const bool stereo = layout == FLACENC_HIP_LAYOUT_STEREO_FRAMES;
  if (!stereo && layout != FLACENC_HIP_LAYOUT_SUBFRAMES) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const size_t n_sub = stereo ? n_units * 4 : n_units;
  const size_t n_rows = stereo ? n_units * 2 : n_units;
  int rc = check_batch_args(h, &cfg->qlpc, samples, n_sub, block_size, stride, params, residual, residual_stride);
  if (rc != FLACENC_HIP_OK || n_units == 0) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t dstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  if ((rc = ensure(h, h->d_samples, n_rows * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_residual, n_sub * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_params, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_autocorr, n_sub * 8)) != FLACENC_HIP_OK) return rc;  // selector keys
  if (bps && !stereo && (rc = ensure(h, h->d_bps, n_sub)) != FLACENC_HIP_OK) return rc;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpy2DAsync(h->d_samples.ptr, dstride * 4, samples, stride * 4,
                              static_cast<size_t>(block_size) * 4, n_rows, hipMemcpyHostToDevice, s));
  if (bps && !stereo) HIP_TRY(h, hipMemcpyAsync(h->d_bps.ptr, bps, n_sub, hipMemcpyHostToDevice, s));
  rc = flacenc_hip_fixed_lpc_batch_async(h, cfg, static_cast<const int32_t*>(h->d_samples.ptr), n_units, block_size,
                                         dstride, (bps && !stereo) ? static_cast<const uint8_t*>(h->d_bps.ptr) : nullptr,
                                         bits_per_sample, layout,
                                         static_cast<flacenc_hip_subframe_params*>(h->d_params.ptr),
                                         static_cast<int32_t*>(h->d_residual.ptr), dstride,
                                         static_cast<uint64_t*>(h->d_autocorr.ptr), s);
  if (rc != FLACENC_HIP_OK) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(residual, residual_stride * 4, h->d_residual.ptr, dstride * 4,
                              static_cast<size_t>(block_size) * 4, n_sub, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(params, h->d_params.ptr, n_sub * sizeof(flacenc_hip_subframe_params),
                            hipMemcpyDeviceToHost, s));
  if (selector_keys)
    HIP_TRY(h, hipMemcpyAsync(selector_keys, h->d_autocorr.ptr, n_sub * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return FLACENC_HIP_OK;
}

size_t flacenc_hip_frame_bytes_bound(uint32_t channels, uint32_t block_size, uint32_t bits_per_sample) {
  return (flacenc_hip::frame_bytes_bound(channels, block_size, bits_per_sample) + 15) & ~static_cast<size_t>(15);
}

size_t flacenc_hip_stereo_frame_bytes_bound(uint32_t block_size, uint32_t bits_per_sample) {
  return (flacenc_hip::stereo_frame_bytes_bound(block_size, bits_per_sample) + 15) & ~static_cast<size_t>(15);
}

// Frame::write for a batch: `results` (2-channel records) or `chan_results` + channels
static int enqueue_pack(flacenc_hip_handle* h, const int32_t* frames, size_t n_frames, uint32_t channels,
                        uint32_t block_size, size_t stride, const flacenc_hip_stereo_frame_result* results,
                        const flacenc_hip_channel_result* chan_results, const int32_t* residual,
                        size_t residual_stride, uint32_t bits_per_sample, uint32_t sample_rate,
                        uint32_t first_frame_number, uint32_t frame_number_step, uint8_t* out,
                        size_t out_stride, uint32_t* out_len, void* stream) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  const size_t bound = chan_results ? flacenc_hip_frame_bytes_bound(channels, block_size, bits_per_sample)
                                    : flacenc_hip_stereo_frame_bytes_bound(block_size, bits_per_sample);
  if (chan_results && (channels < 1 || channels > 8)) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (!frames || (!results && !chan_results) || !residual || !out || !out_len || stride < block_size || residual_stride < block_size ||
      block_size < 1 || block_size > FLACENC_HIP_MAX_BLOCK_SIZE ||
      bits_per_sample < 8 || bits_per_sample > 24 || n_frames > 0x7FFFFFFFull ||
      (reinterpret_cast<uintptr_t>(out) & 15) || (out_stride & 15) ||
      out_stride < bound) {
    h->last_error = "pack_stereo_frames: null pointer, bad size, or out_stride below flacenc_hip_stereo_frame_bytes_bound";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  // frame numbers must stay below 2^31 (encode_fixed_size_frame, coding.rs:587-591)
  const unsigned long long last = static_cast<unsigned long long>(first_frame_number) +
                                  static_cast<unsigned long long>(n_frames - 1) * frame_number_step;
  if (last >= (1ull << 31)) {
    h->last_error = "pack_stereo_frames: frame_number must be below 2^31";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  flacenc_hip::FramePackArgs a;
  a.frames = frames;
  a.stride = stride;
  a.block_size = block_size;
  a.n_frames = static_cast<uint32_t>(n_frames);
  a.results = results;
  a.chan_results = chan_results;
  a.channels = chan_results ? channels : 2u;
  a.residual = residual;
  a.residual_stride = residual_stride;
  a.bits_per_sample = bits_per_sample;
  a.first_frame_number = first_frame_number;
  a.frame_number_step = frame_number_step;
  a.out = out;
  a.out_stride = out_stride;
  a.out_len = out_len;
  fill_header_specs(a, block_size, sample_rate, bits_per_sample);
  a.lds_words = static_cast<uint32_t>(bound / 4 + 4);
  fill_crc_powers(a.lds_words, &a.crc_per, a.crc_pow);
  if (static_cast<size_t>(a.lds_words) * 4 > 150 * 1024) {
    h->last_error = "pack_stereo_frames: frame too large for the LDS bit buffer (block_size x bits_per_sample)";
    return FLACENC_HIP_ERR_UNSUPPORTED;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, flacenc_hip::launch_frame_pack(a, static_cast<hipStream_t>(stream)));
  return FLACENC_HIP_OK;
}

int flacenc_hip_pack_stereo_frames_async(flacenc_hip_handle* h, const int32_t* frames, size_t n_frames,
                                         uint32_t block_size, size_t stride,
                                         const flacenc_hip_stereo_frame_result* results, const int32_t* residual,
                                         size_t residual_stride, uint32_t bits_per_sample, uint32_t sample_rate,
                                         uint32_t first_frame_number, uint32_t frame_number_step, uint8_t* out,
                                         size_t out_stride, uint32_t* out_len, void* stream) {
  if (!results && n_frames) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  return enqueue_pack(h, frames, n_frames, 2, block_size, stride, results, nullptr, residual, residual_stride,
                      bits_per_sample, sample_rate, first_frame_number, frame_number_step, out, out_stride, out_len,
                      stream);
}

int flacenc_hip_pack_frames_async(flacenc_hip_handle* h, const int32_t* frames, size_t n_frames, uint32_t channels,
                                  uint32_t block_size, size_t stride, const flacenc_hip_channel_result* results,
                                  const int32_t* residual, size_t residual_stride, uint32_t bits_per_sample,
                                  uint32_t sample_rate, uint32_t first_frame_number, uint32_t frame_number_step,
                                  uint8_t* out, size_t out_stride, uint32_t* out_len, void* stream) {
  if (!results && n_frames) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  return enqueue_pack(h, frames, n_frames, channels, block_size, stride, nullptr, results, residual, residual_stride,
                      bits_per_sample, sample_rate, first_frame_number, frame_number_step, out, out_stride, out_len,
                      stream);
}

int flacenc_hip_stereo_frame_lengths_async(flacenc_hip_handle* h, const flacenc_hip_stereo_frame_result* results,
                                           size_t n_frames, uint32_t block_size, uint32_t bits_per_sample,
                                           uint32_t sample_rate, uint32_t first_frame_number,
                                           uint32_t frame_number_step, uint32_t* out_len, void* stream) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  if (!results || !out_len || n_frames > 0x7FFFFFFFull) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const unsigned long long last = static_cast<unsigned long long>(first_frame_number) +
                                  static_cast<unsigned long long>(n_frames - 1) * frame_number_step;
  if (last >= (1ull << 31)) {
    h->last_error = "stereo_frame_lengths: frame_number must be below 2^31";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  flacenc_hip::FramePackArgs a{};
  a.results = results;
  a.n_frames = static_cast<uint32_t>(n_frames);
  a.first_frame_number = first_frame_number;
  a.frame_number_step = frame_number_step;
  a.out_len = out_len;
  fill_header_specs(a, block_size, sample_rate, bits_per_sample);
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, flacenc_hip::launch_frame_lengths(a, static_cast<hipStream_t>(stream)));
  return FLACENC_HIP_OK;
}

// 2^finest_partition_order of a block (src/rice.rs:157-165 with a warm-up of at most 64 samples: the bound over all
// predictor orders): how many of a subframe record's rice_params can be non-zero
static uint32_t finest_partitions(uint32_t block_size) {
  uint32_t order = 0, n = block_size;
  while (order < 8 && n % 2 == 0 && n / 2 >= 64) {
    n /= 2;
    ++order;
  }
  return 1u << order;
}

size_t flacenc_hip_frame_wire_bytes(uint32_t block_size) { return 48 + 2 * (96 + static_cast<size_t>(finest_partitions(block_size))); }

int flacenc_hip_stereo_frame_wire_async(flacenc_hip_handle* h, const flacenc_hip_stereo_frame_result* results,
                                        size_t n_frames, uint32_t block_size, uint32_t bits_per_sample,
                                        uint32_t sample_rate, uint32_t first_frame_number, uint32_t frame_number_step,
                                        uint8_t* wire, size_t wire_stride, uint32_t* out_len, void* stream) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  if (!results || !wire || n_frames > 0x7FFFFFFFull || wire_stride < flacenc_hip_frame_wire_bytes(block_size))
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const unsigned long long last = static_cast<unsigned long long>(first_frame_number) +
                                  static_cast<unsigned long long>(n_frames - 1) * frame_number_step;
  if (last >= (1ull << 31)) {
    h->last_error = "stereo_frame_wire: frame_number must be below 2^31";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  flacenc_hip::FramePackArgs a{};
  a.results = results;
  a.n_frames = static_cast<uint32_t>(n_frames);
  a.first_frame_number = first_frame_number;
  a.frame_number_step = frame_number_step;
  a.out_len = out_len;
  fill_header_specs(a, block_size, sample_rate, bits_per_sample);
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, flacenc_hip::launch_frame_wire(a, finest_partitions(block_size), wire, wire_stride,
                                            static_cast<hipStream_t>(stream)));
  return FLACENC_HIP_OK;
}

int flacenc_hip_stream_offsets_async(flacenc_hip_handle* h, const uint32_t* gathered_lengths, size_t n_frames_total,
                                     uint32_t world, uint64_t header_bytes, uint32_t* lengths_stream,
                                     uint64_t* offsets, uint64_t* total, void* stream) {
  if (!h || world == 0 || !total || n_frames_total > 0x7FFFFFFFull) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames_total && (!gathered_lengths || !offsets)) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const uint32_t n = static_cast<uint32_t>(n_frames_total);
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, flacenc_hip::launch_stream_offsets(gathered_lengths, n, world, (n + world - 1) / world, header_bytes,
                                                lengths_stream, offsets, total, static_cast<hipStream_t>(stream)));
  return FLACENC_HIP_OK;
}

int flacenc_hip_place_frames_async(flacenc_hip_handle* h, const uint8_t* src, const uint64_t* src_offsets,
                                   const uint32_t* lengths, size_t n_frames, uint8_t* dst,
                                   const uint64_t* dst_offsets, void* stream) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  if (!src || !src_offsets || !lengths || !dst || !dst_offsets || n_frames > 0x7FFFFFFFull)
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, flacenc_hip::launch_place_frames(src, src_offsets, lengths, dst, dst_offsets,
                                              static_cast<uint32_t>(n_frames), static_cast<hipStream_t>(stream)));
  return FLACENC_HIP_OK;
}

int flacenc_hip_pack_stereo_frames(flacenc_hip_handle* h, const int32_t* frames, size_t n_frames,
                                   uint32_t block_size, size_t stride,
                                   const flacenc_hip_stereo_frame_result* results, const int32_t* residual,
                                   size_t residual_stride, uint32_t bits_per_sample, uint32_t sample_rate,
                                   uint32_t first_frame_number, uint32_t frame_number_step, uint8_t* out,
                                   size_t out_stride, uint32_t* out_len, int memory_kind) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (memory_kind == FLACENC_HIP_MEM_DEVICE) {
    int rc = flacenc_hip_pack_stereo_frames_async(h, frames, n_frames, block_size, stride, results, residual,
                                                  residual_stride, bits_per_sample, sample_rate, first_frame_number,
                                                  frame_number_step, out, out_stride, out_len, h->stream);
    if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FLACENC_HIP_OK;
  }
  if (memory_kind != FLACENC_HIP_MEM_HOST) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  if (!frames || !results || !residual || !out || !out_len || stride < block_size || residual_stride < block_size)
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  int rc;
  const size_t dstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  const size_t ostride = flacenc_hip_stereo_frame_bytes_bound(block_size, bits_per_sample);
  if (out_stride < ostride) {
    h->last_error = "pack_stereo_frames: out_stride below flacenc_hip_stereo_frame_bytes_bound";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  if ((rc = ensure(h, h->d_samples, n_frames * 2 * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_residual, n_frames * 2 * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_results, n_frames * sizeof(flacenc_hip_stereo_frame_result))) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_out, n_frames * ostride)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_outlen, n_frames * 4)) != FLACENC_HIP_OK) return rc;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpy2DAsync(h->d_samples.ptr, dstride * 4, frames, stride * 4, static_cast<size_t>(block_size) * 4,
                              n_frames * 2, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpy2DAsync(h->d_residual.ptr, dstride * 4, residual, residual_stride * 4,
                              static_cast<size_t>(block_size) * 4, n_frames * 2, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(h->d_results.ptr, results, n_frames * sizeof(flacenc_hip_stereo_frame_result),
                            hipMemcpyHostToDevice, s));
  rc = flacenc_hip_pack_stereo_frames_async(h, static_cast<const int32_t*>(h->d_samples.ptr), n_frames, block_size,
                                            dstride, static_cast<const flacenc_hip_stereo_frame_result*>(h->d_results.ptr),
                                            static_cast<const int32_t*>(h->d_residual.ptr), dstride, bits_per_sample,
                                            sample_rate, first_frame_number, frame_number_step,
                                            static_cast<uint8_t*>(h->d_out.ptr), ostride,
                                            static_cast<uint32_t*>(h->d_outlen.ptr), s);
  if (rc != FLACENC_HIP_OK) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(out, out_stride, h->d_out.ptr, ostride, ostride, n_frames, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(out_len, h->d_outlen.ptr, n_frames * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return FLACENC_HIP_OK;
}

int flacenc_hip_encode_frames_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                    const int32_t* frames, size_t n_frames, uint32_t channels,
                                    uint32_t block_size, size_t stride, uint32_t bits_per_sample,
                                    flacenc_hip_channel_result* results, int32_t* residual,
                                    size_t residual_stride, void* stream) {
  if (!h || !cfg || (!results && n_frames) || channels < 1 || channels > 8) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const size_t n_sub = n_frames * channels;
  int rc = check_batch_args(h, &cfg->qlpc, frames, n_sub, block_size, stride,
                            reinterpret_cast<flacenc_hip_subframe_params*>(results), residual, residual_stride, 1);
  if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
  if (bits_per_sample < 8 || bits_per_sample > 24) {
    h->last_error = "bits_per_sample must be in 8..=24";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  if (cfg->use_fixed && (rc = verify_fixed(h, cfg)) != FLACENC_HIP_OK) return rc;
  // too_short (coding.rs:396): neither fixed_lpc nor estimated_qlpc is tried; Constant or Verbatim
  flacenc_hip_frame_config short_cfg;
  if (block_size < FLACENC_HIP_MIN_BLOCK_SIZE) {
    short_cfg = *cfg;
    short_cfg.use_fixed = short_cfg.use_lpc = 0;
    cfg = &short_cfg;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    // block size 4096, order <= 12: one fused kernel, a wave per channel (analysis, fixed-LPC
    // candidate, encode_subframe's choice, only the chosen residual written)
    uint32_t glog = 0;
    bool pow2 = true;
    if (cfg->use_fixed && cfg->fixed_order_sel == FLACENC_HIP_ORDERSEL_APPROXENT) {
      const uint32_t p = cfg->fixed_partitions;
      pow2 = (p & (p - 1)) == 0;
      for (uint32_t lanes = pow2 ? 64u / p : 1u; lanes > 1; lanes >>= 1) ++glog;
    }
    const WindowEntry* win = nullptr;
    static const WindowEntry no_window{};
    if (block_size >= FLACENC_HIP_MIN_BLOCK_SIZE) {
      rc = get_window(h, &cfg->qlpc, block_size, &win);
      if (rc != FLACENC_HIP_OK) return rc;
    } else {
      win = &no_window;  // (never read: short blocks take the candidate-free general path below)
    }
    flacenc_hip::QlpcKernelArgs a;
    a.samples = frames;
    a.stride = stride;
    a.block_size = block_size;
    a.n_subframes = static_cast<uint32_t>(n_sub);
    a.bps = nullptr;
    a.bps_uniform = bits_per_sample;
    a.stereo = 0;
    a.window = win->dev;
    a.flat_lo = win->flat_lo;
    a.flat_hi = win->flat_hi;
    a.lpc_order = cfg->qlpc.lpc_order;
    a.precision = cfg->qlpc.quant_precision;
    a.max_rice_parameter = cfg->qlpc.max_rice_parameter;
  a.rice_finest_only = (cfg->qlpc.flags & FLACENC_HIP_FLAG_FINEST_RICE_ORDER) ? 1u : 0u;
  a.force_generic = (cfg->qlpc.flags & FLACENC_HIP_FLAG_GENERIC_KERNEL) ? 1u : 0u;
  a.reference_order = sum_order_mode(cfg->qlpc.flags);
  set_certify(h, a, cfg->qlpc.flags);
  a.direct_mse = cfg->qlpc.use_direct_mse ? 1u : 0u;  // (keeps the launch off the fused wave kernel)
  a.acorr_in = nullptr;
  a.only_marked = 0;
    a.params = nullptr;
    a.residual = residual;
    a.residual_stride = residual_stride;
    a.autocorr = nullptr;
    a.lpc_coefs = nullptr;
    a.table_scratch = nullptr;
    a.stamps = nullptr;
    a.frame_results = nullptr;
    a.chan_results = results;
    a.use_constant = cfg->use_constant;
    a.use_lpc = cfg->use_lpc;
    a.use_leftside = a.use_rightside = a.use_midside = 0;
    a.use_fixed = cfg->use_fixed;
    a.fixed_max_order = cfg->fixed_max_order;
    a.fixed_order_sel = cfg->fixed_order_sel;
    a.fixed_group_log2 = glog;
    a.fixed_keys = h->fixed_keys;
    a.fixed_mode = a.fixed_partitions = a.forced_uniform = 0;
    a.forced_orders = nullptr;
    a.selector_keys = nullptr;
    a.lpc_stage = 0;
    a.pack_out = nullptr;
  a.pack_out = nullptr;
    a.pred = nullptr;
    a.pred_out = nullptr;
    a.split_scratch = nullptr;
    if (a.reference_order || certify_needs_scratch(a)) {  // R[] of the reference-order pass
      if ((rc = attach_split_scratch(h, a)) != FLACENC_HIP_OK) return rc;
    }
    if ((rc = attach_sumabs_scratch(h, a, cfg->use_fixed && cfg->fixed_order_sel == FLACENC_HIP_ORDERSEL_APPROXENT)) !=
        FLACENC_HIP_OK)
      return rc;
    if (pow2 && flacenc_hip::wave_kernel_eligible(a)) {
      flacenc_hip::QlpcLaunchPlan plan = flacenc_hip::plan_qlpc_launch(block_size, cfg->qlpc.lpc_order);
      return launch_adaptive(h, a, plan, s);
    }
    // blocks of 8 / 16 / 32 finest Rice partitions: qlpc_subwave_kernel's independent-channel variant -- both candidates
    // and encode_subframe's choice of every channel in one launch; what it marks takes the general path below, restricted
    // to the marked subframes (three launches that return at once when the count is 0)
    if (block_size >= FLACENC_HIP_MIN_BLOCK_SIZE && flacenc_hip::subwave_shape(block_size)) {
      flacenc_hip::QlpcKernelArgs m = a;
      m.fixed_partitions = cfg->fixed_partitions;
      m.cert_subwave = (a.certify != 0u && a.reference_order == 0u && !a.direct_mse && cfg->use_lpc && cfg->qlpc.lpc_order <= 12) ? 1u : 0u;
      if ((rc = attach_split_scratch(h, m)) != FLACENC_HIP_OK) return rc;
      if (flacenc_hip::subwave_channels_eligible(m)) {
        const size_t cs = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
        if ((rc = ensure(h, h->d_cparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
        if ((rc = ensure(h, h->d_cresid, n_sub * cs * 4)) != FLACENC_HIP_OK) return rc;
        m.cand_lpc_params = static_cast<const flacenc_hip_subframe_params*>(h->d_cparams.ptr);
        if (cfg->use_fixed) {
          if ((rc = ensure(h, h->d_fparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
          if ((rc = ensure(h, h->d_fresid, n_sub * cs * 4)) != FLACENC_HIP_OK) return rc;
          if ((rc = ensure(h, h->d_fkeys, n_sub * 8)) != FLACENC_HIP_OK) return rc;
          m.cand_fixed_params = static_cast<const flacenc_hip_subframe_params*>(h->d_fparams.ptr);
        }
        HIP_TRY(h, flacenc_hip::launch_subwave_frames(m, s));
        flacenc_hip::ChannelDecideArgs dm{};
        dm.samples = frames;
        dm.stride = stride;
        dm.block_size = block_size;
        dm.n_subframes = static_cast<uint32_t>(n_sub);
        dm.bits_per_sample = bits_per_sample;
        dm.use_constant = cfg->use_constant;
        dm.use_fixed = cfg->use_fixed;
        dm.use_lpc = cfg->use_lpc;
        dm.cand_stride = cs;
        dm.results = results;
        dm.residual = residual;
        dm.residual_stride = residual_stride;
        flacenc_hip::QlpcKernelArgs c = m;
        c.chan_results = nullptr;
        c.cand_lpc_params = c.cand_fixed_params = nullptr;
        c.params = static_cast<flacenc_hip_subframe_params*>(h->d_cparams.ptr);
        c.residual = static_cast<int32_t*>(h->d_cresid.ptr);
        c.residual_stride = cs;
        c.only_marked = 1;
        c.use_fixed = 0;
        c.fixed_keys = nullptr;
        HIP_TRY(h, flacenc_hip::launch_qlpc(c, flacenc_hip::plan_qlpc_launch(block_size, cfg->qlpc.lpc_order), s));
        dm.lpc_params = c.params;
        dm.lpc_residual = c.residual;
        if (cfg->use_fixed) {
          flacenc_hip::QlpcKernelArgs x = c;
          x.params = static_cast<flacenc_hip_subframe_params*>(h->d_fparams.ptr);
          x.residual = static_cast<int32_t*>(h->d_fresid.ptr);
          x.selector_keys = static_cast<unsigned long long*>(h->d_fkeys.ptr);
          x.window = nullptr;
          x.flat_lo = x.flat_hi = 0;
          x.lpc_order = 4;
          x.precision = 0;
          x.use_fixed = 1;
          x.fixed_mode = 1;
          HIP_TRY(h, flacenc_hip::launch_qlpc(x, flacenc_hip::plan_qlpc_launch(block_size, 4), s));
          dm.fixed_params = x.params;
          dm.fixed_residual = x.residual;
          dm.fixed_keys = x.selector_keys;
        }
        dm.only_marked = 1;
        dm.marked_count = m.marked_count;
        dm.marked_list = m.marked_list;  // (entries: subframes)
        dm.marked_cap = m.marked_cap;
        HIP_TRY(h, flacenc_hip::launch_channel_decide(dm, s));
        return FLACENC_HIP_OK;
      }
    }
  }
  const size_t cstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  flacenc_hip::ChannelDecideArgs d{};
  d.samples = frames;
  d.stride = stride;
  d.block_size = block_size;
  d.n_subframes = static_cast<uint32_t>(n_sub);
  d.bits_per_sample = bits_per_sample;
  d.use_constant = cfg->use_constant;
  d.use_fixed = cfg->use_fixed;
  d.use_lpc = cfg->use_lpc;
  d.cand_stride = cstride;
  d.results = results;
  d.residual = residual;
  d.residual_stride = residual_stride;
  if (cfg->use_lpc) {
    if ((rc = ensure(h, h->d_cparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
    if ((rc = ensure(h, h->d_cresid, n_sub * cstride * 4)) != FLACENC_HIP_OK) return rc;
    rc = enqueue(h, &cfg->qlpc, frames, n_sub, block_size, stride, nullptr,
                 static_cast<flacenc_hip_subframe_params*>(h->d_cparams.ptr), static_cast<int32_t*>(h->d_cresid.ptr),
                 cstride, nullptr, nullptr, s, false, bits_per_sample);
    if (rc != FLACENC_HIP_OK) return rc;
    d.lpc_params = static_cast<const flacenc_hip_subframe_params*>(h->d_cparams.ptr);
    d.lpc_residual = static_cast<const int32_t*>(h->d_cresid.ptr);
  }
  if (cfg->use_fixed) {
    if ((rc = ensure(h, h->d_fparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
    if ((rc = ensure(h, h->d_fresid, n_sub * cstride * 4)) != FLACENC_HIP_OK) return rc;
    if ((rc = ensure(h, h->d_fkeys, n_sub * 8)) != FLACENC_HIP_OK) return rc;
    rc = enqueue_fixed(h, cfg, frames, n_sub, block_size, stride, nullptr, bits_per_sample, false,
                       static_cast<flacenc_hip_subframe_params*>(h->d_fparams.ptr),
                       static_cast<int32_t*>(h->d_fresid.ptr), cstride,
                       static_cast<unsigned long long*>(h->d_fkeys.ptr), s);
    if (rc != FLACENC_HIP_OK) return rc;
    d.fixed_params = static_cast<const flacenc_hip_subframe_params*>(h->d_fparams.ptr);
    d.fixed_residual = static_cast<const int32_t*>(h->d_fresid.ptr);
    d.fixed_keys = static_cast<const unsigned long long*>(h->d_fkeys.ptr);
  }
  HIP_TRY(h, flacenc_hip::launch_channel_decide(d, s));
  return FLACENC_HIP_OK;
}

int flacenc_hip_encode_frames(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                              const int32_t* frames, size_t n_frames, uint32_t channels, uint32_t block_size,
                              size_t stride, uint32_t bits_per_sample, flacenc_hip_channel_result* results,
                              int32_t* residual, size_t residual_stride, int memory_kind) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (memory_kind == FLACENC_HIP_MEM_DEVICE) {
    int rc = flacenc_hip_encode_frames_async(h, cfg, frames, n_frames, channels, block_size, stride, bits_per_sample,
                                             results, residual, residual_stride, h->stream);
    if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FLACENC_HIP_OK;
  }
  if (memory_kind != FLACENC_HIP_MEM_HOST || !cfg || channels < 1 || channels > 8) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const size_t n_sub = n_frames * channels;
  int rc = check_batch_args(h, &cfg->qlpc, frames, n_sub, block_size, stride,
                            reinterpret_cast<flacenc_hip_subframe_params*>(results), residual, residual_stride, 1);
  if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t dstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  if ((rc = ensure(h, h->d_samples, n_sub * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_residual, n_sub * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_results, n_sub * sizeof(flacenc_hip_channel_result))) != FLACENC_HIP_OK) return rc;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpy2DAsync(h->d_samples.ptr, dstride * 4, frames, stride * 4, static_cast<size_t>(block_size) * 4,
                              n_sub, hipMemcpyHostToDevice, s));
  rc = flacenc_hip_encode_frames_async(h, cfg, static_cast<const int32_t*>(h->d_samples.ptr), n_frames, channels,
                                       block_size, dstride, bits_per_sample,
                                       static_cast<flacenc_hip_channel_result*>(h->d_results.ptr),
                                       static_cast<int32_t*>(h->d_residual.ptr), dstride, s);
  if (rc != FLACENC_HIP_OK) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(residual, residual_stride * 4, h->d_residual.ptr, dstride * 4,
                              static_cast<size_t>(block_size) * 4, n_sub, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(results, h->d_results.ptr, n_sub * sizeof(flacenc_hip_channel_result),
                            hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return FLACENC_HIP_OK;
}

int flacenc_hip_pack_frames(flacenc_hip_handle* h, const int32_t* frames, size_t n_frames, uint32_t channels,
                            uint32_t block_size, size_t stride, const flacenc_hip_channel_result* results,
                            const int32_t* residual, size_t residual_stride, uint32_t bits_per_sample,
                            uint32_t sample_rate, uint32_t first_frame_number, uint32_t frame_number_step,
                            uint8_t* out, size_t out_stride, uint32_t* out_len, int memory_kind) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (memory_kind == FLACENC_HIP_MEM_DEVICE) {
    int rc = flacenc_hip_pack_frames_async(h, frames, n_frames, channels, block_size, stride, results, residual,
                                           residual_stride, bits_per_sample, sample_rate, first_frame_number,
                                           frame_number_step, out, out_stride, out_len, h->stream);
    if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FLACENC_HIP_OK;
  }
  if (memory_kind != FLACENC_HIP_MEM_HOST || channels < 1 || channels > 8) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  if (!frames || !results || !residual || !out || !out_len || stride < block_size || residual_stride < block_size)
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  int rc;
  const size_t n_sub = n_frames * channels;
  const size_t dstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  const size_t ostride = flacenc_hip_frame_bytes_bound(channels, block_size, bits_per_sample);
  if (out_stride < ostride) {
    h->last_error = "pack_frames: out_stride below flacenc_hip_frame_bytes_bound";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  if ((rc = ensure(h, h->d_samples, n_sub * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_residual, n_sub * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_results, n_sub * sizeof(flacenc_hip_channel_result))) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_out, n_frames * ostride)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_outlen, n_frames * 4)) != FLACENC_HIP_OK) return rc;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpy2DAsync(h->d_samples.ptr, dstride * 4, frames, stride * 4, static_cast<size_t>(block_size) * 4,
                              n_sub, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpy2DAsync(h->d_residual.ptr, dstride * 4, residual, residual_stride * 4,
                              static_cast<size_t>(block_size) * 4, n_sub, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(h->d_results.ptr, results, n_sub * sizeof(flacenc_hip_channel_result),
                            hipMemcpyHostToDevice, s));
  rc = flacenc_hip_pack_frames_async(h, static_cast<const int32_t*>(h->d_samples.ptr), n_frames, channels, block_size,
                                     dstride, static_cast<const flacenc_hip_channel_result*>(h->d_results.ptr),
                                     static_cast<const int32_t*>(h->d_residual.ptr), dstride, bits_per_sample,
                                     sample_rate, first_frame_number, frame_number_step,
                                     static_cast<uint8_t*>(h->d_out.ptr), ostride,
                                     static_cast<uint32_t*>(h->d_outlen.ptr), s);
  if (rc != FLACENC_HIP_OK) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(out, out_stride, h->d_out.ptr, ostride, ostride, n_frames, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(out_len, h->d_outlen.ptr, n_frames * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return FLACENC_HIP_OK;
}

int flacenc_hip_fill_le_bytes_async(flacenc_hip_handle* h, const uint8_t* bytes, uint64_t total_samples,
                                    uint32_t channels, uint32_t bytes_per_sample, size_t n_frames,
                                    uint32_t block_size, int32_t* frames, size_t stride, void* stream) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  if (!bytes || !frames || channels < 1 || channels > 8 || bytes_per_sample < 1 || bytes_per_sample > 4 ||
      block_size < 1 || block_size > FLACENC_HIP_MAX_BLOCK_SIZE || stride < block_size || n_frames > 0xFFFFull) {
    h->last_error = "fill_le_bytes: null pointer, channels not in 1..=8, bytes_per_sample not in 1..=4, or > 65535 frames";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, flacenc_hip::launch_fill_le_bytes(bytes, channels, bytes_per_sample, total_samples,
                                               static_cast<uint32_t>(n_frames), block_size, frames, stride,
                                               static_cast<hipStream_t>(stream)));
  return FLACENC_HIP_OK;
}

int flacenc_hip_fill_le_bytes(flacenc_hip_handle* h, const uint8_t* bytes, uint64_t total_samples,
                              uint32_t channels, uint32_t bytes_per_sample, size_t n_frames, uint32_t block_size,
                              int32_t* frames, size_t stride, int memory_kind) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (memory_kind == FLACENC_HIP_MEM_DEVICE) {
    int rc = flacenc_hip_fill_le_bytes_async(h, bytes, total_samples, channels, bytes_per_sample, n_frames,
                                             block_size, frames, stride, h->stream);
    if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FLACENC_HIP_OK;
  }
  if (memory_kind != FLACENC_HIP_MEM_HOST) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  if (!bytes || !frames || channels < 1 || channels > 8 || bytes_per_sample < 1 || bytes_per_sample > 4 ||
      stride < block_size)
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  int rc;
  const uint64_t wanted = static_cast<uint64_t>(n_frames) * block_size;
  const uint64_t have = total_samples < wanted ? total_samples : wanted;
  const size_t nbytes = static_cast<size_t>(have) * channels * bytes_per_sample;
  const size_t dstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  if ((rc = ensure(h, h->d_out, nbytes + 16)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_samples, n_frames * channels * dstride * 4)) != FLACENC_HIP_OK) return rc;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpyAsync(h->d_out.ptr, bytes, nbytes, hipMemcpyHostToDevice, s));
  rc = flacenc_hip_fill_le_bytes_async(h, static_cast<const uint8_t*>(h->d_out.ptr), have, channels, bytes_per_sample,
                                       n_frames, block_size, static_cast<int32_t*>(h->d_samples.ptr), dstride, s);
  if (rc != FLACENC_HIP_OK) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(frames, stride * 4, h->d_samples.ptr, dstride * 4, static_cast<size_t>(block_size) * 4,
                              n_frames * channels, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return FLACENC_HIP_OK;
}

int flacenc_hip_encode_pack_stereo_frames_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                                const int32_t* frames, size_t n_frames, uint32_t block_size,
                                                size_t stride, uint32_t bits_per_sample, uint32_t sample_rate,
                                                uint32_t first_frame_number, uint32_t frame_number_step,
                                                flacenc_hip_stereo_frame_result* results, uint8_t* out,
                                                size_t out_stride, uint32_t* out_len, void* stream) {
  if (!h || !cfg) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  if (!out || !out_len || (reinterpret_cast<uintptr_t>(out) & 15) || (out_stride & 15) ||
      out_stride < flacenc_hip_stereo_frame_bytes_bound(block_size, bits_per_sample)) {
    h->last_error = "encode_pack_stereo_frames: out must be 16-byte aligned, out_stride a multiple of 16 and at least "
                    "flacenc_hip_stereo_frame_bytes_bound";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  const unsigned long long last = static_cast<unsigned long long>(first_frame_number) +
                                  static_cast<unsigned long long>(n_frames - 1) * frame_number_step;
  if (last >= (1ull << 31)) {
    h->last_error = "encode_pack_stereo_frames: frame_number must be below 2^31";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  // residual rows are only an intermediate here: handle scratch
  int rc;
  const size_t cstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  if ((rc = ensure(h, h->d_presid, n_frames * 2 * cstride * 4)) != FLACENC_HIP_OK) return rc;
  PackTarget pt;
  pt.out = out;
  pt.out_stride = out_stride;
  pt.out_len = out_len;
  pt.sample_rate = sample_rate;
  pt.first_frame_number = first_frame_number;
  pt.frame_number_step = frame_number_step;
  bool packed = false;
  rc = encode_stereo_frames_impl(h, cfg, frames, n_frames, block_size, stride, bits_per_sample, results,
                                 static_cast<int32_t*>(h->d_presid.ptr), cstride, stream, &pt, &packed);
  if (rc != FLACENC_HIP_OK || packed) return rc;
  return flacenc_hip_pack_stereo_frames_async(h, frames, n_frames, block_size, stride, results,
                                              static_cast<const int32_t*>(h->d_presid.ptr), cstride, bits_per_sample,
                                              sample_rate, first_frame_number, frame_number_step, out, out_stride,
                                              out_len, stream);
}

int flacenc_hip_encode_pack_frames_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                         const int32_t* frames, size_t n_frames, uint32_t channels,
                                         uint32_t block_size, size_t stride, uint32_t bits_per_sample,
                                         uint32_t sample_rate, uint32_t first_frame_number,
                                         uint32_t frame_number_step, flacenc_hip_channel_result* results,
                                         uint8_t* out, size_t out_stride, uint32_t* out_len, void* stream) {
  if (!h || !cfg) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return FLACENC_HIP_OK;
  if (channels < 1 || channels > 8) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  // residual rows are only an intermediate here: handle scratch
  int rc;
  const size_t cstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  if ((rc = ensure(h, h->d_presid, n_frames * channels * cstride * 4)) != FLACENC_HIP_OK) return rc;
  rc = flacenc_hip_encode_frames_async(h, cfg, frames, n_frames, channels, block_size, stride, bits_per_sample, results,
                                       static_cast<int32_t*>(h->d_presid.ptr), cstride, stream);
  if (rc != FLACENC_HIP_OK) return rc;
  return flacenc_hip_pack_frames_async(h, frames, n_frames, channels, block_size, stride, results,
                                       static_cast<const int32_t*>(h->d_presid.ptr), cstride, bits_per_sample,
                                       sample_rate, first_frame_number, frame_number_step, out, out_stride, out_len,
                                       stream);
}

void* flacenc_hip_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void flacenc_hip_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int flacenc_hip_set_host_threads(flacenc_hip_handle* h, int threads) {
  if (!h || threads < 0 || threads > 64) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  h->host_threads = threads;
  h->copy_pool.reset();  // rebuilt with the new size at the next pageable call
  return FLACENC_HIP_OK;
}

namespace {
bool is_pinned(const void* p) {
  hipPointerAttribute_t attr{};
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // plain malloc memory: not an error for the caller
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

int ensure_pinned(flacenc_hip_handle* h, void** slot, size_t* cap_field, size_t bytes) {
  if (bytes <= *cap_field && slot[0] && slot[1]) return FLACENC_HIP_OK;
  for (int i = 0; i < 2; ++i) {
    if (slot[i]) HIP_TRY(h, hipHostFree(slot[i]));
    slot[i] = nullptr;
  }
  *cap_field = 0;
  const size_t want = bytes + bytes / 8 + 4096;
  for (int i = 0; i < 2; ++i) HIP_TRY(h, hipHostMalloc(&slot[i], want, hipHostMallocDefault));
  *cap_field = want;
  return FLACENC_HIP_OK;
}
}  // namespace

int flacenc_hip_encode_pcm_stereo(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg, const uint8_t* pcm,
                                  uint64_t total_samples, uint32_t bytes_per_sample, uint32_t bits_per_sample,
                                  uint32_t block_size, uint32_t sample_rate, uint32_t first_frame_number,
                                  uint32_t frame_number_step, uint8_t* out, size_t out_capacity, uint32_t* out_len,
                                  uint64_t* out_total) {
  return flacenc_hip_encode_pcm(h, cfg, pcm, total_samples, 2, bytes_per_sample, bits_per_sample, block_size,
                                sample_rate, first_frame_number, frame_number_step, out, out_capacity, out_len, out_total);
}

int flacenc_hip_encode_pcm(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg, const uint8_t* pcm,
                           uint64_t total_samples, uint32_t channels, uint32_t bytes_per_sample,
                           uint32_t bits_per_sample, uint32_t block_size, uint32_t sample_rate,
                           uint32_t first_frame_number, uint32_t frame_number_step, uint8_t* out, size_t out_capacity,
                           uint32_t* out_len, uint64_t* out_total) {
  if (!h || !cfg || !out_total) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  *out_total = 0;
  if (total_samples == 0) return FLACENC_HIP_OK;
  if (!pcm || !out || !out_len || channels < 1 || channels > 8 || bytes_per_sample < 1 || bytes_per_sample > 4 ||
      block_size < FLACENC_HIP_MIN_BLOCK_SIZE || block_size > FLACENC_HIP_MAX_BLOCK_SIZE) {
    h->last_error = "encode_pcm: null pointer, channels not in 1..=8, bytes_per_sample not in 1..=4 or block_size "
                    "not in 64..=32767";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  const bool stereo = channels == 2;
  const uint64_t n_full = total_samples / block_size;
  const uint32_t tail = static_cast<uint32_t>(total_samples % block_size);
  // (a last block shorter than MIN_BLOCK_SIZE_FOR_PREDICTION is a frame like any other: encode_subframe skips
  // its predictors, coding.rs:396, and the frame-level calls below do the same)
  HIP_TRY(h, hipSetDevice(h->device));
  if (!h->s_in) {
    HIP_TRY(h, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    HIP_TRY(h, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(h, hipEventCreateWithFlags(&h->ev_h2d[i], hipEventDisableTiming));
      HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fill[i], hipEventDisableTiming));
      HIP_TRY(h, hipEventCreateWithFlags(&h->ev_pack[i], hipEventDisableTiming));
      HIP_TRY(h, hipEventCreateWithFlags(&h->ev_d2h[i], hipEventDisableTiming));
    }
  }
  // chunks of whole frames: big enough to run the kernels at full occupancy (>= 768 workgroups),
  // small enough that two slots of staging stay modest and the pipeline has several stages in flight
  const size_t frame_in_bytes = static_cast<size_t>(block_size) * channels * bytes_per_sample;
  size_t chunk = (48u << 20) / frame_in_bytes;
  chunk = chunk < 768 ? 768 : (chunk > 8192 ? 8192 : chunk);
  // never more than the call has: staging, device buffers and the candidates' scratch are all sized from it
  // (a one-frame call of 8 channels x 32767 samples would otherwise pin gigabytes)
  if (chunk > n_full) chunk = n_full ? static_cast<size_t>(n_full) : 1;
  const size_t bound = stereo ? flacenc_hip_stereo_frame_bytes_bound(block_size, bits_per_sample)
                              : flacenc_hip_frame_bytes_bound(channels, block_size, bits_per_sample);
  const size_t ostride = (bound + 15) & ~static_cast<size_t>(15);
  const bool in_pinned = is_pinned(pcm), out_pinned = is_pinned(out);
  int rc;
  const size_t dstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  if (!in_pinned && (rc = ensure_pinned(h, h->pin_in, &h->pin_in_cap, chunk * frame_in_bytes)) != FLACENC_HIP_OK) return rc;
  if (!out_pinned && (rc = ensure_pinned(h, h->pin_out, &h->pin_out_cap, chunk * ostride)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure_pinned(h, h->pin_meta, &h->pin_meta_cap, chunk * 4 + 16)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_samples, chunk * channels * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_results, chunk * (stereo ? sizeof(flacenc_hip_stereo_frame_result)
                                                     : channels * sizeof(flacenc_hip_channel_result)))) != FLACENC_HIP_OK)
    return rc;
  for (int i = 0; i < 2; ++i) {
    if ((rc = ensure(h, h->d_pcm[i], chunk * frame_in_bytes + 16)) != FLACENC_HIP_OK) return rc;
    if ((rc = ensure(h, h->d_pack[i], chunk * ostride)) != FLACENC_HIP_OK) return rc;
    if ((rc = ensure(h, h->d_plen[i], chunk * 4 + 16)) != FLACENC_HIP_OK) return rc;   // lengths, then the total
    if ((rc = ensure(h, h->d_poff[i], chunk * 16 + 16)) != FLACENC_HIP_OK) return rc;  // src + dst offsets
    if ((rc = ensure(h, h->d_cont[i], chunk * ostride)) != FLACENC_HIP_OK) return rc;
  }

  struct Chunk {
    uint64_t first_frame;  // index within this call
    size_t frames;
    uint32_t n;            // block size of its frames
  };
  std::vector<Chunk> chunks;
  for (uint64_t f = 0; f < n_full; f += chunk)
    chunks.push_back({f, static_cast<size_t>(n_full - f < chunk ? n_full - f : chunk), block_size});
  if (tail) chunks.push_back({n_full, 1, tail});

  // staging copies for pageable caller memory run on the caller's thread + the handle's helper threads
  if ((!in_pinned || !out_pinned) && !h->copy_pool) {
    const int want = h->host_threads < 0 ? 4 : h->host_threads;  // total, the caller's thread included
    // (thread creation can throw std::system_error, vector growth std::bad_alloc: nothing unwinds across the ABI)
    try {
      h->copy_pool.reset(new CopyPool(want > 1 ? static_cast<unsigned>(want - 1) : 0u));
    } catch (...) {
      h->copy_pool.reset();
    }
    if (!h->copy_pool) {
      try {
        h->copy_pool.reset(new CopyPool(0u));  // no helper threads: plain memcpy on the caller's thread
      } catch (...) {
        h->last_error = "encode_pcm: out of host memory";
        return FLACENC_HIP_ERR_DEVICE;
      }
    }
  }
  uint64_t written = 0;
  // A chunk's way out has two steps so that the host never idles on a transfer: start_out waits for the
  // chunk's lengths and starts the device -> host copy of exactly its bytes (contiguous on the device
  // already); finish_out -- one chunk later, after the next chunk's staging copy in -- waits for that
  // transfer and hands the bytes to the caller.
  struct Pending {
    uint64_t at = 0, bytes = 0;
  } pending[2];
  auto start_out = [&](size_t ci) -> int {
    const Chunk& c = chunks[ci];
    const int s = static_cast<int>(ci & 1);
    HIP_TRY(h, hipEventSynchronize(h->ev_pack[s]));  // lengths + total are in pin_meta[s]
    const uint32_t* lens = static_cast<const uint32_t*>(h->pin_meta[s]);
    uint64_t bytes = 0;
    for (size_t f = 0; f < c.frames; ++f) bytes += lens[f];
    if (written + bytes > out_capacity) {
      h->last_error = "encode_pcm: out_capacity too small";
      return FLACENC_HIP_ERR_BAD_ARGUMENT;
    }
    std::memcpy(out_len + c.first_frame, lens, c.frames * 4);
    HIP_TRY(h, hipMemcpyAsync(out_pinned ? static_cast<void*>(out + written) : h->pin_out[s], h->d_cont[s].ptr, bytes,
                              hipMemcpyDeviceToHost, h->s_out));
    HIP_TRY(h, hipEventRecord(h->ev_d2h[s], h->s_out));
    pending[s].at = written;
    pending[s].bytes = bytes;
    written += bytes;
    return FLACENC_HIP_OK;
  };
  auto finish_out = [&](size_t ci) -> int {
    if (out_pinned) return FLACENC_HIP_OK;
    const int s = static_cast<int>(ci & 1);
    HIP_TRY(h, hipEventSynchronize(h->ev_d2h[s]));
    h->copy_pool->copy(out + pending[s].at, h->pin_out[s], pending[s].bytes);
    return FLACENC_HIP_OK;
  };

  // an error half way leaves work in flight on three streams: drain them before handing the handle back
  struct Drain {
    flacenc_hip_handle* h;
    bool armed = true;
    ~Drain() {
      if (!armed) return;
      (void)hipStreamSynchronize(h->s_in);
      (void)hipStreamSynchronize(h->stream);
      (void)hipStreamSynchronize(h->s_out);
    }
  } drain_on_error{h};
  for (size_t ci = 0; ci < chunks.size(); ++ci) {
    const Chunk& c = chunks[ci];
    const int s = static_cast<int>(ci & 1);
    const size_t in_bytes = c.frames * static_cast<size_t>(c.n) * channels * bytes_per_sample;
    const uint8_t* src = pcm + c.first_frame * frame_in_bytes;
    // 1. host -> device: packed PCM (2..3 bytes per sample instead of 4)
    if (ci >= 2) HIP_TRY(h, hipStreamWaitEvent(h->s_in, h->ev_fill[s], 0));  // d_pcm[s] has been consumed
    if (in_pinned) {
      HIP_TRY(h, hipMemcpyAsync(h->d_pcm[s].ptr, src, in_bytes, hipMemcpyHostToDevice, h->s_in));
    } else {
      if (ci >= 2) HIP_TRY(h, hipEventSynchronize(h->ev_h2d[s]));  // pin_in[s] has been sent
      h->copy_pool->copy(h->pin_in[s], src, in_bytes);
      HIP_TRY(h, hipMemcpyAsync(h->d_pcm[s].ptr, h->pin_in[s], in_bytes, hipMemcpyHostToDevice, h->s_in));
    }
    HIP_TRY(h, hipEventRecord(h->ev_h2d[s], h->s_in));
    // 2. compute stream: widen + de-interleave, analyse + decide + Frame::write, compact
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_h2d[s], 0));
    const size_t cstride = (static_cast<size_t>(c.n) + 3) & ~static_cast<size_t>(3);
    rc = flacenc_hip_fill_le_bytes_async(h, static_cast<const uint8_t*>(h->d_pcm[s].ptr),
                                         static_cast<uint64_t>(c.frames) * c.n, channels, bytes_per_sample, c.frames, c.n,
                                         static_cast<int32_t*>(h->d_samples.ptr), cstride, h->stream);
    if (rc != FLACENC_HIP_OK) return rc;
    HIP_TRY(h, hipEventRecord(h->ev_fill[s], h->stream));
    if (ci >= 2) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_d2h[s], 0));  // d_cont[s] has been copied out
    const size_t cbound = ((stereo ? flacenc_hip_stereo_frame_bytes_bound(c.n, bits_per_sample)
                                   : flacenc_hip_frame_bytes_bound(channels, c.n, bits_per_sample)) + 15) & ~static_cast<size_t>(15);
    uint32_t* dlen = static_cast<uint32_t*>(h->d_plen[s].ptr);
    const uint32_t number = first_frame_number + static_cast<uint32_t>(c.first_frame) * frame_number_step;
    if (stereo) {
      rc = flacenc_hip_encode_pack_stereo_frames_async(
          h, cfg, static_cast<const int32_t*>(h->d_samples.ptr), c.frames, c.n, cstride, bits_per_sample, sample_rate,
          number, frame_number_step, static_cast<flacenc_hip_stereo_frame_result*>(h->d_results.ptr),
          static_cast<uint8_t*>(h->d_pack[s].ptr), cbound, dlen, h->stream);
    } else {  // Independent(channels) frames, src/coding.rs:537-541
      rc = flacenc_hip_encode_pack_frames_async(
          h, cfg, static_cast<const int32_t*>(h->d_samples.ptr), c.frames, channels, c.n, cstride, bits_per_sample,
          sample_rate, number, frame_number_step, static_cast<flacenc_hip_channel_result*>(h->d_results.ptr),
          static_cast<uint8_t*>(h->d_pack[s].ptr), cbound, dlen, h->stream);
    }
    if (rc != FLACENC_HIP_OK) return rc;
    uint64_t* soff = static_cast<uint64_t*>(h->d_poff[s].ptr);
    uint64_t* doff = soff + c.frames;
    uint64_t* dtotal = reinterpret_cast<uint64_t*>(dlen + ((c.frames + 1) & ~static_cast<size_t>(1)));
    HIP_TRY(h, flacenc_hip::launch_frame_offsets(dlen, static_cast<uint32_t>(c.frames), cbound, soff, doff, dtotal, h->stream));
    HIP_TRY(h, flacenc_hip::launch_place_frames(static_cast<const uint8_t*>(h->d_pack[s].ptr), soff, dlen,
                                                static_cast<uint8_t*>(h->d_cont[s].ptr), doff,
                                                static_cast<uint32_t>(c.frames), h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->pin_meta[s], dlen, c.frames * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipEventRecord(h->ev_pack[s], h->stream));
    // 3. while this chunk runs: the chunk before the previous one reaches the caller (its transfer ran
    // during this chunk's staging copy), then the previous one's transfer starts
    if (ci >= 2 && (rc = finish_out(ci - 2)) != FLACENC_HIP_OK) return rc;
    if (ci >= 1 && (rc = start_out(ci - 1)) != FLACENC_HIP_OK) return rc;
  }
  const size_t nc = chunks.size();
  if (nc >= 2 && (rc = finish_out(nc - 2)) != FLACENC_HIP_OK) return rc;
  if ((rc = start_out(nc - 1)) != FLACENC_HIP_OK) return rc;
  if ((rc = finish_out(nc - 1)) != FLACENC_HIP_OK) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->s_out));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  drain_on_error.armed = false;
  *out_total = written;
  return FLACENC_HIP_OK;
}

int flacenc_hip_synchronize(flacenc_hip_handle* h) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return FLACENC_HIP_OK;
}

int flacenc_hip_qlpc_batch_async(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg,
                                 const int32_t* samples, size_t n_subframes, uint32_t block_size,
                                 size_t stride, const uint8_t* bps,
                                 flacenc_hip_subframe_params* params, int32_t* residual,
                                 size_t residual_stride, double* autocorr, double* lpc_coefs,
                                 void* stream) {
  int rc = check_batch_args(h, cfg, samples, n_subframes, block_size, stride, params, residual,
                            residual_stride);
  if (rc != FLACENC_HIP_OK || n_subframes == 0) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = static_cast<hipStream_t>(stream);  // NULL = HIP's default (null) stream
  return enqueue(h, cfg, samples, n_subframes, block_size, stride, bps, params, residual,
                 residual_stride, autocorr, lpc_coefs, s);
}

int flacenc_hip_qlpc_batch(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg,
                           const int32_t* samples, size_t n_subframes, uint32_t block_size,
                           size_t stride, const uint8_t* bps,
                           flacenc_hip_subframe_params* params, int32_t* residual,
                           size_t residual_stride, double* autocorr, double* lpc_coefs,
                           int memory_kind) {
  int rc = check_batch_args(h, cfg, samples, n_subframes, block_size, stride, params, residual,
                            residual_stride);
  if (rc != FLACENC_HIP_OK || n_subframes == 0) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  if (memory_kind == FLACENC_HIP_MEM_DEVICE) {
    rc = enqueue(h, cfg, samples, n_subframes, block_size, stride, bps, params, residual,
                 residual_stride, autocorr, lpc_coefs, h->stream);
    if (rc != FLACENC_HIP_OK) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FLACENC_HIP_OK;
  }
  if (memory_kind != FLACENC_HIP_MEM_HOST) return FLACENC_HIP_ERR_BAD_ARGUMENT;

  // host pointers: stage through the handle's device scratch (PCIe both ways)
  const size_t dstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  if ((rc = ensure(h, h->d_samples, n_subframes * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_residual, n_subframes * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_params, n_subframes * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK)
    return rc;
  if (bps && (rc = ensure(h, h->d_bps, n_subframes)) != FLACENC_HIP_OK) return rc;
  if (autocorr && (rc = ensure(h, h->d_autocorr, n_subframes * 33 * 8)) != FLACENC_HIP_OK) return rc;
  if (lpc_coefs && (rc = ensure(h, h->d_lpc, n_subframes * 32 * 8)) != FLACENC_HIP_OK) return rc;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpy2DAsync(h->d_samples.ptr, dstride * 4, samples, stride * 4,
                              static_cast<size_t>(block_size) * 4, n_subframes,
                              hipMemcpyHostToDevice, s));
  if (bps) HIP_TRY(h, hipMemcpyAsync(h->d_bps.ptr, bps, n_subframes, hipMemcpyHostToDevice, s));
  rc = enqueue(h, cfg, static_cast<const int32_t*>(h->d_samples.ptr), n_subframes, block_size, dstride,
               bps ? static_cast<const uint8_t*>(h->d_bps.ptr) : nullptr,
               static_cast<flacenc_hip_subframe_params*>(h->d_params.ptr),
               static_cast<int32_t*>(h->d_residual.ptr), dstride,
               autocorr ? static_cast<double*>(h->d_autocorr.ptr) : nullptr,
               lpc_coefs ? static_cast<double*>(h->d_lpc.ptr) : nullptr, s);
  if (rc != FLACENC_HIP_OK) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(residual, residual_stride * 4, h->d_residual.ptr, dstride * 4,
                              static_cast<size_t>(block_size) * 4, n_subframes,
                              hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(params, h->d_params.ptr, n_subframes * sizeof(flacenc_hip_subframe_params),
                            hipMemcpyDeviceToHost, s));
  if (autocorr)
    HIP_TRY(h, hipMemcpyAsync(autocorr, h->d_autocorr.ptr, n_subframes * 33 * 8, hipMemcpyDeviceToHost, s));
  if (lpc_coefs)
    HIP_TRY(h, hipMemcpyAsync(lpc_coefs, h->d_lpc.ptr, n_subframes * 32 * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return FLACENC_HIP_OK;
}

int flacenc_hip_stereo_qlpc_batch_async(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg,
                                        const int32_t* frames, size_t n_frames, uint32_t block_size,
                                        size_t stride, uint32_t bits_per_sample,
                                        flacenc_hip_subframe_params* params, int32_t* residual,
                                        size_t residual_stride, void* stream) {
  int rc = check_batch_args(h, cfg, frames, n_frames * 4, block_size, stride, params, residual,
                            residual_stride);
  if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
  if (bits_per_sample < 8 || bits_per_sample > 24) {
    h->last_error = "bits_per_sample must be in 8..=24";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = static_cast<hipStream_t>(stream);  // NULL = HIP's default (null) stream
  return enqueue(h, cfg, frames, n_frames * 4, block_size, stride, nullptr, params, residual,
                 residual_stride, nullptr, nullptr, s, true, bits_per_sample);
}

int flacenc_hip_stereo_qlpc_batch(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg,
                                  const int32_t* frames, size_t n_frames, uint32_t block_size,
                                  size_t stride, uint32_t bits_per_sample,
                                  flacenc_hip_subframe_params* params, int32_t* residual,
                                  size_t residual_stride, int memory_kind) {
  if (memory_kind == FLACENC_HIP_MEM_DEVICE) {
    if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
    int rc = flacenc_hip_stereo_qlpc_batch_async(h, cfg, frames, n_frames, block_size, stride,
                                                 bits_per_sample, params, residual, residual_stride,
                                                 h->stream);
    if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FLACENC_HIP_OK;
  }
  if (memory_kind != FLACENC_HIP_MEM_HOST) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  int rc = check_batch_args(h, cfg, frames, n_frames * 4, block_size, stride, params, residual,
                            residual_stride);
  if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t dstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  const size_t n_sub = n_frames * 4;
  if ((rc = ensure(h, h->d_samples, n_frames * 2 * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_residual, n_sub * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_params, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK)
    return rc;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpy2DAsync(h->d_samples.ptr, dstride * 4, frames, stride * 4,
                              static_cast<size_t>(block_size) * 4, n_frames * 2,
                              hipMemcpyHostToDevice, s));
  rc = flacenc_hip_stereo_qlpc_batch_async(h, cfg, static_cast<const int32_t*>(h->d_samples.ptr),
                                           n_frames, block_size, dstride, bits_per_sample,
                                           static_cast<flacenc_hip_subframe_params*>(h->d_params.ptr),
                                           static_cast<int32_t*>(h->d_residual.ptr), dstride, s);
  if (rc != FLACENC_HIP_OK) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(residual, residual_stride * 4, h->d_residual.ptr, dstride * 4,
                              static_cast<size_t>(block_size) * 4, n_sub, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(params, h->d_params.ptr, n_sub * sizeof(flacenc_hip_subframe_params),
                            hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return FLACENC_HIP_OK;
}

int flacenc_hip_encode_stereo_frames_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                           const int32_t* frames, size_t n_frames, uint32_t block_size,
                                           size_t stride, uint32_t bits_per_sample,
                                           flacenc_hip_stereo_frame_result* results, int32_t* residual,
                                           size_t residual_stride, void* stream) {
  return encode_stereo_frames_impl(h, cfg, frames, n_frames, block_size, stride, bits_per_sample, results, residual,
                                   residual_stride, stream, nullptr, nullptr);
}

// `pack` non-null: if the launch can run as the fused kernel with the bit writer, the frames are
// packed there (*packed = true, `residual` untouched); otherwise *packed = false and the call
// behaves like flacenc_hip_encode_stereo_frames_async (the caller packs in a second launch).
static int encode_stereo_frames_impl(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                     const int32_t* frames, size_t n_frames, uint32_t block_size, size_t stride,
                                     uint32_t bits_per_sample, flacenc_hip_stereo_frame_result* results,
                                     int32_t* residual, size_t residual_stride, void* stream,
                                     const PackTarget* pack, bool* packed) {
  if (packed) *packed = false;
  if (!h || !cfg || (!results && n_frames)) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  int rc = check_batch_args(h, &cfg->qlpc, frames, n_frames * 4, block_size, stride,
                            reinterpret_cast<flacenc_hip_subframe_params*>(results), residual, residual_stride, 1);
  if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
  if (bits_per_sample < 8 || bits_per_sample > 24) {
    h->last_error = "bits_per_sample must be in 8..=24";
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  // too_short (coding.rs:396): neither fixed_lpc nor estimated_qlpc is tried; Constant or Verbatim
  flacenc_hip_frame_config short_cfg;
  if (block_size < FLACENC_HIP_MIN_BLOCK_SIZE) {
    short_cfg = *cfg;
    short_cfg.use_fixed = short_cfg.use_lpc = 0;
    cfg = &short_cfg;
  }
  uint32_t fixed_group_log2 = 0;
  bool fixed_composite = false;
  if (cfg->use_fixed) {
    if ((rc = verify_fixed(h, cfg)) != FLACENC_HIP_OK) return rc;
    if (cfg->fixed_order_sel == FLACENC_HIP_ORDERSEL_APPROXENT) {
      // the estimator's partitions must be whole groups of 64-sample lanes
      const uint32_t p = cfg->fixed_partitions;
      if ((p & (p - 1)) != 0) fixed_composite = true;  // partitions not whole lane groups: general path
      uint32_t lanes = fixed_composite ? 1u : 64u / p;
      while (lanes > 1) {
        ++fixed_group_log2;
        lanes >>= 1;
      }
    }
  }
  HIP_TRY(h, hipSetDevice(h->device));
  const WindowEntry* win = nullptr;
  static const WindowEntry no_window{};
  if (block_size >= FLACENC_HIP_MIN_BLOCK_SIZE) {
    rc = get_window(h, &cfg->qlpc, block_size, &win);
    if (rc != FLACENC_HIP_OK) return rc;
  } else {
    win = &no_window;  // (never read: short blocks take the candidate-free general path below)
  }
  flacenc_hip::QlpcKernelArgs a;
  a.samples = frames;
  a.stride = stride;
  a.block_size = block_size;
  a.n_subframes = static_cast<uint32_t>(n_frames * 4);
  a.bps = nullptr;
  a.bps_uniform = bits_per_sample;
  a.stereo = 1;
  a.window = win->dev;
  a.flat_lo = win->flat_lo;
  a.flat_hi = win->flat_hi;
  a.lpc_order = cfg->qlpc.lpc_order;
  a.precision = cfg->qlpc.quant_precision;
  a.max_rice_parameter = cfg->qlpc.max_rice_parameter;
  a.rice_finest_only = (cfg->qlpc.flags & FLACENC_HIP_FLAG_FINEST_RICE_ORDER) ? 1u : 0u;
  a.force_generic = (cfg->qlpc.flags & FLACENC_HIP_FLAG_GENERIC_KERNEL) ? 1u : 0u;
  a.reference_order = sum_order_mode(cfg->qlpc.flags);
  set_certify(h, a, cfg->qlpc.flags);
  a.direct_mse = cfg->qlpc.use_direct_mse ? 1u : 0u;  // (keeps the launch off the fused wave kernel)
  a.acorr_in = nullptr;
  a.only_marked = 0;
  a.params = nullptr;
  a.residual = residual;
  a.residual_stride = residual_stride;
  a.autocorr = nullptr;
  a.lpc_coefs = nullptr;
  a.table_scratch = nullptr;
  a.stamps = h->stamps;
  a.frame_results = results;
  a.chan_results = nullptr;
  a.use_constant = cfg->use_constant;
  a.use_lpc = cfg->use_lpc;
  a.use_leftside = cfg->use_leftside;
  a.use_rightside = cfg->use_rightside;
  a.use_midside = cfg->use_midside;
  a.use_fixed = cfg->use_fixed;
  a.fixed_max_order = cfg->fixed_max_order;
  a.fixed_order_sel = cfg->fixed_order_sel;
  a.fixed_group_log2 = fixed_group_log2;
  a.fixed_keys = h->fixed_keys;
  a.fixed_mode = a.fixed_partitions = a.forced_uniform = 0;
  a.forced_orders = nullptr;
  a.selector_keys = nullptr;
  a.lpc_stage = 0;
  a.pack_out = nullptr;
  a.pred = nullptr;
  a.pred_out = nullptr;
  a.split_scratch = nullptr;
  if (a.reference_order) {  // R[] of the reference-order pass
    int rc2 = attach_split_scratch(h, a);
    if (rc2 != FLACENC_HIP_OK) return rc2;
    rc2 = attach_sumabs_scratch(h, a, a.use_fixed && a.fixed_order_sel == FLACENC_HIP_ORDERSEL_APPROXENT);
    if (rc2 != FLACENC_HIP_OK) return rc2;
  }
  // One kernel or two?  Measured on MI355X (24576 frames, order 8): with the fixed-LPC candidate the fused
  // bit writer takes 1.80 ms against 1.13 + 0.63 ms for the deciding kernel followed by the stand-alone
  // packer; without it 1.62 against 0.72 + 0.63 ms.  The deciding kernels run at three workgroups per CU,
  // the fused one -- whose packing tail keeps two of four waves busy -- fits only two, so the two-launch
  // form is the default; FLACENC_HIP_FLAG_FUSED_PACK / _TWO_STAGE_PACK in cfg->qlpc.flags override it.
  bool want_fused = false;
  if (cfg->qlpc.flags & FLACENC_HIP_FLAG_FUSED_PACK) want_fused = true;
  if (cfg->qlpc.flags & FLACENC_HIP_FLAG_TWO_STAGE_PACK) want_fused = false;
  if (pack && want_fused && !fixed_composite && block_size == 4096 && flacenc_hip::wave_kernel_eligible(a)) {
    const size_t bound = flacenc_hip_stereo_frame_bytes_bound(block_size, bits_per_sample);
    flacenc_hip::FramePackArgs pa{};
    fill_header_specs(pa, block_size, pack->sample_rate, bits_per_sample);
    a.pack_out = pack->out;
    a.pack_out_stride = pack->out_stride;
    a.pack_out_len = pack->out_len;
    a.pack_header_mid = pa.header_mid;
    a.pack_extra_len = pa.extra_len;
    for (int i = 0; i < 4; ++i) a.pack_extra[i] = pa.extra[i];
    a.pack_first_frame = pack->first_frame_number;
    a.pack_frame_step = pack->frame_number_step;
    a.pack_lds_words = static_cast<uint32_t>(bound / 4 + 4);
    fill_crc_powers(a.pack_lds_words, &a.pack_crc_per, a.pack_crc_pow);
    if (static_cast<size_t>(a.pack_lds_words) * 4 <= 35424) {  // the bit buffer reuses the two channel images
      if (packed) *packed = true;
    } else {
      a.pack_out = nullptr;
    }
  }
  if (a.split_scratch == nullptr && certify_needs_scratch(a)) {  // (the fused bit writer is handed the reference's R[])
    int rc2 = attach_split_scratch(h, a);
    if (rc2 != FLACENC_HIP_OK) return rc2;
  }
  if (!flacenc_hip::wave_kernel_eligible(a) || fixed_composite) {
    // General shapes: the same result from candidate batches (4 QLPC + 4 fixed-LPC candidates per
    // frame in handle scratch) and the stand-alone controller kernel (frame_decide.cpp).
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n_sub = n_frames * 4;
    const size_t cstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
    flacenc_hip::FrameDecideArgs d{};
    d.frames = frames;
    d.stride = stride;
    d.block_size = block_size;
    d.n_frames = static_cast<uint32_t>(n_frames);
    d.bits_per_sample = bits_per_sample;
    d.use_constant = cfg->use_constant;
    d.use_fixed = cfg->use_fixed;
    d.use_lpc = cfg->use_lpc;
    d.use_leftside = cfg->use_leftside;
    d.use_rightside = cfg->use_rightside;
    d.use_midside = cfg->use_midside;
    d.cand_stride = cstride;
    d.results = results;
    d.residual = residual;
    d.residual_stride = residual_stride;
    // Big-block shapes: two analyse-only passes (QLPC, fixed_lpc: records, no residual rows), then one kernel that
    // decides and writes only the two rows the frame keeps (bigblock_residual_kernel, modes 1 and 2) -- eight candidate
    // rows per frame stay off HBM, and so does the copy of the chosen two.
    const bool big_shape =
        cfg->use_lpc && (block_size == 8192 || block_size == 16384 || (block_size == 4096 && cfg->qlpc.lpc_order >= 13)) &&
        !(cfg->qlpc.flags & FLACENC_HIP_FLAG_GENERIC_KERNEL) && !cfg->qlpc.use_direct_mse &&
        (reinterpret_cast<uintptr_t>(frames) & 15) == 0 && (stride & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(residual) & 15) == 0 && (residual_stride & 3) == 0;
    const uint32_t fparts = cfg->fixed_partitions;
    const bool fixed_big = !cfg->use_fixed ||
                           (cfg->fixed_order_sel == FLACENC_HIP_ORDERSEL_APPROXENT && fparts != 0 && (fparts & (fparts - 1)) == 0 &&
                            block_size / fparts >= 64 && block_size / fparts <= 4096 && cfg->fixed_max_order <= 4);
    if (big_shape && fixed_big) {
      if ((rc = ensure(h, h->d_cparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
      if ((rc = ensure(h, h->d_cresid, n_sub * cstride * 4)) != FLACENC_HIP_OK) return rc;
      if ((rc = ensure(h, h->d_minmax, n_sub * 2 * sizeof(int32_t))) != FLACENC_HIP_OK) return rc;
      bool analysed = false;
      rc = enqueue(h, &cfg->qlpc, frames, n_sub, block_size, stride, nullptr,
                   static_cast<flacenc_hip_subframe_params*>(h->d_cparams.ptr), static_cast<int32_t*>(h->d_cresid.ptr),
                   cstride, nullptr, nullptr, s, true, bits_per_sample, nullptr, 0,
                   static_cast<int32_t*>(h->d_minmax.ptr), &analysed, 1u);
      if (rc != FLACENC_HIP_OK) return rc;
      if (!analysed) {
        h->last_error = "internal: analyse-only QLPC batch on a shape the big-block kernels do not take";
        return FLACENC_HIP_ERR_UNSUPPORTED;
      }
      flacenc_hip::QlpcKernelArgs m = a;  // (shape, width and the encode_frame switches are set above)
      m.frame_results = results;
      m.residual = residual;
      m.residual_stride = residual_stride;
      m.residual_mode = 2u;
      m.stamps = nullptr;
      m.cand_lpc_params = static_cast<const flacenc_hip_subframe_params*>(h->d_cparams.ptr);
      m.cand_lpc_rows = static_cast<const int32_t*>(h->d_cresid.ptr);
      m.cand_minmax = static_cast<const int32_t*>(h->d_minmax.ptr);
      m.cand_stride = cstride;
      if (cfg->use_fixed) {
        if ((rc = ensure(h, h->d_fparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
        if ((rc = ensure(h, h->d_fresid, n_sub * cstride * 4)) != FLACENC_HIP_OK) return rc;
        if ((rc = ensure(h, h->d_fkeys, n_sub * 8)) != FLACENC_HIP_OK) return rc;
        rc = enqueue_fixed(h, cfg, frames, n_sub, block_size, stride, nullptr, bits_per_sample, true,
                           static_cast<flacenc_hip_subframe_params*>(h->d_fparams.ptr),
                           static_cast<int32_t*>(h->d_fresid.ptr), cstride,
                           static_cast<unsigned long long*>(h->d_fkeys.ptr), s, 1u);
        if (rc != FLACENC_HIP_OK) return rc;
        m.cand_fixed_params = static_cast<const flacenc_hip_subframe_params*>(h->d_fparams.ptr);
        m.cand_fixed_rows = static_cast<const int32_t*>(h->d_fresid.ptr);
        m.cand_fixed_keys = static_cast<const unsigned long long*>(h->d_fkeys.ptr);
      }
      HIP_TRY(h, flacenc_hip::launch_bigblock_residual(m, s));
      return FLACENC_HIP_OK;
    }
    // Blocks of 8 / 16 / 32 finest Rice partitions (512 .. 2304 samples): qlpc_subwave_kernel's frame variant does the
    // whole of encode_frame -- both candidates of the four roles, the decision, the two chosen rows -- several frames
    // per workgroup.  A frame with a candidate beyond its exact sums (residuals of 2^25 and more) comes back marked
    // and takes the general path below, whose three kernels return at once when the count of marked frames is 0.
    if (block_size >= FLACENC_HIP_MIN_BLOCK_SIZE && flacenc_hip::subwave_shape(block_size)) {
      flacenc_hip::QlpcKernelArgs m = a;
      m.stamps = nullptr;
      m.fixed_partitions = cfg->fixed_partitions;
      // (the unflagged order on these shapes is the reference's: its chains for every QLPC candidate in front, QlpcKernelArgs::cert_subwave)
      m.cert_subwave = (a.certify != 0u && a.reference_order == 0u && !a.direct_mse && cfg->use_lpc && cfg->qlpc.lpc_order <= 12) ? 1u : 0u;
      if ((rc = attach_split_scratch(h, m)) != FLACENC_HIP_OK) return rc;
      m.marked_unit = 4;  // (the kernel lists marked FRAMES; the candidate clean-ups visit their four roles)
      if (flacenc_hip::subwave_frame_eligible(m)) {
        if ((rc = ensure(h, h->d_cparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
        if ((rc = ensure(h, h->d_cresid, n_sub * cstride * 4)) != FLACENC_HIP_OK) return rc;
        m.cand_lpc_params = static_cast<const flacenc_hip_subframe_params*>(h->d_cparams.ptr);
        if (cfg->use_fixed) {
          if ((rc = ensure(h, h->d_fparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
          if ((rc = ensure(h, h->d_fresid, n_sub * cstride * 4)) != FLACENC_HIP_OK) return rc;
          if ((rc = ensure(h, h->d_fkeys, n_sub * 8)) != FLACENC_HIP_OK) return rc;
          m.cand_fixed_params = static_cast<const flacenc_hip_subframe_params*>(h->d_fparams.ptr);
        }
        HIP_TRY(h, flacenc_hip::launch_subwave_frames(m, s));
        // the marked frames' candidates, by the generic kernel's clean-up launches (status -1 in the scratch records)
        flacenc_hip::QlpcKernelArgs c = m;
        c.frame_results = nullptr;
        c.cand_lpc_params = c.cand_fixed_params = nullptr;
        c.params = static_cast<flacenc_hip_subframe_params*>(h->d_cparams.ptr);
        c.residual = static_cast<int32_t*>(h->d_cresid.ptr);
        c.residual_stride = cstride;
        c.only_marked = 1;
        c.use_fixed = 0;
        c.fixed_keys = nullptr;
        HIP_TRY(h, flacenc_hip::launch_qlpc(c, flacenc_hip::plan_qlpc_launch(block_size, cfg->qlpc.lpc_order), s));
        d.lpc_params = c.params;
        d.lpc_residual = c.residual;
        if (cfg->use_fixed) {
          flacenc_hip::QlpcKernelArgs x = c;
          x.params = static_cast<flacenc_hip_subframe_params*>(h->d_fparams.ptr);
          x.residual = static_cast<int32_t*>(h->d_fresid.ptr);
          x.selector_keys = static_cast<unsigned long long*>(h->d_fkeys.ptr);
          x.window = nullptr;
          x.flat_lo = x.flat_hi = 0;
          x.lpc_order = 4;
          x.precision = 0;
          x.use_fixed = 1;
          x.fixed_mode = 1;
          HIP_TRY(h, flacenc_hip::launch_qlpc(x, flacenc_hip::plan_qlpc_launch(block_size, 4), s));
          d.fixed_params = x.params;
          d.fixed_residual = x.residual;
          d.fixed_keys = x.selector_keys;
        }
        d.only_marked = 1;
        d.marked_count = m.marked_count;
        d.marked_list = m.marked_list;  // (entries: frames)
        d.marked_cap = m.marked_cap;
        HIP_TRY(h, flacenc_hip::launch_frame_decide(d, s));
        return FLACENC_HIP_OK;
      }
    }
    if (cfg->use_lpc) {
      if ((rc = ensure(h, h->d_cparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
      if ((rc = ensure(h, h->d_cresid, n_sub * cstride * 4)) != FLACENC_HIP_OK) return rc;
      if ((rc = ensure(h, h->d_minmax, n_sub * 2 * sizeof(int32_t))) != FLACENC_HIP_OK) return rc;
      bool placed_lr = false;
      rc = enqueue(h, &cfg->qlpc, frames, n_sub, block_size, stride, nullptr,
                   static_cast<flacenc_hip_subframe_params*>(h->d_cparams.ptr),
                   static_cast<int32_t*>(h->d_cresid.ptr), cstride, nullptr, nullptr, s, true, bits_per_sample,
                   residual, residual_stride, static_cast<int32_t*>(h->d_minmax.ptr), &placed_lr);
      if (rc != FLACENC_HIP_OK) return rc;
      d.lpc_lr_in_place = placed_lr ? 1u : 0u;
      d.minmax = placed_lr ? static_cast<const int32_t*>(h->d_minmax.ptr) : nullptr;
      d.lpc_params = static_cast<const flacenc_hip_subframe_params*>(h->d_cparams.ptr);
      d.lpc_residual = static_cast<const int32_t*>(h->d_cresid.ptr);
    }
    if (cfg->use_fixed) {
      if ((rc = ensure(h, h->d_fparams, n_sub * sizeof(flacenc_hip_subframe_params))) != FLACENC_HIP_OK) return rc;
      if ((rc = ensure(h, h->d_fresid, n_sub * cstride * 4)) != FLACENC_HIP_OK) return rc;
      if ((rc = ensure(h, h->d_fkeys, n_sub * 8)) != FLACENC_HIP_OK) return rc;
      rc = enqueue_fixed(h, cfg, frames, n_sub, block_size, stride, nullptr, bits_per_sample, true,
                         static_cast<flacenc_hip_subframe_params*>(h->d_fparams.ptr),
                         static_cast<int32_t*>(h->d_fresid.ptr), cstride,
                         static_cast<unsigned long long*>(h->d_fkeys.ptr), s);
      if (rc != FLACENC_HIP_OK) return rc;
      d.fixed_params = static_cast<const flacenc_hip_subframe_params*>(h->d_fparams.ptr);
      d.fixed_residual = static_cast<const int32_t*>(h->d_fresid.ptr);
      d.fixed_keys = static_cast<const unsigned long long*>(h->d_fkeys.ptr);
    }
    HIP_TRY(h, flacenc_hip::launch_frame_decide(d, s));
    return FLACENC_HIP_OK;
  }
  flacenc_hip::QlpcLaunchPlan plan = flacenc_hip::plan_qlpc_launch(block_size, cfg->qlpc.lpc_order);
  return launch_adaptive(h, a, plan, static_cast<hipStream_t>(stream));
}

int flacenc_hip_encode_stereo_frames(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                     const int32_t* frames, size_t n_frames, uint32_t block_size,
                                     size_t stride, uint32_t bits_per_sample,
                                     flacenc_hip_stereo_frame_result* results, int32_t* residual,
                                     size_t residual_stride, int memory_kind) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (memory_kind == FLACENC_HIP_MEM_DEVICE) {
    int rc = flacenc_hip_encode_stereo_frames_async(h, cfg, frames, n_frames, block_size, stride,
                                                    bits_per_sample, results, residual, residual_stride,
                                                    h->stream);
    if (rc != FLACENC_HIP_OK || n_frames == 0) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FLACENC_HIP_OK;
  }
  if (memory_kind != FLACENC_HIP_MEM_HOST || !cfg) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  if (n_frames == 0) return flacenc_hip_verify_config(&cfg->qlpc);
  if (!frames || !results || !residual) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t dstride = (static_cast<size_t>(block_size) + 3) & ~static_cast<size_t>(3);
  int rc;
  if ((rc = ensure(h, h->d_samples, n_frames * 2 * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_residual, n_frames * 2 * dstride * 4)) != FLACENC_HIP_OK) return rc;
  if ((rc = ensure(h, h->d_params, n_frames * sizeof(flacenc_hip_stereo_frame_result))) != FLACENC_HIP_OK)
    return rc;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpy2DAsync(h->d_samples.ptr, dstride * 4, frames, stride * 4,
                              static_cast<size_t>(block_size) * 4, n_frames * 2, hipMemcpyHostToDevice, s));
  rc = flacenc_hip_encode_stereo_frames_async(h, cfg, static_cast<const int32_t*>(h->d_samples.ptr), n_frames,
                                              block_size, dstride, bits_per_sample,
                                              static_cast<flacenc_hip_stereo_frame_result*>(h->d_params.ptr),
                                              static_cast<int32_t*>(h->d_residual.ptr), dstride, s);
  if (rc != FLACENC_HIP_OK) return rc;
  HIP_TRY(h, hipMemcpy2DAsync(residual, residual_stride * 4, h->d_residual.ptr, dstride * 4,
                              static_cast<size_t>(block_size) * 4, n_frames * 2, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(results, h->d_params.ptr, n_frames * sizeof(flacenc_hip_stereo_frame_result),
                            hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return FLACENC_HIP_OK;
}

}  // extern "C"
