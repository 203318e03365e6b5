// qlpc_dispatch.cpp -- launch planning (workgroup size, LDS budget, order bucket)
// and dispatch to the per-bucket kernel instantiations.
#include "qlpc_kernel.h"

#include "acorr_reference.h"
#include "direct_mse.h"
#include "sumabs_reference.h"

namespace flacenc_hip {
namespace {

constexpr int kLeadRows = 2;
constexpr int kMiscCount = 16;

int bucket_order(int P) {
  if (P <= 8) return 8;
  if (P <= 10) return 10;
  if (P <= 12) return 12;
  if (P <= 16) return 16;
  if (P <= 24) return 24;
  return 32;
}

}  // namespace

bool wave_kernel_eligible(const QlpcKernelArgs& a) {
  if ((a.block_size != 4096 && a.block_size != 4608) || a.lpc_order > 12) return false;
  if (a.block_size == 4608 && a.pack_out != nullptr) return false;  // (the fused bit writer exists for 4096 only)
  if (a.fixed_mode != 0) return false;  // fixed_lpc as a stand-alone batch: generic kernel
  if (a.direct_mse) return false;       // experimental estimators: direct_mse_kernel + the split pipeline
  if (a.force_generic) return false;
  if ((reinterpret_cast<uintptr_t>(a.samples) & 15) || (a.stride & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(a.residual) & 15) || (a.residual_stride & 3)) return false;
  if (a.stereo && (a.n_subframes & 3)) return false;
  return true;
}

bool cert_shape(const QlpcKernelArgs& a) {
  return (a.block_size == 4096 || a.block_size == 4608) && a.lpc_order >= 1 && a.lpc_order <= 12;
}

bool subwave_shape(uint32_t n) {
  return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 288 || n == 576 || n == 1152 || n == 2304;
}

static bool subwave_buffers_ok(const QlpcKernelArgs& a) {
  if (!subwave_shape(a.block_size) || a.n_subframes == 0 || a.force_generic || a.only_marked || a.direct_mse) return false;
  if (a.sumabs_in != nullptr || a.lpc_stage != 0) return false;
  if ((reinterpret_cast<uintptr_t>(a.samples) & 15) || (a.stride & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(a.residual) & 15) || (a.residual_stride & 3)) return false;
  if (a.stereo && (a.n_subframes & 3)) return false;
  return a.marked_count != nullptr && a.pack_out == nullptr;
}

// the estimator's partitions: a whole number of the selector's sub-sums (an eighth of a lane's samples), and either
// 1 / 2 / 4 per lane or a power-of-two group of lanes inside the segment
static bool subwave_fixed_ok(const QlpcKernelArgs& a) {
  if (a.fixed_order_sel != 1u || a.fixed_max_order > 4u || a.fixed_partitions == 0u) return false;
  const uint32_t n = a.block_size, spl = (n % 72u) == 0 ? 72u : 64u, parts = a.fixed_partitions;
  if (n % parts) return false;
  const uint32_t psz = n / parts;
  if (psz < spl) return psz * 2 == spl || psz * 4 == spl;
  const uint32_t lanes = psz / spl;
  return psz % spl == 0 && (lanes & (lanes - 1)) == 0;
}

// the plain analysis (records + residual rows, no on-device decision) in the build's canonical summation order
bool subwave_eligible(const QlpcKernelArgs& a) {
  if (!subwave_buffers_ok(a) || a.lpc_order > 12 || a.lpc_order == 0 || a.fixed_mode != 0) return false;
  // (a flagged summation order reaches the kernel as R[] of the launch in front: acorr_in)
  if (a.reference_order != 0 && a.acorr_in == nullptr) return false;
  return a.frame_results == nullptr && a.chan_results == nullptr && a.params != nullptr;
}

bool subwave_fixed_eligible(const QlpcKernelArgs& a) {
  if (!subwave_buffers_ok(a) || a.fixed_mode != 1u || !subwave_fixed_ok(a)) return false;
  if (a.reference_order != 0 || a.acorr_in != nullptr) return false;
  return a.frame_results == nullptr && a.chan_results == nullptr && a.params != nullptr;
}

bool subwave_frame_eligible(const QlpcKernelArgs& a) {
  if (!subwave_buffers_ok(a) || !a.stereo || a.fixed_mode != 0 || a.frame_results == nullptr) return false;
  if (a.reference_order != 0 || a.acorr_in != nullptr) return false;  // (flagged orders: the selector's sums are the reference's too)
  if (!a.use_lpc || a.lpc_order > 12 || a.lpc_order == 0) return false;
  return !a.use_fixed || subwave_fixed_ok(a);
}

bool subwave_channels_eligible(const QlpcKernelArgs& a) {
  if (!subwave_buffers_ok(a) || a.stereo || a.fixed_mode != 0 || a.chan_results == nullptr) return false;
  if (a.reference_order != 0 || a.acorr_in != nullptr) return false;
  if (!a.use_lpc || a.lpc_order > 12 || a.lpc_order == 0) return false;
  return !a.use_fixed || subwave_fixed_ok(a);
}

// R[] of every subframe of `a` in the stable build's order (acorr_reference_mfma_kernel) into a.split_scratch
static hipError_t reference_chains_for_all(const QlpcKernelArgs& a, double* racc, hipStream_t stream) {
  AcorrRefArgs r{};
  r.samples = a.samples;
  r.stride = a.stride;
  r.block_size = a.block_size;
  r.n_subframes = a.n_subframes;
  r.stereo = a.stereo;
  r.window = a.window;
  r.lpc_order = a.lpc_order;
  r.nightly = 0u;
  r.out = racc;
  return launch_acorr_reference(r, stream);
}

hipError_t launch_subwave_frames(const QlpcKernelArgs& a, hipStream_t stream) {
  if (a.cert_subwave != 0u && a.acorr_in == nullptr) {
    // the two-pass form of the unflagged order on these shapes: the reference's chains for every QLPC candidate in front,
    // the kernel's own autocorrelation skipped (the fixed-LPC candidate keeps the canonical sums of the unflagged mode)
    if (a.split_scratch == nullptr) return hipErrorInvalidValue;
    double* racc = reinterpret_cast<double*>(a.split_scratch);
    if (hipError_t err = reference_chains_for_all(a, racc, stream); err != hipSuccess) return err;
    QlpcKernelArgs b = a;
    b.acorr_in = racc;
    return launch_subwave_frames(b, stream);
  }
  const int mp = a.lpc_order <= 8 ? 8 : (a.lpc_order <= 10 ? 10 : 12);
  const int spl = (a.block_size % 72u) == 0 ? 72 : 64;
  const int var = a.stereo ? 2 : 3;
#define FLACENC_HIP_SUBFRAMES(MP, ST, SP, V) \
  if (V == var && mp == MP && spl == SP) return launch_qlpc_subwave_##MP##_##ST##_##SP##_##V(a, stream);
  FLACENC_HIP_FOR_EACH_SUBWAVE_INSTANCE(FLACENC_HIP_SUBFRAMES)
#undef FLACENC_HIP_SUBFRAMES
  return hipErrorInvalidValue;
}

QlpcLaunchPlan plan_qlpc_launch(uint32_t block_size, uint32_t lpc_order) {
  QlpcLaunchPlan plan;
  plan.wave = false;
  const int n = static_cast<int>(block_size);
  const int rows = (n + 15) / 16;
  plan.big = n > 16384;
  plan.maxp = bucket_order(static_cast<int>(lpc_order));
  if (plan.big) plan.maxp = plan.maxp <= 12 ? 12 : 32;
  // must match the kernel's __launch_bounds__
  const int max_threads = plan.maxp <= 12 ? 1024 : (plan.maxp <= 16 ? 512 : 256);
  int threads = 64;
  while (threads < rows && threads < max_threads) threads <<= 1;
  // a block a little over half a workgroup's worth of 16-sample rows (1152 = 72 rows, 2304 = 144) takes the smaller
  // workgroup and a second, mostly idle chunk round: the second wave's idle lanes cost issue slots in every phase,
  // the extra round only in two (measured: 2304 at order 8 / 10 78 -> 90 / 59 -> 82 G samples/s, 1152 at order 10 56 -> 65)
  if (n < 4096 && threads > 64 && rows <= threads / 2 + threads / 8) threads >>= 1;
  plan.threads = threads;
  const int J = (rows + threads - 1) / threads;
  int Jp = 1;
  while (Jp < J) Jp <<= 1;
  const int W = threads / 64;
  const int rowstride = plan.big ? 16 : 20;
  size_t bytes = static_cast<size_t>(rows + kLeadRows) * rowstride * 4;
  bytes += static_cast<size_t>(Jp) * W * (plan.maxp + 1) * 8;
  bytes = (bytes + 15) & ~static_cast<size_t>(15);
  bytes += 40 * 8 + 16 * 8 + 32 * 4 + kMiscCount * 4;
  bytes += (5 * 64 + 16) * 8;  // kFixedSumWords (qlpc_kernel_impl.h): fixed-LPC order selection
  if (!plan.big) {
    // finest_partition_order, rice.rs:157-165 (warm-up <= 32 < 64)
    int lg = 31 - __builtin_clz(static_cast<unsigned>(n / 64));
    int tz = __builtin_ctz(static_cast<unsigned>(n));
    int fo = lg < tz ? lg : tz;
    if (fo > 8) fo = 8;
    bytes += static_cast<size_t>(((2 << fo) + 15) & ~15);  // the levels' Rice parameters, 2 bytes per finest partition
    bytes += static_cast<size_t>(1 << fo) * 32 * 4;
  } else {
    bytes += 2 * FLACENC_HIP_MAX_RICE_PARTITIONS;
  }
  plan.smem_bytes = bytes;
  plan.table_scratch_bytes_per_subframe =
      plan.big ? static_cast<size_t>(FLACENC_HIP_MAX_RICE_PARTITIONS) * 32 * 4 : 0;
  return plan;
}

hipError_t launch_qlpc(const QlpcKernelArgs& a, const QlpcLaunchPlan& plan, hipStream_t stream) {
  if (a.n_subframes == 0) return hipSuccess;
  if ((a.frame_results || a.chan_results) && !wave_kernel_eligible(a)) return hipErrorNotSupported;
  if (a.reference_order && a.sumabs_in == nullptr && a.sumabs_scratch != nullptr && a.lpc_stage == 0 &&
      (a.fixed_mode == 1u || (a.fixed_mode == 0 && a.use_fixed && a.fixed_order_sel == 1u && (a.frame_results || a.chan_results)))) {
    if (a.fixed_mode == 0 && a.sumabs_mode == 0 && a.bps == nullptr && a.bps_uniform <= 16u && a.pack_out == nullptr) {
      // the fused kernel on material of at most 16 bits (side channel: 17): its exact sums are the reference's
      // f32 chains unless a partition reaches 2^24, and it walks those itself (QlpcKernelArgs::sumabs_mode)
      QlpcKernelArgs b = a;
      b.sumabs_mode = a.reference_order;
      b.sumabs_scratch = nullptr;
      return launch_qlpc(b, plan, stream);
    }
    // Reference summation order for fixed_lpc's ApproxEnt selector: every estimator partition's sum of |e|
    // as find_sum_abs_f32's own f32 chain, one (subframe, partition) per lane; the selecting kernels then
    // read the sums instead of adding them up themselves.
    SumAbsRefArgs r{};
    r.samples = a.samples;
    r.stride = a.stride;
    r.block_size = a.block_size;
    r.n_subframes = a.n_subframes;
    r.stereo = a.stereo;
    r.partitions = a.fixed_mode == 1u ? a.fixed_partitions : (64u >> a.fixed_group_log2);
    r.nightly = a.reference_order == 2u ? 1u : 0u;
    r.out = a.sumabs_scratch;
    hipError_t err = launch_sumabs_reference(r, stream);
    if (err != hipSuccess) return err;
    QlpcKernelArgs b = a;
    b.sumabs_in = a.sumabs_scratch;
    return launch_qlpc(b, plan, stream);
  }
  if (a.direct_mse && a.fixed_mode == 0 && a.lpc_stage == 0) {
    // perform_qlpc's experimental branches (src/coding.rs:337-347): the predictor record comes from
    // direct_mse_kernel, then the residual + Rice kernels of the split pipeline take over
    if (a.split_scratch == nullptr) return hipErrorInvalidValue;
    double* racc = reinterpret_cast<double*>(a.split_scratch);
    int32_t* pred = reinterpret_cast<int32_t*>(racc + static_cast<size_t>(a.n_subframes) * 33);
    DirectMseArgs d{};
    d.samples = a.samples;
    d.stride = a.stride;
    d.block_size = a.block_size;
    d.n_subframes = a.n_subframes;
    d.stereo = a.stereo;
    d.window = a.window;
    d.lpc_order = a.lpc_order;
    d.precision = a.precision;
    d.mae_steps = a.mae_steps;
    d.pred_out = pred;
    d.autocorr = a.autocorr;
    d.lpc_coefs = a.lpc_coefs;
    d.weight_scratch = a.irls_weight_scratch;
    d.gram_scratch = a.direct_mse_scratch;
    d.irls_state = (a.mae_steps != 0 && a.direct_mse_scratch != nullptr)
                       ? a.direct_mse_scratch + static_cast<size_t>(a.n_subframes) * direct_mse_gram_stride(a.lpc_order)
                       : nullptr;
    hipError_t err = launch_direct_mse(d, stream);
    if (err != hipSuccess) return err;
    QlpcKernelArgs s3 = a;
    s3.direct_mse = 0;
    s3.reference_order = 0;  // (the estimator has one summation order; nothing left for the order flags here)
    s3.lpc_stage = 3;
    s3.pred = pred;
    s3.autocorr = nullptr;
    s3.lpc_coefs = nullptr;
    s3.acorr_in = nullptr;
    if (bigblock_shape_eligible(a)) {
      err = launch_bigblock_residual(s3, stream);
      if (err != hipSuccess) return err;
      s3.only_marked = 1;  // residuals of 2^26 and more: redone by the generic kernel (see below)
    }
#define FLACENC_HIP_DM3(MP, BG) \
  if (plan.maxp == MP && plan.big == (BG != 0)) return launch_qlpc_##MP##_##BG(s3, plan.threads, plan.smem_bytes, stream);
    FLACENC_HIP_FOR_EACH_INSTANCE(FLACENC_HIP_DM3)
#undef FLACENC_HIP_DM3
    return hipErrorInvalidValue;
  }
  // Blocks of 4096 / 8192 / 16384 samples at orders from 16: the unflagged order IS the reference's stable one -- its
  // chains on the f64 matrix cores cost no more than the chunk tree's fma there (BASELINE configs[2] / [4]; 4096-sample
  // blocks: 344 against 404 us per 50 M samples at order 24), so those shapes emit
  // the stable build's bytes by default (the oracle's orc_default_order_is_stable states the same rule)
  // (a function of the shape alone: unaligned rows or FLACENC_HIP_FLAG_GENERIC_KERNEL change the kernels, not the sums)
  // The unflagged order on blocks of 4096 / 4608 samples at orders up to 12 (the oracle's orc_default_order_is_certified):
  // the fused kernel keeps its own sums where they certify the quantised parameters against the reference's chains and
  // runs those chains itself where not (QlpcKernelArgs::certify).  Launches of these shapes that the fused kernel cannot
  // take (unaligned rows, FLACENC_HIP_FLAG_GENERIC_KERNEL) or that cannot certify inside it (the fused bit writer, which
  // returns bytes alone) are simply given the reference's R[] from acorr_reference_kernel: the same integers -- the
  // reference's -- with R[] and the unquantised coefficients in the reference's own order.
  // The clean-up launch behind the sub-wave kernel's order certificate (round 6): the records it marked with status -2 take
  // the reference's chains -- acorr_reference_mfma_kernel restricted to them (a wave without such a record returns at once,
  // the launch when nothing was marked) -- and the generic kernel redoes them from that R[] (`acorr_marked`).
  if (a.only_marked && a.cert_subwave != 0u && a.acorr_marked == nullptr && a.fixed_mode == 0 && a.split_scratch != nullptr &&
      a.params != nullptr) {
    double* racc = reinterpret_cast<double*>(a.split_scratch);
    AcorrRefArgs r{};
    r.samples = a.samples;
    r.stride = a.stride;
    r.block_size = a.block_size;
    r.n_subframes = a.n_subframes;
    r.stereo = a.stereo;
    r.window = a.window;
    r.lpc_order = a.lpc_order;
    r.nightly = 0u;
    r.out = racc;
    r.marked_params = a.params;
    r.marked_count = a.marked_count;
    r.marked_list = a.marked_list;
    r.marked_cap = a.marked_cap;
    r.marked_unit = a.marked_unit;
    hipError_t err = launch_acorr_reference(r, stream);
    if (err != hipSuccess) return err;
    QlpcKernelArgs b = a;
    b.acorr_marked = racc;
    return launch_qlpc(b, plan, stream);
  }
  bool certified_fused = false;  // this launch certifies inside the fused kernel: no reference-order pass in front of it
  if (a.certify != 0u) {
    // (FLACENC_HIP_FLAG_INTEGER_PARITY_ONLY: the stable order is asked for, but only for the integers -- the certificate
    // gives those; reference_order stays set for the fixed-LPC selector's sums)
    const bool order_ok = a.reference_order == 0u || (a.reference_order == 1u && a.integer_parity_only != 0u);
    const bool shape = cert_shape(a) && order_ok && !a.direct_mse && a.fixed_mode == 0 && a.lpc_stage == 0 &&
                       a.acorr_in == nullptr && !a.only_marked;
    const bool wave = shape && wave_kernel_eligible(a);
    // Every other shape (round 6): the reference's chains for every subframe in a pass of their own in front of the kernel
    // that takes the shape -- the sub-wave kernel (its autocorrelation skipped: acorr_in; an order certificate inside it was
    // 6-13 % faster on noise-like material and 20 to 140 x slower on music, profiles/r06_subwave_two_pass.txt), the
    // big-block kernels, the generic kernel -- so that the unflagged integers are the stable build's on ANY shape
    // (the oracle's orc_default_order_is_two_pass); FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER keeps the one-pass orders.
    const bool two_pass = a.reference_order == 0u && !a.direct_mse && a.fixed_mode == 0 && a.lpc_stage == 0 &&
                          a.acorr_in == nullptr && !a.only_marked && a.lpc_order >= 1 && a.split_scratch != nullptr;
    if (!shape || !wave || a.pack_out != nullptr) {
      QlpcKernelArgs b = a;
      b.certify = 0;
      if (shape || two_pass) b.reference_order = 1u;  // the stable build's order, by the two-pass pipeline below
      return launch_qlpc(b, plan, stream);
    }
    certified_fused = true;
  }
  const bool stable_by_default = a.reference_order == 0u && !a.direct_mse &&
                                 (a.block_size == 4096u || a.block_size == 8192u || a.block_size == 16384u) &&
                                 a.lpc_order >= 16u && a.split_scratch != nullptr;
  if ((a.reference_order || stable_by_default) && !certified_fused && a.fixed_mode == 0 && a.lpc_stage == 0 && a.acorr_in == nullptr) {
    // Reference summation order: R[] by the lane-per-subframe kernel, then the usual pipeline from
    // Levinson on (below: the fused wave kernel with its phase 1 skipped, the big-block kernels from
    // their second launch, or the generic kernel's three-launch split from its second launch).
    if (a.split_scratch == nullptr) return hipErrorInvalidValue;
    double* racc = a.autocorr ? a.autocorr : reinterpret_cast<double*>(a.split_scratch);
    AcorrRefArgs r{};
    r.samples = a.samples;
    r.stride = a.stride;
    r.block_size = a.block_size;
    r.n_subframes = a.n_subframes;
    r.stereo = a.stereo;
    r.window = a.window;
    r.lpc_order = a.lpc_order;
    r.nightly = a.reference_order == 2u ? 1u : 0u;
    r.out = racc;
    hipError_t err = launch_acorr_reference(r, stream);
    if (err != hipSuccess) return err;
    QlpcKernelArgs b = a;
    b.acorr_in = racc;
    b.autocorr = nullptr;  // already written
    return launch_qlpc(b, plan, stream);
  }
  if (bigblock_fixed_eligible(a)) {
    // fixed_lpc on the big-block shapes: order selection -> predictor record -> the residual + Rice kernel with
    // FixedLpc's bit count; subframes with residuals of 2^26 and more are redone by the generic kernel
    int32_t* pred = reinterpret_cast<int32_t*>(reinterpret_cast<double*>(a.split_scratch) +
                                                static_cast<size_t>(a.n_subframes) * 33);
    QlpcKernelArgs s1 = a, s3 = a;
    s1.pred_out = pred;
    hipError_t err = launch_bigblock_fixed_select(s1, stream);
    if (err != hipSuccess) return err;
    s3.pred = pred;
    s3.selector_keys = nullptr;  // written by the selection
    s3.fixed_keys = nullptr;
    err = launch_bigblock_fixed_residual(s3, stream);
    if (err != hipSuccess) return err;
    QlpcKernelArgs s4 = a;
    s4.only_marked = 1;
    s4.split_scratch = nullptr;
#define FLACENC_HIP_FIXCLEAN(MP, BG) \
  if (plan.maxp == MP && plan.big == (BG != 0)) return launch_qlpc_##MP##_##BG(s4, plan.threads, plan.smem_bytes, stream);
    FLACENC_HIP_FOR_EACH_INSTANCE(FLACENC_HIP_FIXCLEAN)
#undef FLACENC_HIP_FIXCLEAN
    return hipErrorInvalidValue;
  }
  if (bigblock_eligible(a) || (a.acorr_in != nullptr && !wave_kernel_eligible(a) && !subwave_eligible(a) && a.lpc_stage == 0)) {
    // R[] -> levinson_batch_kernel (one subframe per lane) -> residual + Rice search
    if (a.split_scratch == nullptr) return hipErrorInvalidValue;
    const bool big = bigblock_eligible(a);
    const double* racc = a.acorr_in;
    hipError_t err;
    if (racc == nullptr) {
      QlpcKernelArgs s1 = a;
      s1.autocorr = a.autocorr ? a.autocorr : reinterpret_cast<double*>(a.split_scratch);
      racc = s1.autocorr;
      err = launch_bigblock_acorr(s1, stream);
      if (err != hipSuccess) return err;
    }
    int32_t* pred = reinterpret_cast<int32_t*>(reinterpret_cast<double*>(a.split_scratch) +
                                                static_cast<size_t>(a.n_subframes) * 33);
    QlpcKernelArgs s2 = a, s3 = a;
    s2.autocorr = const_cast<double*>(racc);
    s2.pred_out = pred;
    s3.lpc_stage = 3;
    s3.pred = pred;
    s3.autocorr = nullptr;
    s3.lpc_coefs = nullptr;  // written by the batch kernel
    s3.acorr_in = nullptr;
    err = hipErrorInvalidValue;
#define FLACENC_HIP_LEV(MP, BG) \
  if (plan.maxp == MP && plan.big == (BG != 0)) err = launch_levinson_##MP##_##BG(s2, stream);
    FLACENC_HIP_FOR_EACH_INSTANCE(FLACENC_HIP_LEV)
#undef FLACENC_HIP_LEV
    if (err != hipSuccess) return err;
    if (big) {
      err = launch_bigblock_residual(s3, stream);
      if (err != hipSuccess) return err;
      // clean-up launch: the generic kernel redoes the subframes the big-block kernel marked (residuals
      // of 2^26 and more need the literal chunk-clamped bit tables); every other workgroup exits at once
      s3.only_marked = 1;
    }
#define FLACENC_HIP_STAGE3(MP, BG) \
  if (plan.maxp == MP && plan.big == (BG != 0)) return launch_qlpc_##MP##_##BG(s3, plan.threads, plan.smem_bytes, stream);
    FLACENC_HIP_FOR_EACH_INSTANCE(FLACENC_HIP_STAGE3)
#undef FLACENC_HIP_STAGE3
    return hipErrorInvalidValue;
  }
  if (subwave_eligible(a) || subwave_fixed_eligible(a)) {
    // several subframes per wave; what it marks (residuals of 2^25 and more, saturated Rice tables) is redone by the
    // generic kernel's clean-up launch, which returns at once when nothing was marked
    const int var = a.fixed_mode == 1u ? 1 : 0;
    const int mp = (var == 1 || a.lpc_order <= 8) ? 8 : (a.lpc_order <= 10 ? 10 : 12);
    const int st = a.stereo ? 1 : 0;
    const int spl = (a.block_size % 72u) == 0 ? 72 : 64;
    hipError_t err = hipErrorInvalidValue;
#define FLACENC_HIP_SUBCASE(MP, ST, SP, V) \
  if (mp == MP && st == ST && spl == SP && var == V) err = launch_qlpc_subwave_##MP##_##ST##_##SP##_##V(a, stream);
    FLACENC_HIP_FOR_EACH_SUBWAVE_INSTANCE(FLACENC_HIP_SUBCASE)
#undef FLACENC_HIP_SUBCASE
    if (err != hipSuccess) return err;
    QlpcKernelArgs c = a;
    c.only_marked = 1;
    c.autocorr = nullptr;  // (written; the clean-up rewrites records and rows only)
    c.lpc_coefs = nullptr;
    c.selector_keys = a.selector_keys;
    if (a.acorr_in != nullptr) {
      // R[] came from the launch in front (the reference's chains of the unflagged order, or a flagged order): what the
      // kernel marked (-2) is redone from that very R[]
      c.acorr_marked = a.acorr_in;
      c.acorr_in = nullptr;
      c.reference_order = 0u;
    }
#define FLACENC_HIP_SUBCLEAN(MP, BG) \
  if (plan.maxp == MP && plan.big == (BG != 0)) return launch_qlpc_##MP##_##BG(c, plan.threads, plan.smem_bytes, stream);
    FLACENC_HIP_FOR_EACH_INSTANCE(FLACENC_HIP_SUBCLEAN)
#undef FLACENC_HIP_SUBCLEAN
    return hipErrorInvalidValue;
  }
  if (wave_kernel_eligible(a)) {
    const int mp = a.lpc_order <= 8 ? 8 : (a.lpc_order <= 10 ? 10 : 12);
    int variant = a.stereo ? (a.frame_results ? (a.pack_out ? 5 : (a.use_fixed ? 3 : 2)) : 1)
                           : (a.chan_results ? 4 : 0);
    if (a.sumabs_mode != 0u && (variant == 3 || variant == 4)) variant += 3;  // the instances with the chain walk
    if (a.block_size == 4608) {
#define FLACENC_HIP_W72CASE(MP, ST) \
  if (mp == MP && variant == ST) return launch_qlpc_wave72_##MP##_##ST(a, stream);
      FLACENC_HIP_FOR_EACH_WAVE72_INSTANCE(FLACENC_HIP_W72CASE)
#undef FLACENC_HIP_W72CASE
      return hipErrorInvalidValue;
    }
#define FLACENC_HIP_WCASE(MP, ST) \
  if (mp == MP && variant == ST) return launch_qlpc_wave_##MP##_##ST(a, stream);
    FLACENC_HIP_FOR_EACH_WAVE_INSTANCE(FLACENC_HIP_WCASE)
#undef FLACENC_HIP_WCASE
  }
  // Large orders: the recursion is ~5 P^2 / 2 serial instructions, during which a whole workgroup
  // would idle -- run it one subframe per lane between two launches of the generic kernel instead
  // (R[] and the quantised predictor go through `split_scratch`; the samples are read twice).
  if (a.fixed_mode == 0 && a.lpc_stage == 0 && a.lpc_order >= 16 && a.split_scratch != nullptr) {
    QlpcKernelArgs s1 = a, s2 = a, s3 = a;
    double* racc = reinterpret_cast<double*>(a.split_scratch);
    int32_t* pred = reinterpret_cast<int32_t*>(racc + static_cast<size_t>(a.n_subframes) * 33);
    s1.lpc_stage = 1;
    s1.autocorr = a.autocorr ? a.autocorr : racc;
    s2.autocorr = s1.autocorr;
    s2.pred_out = pred;
    s3.lpc_stage = 3;
    s3.pred = pred;
    s3.autocorr = nullptr;  // already written by the first launch
    s3.lpc_coefs = nullptr;  // written by the batch kernel
#define FLACENC_HIP_SPLIT(MP, BG)                                                              \
  if (plan.maxp == MP && plan.big == (BG != 0)) {                                              \
    hipError_t err = launch_qlpc_##MP##_##BG(s1, plan.threads, plan.smem_bytes, stream);       \
    if (err != hipSuccess) return err;                                                         \
    err = launch_levinson_##MP##_##BG(s2, stream);                                             \
    if (err != hipSuccess) return err;                                                         \
    return launch_qlpc_##MP##_##BG(s3, plan.threads, plan.smem_bytes, stream);                 \
  }
    FLACENC_HIP_FOR_EACH_INSTANCE(FLACENC_HIP_SPLIT)
#undef FLACENC_HIP_SPLIT
  }
#define FLACENC_HIP_CASE(MP, BG)                       \
  if (plan.maxp == MP && plan.big == (BG != 0))        \
    return launch_qlpc_##MP##_##BG(a, plan.threads, plan.smem_bytes, stream);
  FLACENC_HIP_FOR_EACH_INSTANCE(FLACENC_HIP_CASE)
#undef FLACENC_HIP_CASE
  return hipErrorInvalidValue;
}

}  // namespace flacenc_hip
