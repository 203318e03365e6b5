// qlpc_kernel.h -- launch interface between the C-ABI layer and the HIP kernels.
#ifndef FLACENC_HIP_QLPC_KERNEL_H_
#define FLACENC_HIP_QLPC_KERNEL_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>


#include "flacenc_hip.h"
#include "lds_opt_in.h"

namespace flacenc_hip {

// Autocorrelation summation order implemented by the kernels (the "canonical
// order" of DESIGN.md): 16-sample chunk chains + balanced tree over chunk index.
constexpr int kAcorrChunk = 16;

struct QlpcKernelArgs {
  const int32_t* samples;   // device; subframe k at samples + k*stride
  size_t stride;
  uint32_t block_size;
  uint32_t n_subframes;
  const uint8_t* bps;       // device, per subframe (nullable -> bps_uniform)
  uint32_t bps_uniform;
  uint32_t stereo;          // 1: workgroups 4f..4f+3 = L, R, M, S of 2-channel frame f
  const float* window;      // device table with 32 leading pad floats, nullptr = all ones
  int32_t flat_lo;          // window[t] == 1.0f for flat_lo <= t < flat_hi
  int32_t flat_hi;
  uint32_t lpc_order;
  uint32_t precision;
  uint32_t max_rice_parameter;
  uint32_t rice_finest_only;  // FLACENC_HIP_FLAG_FINEST_RICE_ORDER: no merging below the finest order
  uint32_t force_generic;     // FLACENC_HIP_FLAG_GENERIC_KERNEL: never the wave-per-subframe kernel
  // FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER: R[] comes from acorr_reference_kernel (the reference's stable
  // summation order, one sequential chain per lag) instead of the kernels' canonical chunk tree;
  // launch_qlpc runs it into split_scratch (or `autocorr`) and hands the result on as `acorr_in`
  uint32_t reference_order;
  const double* acorr_in;     // device, [n][33]: precomputed R[], skips phase 1 (wave kernel)
  // The unflagged order on blocks of 4096 / 4608 samples at orders up to 12 (set by launch_qlpc): the chunk tree's R[]
  // is kept where it CERTIFIES the quantised parameters against the reference's chains (levinson_quantize<.., CERT>) and
  // the subframe is redone from those chains where it does not -- by the fused kernel itself (reference_chains_from_lds);
  // launches of these shapes on other kernels take the reference's order outright.  cert_stats (nullable, test / bench
  // hook): [0] subframes analysed, [1] certificates that needed the rows of T^-1, [2] subframes redone.
  uint32_t certify = 0;
  // ... and on the sub-wave kernel's shapes (blocks of 256 .. 2304 samples, orders up to 12; round 6) the unflagged order is
  // the reference's by two passes: launch_subwave_frames runs the reference's chains for every QLPC candidate in front
  // (acorr_reference_mfma_kernel) and qlpc_subwave_kernel takes them as acorr_in; the clean-up launch behind it
  // (only_marked) redoes what the kernel marked (-2) from those chains, which acorr_reference_mfma_kernel recomputes for
  // exactly the marked records and hands to the generic kernel as `acorr_marked`.  (Candidate-level launches of these
  // shapes simply run with reference_order = 1: qlpc_dispatch.cpp.)
  uint32_t cert_subwave = 0;
  const double* acorr_marked = nullptr;  // device, [n][33]: R[] of the records with status -2 (clean-up launch only)
  uint32_t integer_parity_only = 0;  // FLACENC_HIP_FLAG_INTEGER_PARITY_ONLY with reference_order 1: certified shapes keep their order
  uint32_t* cert_stats = nullptr;
  // ... and, with the ApproxEnt order selector of fixed_lpc, every estimator partition's sum of |e| comes
  // from sumabs_reference_kernel (find_sum_abs_f32's sequential f32 chain, arrayutils.rs:496-506) instead of
  // the kernels' exact integer sums: launch_qlpc runs it into `sumabs_scratch` and hands it on as `sumabs_in`
  const float* sumabs_in = nullptr;  // device, [n][5][64]: order k, partition p at [(sf * 5 + k) * 64 + p]
  float* sumabs_scratch = nullptr;   // device, n * 5 * 64 floats, or nullptr
  // ... except in front of the fused kernel on material of at most 16 bits: there the exact sums ARE the
  // reference's while a partition stays below 2^24, and the kernel itself walks the rare partition that does not
  // (1: the stable build's chain, 2: simd-nightly's lane chains) -- set by launch_qlpc, no pre-pass
  uint32_t sumabs_mode = 0;
  // config::Qlpc::use_direct_mse / mae_optimization_steps (experimental in the reference, src/coding.rs:337-347):
  // the predictor comes from direct_mse_kernel (covariance-method LPC, optionally IRLS) instead of
  // autocorrelation + Levinson; launch_qlpc runs it into split_scratch and continues with the residual kernels
  uint32_t direct_mse = 0;
  uint32_t mae_steps = 0;
  double* direct_mse_scratch = nullptr;  // device, n * direct_mse_gram_stride(order) doubles: R[] and the matrix between direct_mse_kernel's chains and the batched solve (no IRLS steps)
  float* irls_weight_scratch = nullptr;  // device, n * ((block_size + 3) & ~3) floats: IRLS weights of blocks above 16384 samples
  // Frame-level calls on the big-block shapes (bigblock_residual_kernel, stereo): the candidates of roles L and R
  // go straight to the OUTPUT rows 2f and 2f + 1 (L can only ever fill output channel 0, R only channel 1:
  // ChannelAssignment::select_channels, datatype.rs:1173-1185), M and S to `residual` as usual, and the role's
  // min / max (is_constant, arrayutils.rs:382) to minmax_out[sf][2] -- frame_decide_kernel then copies a row only
  // when the decision picked something else for that channel, and scans nothing.
  int32_t* residual_lr = nullptr;
  size_t residual_lr_stride = 0;
  int32_t* minmax_out = nullptr;
  // bigblock_residual_kernel: 0 = analyse + store four candidate rows per frame; 1 = analyse only (records, no rows);
  // 2 = decide + store: encode_subframe / try_stereo_coding over the records two mode-1 launches left, then only the
  // two chosen rows are produced (frame-level calls on the big-block shapes)
  uint32_t residual_mode = 0;
  const flacenc_hip_subframe_params* cand_lpc_params = nullptr;    // mode 2, [n_subframes]; null: candidate kind off
  const flacenc_hip_subframe_params* cand_fixed_params = nullptr;
  const unsigned long long* cand_fixed_keys = nullptr;             // the order selector's key of each fixed candidate
  const int32_t* cand_minmax = nullptr;                            // [n_subframes][2]: min, max of each role
  const int32_t* cand_lpc_rows = nullptr;                          // rows the generic clean-up wrote for marked subframes
  const int32_t* cand_fixed_rows = nullptr;
  size_t cand_stride = 0;
  uint32_t only_marked;       // generic kernel: redo only subframes whose record says status == -1
  // subframes bigblock_residual_kernel marked for that clean-up launch: qlpc_marked_kernel returns at once on 0.  The
  // handle alternates between two counters from pipeline to pipeline; the clean-up launch clears `marked_next`, the one
  // the following pipeline counts into (a pipeline without a clean-up launch leaves a stale count behind at worst: a
  // scan that finds nothing).  nullptr: always scan.
  uint32_t* marked_count = nullptr;
  uint32_t* marked_next = nullptr;
  // The first `marked_cap` of those marks, in the order they were counted (slot = what the mark's atomicAdd returned): the
  // clean-up launches visit these records directly instead of reading the status of all n_subframes records -- a scan of
  // 262 144 records for one marked subframe took a 256-sample launch 90 us (round 6).  An entry counts in units of
  // `marked_unit` records: 1, or 4 for stereo frames (whose four roles are marked together: entry = the frame).  Consumers
  // check an entry against n_subframes and the record's own status, so that entries a pipeline without a clean-up launch
  // left behind cost a look and nothing else.  nullptr, or more marks than the list holds: the scan.
  uint32_t* marked_list = nullptr;
  uint32_t marked_cap = 0;
  uint32_t marked_unit = 1;
  flacenc_hip_subframe_params* params;  // device
  int32_t* residual;                    // device
  size_t residual_stride;
  double* autocorr;                     // device, nullable, [n][33]
  double* lpc_coefs;                    // device, nullable, [n][32]
  uint32_t* table_scratch;              // device, only for blocks > 16384 samples
  unsigned long long* stamps;           // device, nullable: [n][8] phase timestamps (profiling)
  // on-device encode_frame decision (stereo, wave kernel only)
  flacenc_hip_stereo_frame_result* frame_results;  // device, [n_frames]; non-null selects DECIDE
  flacenc_hip_channel_result* chan_results;        // device, [n_subframes]; non-null selects the
                                                   // independent-channel DECIDE variant (stereo = 0)
  uint32_t use_constant, use_lpc, use_leftside, use_rightside, use_midside;
  // fixed-LPC candidate (config::Fixed, src/config.rs:236-244); wave kernel variant 3 only
  uint32_t use_fixed;
  uint32_t fixed_max_order;    // 0..4
  uint32_t fixed_order_sel;    // 0 BitCount, 1 ApproxEnt
  uint32_t fixed_group_log2;   // ApproxEnt: lanes per estimator partition = 2^g (4096 / partitions / 64)
  unsigned long long* fixed_keys;  // device, nullable: [n][8] the selector's key per order (tests)
  // generic kernel as `fixed_lpc` (src/coding.rs:298-331): 0 = QLPC analysis (default),
  // 1 = ApproxEnt order selection + coding, 2 = code order forced_uniform, 3 = code forced_orders[sf]
  uint32_t fixed_mode;
  uint32_t fixed_partitions;       // mode 1: OrderSel::ApproxEnt.partitions (1..64)
  uint32_t forced_uniform;         // mode 2
  const uint8_t* forced_orders;    // mode 3, device, [n]
  unsigned long long* selector_keys;  // device, nullable, [n]: the chosen order's selector key
  // three-launch split of the generic kernel for large orders (launch_qlpc): 0 = fused,
  // 1 = window + autocorrelation only (R[] to `autocorr`), 3 = residual + Rice with the predictor
  // levinson_batch_kernel left in `pred` ([n][36] int32: qc[32], order, shift, status, 0)
  uint32_t lpc_stage;
  const int32_t* pred;
  int32_t* pred_out;
  void* split_scratch;  // device, n * (33 * 8 + 36 * 4) bytes, or nullptr: no split
  // Frame::write fused into the deciding wave kernel (variant 5): non-null pack_out selects it.
  // Header constants and CRC combination powers as in FramePackArgs (frame_pack.h).
  uint8_t* pack_out;          // device, 16-byte aligned; frame f at pack_out + f*pack_out_stride
  size_t pack_out_stride;     // multiple of 16
  uint32_t* pack_out_len;     // device, [n_frames]
  uint32_t pack_header_mid, pack_extra_len;
  uint8_t pack_extra[4];
  uint32_t pack_first_frame, pack_frame_step;
  uint32_t pack_lds_words, pack_crc_per;
  uint16_t pack_crc_pow[32];
};

// A record (or, unit 4, a stereo frame) is marked for the clean-up launches: counted, and entered in the list while it has room.
__device__ __forceinline__ void count_marked(const QlpcKernelArgs& a, uint32_t index) {
  if (a.marked_count == nullptr) return;
  const uint32_t slot = atomicAdd(a.marked_count, 1u);
  if (a.marked_list != nullptr && slot < a.marked_cap) a.marked_list[slot] = index;
}

struct QlpcLaunchPlan {
  bool wave;       // wave-per-subframe kernel (block_size 4096, order <= 12, aligned buffers)
  int maxp;        // template bucket for the LPC order
  bool big;        // block_size > 16384: unpadded LDS image, bit tables in HBM scratch
  int threads;     // workgroup size (power of two, 64..1024)
  size_t smem_bytes;
  size_t table_scratch_bytes_per_subframe;
};

QlpcLaunchPlan plan_qlpc_launch(uint32_t block_size, uint32_t lpc_order);
// true if the wave-per-subframe kernel can take this launch (decided per call: alignment)
bool wave_kernel_eligible(const QlpcKernelArgs& args);
// blocks of 4096 / 4608 samples at orders up to 12: the shapes whose unflagged order is certified (QlpcKernelArgs::certify)
bool cert_shape(const QlpcKernelArgs& args);
hipError_t launch_qlpc(const QlpcKernelArgs& args, const QlpcLaunchPlan& plan, hipStream_t stream);
// blocks of 8192 / 16384 samples at order 13..32 (qlpc_bigblock.cpp): autocorrelation and residual + Rice
// search as two pass-structured kernels either side of levinson_batch_kernel
bool bigblock_eligible(const QlpcKernelArgs& args);
// the same without the order bounds (any order 1..32: the residual kernel alone, behind a predictor record)
bool bigblock_shape_eligible(const QlpcKernelArgs& args);
hipError_t launch_bigblock_acorr(const QlpcKernelArgs& args, hipStream_t stream);     // R[] -> args.autocorr
hipError_t launch_bigblock_residual(const QlpcKernelArgs& args, hipStream_t stream);  // args.pred -> records
// fixed_lpc with OrderSel::ApproxEnt on those shapes: order selection -> args.pred_out, then the residual kernel
bool bigblock_fixed_eligible(const QlpcKernelArgs& args);
hipError_t launch_bigblock_fixed_select(const QlpcKernelArgs& args, hipStream_t stream);
hipError_t launch_bigblock_fixed_residual(const QlpcKernelArgs& args, hipStream_t stream);

// one per (order bucket, big) instantiation, each defined by its own translation unit
#define FLACENC_HIP_FOR_EACH_INSTANCE(X) \
  X(8, 0) X(10, 0) X(12, 0) X(16, 0) X(24, 0) X(32, 0) X(12, 1) X(32, 1)
#define FLACENC_HIP_DECLARE_INSTANCE(MP, BG) \
  hipError_t launch_qlpc_##MP##_##BG(const QlpcKernelArgs&, int threads, size_t smem, hipStream_t); \
  hipError_t launch_levinson_##MP##_##BG(const QlpcKernelArgs&, hipStream_t);
FLACENC_HIP_FOR_EACH_INSTANCE(FLACENC_HIP_DECLARE_INSTANCE)

#define FLACENC_HIP_FOR_EACH_WAVE_INSTANCE(X) \
  X(8, 0) X(8, 1) X(8, 2) X(8, 3) X(8, 4) X(8, 5) X(8, 6) X(8, 7) X(10, 0) X(10, 1) X(10, 2) X(10, 3) X(10, 4) X(10, 5) \
  X(10, 6) X(10, 7) X(12, 0) X(12, 1) X(12, 2) X(12, 3) X(12, 4) X(12, 5) X(12, 6) X(12, 7)
#define FLACENC_HIP_DECLARE_WAVE_INSTANCE(MP, ST) \
  hipError_t launch_qlpc_wave_##MP##_##ST(const QlpcKernelArgs&, hipStream_t);
FLACENC_HIP_FOR_EACH_WAVE_INSTANCE(FLACENC_HIP_DECLARE_WAVE_INSTANCE)
// ... and for blocks of 4608 samples (72 per lane; no fused bit writer)
#define FLACENC_HIP_FOR_EACH_WAVE72_INSTANCE(X) \
  X(8, 0) X(8, 1) X(8, 2) X(8, 3) X(8, 4) X(8, 6) X(8, 7) X(10, 0) X(10, 1) X(10, 2) X(10, 3) X(10, 4) X(10, 6) X(10, 7) \
  X(12, 0) X(12, 1) X(12, 2) X(12, 3) X(12, 4) X(12, 6) X(12, 7)
#define FLACENC_HIP_DECLARE_WAVE72_INSTANCE(MP, ST) \
  hipError_t launch_qlpc_wave72_##MP##_##ST(const QlpcKernelArgs&, hipStream_t);
FLACENC_HIP_FOR_EACH_WAVE72_INSTANCE(FLACENC_HIP_DECLARE_WAVE72_INSTANCE)

// bigblock_residual_kernel: one translation unit per (passes of 4096 samples, byte limbs per sample)
#define FLACENC_HIP_FOR_EACH_BIGRES_INSTANCE(X) X(1, 2) X(1, 3) X(1, 4) X(2, 2) X(2, 3) X(2, 4) X(4, 2) X(4, 3) X(4, 4)
#define FLACENC_HIP_DECLARE_BIGRES_INSTANCE(K_, NLB_) \
  hipError_t launch_bigblock_residual_##K_##_##NLB_(const QlpcKernelArgs&, hipStream_t);
FLACENC_HIP_FOR_EACH_BIGRES_INSTANCE(FLACENC_HIP_DECLARE_BIGRES_INSTANCE)

// qlpc_subwave_kernel: several subframes per wave for blocks of 4 / 8 / 16 / 32 finest Rice partitions (256 .. 2048,
// 288 .. 2304 samples) at orders up to 12; one translation unit per (order bucket, stereo, samples per lane,
// variant: 0 QLPC candidates, 1 fixed_lpc batch with the ApproxEnt selector, 2 the 2-channel frame decision, 3 frames of
// independent channels)
bool subwave_shape(uint32_t block_size);
bool subwave_eligible(const QlpcKernelArgs& args);        // variant 0
bool subwave_fixed_eligible(const QlpcKernelArgs& args);  // variant 1 (args.fixed_mode == 1)
bool subwave_frame_eligible(const QlpcKernelArgs& args);  // variant 2 (args.frame_results set)
// variant 2: results + the two chosen rows per frame; frames it could not decide (a candidate beyond the exact sums) are
// marked -- channel_assignment 0xFF, status -1 in args.cand_lpc_params / cand_fixed_params, args.marked_count -- for the
// caller's general path
hipError_t launch_subwave_frames(const QlpcKernelArgs& args, hipStream_t stream);
// variant 3: Independent(n) frames (args.chan_results set, args.stereo == 0); a subframe it could not decide is marked by
// kind 0xFF in its result (+ the scratch records and the count, as above)
bool subwave_channels_eligible(const QlpcKernelArgs& args);
#define FLACENC_HIP_FOR_EACH_SUBWAVE_INSTANCE(X)                                                                      \
  X(8, 0, 64, 0) X(8, 0, 72, 0) X(8, 1, 64, 0) X(8, 1, 72, 0) X(10, 0, 64, 0) X(10, 0, 72, 0) X(10, 1, 64, 0) X(10, 1, 72, 0)     \
  X(12, 0, 64, 0) X(12, 0, 72, 0) X(12, 1, 64, 0) X(12, 1, 72, 0) X(8, 0, 64, 1) X(8, 0, 72, 1) X(8, 1, 64, 1) X(8, 1, 72, 1)     \
  X(8, 1, 64, 2) X(8, 1, 72, 2) X(10, 1, 64, 2) X(10, 1, 72, 2) X(12, 1, 64, 2) X(12, 1, 72, 2)                                 \
  X(8, 0, 64, 3) X(8, 0, 72, 3) X(10, 0, 64, 3) X(10, 0, 72, 3) X(12, 0, 64, 3) X(12, 0, 72, 3)
#define FLACENC_HIP_DECLARE_SUBWAVE_INSTANCE(MP, ST, SP, V) \
  hipError_t launch_qlpc_subwave_##MP##_##ST##_##SP##_##V(const QlpcKernelArgs&, hipStream_t);
FLACENC_HIP_FOR_EACH_SUBWAVE_INSTANCE(FLACENC_HIP_DECLARE_SUBWAVE_INSTANCE)

}  // namespace flacenc_hip
#endif
