/*
 * flacenc_hip.h -- C ABI of the MI355X (gfx950) implementation of flacenc-rs's
 * per-subframe quantised-LPC analysis path and of the stages either side of it.
 *
 * Three levels, each a superset of the one before:
 *   candidates   flacenc_hip_qlpc_batch / _stereo_qlpc_batch   (coding::estimated_qlpc)
 *                flacenc_hip_fixed_lpc_batch                   (coding::fixed_lpc)
 *   frames       flacenc_hip_encode_stereo_frames / _encode_frames   (coding::encode_frame: the
 *                candidates plus encode_subframe and try_stereo_coding on the device)
 *   bytes        flacenc_hip_pack_stereo_frames / _pack_frames (Frame::write, both CRCs),
 *                flacenc_hip_encode_pack_*_frames_async (PCM in HBM -> frame bytes in one call),
 *                flacenc_hip_fill_le_bytes (packed interleaved PCM -> FrameBuf layout)
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Each entry point names the reference interface it replaces (paths relative
 * to the flacenc-rs v0.5.1 source tree).  The Rust-side binding a maintainer
 * would add is shown in INTEGRATION.md and shipped as source in rust/.
 *
 * Threading: a handle owns one HIP stream, its device scratch and its window
 * cache; it is NOT thread-safe -- use one handle per host thread / per GPU,
 * like the reference's per-thread `reusable!` scratch (src/lib.rs:92-116).
 * ONE STREAM AT A TIME PER HANDLE: the *_async entry points take a stream per call, but the handle's scratch
 * (and the marked-subframe counters its clean-up launches alternate between) is shared by all of them: calls
 * on one handle must be ordered on one stream (or by events) -- two streams need two handles.
 * Errors never unwind across this boundary: every call returns an int status
 * (0 = OK, negative = error) and per-subframe `status` fields carry the
 * conditions on which the reference would panic.
 */
#ifndef FLACENC_HIP_H_
#define FLACENC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6: without a summation-order flag the autocorrelation is the stable build's on every shape (round 5: certified on blocks
 *    of 4096 / 4608 at orders up to 12; round 6: the reference's chains in a pass of their own everywhere else) -- other bytes
 *    for the same config than revision 5 on a fraction of a per mille of subframes; the product library exports the
 *    declarations of this header and nothing else (the debug hooks live in libflacenc_hip_hooks.so).  No signature changed.
 * 5: blocks of 4096 / 8192 / 16384 samples at orders from 16 sum in the stable build's order by default (other bytes
 *    for the same config than revision 4); the two flacenc_hip_debug_* symbols left the public ABI; new exports
 *    flacenc_hip_frame_wire_bytes, flacenc_hip_stereo_frame_wire_async, flacenc_hip_stream_offsets_async (round 4) and
 *    the flacenc_hip_comm_* / flacenc_hip_allgather_* collective calls (round 5).
 * 4: use_direct_mse / mae_optimization_steps in flacenc_hip_qlpc_config, NIGHTLY_SUM_ORDER, frame-level calls take
 *    blocks below 64 samples */
#define FLACENC_HIP_ABI_VERSION 6

/* FLAC allows LPC order 32; the reference's config verifier caps it at 24
 * (src/constant.rs:118, src/config.rs:304).  Orders 25..32 are an extension
 * and are rejected unless FLACENC_HIP_FLAG_ALLOW_ORDER_32 is set. */
#define FLACENC_HIP_MAX_LPC_ORDER 32
#define FLACENC_HIP_REF_MAX_LPC_ORDER 24
#define FLACENC_HIP_MAX_PRECISION 15      /* src/constant.rs qlpc::MAX_PRECISION */
#define FLACENC_HIP_MAX_RICE_PARAMETER 30 /* src/constant.rs:143 */
#define FLACENC_HIP_MAX_MAE_STEPS 64    /* (the reference has no bound; its experimental preset uses 20, src/config.rs:535) */
#define FLACENC_HIP_MIN_BLOCK_SIZE 64     /* MIN_BLOCK_SIZE_FOR_PREDICTION, src/constant.rs:51 */
#define FLACENC_HIP_MAX_BLOCK_SIZE 32767  /* src/constant.rs:57 */
#define FLACENC_HIP_MAX_RICE_PARTITIONS 256 /* 2^8: finest order for any block <= 32767 */
#define FLACENC_HIP_MAX_FIXED_LPC_ORDER 4   /* src/constant.rs:95 */

/* return codes */
#define FLACENC_HIP_OK 0
#define FLACENC_HIP_ERR_BAD_CONFIG (-1)   /* -> EncodeError::Config(VerifyError), src/error.rs:458 */
#define FLACENC_HIP_ERR_BAD_ARGUMENT (-2)
#define FLACENC_HIP_ERR_DEVICE (-3)       /* HIP runtime failure; see flacenc_hip_last_error */
#define FLACENC_HIP_ERR_UNSUPPORTED (-4)
#define FLACENC_HIP_ERR_NO_DEVICE (-5)

/* per-subframe status bits (the reference panics on these) */
#define FLACENC_HIP_SUBFRAME_OK 0
#define FLACENC_HIP_SUBFRAME_NONFINITE 1  /* assert at src/lpc.rs:786-799 */
#define FLACENC_HIP_SUBFRAME_NEG_ENERGY 2 /* assert at src/lpc.rs:646-655 */

/* config::Window, src/config.rs:344-359 */
#define FLACENC_HIP_WINDOW_RECTANGLE 0
#define FLACENC_HIP_WINDOW_TUKEY 1

#define FLACENC_HIP_FLAG_ALLOW_ORDER_32 1u
/* Build extension, NOT a reference mode (the reference always runs the exhaustive search,
 * src/rice.rs:246-298): keep the finest Rice partition order instead of merging down to order 0 --
 * BASELINE config 2's "fixed Rice partition order".  Output stays valid, lossless FLAC; it is just
 * not the partition the reference would pick when a coarser order is cheaper. */
#define FLACENC_HIP_FLAG_FINEST_RICE_ORDER 2u
/* Kernel selection overrides (no reference counterpart; every choice produces identical results):
 * GENERIC_KERNEL keeps block-4096 / order <= 12 launches off the fused wave-per-subframe kernel;
 * FUSED_PACK / TWO_STAGE_PACK force flacenc_hip_encode_pack_stereo_frames_async's one-kernel or
 * two-kernel form (default: two kernels, the faster choice measured since the deciding kernels run at
 * three workgroups per CU). */
#define FLACENC_HIP_FLAG_GENERIC_KERNEL 4u
#define FLACENC_HIP_FLAG_FUSED_PACK 8u
#define FLACENC_HIP_FLAG_TWO_STAGE_PACK 16u
/* Autocorrelation in the summation order of the reference's stable build: one sequential mul_add chain
 * per lag over t = order .. n-1 (weighted_auto_correlation_nosimd, src/lpc.rs:533-548), computed one
 * subframe per lane by a kernel of its own, instead of the kernels' own orders (a chain per lane or per 16-sample
 * chunk, combined by a tree -- same FMA count, different roundings; certified against the reference's on the fused
 * kernel's shapes, see FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER below and DESIGN.md section 2).
 * With this flag R[], the LPC coefficients and therefore every integer output are those of the stable
 * reference build for everything the oracle pins.  Costs one extra pass over the samples.
 * The flag also reaches the path's other order-sensitive sum: with OrderSel::ApproxEnt, fixed_lpc's
 * estimator takes every partition's sum of |e| from find_sum_abs_f32 (src/arrayutils.rs:496-506), which in
 * the stable build is one sequential f32 chain per partition; a kernel that gives a lane one (subframe,
 * partition) pair reproduces those chains for the five orders, so the selector's keys -- and with them the
 * chosen candidate and the frame bytes of the reference's default configuration -- are the stable
 * build's, 24-bit material included (tests/test_gpu_reference_order.py). */
#define FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER 32u
/* The same two sums in the order of the reference's `simd-nightly` build -- the build its published numbers and
 * compression ratios come from (report/report.nightly.md).  Autocorrelation (weighted_auto_correlation_simd,
 * src/lpc.rs:510-531 over weighted_delay_prod_sum_impl :439-500): per lag 8 (lags 0..7) or 16 (lags 8..15)
 * strided f64 lane chains over the 64-byte-aligned body of the windowed buffer, a scalar chain over head and
 * foot, an ordered lane sum.  find_sum_abs_f32 (src/arrayutils.rs:459-506): 16 f32 lane chains + head / foot.
 * Defined up to lpc_order 15 only: from lag 16 on the vectors are 128 / 256 bytes wide and where `as_simd`
 * splits the buffer depends on the allocator; flacenc_hip_verify_config answers ERR_UNSUPPORTED there, and
 * BAD_CONFIG if both order flags are set. */
#define FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER 64u

/* The kernels' own order as it is.  Without a summation-order flag, blocks of 4096 / 4608 samples at orders up to 12 (the
 * fused kernel's shapes) are CERTIFIED: the kernels keep their own autocorrelation sums where a perturbation bound on the
 * Toeplitz solve, held against the distance of every coefficient to its rounding boundary, says that the quantised LPC
 * parameters equal those of the reference's sequential chains, and recompute the subframe from those chains where it
 * does not (DESIGN.md section 2 lists what of the bound is shown and what is assumed and measured: it is evidence-backed,
 * not a theorem about the floating-point recursion; systems the recursion does not find positive definite are never
 * certified) -- so that every integer output (coefficients, shift, order, residual,
 * Rice partition, bit counts, frame bytes) is the stable build's, at the chunk tree's speed on material that is not
 * strongly tonal.  On material that IS (music, mostly: the certificate's second tier and the recomputation are serial
 * work), launches that return integers only -- no autocorr, no lpc_coefs -- switch, by what the certificate's counters
 * said about the launches before them (a verdict per 4096 subframes or more), to two passes (the reference's chains for every
 * subframe on the matrix cores, then the fused kernel): the same integers at a flat 1.4 x the certified kernel's best
 * time; a choice of speed, never of result ("the order mode by material").
 * EVERY OTHER SHAPE takes those two passes always (round 6; the pass of chains in front of whatever kernel takes the
 * shape -- an order certificate inside the small blocks' kernel was up to 140 x slower on music): there the unflagged R[],
 * coefficients and integers are the stable build's outright.
 * This flag switches the certificate -- and the other shapes' pass of reference chains -- off (blocks of 4096 / 8192 /
 * 16384 at orders from 16 keep the chains: they cost nothing extra there): a valid encoding of the same configuration
 * whose coefficients may differ from the reference's in the last quantisation step on a fraction of a per mille of
 * subframes. */
#define FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER 128u

/* A modifier of FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER for callers that consume the INTEGER outputs only (a drop-in under
 * encode_with_fixed_block_size does: QuantizedParameters, residuals, Rice partitions, frame bytes): shapes whose unflagged
 * order is certified to give the stable build's integers (blocks of 4096 / 4608 samples at orders up to 12) keep it -- no
 * second pass over the samples there -- and every other shape takes the stable build's chains as the flag alone would.
 * What is given up is only the bit-equality of the optional floating-point outputs (autocorr, lpc_coefs). */
#define FLACENC_HIP_FLAG_INTEGER_PARITY_ONLY 256u

/* where the caller's sample / output buffers live */
#define FLACENC_HIP_MEM_HOST 0
#define FLACENC_HIP_MEM_DEVICE 1

typedef struct flacenc_hip_handle flacenc_hip_handle;

/* The fields of config::Qlpc (src/config.rs:271-288) and config::Prc
 * (src/config.rs:211-214) that parameterise the path.  Defaults: order 10,
 * precision 15, Tukey(0.4), max_parameter 30 (src/constant.rs:109-115,
 * src/config.rs:216-221).
 * `use_direct_mse` / `mae_optimization_steps` (src/config.rs:280, :285; both off by default) select the
 * reference's EXPERIMENTAL estimators in perform_qlpc (src/coding.rs:333-351): the covariance-method LPC of
 * LpcEstimator::weighted_lpc_with_direct_mse (src/lpc.rs:853-903: weighted_lagged_outer_prod_sum :573-600, a
 * Cholesky solve with a doubling diagonal regulariser) and, with steps > 0, its IRLS re-weighting towards the
 * mean absolute error (lpc_with_irls_mae, :814-850; steps is ignored unless use_direct_mse is set, as there).
 * The reference's stable `Verify` rejects both switches outside its `experimental` feature (config.rs:302-326);
 * this ABI follows the experimental build.  Its linear solver is nalgebra's (not part of the reference tree):
 * results equal the oracle's restatement of nalgebra 0.32's published algorithm bit for bit, which is as far
 * as this estimator can be pinned.  mae_optimization_steps is limited to FLACENC_HIP_MAX_MAE_STEPS. */
typedef struct flacenc_hip_qlpc_config {
  uint32_t lpc_order;          /* 1..=24 (..=32 with ALLOW_ORDER_32) */
  uint32_t quant_precision;    /* 1..=15 */
  uint32_t window_type;        /* FLACENC_HIP_WINDOW_* */
  float tukey_alpha;           /* 0.0..=1.0 */
  uint32_t max_rice_parameter; /* ..=30 */
  uint32_t flags;
  uint32_t use_direct_mse;         /* 0 / 1 */
  uint32_t mae_optimization_steps; /* 0..=FLACENC_HIP_MAX_MAE_STEPS */
} flacenc_hip_qlpc_config;

/* One record per analysed subframe: everything `estimated_qlpc`
 * (src/coding.rs:360-381) returns inside SubFrame::Lpc except the residual
 * samples (written separately) and the warm-up samples (= the first `order`
 * input samples).  Fixed size (352 bytes) so that records can be all-gathered
 * across GPUs as-is.
 *   coefs/order/shift/precision = component::QuantizedParameters
 *                                 (src/component/datatype.rs:2164-2170)
 *   rice_order/rice_params      = component::Residual partition_order / rice_params
 *                                 (src/component/datatype.rs:2269-2284)
 *   code_bits                   = rice::PrcParameter::code_bits (src/rice.rs:220-224)
 *   sum_quotients               = Residual::sum_quotients (datatype.rs:2325-2331)
 *   subframe_bits               = BitRepr for Lpc::count_bits (src/component/bitrepr.rs:492-499)
 */
typedef struct flacenc_hip_subframe_params {
  int16_t coefs[32];
  uint8_t order;
  int8_t shift;
  uint8_t precision;
  uint8_t rice_order;
  int32_t status;
  uint64_t code_bits;
  uint64_t subframe_bits;
  uint64_t sum_quotients;
  uint8_t rice_params[FLACENC_HIP_MAX_RICE_PARTITIONS];
} flacenc_hip_subframe_params;

/* ---- lifetime --------------------------------------------------------- */
int flacenc_hip_abi_version(void);
int flacenc_hip_device_count(void);
/* Replaces the per-thread scratch set-up of the reference (`reusable!`
 * LPC_ESTIMATOR src/lpc.rs:916, WINDOW_CACHE :219, QLPC_ERROR_BUFFER
 * src/coding.rs:353, PRC_FINDER src/rice.rs:301). */
int flacenc_hip_create(flacenc_hip_handle** out, int device_id);
void flacenc_hip_destroy(flacenc_hip_handle* h);
const char* flacenc_hip_last_error(const flacenc_hip_handle* h);

/* config::Qlpc::verify + config::Prc::verify, src/config.rs:302-326, 224-229 */
int flacenc_hip_verify_config(const flacenc_hip_qlpc_config* cfg);

/* lpc::window_weights (src/lpc.rs:96-120), evaluated on the host in f32 with
 * libm cosf exactly as the reference does; this is the table the kernels use. */
int flacenc_hip_window_weights(const flacenc_hip_qlpc_config* cfg, uint32_t block_size, float* out);

/* ---- the hot path ----------------------------------------------------- */
/*
 * Batched `estimated_qlpc` (src/coding.rs:360-381): for k in 0..n_subframes the
 * subframe is the `block_size` samples at `samples + k*stride` (the layout of
 * FrameBuf::channel_slice, src/source.rs:251-253, batched).  Requires
 * 64 <= block_size <= 32767 (blocks < 64 never reach the path, coding.rs:396; the frame-level calls
 * flacenc_hip_encode_[stereo_]frames / pack_* take them -- 1 <= block_size -- and run encode_subframe's
 * `too_short` branch: Constant or Verbatim only).
 *   bps[k]        bits per sample of subframe k (8..=25; side channels carry +1,
 *                 src/coding.rs:444); only enters subframe_bits.
 *   params[k]     output record (see above)
 *   residual      output, subframe k at residual + k*residual_stride; first
 *                 `order` slots are zero (src/lpc.rs:349)
 *   autocorr      optional [n_subframes][33] f64: R[0..=lpc_order] (src/lpc.rs:780-785)
 *   lpc_coefs     optional [n_subframes][32] f64: unquantised coefficients (src/lpc.rs:792-796)
 *   memory_kind   FLACENC_HIP_MEM_HOST: all pointers are host memory, the call
 *                 stages through the handle's device scratch and returns after
 *                 the results are back on the host.
 *                 FLACENC_HIP_MEM_DEVICE: all pointers are device memory on the
 *                 handle's GPU; the call returns after the kernel completed.
 */
int flacenc_hip_qlpc_batch(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg,
                           const int32_t* samples, size_t n_subframes, uint32_t block_size,
                           size_t stride, const uint8_t* bps,
                           flacenc_hip_subframe_params* params, int32_t* residual,
                           size_t residual_stride, double* autocorr, double* lpc_coefs,
                           int memory_kind);

/* Same with device pointers only, enqueued on `stream` (a hipStream_t; NULL =
 * HIP's default stream) without synchronising. */
int flacenc_hip_qlpc_batch_async(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg,
                                 const int32_t* samples, size_t n_subframes, uint32_t block_size,
                                 size_t stride, const uint8_t* bps,
                                 flacenc_hip_subframe_params* params, int32_t* residual,
                                 size_t residual_stride, double* autocorr, double* lpc_coefs,
                                 void* stream);

/*
 * The four `estimated_qlpc` calls `encode_frame` makes for a 2-channel frame
 * (src/coding.rs:530-544): encode_frame_impl(Independent(2)) on L and R, then
 * try_stereo_coding's MidSide frame on M = (l + r) >> 1 and S = l - r
 * (src/coding.rs:476-491).  `frames` is batched FrameBuf layout
 * (src/source.rs:115-127): channel c of frame f at frames + (2f + c)*stride.
 * M and S are formed on the GPU.  Outputs are indexed 4f + {0:L, 1:R, 2:M, 3:S};
 * the S record's subframe_bits uses bits_per_sample + 1 (src/coding.rs:444).
 * This is the candidate-level call: all four records and residual rows come back and the choice between
 * L+R / L+S / R+S / M+S (src/coding.rs:493-522) is the caller's.  flacenc_hip_encode_stereo_frames below makes
 * it on the GPU (with the fixed-LPC candidate) and returns only the two chosen rows.
 */
int flacenc_hip_stereo_qlpc_batch(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg,
                                  const int32_t* frames, size_t n_frames, uint32_t block_size,
                                  size_t stride, uint32_t bits_per_sample,
                                  flacenc_hip_subframe_params* params, int32_t* residual,
                                  size_t residual_stride, int memory_kind);
int flacenc_hip_stereo_qlpc_batch_async(flacenc_hip_handle* h, const flacenc_hip_qlpc_config* cfg,
                                        const int32_t* frames, size_t n_frames, uint32_t block_size,
                                        size_t stride, uint32_t bits_per_sample,
                                        flacenc_hip_subframe_params* params, int32_t* residual,
                                        size_t residual_stride, void* stream);

/* ---- encode_frame for 2-channel frames, decision on the device ------------------------ */
/*
 * config::Encoder fields that steer `encode_frame` (src/coding.rs:530-544): SubFrameCoding's
 * candidate switches (src/config.rs:167-183), its config::Fixed (src/config.rs:236-244: max_order
 * and OrderSel, :400-409) and StereoCoding (src/config.rs:137-144).  Reference defaults: every
 * switch on, fixed_max_order 4, ApproxEnt with 16 partitions (src/constant.rs:35, :95).
 */
#define FLACENC_HIP_ORDERSEL_BITCOUNT 0  /* OrderSel::BitCount: code every order, count the bits */
#define FLACENC_HIP_ORDERSEL_APPROXENT 1 /* OrderSel::ApproxEnt { partitions } (the default) */
/* ApproxEnt feeds estimate_entropy (src/coding.rs:200-227) the partition's sum of |e| as an f32.  Here that
 * is the exact integer sum rounded to f32 once; the reference accumulates in f32 (find_sum_abs_f32,
 * src/arrayutils.rs:496: one sequential chain in the stable build, 16 lanes in simd-nightly).  All three
 * coincide while the sum stays below 2^24 and differ by f32 rounding above it (24-bit material, loud 16-bit
 * material at higher fixed orders): without FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER the selected fixed order --
 * and hence the frame bytes -- are then not guaranteed identical to a CPU build's; with it the sums are the
 * stable build's sequential chains, bit for bit.  BitCount has no such dependence. */

typedef struct flacenc_hip_frame_config {
  flacenc_hip_qlpc_config qlpc;
  uint32_t use_constant;
  uint32_t use_fixed;
  uint32_t use_lpc;
  uint32_t use_leftside;
  uint32_t use_rightside;
  uint32_t use_midside;
  uint32_t fixed_max_order;  /* 0..=4 */
  uint32_t fixed_order_sel;  /* FLACENC_HIP_ORDERSEL_* */
  uint32_t fixed_partitions; /* ApproxEnt.partitions */
  uint32_t reserved;
} flacenc_hip_frame_config;

#define FLACENC_HIP_KIND_CONSTANT 0 /* SubFrame::Constant */
#define FLACENC_HIP_KIND_VERBATIM 1 /* SubFrame::Verbatim */
#define FLACENC_HIP_KIND_FIXED 2    /* SubFrame::FixedLpc */
#define FLACENC_HIP_KIND_LPC 3      /* SubFrame::Lpc */

/* What `encode_frame` decides for one stereo frame: `try_stereo_coding`'s channel assignment
 * (src/coding.rs:493-522; 0 Independent(2), 1 LeftSide, 2 RightSide, 3 MidSide), and for each of
 * the two output channels (ChannelAssignment::select_channels, datatype.rs:1173-1185) which of
 * L, R, M, S it is (`role` 0..3), which SubFrame variant `encode_subframe` picked (`kind`), the
 * Constant's value, and the predictor record when kind == LPC or FIXED.  A FixedLpc subframe's
 * record holds FIXED_LPC_COEFS[order] (src/component/decode.rs:179-185) with shift 0 and
 * precision 0, the Rice partition of its error signal, and FixedLpc::count_bits
 * (src/component/bitrepr.rs:473-477) in subframe_bits.  `bits` are SubFrame::count_bits of the
 * four candidates L, R, M, S after encode_subframe. */
typedef struct flacenc_hip_stereo_frame_result {
  uint8_t channel_assignment;
  uint8_t kind[2];
  uint8_t role[2];
  uint8_t analysis_status; /* OR of the FLACENC_HIP_SUBFRAME_* bits of the four LPC analyses (L, R, M, S): non-zero
                              where the reference panics (lpc.rs:646, :786-799); the frame is still valid FLAC --
                              the affected candidate was dropped -- but a drop-in should raise */
  uint8_t pad[2];
  int32_t dc_offset[2];
  uint64_t bits[4];
  flacenc_hip_subframe_params lpc[2];
} flacenc_hip_stereo_frame_result;

/*
 * `encode_frame` (src/coding.rs:530-544) for a batch of 2-channel frames with the whole decision
 * on the GPU: L, R, M, S analysed, `encode_subframe` (coding.rs:384-418) applied to each,
 * `try_stereo_coding` picks the assignment, and only the two chosen residual rows are written:
 * residual + (2f + c)*residual_stride for output channel c of frame f (the LPC residual or the
 * fixed-LPC error signal, warm-up slots zero; all zeros for Constant / Verbatim).
 * With use_fixed the other input of encode_subframe, `fixed_lpc` (src/coding.rs:298-331:
 * reset_fixed_lpc_errors :182-197, estimate_entropy :200-227, select_order_and_encode_residual
 * :230-288), runs on the GPU too, so the whole default-configuration decision is on the device.
 * Block size 4096 with lpc_order <= 12, 16-byte aligned rows and power-of-two ApproxEnt.partitions
 * runs as ONE fused kernel (a wave per candidate, samples read from HBM once, losing candidates never
 * leave the CU); so do blocks of 256 / 512 / 1024 / 2048 and 288 / 576 / 1152 / 2304 samples (4 .. 32 finest Rice
 * partitions: several frames per workgroup) with the ApproxEnt selector and estimator partitions of a quarter,
 * half or whole number of those; blocks of 8192 / 16384 samples (and 4096 from order 13) take two analysing
 * passes and a deciding store pass; every other shape (any block size 64..32767, any order, any partition
 * count) runs the candidate batches into handle scratch followed by a controller kernel -- same outputs.
 */
/* Precondition (the reference checks it in FrameBuf::verify_samples, src/source.rs:262-275, before the path
 * is reached): every sample lies in [-2^(bits_per_sample-1), 2^(bits_per_sample-1)).  The frame entry points
 * do not re-check it; out-of-range samples give frames whose warm-up / Verbatim fields are truncated to
 * bits_per_sample bits.  The same precondition holds for every entry point that takes a width (`bps` /
 * `bits_per_sample`, 8..25 bits with the side channel): the fixed-LPC order selector's per-lane sums (v_sad_u32 on
 * biased values) are exact for such samples only, and out-of-width input may select a different fixed order on
 * power-of-two blocks than on ragged ones.  The big-block residual kernel checks each row's extremes against its
 * declared width (its byte planes carry no more) and sends an out-of-width subframe to the generic kernel instead. */
int flacenc_hip_encode_stereo_frames(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                     const int32_t* frames, size_t n_frames, uint32_t block_size,
                                     size_t stride, uint32_t bits_per_sample,
                                     flacenc_hip_stereo_frame_result* results, int32_t* residual,
                                     size_t residual_stride, int memory_kind);
int flacenc_hip_encode_stereo_frames_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                           const int32_t* frames, size_t n_frames, uint32_t block_size,
                                           size_t stride, uint32_t bits_per_sample,
                                           flacenc_hip_stereo_frame_result* results, int32_t* residual,
                                           size_t residual_stride, void* stream);

/* ---- fixed_lpc as a stand-alone batch (any block size, any channel count) ---------------- */
#define FLACENC_HIP_LAYOUT_SUBFRAMES 0     /* unit = one subframe at samples + k*stride */
#define FLACENC_HIP_LAYOUT_STEREO_FRAMES 1 /* unit = one 2-channel frame; outputs 4f + {L, R, M, S} */
/*
 * `fixed_lpc` (src/coding.rs:298-331) for a batch: reset_fixed_lpc_errors (:182-197), the order
 * selector of cfg->fixed_* (select_order_and_encode_residual :230-288; ApproxEnt partitions may
 * be any 1..=64 here) and the Rice-coded error signal of the selected order.  Only the fixed_* and
 * qlpc.max_rice_parameter fields of `cfg` are used.  Per subframe:
 *   params[k]        order = selected fixed order, coefs = FIXED_LPC_COEFS[order]
 *                    (src/component/decode.rs:179-185), shift 0, precision 0, the Rice partition,
 *                    code_bits, sum_quotients, subframe_bits = FixedLpc::count_bits (bitrepr.rs:473-477)
 *   residual         the order's error signal, first `order` slots zero
 *   selector_keys[k] optional: the selector's key of the selected order (estimate_entropy +
 *                    bps*order, or bps*order + code_bits for BitCount).  `fixed_lpc` returns Some(..)
 *                    iff this key is < baseline_bits (coding.rs:262, :284): that comparison, and
 *                    encode_subframe's choice (coding.rs:384-418), stay with the caller.
 *   bps              per-subframe bits per sample (SUBFRAMES layout; NULL -> bits_per_sample);
 *                    STEREO_FRAMES uses bits_per_sample (+1 for the side channel, coding.rs:444).
 * This is the general-shape companion of flacenc_hip_encode_stereo_frames, which fuses the same
 * computation into the frame decision for block_size 4096.
 */
int flacenc_hip_fixed_lpc_batch(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                const int32_t* samples, size_t n_units, uint32_t block_size, size_t stride,
                                const uint8_t* bps, uint32_t bits_per_sample, int layout,
                                flacenc_hip_subframe_params* params, int32_t* residual, size_t residual_stride,
                                uint64_t* selector_keys, int memory_kind);
int flacenc_hip_fixed_lpc_batch_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                      const int32_t* samples, size_t n_units, uint32_t block_size,
                                      size_t stride, const uint8_t* bps, uint32_t bits_per_sample, int layout,
                                      flacenc_hip_subframe_params* params, int32_t* residual,
                                      size_t residual_stride, uint64_t* selector_keys, void* stream);

/* ---- Frame::write on the GPU ---------------------------------------- */
/*
 * BitRepr::write for the frames flacenc_hip_encode_stereo_frames decided (src/component/bitrepr.rs:
 * Frame :289-319, FrameHeader :373-419 with fixed blocking and FrameOffset::Frame as
 * encode_fixed_size_frame sets it (src/coding.rs:581-606), Constant :449-454, Verbatim :463-470,
 * FixedLpc :479-487, Lpc :501-527, Residual :550-597; CRC_8_SMBUS over the header, CRC_16_UMTS over
 * the byte-aligned frame, :39-40).  Inputs are the outputs of flacenc_hip_encode_stereo_frames plus
 * its input samples (warm-up and Verbatim bodies); frame f gets frame number
 * first_frame_number + f*frame_number_step (step > 1 when frames were dealt round-robin over GPUs).
 *   out       frame f's bytes at out + f*out_stride; out_stride >= flacenc_hip_stereo_frame_bytes_bound
 *             (a multiple of 16; out 16-byte aligned); bytes beyond out_len[f] are unspecified
 *   out_len   [n_frames] byte length of each frame = Frame::count_bits / 8 (bitrepr.rs:275-287)
 * The frame is assembled in LDS: frames whose worst case (flacenc_hip_*_frame_bytes_bound) exceeds
 * 150 KiB -- e.g. 8 channels x 16384 samples x 24 bits -- are FLACENC_HIP_ERR_UNSUPPORTED.
 * sample_rate / bits_per_sample go into the header specs exactly as encode_frame_impl chooses them
 * (src/coding.rs:431-436); a rate or size without a code becomes "Unspecified".
 */
size_t flacenc_hip_stereo_frame_bytes_bound(uint32_t block_size, uint32_t bits_per_sample);
int flacenc_hip_pack_stereo_frames(flacenc_hip_handle* h, const int32_t* frames, size_t n_frames,
                                   uint32_t block_size, size_t stride,
                                   const flacenc_hip_stereo_frame_result* results, const int32_t* residual,
                                   size_t residual_stride, uint32_t bits_per_sample, uint32_t sample_rate,
                                   uint32_t first_frame_number, uint32_t frame_number_step, uint8_t* out,
                                   size_t out_stride, uint32_t* out_len, int memory_kind);
int flacenc_hip_pack_stereo_frames_async(flacenc_hip_handle* h, const int32_t* frames, size_t n_frames,
                                         uint32_t block_size, size_t stride,
                                         const flacenc_hip_stereo_frame_result* results, const int32_t* residual,
                                         size_t residual_stride, uint32_t bits_per_sample, uint32_t sample_rate,
                                         uint32_t first_frame_number, uint32_t frame_number_step, uint8_t* out,
                                         size_t out_stride, uint32_t* out_len, void* stream);

/* ---- encode_frame / Frame::write for Independent(channels) frames: mono and multi-channel -- */
/* What encode_subframe (src/coding.rs:384-418) decided for one channel of an Independent(n) frame
 * (encode_frame for channels != 2, src/coding.rs:537-541): the SubFrame variant, the Constant's
 * value, SubFrame::count_bits and the predictor record when kind == LPC or FIXED.  368 bytes. */
typedef struct flacenc_hip_channel_result {
  uint8_t kind; /* FLACENC_HIP_KIND_* */
  uint8_t analysis_status; /* FLACENC_HIP_SUBFRAME_* bits of this channel's LPC analysis (see above) */
  uint8_t pad[2];
  int32_t dc_offset;
  uint64_t bits;
  flacenc_hip_subframe_params params;
} flacenc_hip_channel_result;

/*
 * encode_frame for frames of 1..8 independent channels (no stereo decorrelation: 2-channel
 * streams should use flacenc_hip_encode_stereo_frames).  `frames` is batched FrameBuf layout:
 * channel c of frame f at frames + (f*channels + c)*stride; results and residual rows use the same
 * index f*channels + c.  Candidates are computed as by flacenc_hip_qlpc_batch /
 * flacenc_hip_fixed_lpc_batch (handle scratch), then encode_subframe's choice runs on the GPU.
 */
int flacenc_hip_encode_frames(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                              const int32_t* frames, size_t n_frames, uint32_t channels, uint32_t block_size,
                              size_t stride, uint32_t bits_per_sample, flacenc_hip_channel_result* results,
                              int32_t* residual, size_t residual_stride, int memory_kind);
int flacenc_hip_encode_frames_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                    const int32_t* frames, size_t n_frames, uint32_t channels,
                                    uint32_t block_size, size_t stride, uint32_t bits_per_sample,
                                    flacenc_hip_channel_result* results, int32_t* residual,
                                    size_t residual_stride, void* stream);
/* Frame::write for those frames; arguments as flacenc_hip_pack_stereo_frames. */
size_t flacenc_hip_frame_bytes_bound(uint32_t channels, uint32_t block_size, uint32_t bits_per_sample);
int flacenc_hip_pack_frames(flacenc_hip_handle* h, const int32_t* frames, size_t n_frames, uint32_t channels,
                            uint32_t block_size, size_t stride, const flacenc_hip_channel_result* results,
                            const int32_t* residual, size_t residual_stride, uint32_t bits_per_sample,
                            uint32_t sample_rate, uint32_t first_frame_number, uint32_t frame_number_step,
                            uint8_t* out, size_t out_stride, uint32_t* out_len, int memory_kind);
int flacenc_hip_pack_frames_async(flacenc_hip_handle* h, const int32_t* frames, size_t n_frames, uint32_t channels,
                                  uint32_t block_size, size_t stride, const flacenc_hip_channel_result* results,
                                  const int32_t* residual, size_t residual_stride, uint32_t bits_per_sample,
                                  uint32_t sample_rate, uint32_t first_frame_number, uint32_t frame_number_step,
                                  uint8_t* out, size_t out_stride, uint32_t* out_len, void* stream);

/*
 * encode_frame + Frame::write in one call, device pointers: PCM frames in, FLAC frame bytes out
 * (= flacenc_hip_encode_stereo_frames_async followed by flacenc_hip_pack_stereo_frames_async, same
 * arguments and outputs, no residual rows): the residual rows live in handle scratch between the two
 * kernels.  With FLACENC_HIP_FLAG_FUSED_PACK, block size 4096 and LPC order <= 12 it is ONE kernel: the
 * chosen residuals go from the deciding wave's registers straight into the frame's bit buffer in LDS and
 * never touch HBM (6 instead of 14 bytes of traffic per input sample, but fewer resident waves: slower).
 */
int flacenc_hip_encode_pack_stereo_frames_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                                const int32_t* frames, size_t n_frames, uint32_t block_size,
                                                size_t stride, uint32_t bits_per_sample, uint32_t sample_rate,
                                                uint32_t first_frame_number, uint32_t frame_number_step,
                                                flacenc_hip_stereo_frame_result* results, uint8_t* out,
                                                size_t out_stride, uint32_t* out_len, void* stream);

/* The same for Independent(channels) frames: flacenc_hip_encode_frames_async +
 * flacenc_hip_pack_frames_async with the residual rows in handle scratch. */
int flacenc_hip_encode_pack_frames_async(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg,
                                         const int32_t* frames, size_t n_frames, uint32_t channels,
                                         uint32_t block_size, size_t stride, uint32_t bits_per_sample,
                                         uint32_t sample_rate, uint32_t first_frame_number,
                                         uint32_t frame_number_step, flacenc_hip_channel_result* results,
                                         uint8_t* out, size_t out_stride, uint32_t* out_len, void* stream);

/* Frame::count_bits / 8 (src/component/bitrepr.rs:275-287) of every frame from the decision records
 * alone: exactly the out_len flacenc_hip_pack_stereo_frames will produce.  Device pointers.  These
 * 4 bytes per frame are all an ordered multi-GPU gather has to exchange to place every frame in
 * the output stream (the role of ParSink, src/par.rs:67-95); the bytes themselves stay local. */
int flacenc_hip_stereo_frame_lengths_async(flacenc_hip_handle* h, const flacenc_hip_stereo_frame_result* results,
                                           size_t n_frames, uint32_t block_size, uint32_t bits_per_sample,
                                           uint32_t sample_rate, uint32_t first_frame_number,
                                           uint32_t frame_number_step, uint32_t* out_len, void* stream);

/* ParSink's reordering (src/par.rs:67-95: finished frames are put back into frame-number order for the
 * writer) for packed frames in HBM: frame i = lengths[i] bytes at src + src_offsets[i] is copied to
 * dst + dst_offsets[i].  Device pointers, any byte alignment.  Used twice by the multi-GPU gather of
 * packed frames: to compact a rank's strided pack output into one contiguous run for the RCCL
 * all-gather, and to place every rank's frames at their stream offsets afterwards. */
int flacenc_hip_place_frames_async(flacenc_hip_handle* h, const uint8_t* src, const uint64_t* src_offsets,
                                   const uint32_t* lengths, size_t n_frames, uint8_t* dst,
                                   const uint64_t* dst_offsets, void* stream);

/* The ordered gather's two device steps (what ParSink, src/par.rs:67-95, does with a BTreeMap on one host).
 * flacenc_hip_stereo_frame_wire_async turns this rank's decision records into their wire form -- the 48 bytes of frame
 * fields and, of each of the two 352-byte subframe records, the 96 fixed bytes + the first 2^finest_order Rice
 * parameters (src/rice.rs:157-165; the rest of rice_params[256] is always zero): flacenc_hip_frame_wire_bytes(block)
 * bytes, 368 of 752 for blocks of 4096 -- at wire + f * wire_stride, and in the same pass writes the frames' byte
 * lengths (flacenc_hip_stereo_frame_lengths_async's values) to out_len when that is not NULL.
 * flacenc_hip_stream_offsets_async takes the all-gathered lengths as the collective delivers them -- rank-major,
 * gathered_lengths[r * ceil(n / world) + j] = stream frame j * world + r, shorter ranks zero-padded; world = 1 is
 * plain stream order -- and writes, in stream order, offsets[f] = header_bytes + sum of the lengths of frames below f,
 * lengths_stream[f] (optional, NULL to skip) and total[0] = header_bytes + sum of all lengths. */
size_t flacenc_hip_frame_wire_bytes(uint32_t block_size);
int flacenc_hip_stereo_frame_wire_async(flacenc_hip_handle* h, const flacenc_hip_stereo_frame_result* results,
                                        size_t n_frames, uint32_t block_size, uint32_t bits_per_sample,
                                        uint32_t sample_rate, uint32_t first_frame_number, uint32_t frame_number_step,
                                        uint8_t* wire, size_t wire_stride, uint32_t* out_len, void* stream);
int flacenc_hip_stream_offsets_async(flacenc_hip_handle* h, const uint32_t* gathered_lengths, size_t n_frames_total,
                                     uint32_t world, uint64_t header_bytes, uint32_t* lengths_stream,
                                     uint64_t* offsets, uint64_t* total, void* stream);

/* ---- the collective of the ordered gather (one process per GPU) -------------------------------------------- */
/*
 * ParSink (src/par.rs:67-95) re-orders the frames its worker threads finish; with one PROCESS per GPU (frame f on
 * rank f mod world) that re-ordering needs one exchange, and this is it: an RCCL communicator owned by the handle
 * and an all-gather of fixed-size per-frame records on the caller's stream.  librccl is opened at run time on the
 * first of these calls (ERR_UNSUPPORTED when it cannot be found; FLACENC_HIP_RCCL names another path) -- a
 * single-GPU drop-in never loads it.
 *   flacenc_hip_comm_unique_id   ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to the other ranks by
 *                                whatever means the host has (a file, a socket, MPI, torch.distributed's store)
 *   flacenc_hip_comm_create      ncclCommInitRank on the handle's device; collective over all `world` ranks
 *   flacenc_hip_comm_info        rank / world of the handle's communicator (world = 0: none)
 *   flacenc_hip_allgather_async  ncclAllGather of bytes_per_rank bytes: recv[r * bytes_per_rank ..] = rank r's send
 *   flacenc_hip_allgather_records_async
 *                                the ordered gather's exchange for n_total frames dealt round-robin: `local` holds this
 *                                rank's n_local = ceil((n_total - rank) / world) records of record_bytes each (wire
 *                                records from flacenc_hip_stereo_frame_wire_async, or its 4-byte lengths); `gathered`
 *                                receives world * ceil(n_total / world) records rank-major -- record r * per_rank + j =
 *                                stream frame j * world + r, ranks one frame short zero-padded -- the layout
 *                                flacenc_hip_stream_offsets_async reads.  `local` may be the rank's own slot of
 *                                `gathered` (in place).  Device pointers; enqueued on `stream`.
 */
#define FLACENC_HIP_COMM_ID_BYTES 128
int flacenc_hip_comm_unique_id(uint8_t id[FLACENC_HIP_COMM_ID_BYTES]);
int flacenc_hip_comm_create(flacenc_hip_handle* h, const uint8_t id[FLACENC_HIP_COMM_ID_BYTES], int rank, int world);
int flacenc_hip_comm_destroy(flacenc_hip_handle* h);
int flacenc_hip_comm_info(flacenc_hip_handle* h, int* rank, int* world);
int flacenc_hip_allgather_async(flacenc_hip_handle* h, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int flacenc_hip_allgather_records_async(flacenc_hip_handle* h, const void* local, size_t n_local, size_t n_total,
                                        size_t record_bytes, void* gathered, void* stream);

/* ---- input side ------------------------------------------------------- */
/*
 * FrameBuf::fill_le_bytes (src/source.rs:288-298) for a run of consecutive frames of one stream, on
 * the GPU: le_bytes_to_i32s (src/arrayutils.rs:273-290; little-endian samples of 1..4 bytes, sign-
 * extended) and deinterleave (src/arrayutils.rs:248-264).  `bytes` holds `total_samples` inter-channel
 * samples of packed interleaved PCM starting at the first frame of the run; frame f, channel c goes
 * to frames + (f*channels + c)*stride, zero-filled beyond total_samples (the short last block).
 * Uploading packed 16- / 24-bit PCM and widening here halves (or better) the PCIe traffic of the
 * host-pointer paths; the MD5 of the input stays with the caller (src/source.rs:406-428).
 */
int flacenc_hip_fill_le_bytes(flacenc_hip_handle* h, const uint8_t* bytes, uint64_t total_samples,
                              uint32_t channels, uint32_t bytes_per_sample, size_t n_frames, uint32_t block_size,
                              int32_t* frames, size_t stride, int memory_kind);
int flacenc_hip_fill_le_bytes_async(flacenc_hip_handle* h, const uint8_t* bytes, uint64_t total_samples,
                                    uint32_t channels, uint32_t bytes_per_sample, size_t n_frames,
                                    uint32_t block_size, int32_t* frames, size_t stride, void* stream);

/* ---- host-memory streaming path: what a drop-in under encode_with_fixed_block_size experiences ---- */
/*
 * Packed interleaved little-endian 2-channel PCM in host memory -> FLAC frame bytes in host memory, for a
 * whole stream (or a long run of it) in one call: the feed loop of the reference's par-mode encoder
 * (src/par.rs:288-325: fill a FrameBuf, hand it to a worker, collect the frame) with the GPU as the
 * worker pool.  The run is cut into chunks of whole frames; per chunk the packed PCM (2..3 bytes per sample
 * instead of 4) goes host -> device through pinned staging on a copy stream, flacenc_hip_fill_le_bytes +
 * flacenc_hip_encode_pack_stereo_frames run on the compute stream, the frames are compacted on the device
 * (flacenc_hip_place_frames) and exactly their bytes come back on a second copy stream -- two slots, so the
 * upload of chunk k + 1, the analysis of chunk k and the download of chunk k - 1 overlap.
 *   pcm            total_samples inter-channel samples, bytes_per_sample each, L R L R ...
 *   block_size     every frame has block_size samples except the last (total_samples % block_size; a last
 *                  block of fewer than 64 samples is coded as the reference codes it: encode_subframe skips
 *                  both predictors, `too_short` src/coding.rs:389-418, and emits Constant or Verbatim)
 *   out/out_len    frames back to back in frame order (what the stream writer appends after the
 *                  metadata blocks) and each frame's byte length; out_total = bytes written
 * `pcm` / `out` may be ordinary (pageable) memory -- then they are staged through the handle's pinned
 * buffers with a host memcpy -- or memory from flacenc_hip_host_alloc, which is transferred directly.
 * The MD5 of the input and STREAMINFO stay with the caller (src/source.rs:406-428).
 */
int flacenc_hip_encode_pcm_stereo(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg, const uint8_t* pcm,
                                  uint64_t total_samples, uint32_t bytes_per_sample, uint32_t bits_per_sample,
                                  uint32_t block_size, uint32_t sample_rate, uint32_t first_frame_number,
                                  uint32_t frame_number_step, uint8_t* out, size_t out_capacity, uint32_t* out_len,
                                  uint64_t* out_total);
/* The same for any channel count 1..=8 (samples interleaved c0 c1 .. c0 c1 ..): 2 channels as above, other
 * counts as Independent(channels) frames (src/coding.rs:537-541: flacenc_hip_encode_pack_frames_async). */
int flacenc_hip_encode_pcm(flacenc_hip_handle* h, const flacenc_hip_frame_config* cfg, const uint8_t* pcm,
                           uint64_t total_samples, uint32_t channels, uint32_t bytes_per_sample,
                           uint32_t bits_per_sample, uint32_t block_size, uint32_t sample_rate,
                           uint32_t first_frame_number, uint32_t frame_number_step, uint8_t* out, size_t out_capacity,
                           uint32_t* out_len, uint64_t* out_total);
/* page-locked host memory for the calls above (NULL on failure) */
void* flacenc_hip_host_alloc(size_t bytes);
void flacenc_hip_host_free(void* p);
/* Host threads (the caller's included) that share the staging copies of flacenc_hip_encode_pcm* when `pcm` or
 * `out` is pageable memory; 0 or 1 = the caller's thread alone, default 4.  The helper threads belong to the
 * handle, sleep between calls and end with flacenc_hip_destroy -- the counterpart of the reference's worker
 * threads feeding and draining FrameBufs (src/par.rs:288-325), which only move bytes here. */
int flacenc_hip_set_host_threads(flacenc_hip_handle* h, int threads);

int flacenc_hip_synchronize(flacenc_hip_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* FLACENC_HIP_H_ */
