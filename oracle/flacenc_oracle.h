/*
 * flacenc_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the per-subframe quantised-LPC analysis path of
 * yotarok/flacenc-rs v0.5.1 (src/lpc.rs, src/rice.rs, src/coding.rs, and the
 * bit-count formulas of src/component/bitrepr.rs, the decoder of
 * src/component/decode.rs).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (flacenc-rs_amd/) never links, imports or calls it.
 *
 * Parity pin: the reference is pure Rust and there is no rustc/cargo in the
 * build image, so the reference itself cannot be executed here.  This
 * restatement is pinned by every in-source known-answer test the reference
 * holds for the path (tests/test_oracle_kat.py cites each one); the
 * coefficient-level behaviour on arbitrary audio has no reference-produced
 * vectors to pin against ("parity unpinned" beyond the KATs).
 *
 * All `file:line` citations are relative to /root/reference/.
 */
#ifndef FLACENC_ORACLE_H_
#define FLACENC_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LPC_ORDER 32        /* FLAC limit; reference caps at 24 (src/constant.rs:118) */
#define ORC_REF_MAX_LPC_ORDER 24    /* src/constant.rs:118 */
#define ORC_QLPC_MAX_SHIFT 15       /* src/constant.rs:124 */
#define ORC_QLPC_MIN_SHIFT 0        /* src/constant.rs:131 */
#define ORC_MAX_RICE_PARAMETER 30   /* src/constant.rs:143 */
#define ORC_MAX_RICE_PARTITION_ORDER 15 /* src/constant.rs:146 */
#define ORC_MIN_RICE_PARTITION_SIZE 64  /* src/constant.rs:152 */
#define ORC_MAX_P_TO_BITS ((1u << 27) - 1u) /* src/rice.rs:51 */
#define ORC_MAX_RICE_PARTITIONS 32768

#define ORC_WINDOW_RECTANGLE 0
#define ORC_WINDOW_TUKEY 1

/* autocorrelation summation orders */
#define ORC_ACORR_REFERENCE 0 /* weighted_auto_correlation_nosimd, src/lpc.rs:533-548 */
#define ORC_ACORR_CANONICAL 1 /* the build's unflagged order: 16-sample chunk chains + balanced tree -- except where
                                  orc_default_order_is_stable() says it is ORC_ACORR_REFERENCE, and certified (else
                                  recomputed as ORC_ACORR_REFERENCE) where orc_default_order_is_certified() says so */
#define ORC_ACORR_NIGHTLY 2   /* weighted_auto_correlation_simd, src/lpc.rs:510-531 (aligned buffer) */
/* config::Qlpc::use_direct_mse (src/config.rs:280): covariance-method LPC, src/lpc.rs:853-903; bits 8.. of
 * acorr_order carry config::Qlpc::mae_optimization_steps (src/config.rs:285; IRLS, src/lpc.rs:814-850).
 * Experimental in the reference, solver from nalgebra: parity unpinned (see flacenc_oracle.c). */
#define ORC_ACORR_DIRECT_MSE 3
/* the kernels' own order as it is, without the certificate of orc_default_order_is_certified() (the product's
 * FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER): the fused kernel's lane order on its shapes, the chunk tree elsewhere */
#define ORC_ACORR_CHUNK_TREE 4
/* the 16-sample chunk tree on every shape */
#define ORC_ACORR_GENERIC_TREE 5

/* find_sum_abs_f32 summation orders (src/arrayutils.rs:496-506) */
#define ORC_SUMABS_STABLE 0    /* stable build: one sequential f32 chain */
#define ORC_SUMABS_NIGHTLY 1   /* simd-nightly: 16 f32 lanes + head/foot */
#define ORC_SUMABS_CANONICAL 2 /* the build's definition: exact integer sum, rounded to f32 once */

/* config::OrderSel, src/config.rs:400-409 */
#define ORC_ORDERSEL_BITCOUNT 0
#define ORC_ORDERSEL_APPROXENT 1

/* OR-ed into max_rice_parameter: stop the partitioned-Rice search at the finest order (a build
 * extension mirrored from FLACENC_HIP_FLAG_FINEST_RICE_ORDER; the reference always searches all orders) */
#define ORC_RICE_FINEST_ONLY 0x100u

#define ORC_STATUS_OK 0
#define ORC_STATUS_NONFINITE 1   /* the reference would panic (src/lpc.rs:786-799) */
#define ORC_STATUS_NEG_ENERGY 2  /* the reference would panic (src/lpc.rs:646) */

typedef struct {
  uint32_t lpc_order;          /* config::Qlpc::lpc_order, src/config.rs:273 */
  uint32_t quant_precision;    /* config::Qlpc::quant_precision, src/config.rs:275 */
  uint32_t window_type;        /* config::Window, src/config.rs:344 */
  float tukey_alpha;           /* config::Window::Tukey::alpha */
  uint32_t max_rice_parameter; /* config::Prc::max_parameter, src/config.rs:213 */
  uint32_t acorr_order;        /* ORC_ACORR_* */
} orc_qlpc_config;

/* component::QuantizedParameters, src/component/datatype.rs:2164-2170 */
typedef struct {
  int16_t coefs[ORC_MAX_LPC_ORDER];
  uint32_t order;
  int32_t shift;
  uint32_t precision;
} orc_qparams;

/* rice::PrcParameter (src/rice.rs:220-224) + the derived Residual scalars
 * (src/component/datatype.rs:2269-2284). */
typedef struct {
  uint32_t order;
  uint64_t code_bits;
  uint8_t ps[256 * 128]; /* up to 2^15 partitions */
} orc_prc_parameter;

typedef struct {
  orc_qparams qp;
  uint32_t rice_order;
  uint64_t code_bits;       /* PrcParameter::code_bits (search estimate) */
  uint64_t sum_quotients;   /* Residual::sum_quotients */
  uint64_t sum_rice_params; /* Residual::sum_rice_params */
  uint64_t residual_bits;   /* Residual::count_bits, bitrepr.rs:533-544 */
  uint64_t subframe_bits;   /* Lpc::count_bits, bitrepr.rs:492-499 */
  int32_t status;
  double autocorr[ORC_MAX_LPC_ORDER + 1];
  double lpc_coefs[ORC_MAX_LPC_ORDER];
} orc_qlpc_result;

/* config::Fixed (src/config.rs:236-244) */
typedef struct {
  uint32_t max_order;  /* ..=4, src/constant.rs:95 */
  uint32_t order_sel;  /* ORC_ORDERSEL_* (default ApproxEnt) */
  uint32_t partitions; /* ApproxEnt.partitions, 1..=64, default 16 (src/constant.rs:35, :63) */
  uint32_t sum_mode;   /* ORC_SUMABS_* */
} orc_fixed_config;

/* config::SubFrameCoding (src/config.rs:167-183) + config::StereoCoding (:137-144) */
typedef struct {
  orc_qlpc_config qlpc;
  uint32_t use_constant, use_fixed, use_lpc;
  uint32_t use_leftside, use_rightside, use_midside;
  orc_fixed_config fixed;
} orc_frame_config;

/* what fixed_lpc (src/coding.rs:298-331) produced */
typedef struct {
  int32_t selected;     /* Some(..) / None */
  uint32_t order;       /* the selector's argmin (first minimum) */
  uint64_t estimate[5]; /* selector key per order: ApproxEnt estimate or BitCount bits, + bps*order */
  uint32_t rice_order;
  uint64_t code_bits, sum_quotients, sum_rice_params, residual_bits;
  uint64_t subframe_bits; /* FixedLpc::count_bits, bitrepr.rs:473-477 */
} orc_fixed_result;

/* ---- lpc.rs ---- */
void orc_window_weights(uint32_t window_type, float alpha, size_t len, float* out);
void orc_fill_windowed_signal(const int32_t* signal, const float* window, size_t n, float* out);
void orc_auto_correlation_f64(size_t order, const float* signal, size_t n, double* dest);
void orc_auto_correlation_f32(size_t order, const float* signal, size_t n, float* dest);
void orc_auto_correlation_canonical_f64(size_t order, const float* signal, size_t n, double* dest);
void orc_auto_correlation_lane_order_f64(size_t order, const float* signal, size_t n, double* dest);
int orc_default_order_is_stable(size_t n, size_t lpc_order);
int orc_default_order_is_certified(size_t n, size_t lpc_order);
int orc_default_order_is_two_pass(size_t n, size_t lpc_order);
extern unsigned long orc_cert_stats[3];
int orc_quant_certified(const double* R, const double* a, const double* fwd, size_t P, uint32_t max_abs_s, size_t n,
                        uint32_t precision, int* tier2);
void orc_auto_correlation_nightly_f64(size_t order, const float* signal, size_t n, double* dest,
                                      size_t base_mod);
int orc_symmetric_levinson_f64(const double* coefs, const double* ys, size_t order, double* dest);
int orc_symmetric_levinson_f32(const float* coefs, const float* ys, size_t order, float* dest);
/* tool / test hook: the order certificate's bounds for one subframe (flacenc_oracle.c) */
int orc_certificate_bounds(const int32_t* signal, size_t n, const orc_qlpc_config* cfg, double* corr_out, double* coefs_out,
                           double* da_out, double* tier1_out);
int32_t orc_find_shift(const double* coefs, size_t n, uint32_t precision);
void orc_quantize_parameters(const double* coefs, size_t n, uint32_t precision, orc_qparams* out);
void orc_compute_error(const orc_qparams* qp, const int32_t* signal, size_t n, int32_t* errors);
void orc_weighted_auto_correlation_nosimd_f64(size_t order, const float* signal, size_t n, const float* weight,
                                              double* dest);
void orc_weighted_lagged_outer_prod_sum_f64(size_t order, const float* signal, size_t len, const float* weight,
                                            size_t wshift, double* dest);
int orc_cholesky_solve(const double* mat, size_t n, double* v);
int orc_weighted_lpc_with_direct_mse(const int32_t* signal, size_t n, const orc_qlpc_config* cfg, const float* weight,
                                     double* autocorr_out, double* gram_out, double* coefs_out);
void orc_compute_raw_errors(const int32_t* signal, size_t n, const double* coefs, size_t order, float* errors);
int orc_lpc_with_irls_mae(const int32_t* signal, size_t n, const orc_qlpc_config* cfg, size_t steps,
                          double* autocorr_out, double* coefs_out);
int orc_perform_qlpc(const int32_t* signal, size_t n, const orc_qlpc_config* cfg, double* autocorr_out,
                     double* coefs_out);
int orc_lpc_from_autocorr(const int32_t* signal, size_t n, const orc_qlpc_config* cfg,
                          double* autocorr_out, double* coefs_out);

/* ---- rice.rs ---- */
uint32_t orc_encode_signbit(int32_t v);
int32_t orc_decode_signbit(uint32_t v);
void orc_prc_bit_table_from_errors(const uint32_t* errors, size_t len, uint32_t offset,
                                   uint32_t table[32]);
void orc_prc_minimizer(const uint32_t table[32], uint32_t max_p, uint32_t* p_out,
                       uint32_t* bits_out);
void orc_prc_merge(const uint32_t a[32], const uint32_t b[32], uint32_t offset, uint32_t out[32]);
uint32_t orc_finest_partition_order(size_t size, size_t min_part_size);
void orc_find_partitioned_rice_parameter(const int32_t* signal, size_t n, size_t warmup_length,
                                         uint32_t max_p, orc_prc_parameter* out);

/* ---- coding.rs / bitrepr.rs / decode.rs ---- */
void orc_encode_residual_with_prc_parameter(const int32_t* errors, size_t n, size_t warmup_length,
                                            const orc_prc_parameter* prc, uint32_t* quotients,
                                            uint32_t* remainders, uint64_t* sum_quotients,
                                            uint64_t* sum_rice_params);
uint64_t orc_residual_count_bits(size_t block_size, size_t warmup_length, uint32_t partition_order,
                                 const uint8_t* rice_params, uint64_t sum_quotients,
                                 uint64_t sum_rice_params);
uint64_t orc_lpc_count_bits(uint32_t bits_per_sample, uint32_t order, uint32_t precision,
                            uint64_t residual_bits);
uint64_t orc_verbatim_count_bits(size_t n, uint32_t bits_per_sample);
void orc_decode_residual(size_t block_size, uint32_t partition_order, const uint8_t* rice_params,
                         const uint32_t* quotients, const uint32_t* remainders, int32_t* dest);
void orc_decode_lpc(const int32_t* warm_up, size_t order, const int16_t* coefs, uint32_t shift,
                    const int32_t* residual, size_t n, int32_t* dest);

/* estimated_qlpc, src/coding.rs:360-381.  `errors`, `quotients`, `remainders`
 * are n-element caller buffers (quotients/remainders may be NULL). */
void orc_estimated_qlpc(const int32_t* signal, size_t n, uint32_t bits_per_sample,
                        const orc_qlpc_config* cfg, orc_qlpc_result* res, uint8_t* rice_params,
                        int32_t* errors, uint32_t* quotients, uint32_t* remainders);

/* batch helpers (one subframe = n samples at samples + k*stride) */
typedef struct {
  int16_t coefs[32];
  uint8_t order;
  int8_t shift;
  uint8_t precision;
  uint8_t rice_order;
  int32_t status;
  uint64_t code_bits;
  uint64_t subframe_bits;
  uint64_t sum_quotients;
  uint8_t rice_params[256];
} orc_subframe_record; /* same layout as flacenc_hip_subframe_params (include/flacenc_hip.h) */

void orc_qlpc_batch(const int32_t* samples, size_t n_subframes, size_t n, size_t stride,
                    const uint8_t* bps, const orc_qlpc_config* cfg, orc_subframe_record* recs,
                    int32_t* residual, size_t residual_stride, double* autocorr, double* lpc_coefs,
                    int nthreads);

/* stereo helpers, src/coding.rs:476-484 and src/component/decode.rs:91-103 */
void orc_stereo_to_midside(const int32_t* l, const int32_t* r, size_t n, int32_t* m, int32_t* s);
void orc_midside_to_stereo(const int32_t* m, const int32_t* s, size_t n, int32_t* l, int32_t* r);

/* encode_subframe restricted to {Constant, Verbatim, Lpc} candidates
 * (src/coding.rs:384-418 with use_fixed = false).  Returns the chosen kind:
 * 0 = Constant, 1 = Verbatim, 3 = Lpc; *bits_out = SubFrame::count_bits. */
int orc_is_constant(const int32_t* samples, size_t n);

/* multi-thread timing helper for bench.py's cpu_baseline ("port" kind). */
double orc_bench_qlpc(const int32_t* samples, size_t n_subframes, size_t n, size_t stride,
                      uint32_t bits_per_sample, const orc_qlpc_config* cfg, int nthreads,
                      int repeats);

/* mirrors flacenc_hip_stereo_frame_result (include/flacenc_hip.h) */
typedef struct {
  uint8_t channel_assignment; /* 0 Independent(2), 1 LeftSide, 2 RightSide, 3 MidSide */
  uint8_t kind[2];            /* 0 Constant, 1 Verbatim, 2 FixedLpc, 3 Lpc */
  uint8_t role[2];            /* 0 L, 1 R, 2 M, 3 S */
  uint8_t pad[3];
  int32_t dc_offset[2];
  uint64_t bits[4];
  orc_subframe_record lpc[2];
} orc_stereo_frame_result;

float orc_log2f(float x);
void orc_reset_fixed_lpc_errors(const int32_t* signal, size_t n, int32_t* errors /* [5][n] */);
float orc_find_sum_abs_f32(const int32_t* data, size_t len, int mode, size_t base_mod);
uint64_t orc_estimate_entropy(const int32_t* errors, size_t n, size_t warmup_len, size_t partitions,
                              int mode);
extern const int16_t orc_fixed_lpc_coefs[5][4];
int orc_fixed_lpc(const int32_t* signal, size_t n, uint32_t bps, uint64_t baseline_bits,
                  const orc_fixed_config* fc, uint32_t max_rice_p, orc_fixed_result* res,
                  uint8_t* rice_params, int32_t* errors_out);
int orc_encode_subframe(const int32_t* samples, size_t n, uint32_t bps, const orc_frame_config* fc,
                        uint64_t* bits_out, orc_qlpc_result* lpc, orc_fixed_result* fixed,
                        uint8_t* rice_params, int32_t* errors);
void orc_encode_stereo_frame_cfg(const int32_t* l, const int32_t* r, size_t n, uint32_t bps,
                                 const orc_frame_config* fc, orc_stereo_frame_result* out,
                                 int32_t* residual0, int32_t* residual1);

int orc_encode_subframe_nofixed(const int32_t* samples, size_t n, uint32_t bps, int use_constant,
                                int use_lpc, const orc_qlpc_config* cfg, uint64_t* bits_out,
                                orc_qlpc_result* lpc, uint8_t* rice_params, int32_t* errors);
void orc_encode_stereo_frame(const int32_t* l, const int32_t* r, size_t n, uint32_t bps,
                             const orc_qlpc_config* cfg, int use_constant, int use_lpc,
                             int use_leftside, int use_rightside, int use_midside,
                             orc_stereo_frame_result* out, int32_t* residual0, int32_t* residual1);

/* ---- bit writer, src/component/bitrepr.rs ---- */
typedef struct {
  uint32_t kind; /* 0 Constant, 1 Verbatim, 2 FixedLpc, 3 Lpc */
  uint32_t bps;  /* bits per sample of this subframe (side channel: +1) */
  int32_t dc_offset;
  const int32_t* samples; /* the subframe's input: Verbatim body / warm-up source */
  uint32_t order;
  int32_t shift;
  uint32_t precision;
  const int16_t* coefs;
  uint32_t rice_order;
  const uint8_t* rice_params;
  const int32_t* residual; /* error signal, warm-up slots ignored */
} orc_subframe_desc;

uint8_t orc_crc8(const uint8_t* data, size_t len);
uint16_t orc_crc16(const uint8_t* data, size_t len);
size_t orc_encode_to_utf8like(uint64_t val, uint8_t out[7]);
size_t orc_write_frame_header(uint32_t block_size, uint32_t channel_tag, uint32_t bits_per_sample,
                              uint32_t sample_rate, int variable, uint64_t offset, uint8_t* out);
size_t orc_write_subframe(const orc_subframe_desc* d, size_t n, uint8_t* out, size_t cap);
size_t orc_write_frame(uint32_t block_size, uint32_t channel_assignment, uint32_t nch,
                       uint32_t bits_per_sample, uint32_t sample_rate, uint32_t frame_number,
                       const orc_subframe_desc* subframes, uint8_t* out, size_t cap);
size_t orc_write_stereo_frame(const orc_stereo_frame_result* fr, const int32_t* l, const int32_t* r, size_t n,
                              uint32_t bits_per_sample, uint32_t sample_rate, uint32_t frame_number,
                              const int32_t* residual0, const int32_t* residual1, uint8_t* out, size_t cap);

/* ---- input side, src/arrayutils.rs ---- */
void orc_le_bytes_to_i32s(const uint8_t* bytes, size_t nbytes, int32_t* dest, uint32_t bytes_per_sample);
void orc_deinterleave(const int32_t* interleaved, size_t len, size_t channels, size_t channel_stride,
                      int32_t* dest);

double orc_bench_stereo_qlpc(const int32_t* frames, size_t n_frames, size_t n, size_t stride,
                             uint32_t bits_per_sample, const orc_qlpc_config* cfg, int nthreads,
                             int repeats, uint64_t* checksum_out);

#ifdef __cplusplus
}
#endif
#endif
