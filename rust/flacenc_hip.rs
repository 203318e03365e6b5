//! flacenc_hip.rs -- Rust side of the drop-in boundary (SOURCE ONLY: there is no rustc in the
//! build image, so this file has never been compiled; the executable host mirror is the C++
//! header `flacenc_rs_amd/host/flacenc.hpp`, which follows the same structure, and
//! `tests/test_rust_binding.py` holds the `extern "C"` block and the `#[repr(C)]` structs below to
//! `include/flacenc_hip.h` -- every export, arity, argument type, field and constant).
//!
//! A maintainer adds this module to the crate (`mod gpu;` behind a `hip` cargo feature), links
//! `libflacenc_hip.so` (`build.rs`: `println!("cargo:rustc-link-lib=dylib=flacenc_hip")`), and
//! branches to `gpu::encode_with_fixed_block_size` where `encode_with_fixed_block_size` already
//! branches on `config.multithread` (`src/coding.rs:650-655`).  Three levels, as in the header:
//!
//!   * `encode_with_fixed_block_size`             the whole of `encode_fixed_size_frame` + `Frame::write` on the GPU;
//!                                                frames come back as bytes (`flacenc_hip_encode_pcm`)
//!   * `encode_with_fixed_block_size_components`  `encode_frame`'s decisions on the GPU, `component::SubFrame`s rebuilt
//!                                                on the host (`flacenc_hip_encode_[stereo_]frames` + `*_from_record`)
//!   * `estimated_qlpc_batch`                     candidates only: the controller (`encode_subframe` `src/coding.rs:384`,
//!                                                `try_stereo_coding` `:469`) stays on the host
//!
//! The one change outside this module: `Frame::set_precomputed_bitstream` (see the comment above
//! `encode_with_fixed_block_size` below).

use std::os::raw::{c_char, c_int, c_void};

use crate::component::{
    BlockSizeSpec, ChannelAssignment, Constant, Frame, FrameOffset, Lpc, QuantizedParameters, Residual, SampleRateSpec,
    SampleSizeSpec, Stream, SubFrame, Verbatim,
};
use crate::config;
use crate::error::{EncodeError, SourceError, SourceErrorReason, Verified, VerifyError};
use crate::source::{Context, Fill, FrameBuf, Source};

pub const OK: c_int = 0;
pub const ERR_BAD_CONFIG: c_int = -1;
pub const MEM_HOST: c_int = 0;
pub const MEM_DEVICE: c_int = 1;
/// `flags` of `QlpcConfig` (include/flacenc_hip.h).  The two order flags make the order-sensitive sums of the path
/// the ones of a CPU build, bit for bit: the stable build's (`weighted_auto_correlation_nosimd`, `src/lpc.rs:533-548`;
/// `find_sum_abs_f32` over `slice_as_simd = (data, [], [])`, `src/arrayutils.rs:435-506`) or the `simd-nightly`
/// build's (`weighted_auto_correlation_simd`, `src/lpc.rs:510-531`; LPC orders up to 15).  A crate built with the
/// `simd-nightly` feature would pass `FLAG_NIGHTLY_SUM_ORDER`, any other `FLAG_REFERENCE_SUM_ORDER`, to get the
/// bytes its own CPU path produces.
pub const FLAG_ALLOW_ORDER_32: u32 = 1;
pub const FLAG_REFERENCE_SUM_ORDER: u32 = 32;
pub const FLAG_NIGHTLY_SUM_ORDER: u32 = 64;
/// `FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER`: the kernels' own order without the order certificate.
pub const FLAG_CANONICAL_SUM_ORDER: u32 = 128;
/// `FLACENC_HIP_FLAG_INTEGER_PARITY_ONLY`: with `FLAG_REFERENCE_SUM_ORDER`, shapes whose own order is certified to give the
/// stable build's integers keep it (no second pass over the samples); this binding consumes integers only.
pub const FLAG_INTEGER_PARITY_ONLY: u32 = 256;

/// `flacenc_hip_qlpc_config` (include/flacenc_hip.h): the path's fields of `config::Qlpc` /
/// `config::Prc` (`src/config.rs:271-288`, `211-214`).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct QlpcConfig {
    pub lpc_order: u32,
    pub quant_precision: u32,
    pub window_type: u32, // 0 = Rectangle, 1 = Tukey
    pub tukey_alpha: f32,
    pub max_rice_parameter: u32,
    pub flags: u32,
    pub use_direct_mse: u32,         // config::Qlpc::use_direct_mse (src/config.rs:280, `experimental`)
    pub mae_optimization_steps: u32, // config::Qlpc::mae_optimization_steps (src/config.rs:285)
}

/// `flacenc_hip_subframe_params`: 352 bytes, one per analysed subframe.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct SubframeParams {
    pub coefs: [i16; 32],
    pub order: u8,
    pub shift: i8,
    pub precision: u8,
    pub rice_order: u8,
    pub status: i32,
    pub code_bits: u64,
    pub subframe_bits: u64,
    pub sum_quotients: u64,
    pub rice_params: [u8; 256],
}

/// `flacenc_hip_frame_config`: `config::SubFrameCoding` switches, `config::Fixed`
/// (`src/config.rs:236-244`) and `config::StereoCoding` (`:137-144`).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct FrameConfig {
    pub qlpc: QlpcConfig,
    pub use_constant: u32,
    pub use_fixed: u32,
    pub use_lpc: u32,
    pub use_leftside: u32,
    pub use_rightside: u32,
    pub use_midside: u32,
    pub fixed_max_order: u32,
    pub fixed_order_sel: u32, // 0 = OrderSel::BitCount, 1 = OrderSel::ApproxEnt
    pub fixed_partitions: u32,
    pub reserved: u32,
}

impl FrameConfig {
    /// `order`: whose floating-point summation order the analysis reproduces (`Gpu::sum_order()` of the handle the
    /// configuration is used with; `SumOrder::CrateBuild` = the bytes of the build this file is compiled into).
    pub fn from_encoder(config: &config::Encoder, order: SumOrder) -> Self {
        let sc = &config.subframe_coding;
        let (sel, partitions) = match sc.fixed.order_sel {
            config::OrderSel::BitCount => (0, 0),
            config::OrderSel::ApproxEnt { partitions } => (1, partitions as u32),
        };
        Self {
            qlpc: abi_config_with(sc, order),
            use_constant: sc.use_constant as u32,
            use_fixed: sc.use_fixed as u32,
            use_lpc: sc.use_lpc as u32,
            use_leftside: config.stereo_coding.use_leftside as u32,
            use_rightside: config.stereo_coding.use_rightside as u32,
            use_midside: config.stereo_coding.use_midside as u32,
            fixed_max_order: sc.fixed.max_order as u32,
            fixed_order_sel: sel,
            fixed_partitions: partitions,
            reserved: 0,
        }
    }
}

/// `flacenc_hip_stereo_frame_result`: 752 bytes, what `encode_frame` decided for one frame.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct StereoFrameResult {
    pub channel_assignment: u8, // 0 Independent(2), 1 LeftSide, 2 RightSide, 3 MidSide
    pub kind: [u8; 2],          // 0 Constant, 1 Verbatim, 2 FixedLpc, 3 Lpc
    pub role: [u8; 2],          // 0 L, 1 R, 2 M, 3 S
    pub analysis_status: u8,    // OR of the four analyses' status bits: non-zero where the crate panics (lpc.rs:646, :786)
    pub pad: [u8; 2],
    pub dc_offset: [i32; 2],
    pub bits: [u64; 4],
    pub lpc: [SubframeParams; 2],
}

/// `flacenc_hip_channel_result`: 368 bytes, what `encode_subframe` (`src/coding.rs:384-418`) decided for one channel of
/// an `Independent(n)` frame (`encode_frame` for channel counts other than 2, `src/coding.rs:537-541`).
#[repr(C)]
#[derive(Clone, Copy)]
pub struct ChannelResult {
    pub kind: u8,            // 0 Constant, 1 Verbatim, 2 FixedLpc, 3 Lpc
    pub analysis_status: u8, // status bits of this channel's LPC analysis
    pub pad: [u8; 2],
    pub dc_offset: i32,
    pub bits: u64,
    pub params: SubframeParams,
}

pub const KIND_CONSTANT: u8 = 0;
pub const KIND_VERBATIM: u8 = 1;
pub const KIND_FIXED: u8 = 2;
pub const KIND_LPC: u8 = 3;

#[repr(C)]
pub struct Handle {
    _private: [u8; 0],
}

/// `FLACENC_HIP_ABI_VERSION` of `include/flacenc_hip.h` this binding was written against.
pub const ABI_VERSION: c_int = 6;

extern "C" {
    pub fn flacenc_hip_abi_version() -> c_int;
    pub fn flacenc_hip_create(out: *mut *mut Handle, device_id: c_int) -> c_int;
    pub fn flacenc_hip_destroy(h: *mut Handle);
    pub fn flacenc_hip_last_error(h: *const Handle) -> *const c_char;
    pub fn flacenc_hip_verify_config(cfg: *const QlpcConfig) -> c_int;
    pub fn flacenc_hip_qlpc_batch(
        h: *mut Handle, cfg: *const QlpcConfig, samples: *const i32, n_subframes: usize,
        block_size: u32, stride: usize, bps: *const u8, params: *mut SubframeParams,
        residual: *mut i32, residual_stride: usize, autocorr: *mut f64, lpc_coefs: *mut f64,
        memory_kind: c_int,
    ) -> c_int;
    /// Packed interleaved LE stereo PCM in host memory -> FLAC frame bytes in host memory (chunked, pinned
    /// staging, copies overlapped with the analysis): the call `encode_with_fixed_block_size` makes per run.
    pub fn flacenc_hip_encode_pcm_stereo(
        h: *mut Handle, cfg: *const FrameConfig, pcm: *const u8, total_samples: u64, bytes_per_sample: u32,
        bits_per_sample: u32, block_size: u32, sample_rate: u32, first_frame_number: u32, frame_number_step: u32,
        out: *mut u8, out_capacity: usize, out_len: *mut u32, out_total: *mut u64,
    ) -> c_int;
    /// The same for 1..=8 interleaved channels (2: stereo decision; others: Independent(n) frames).
    pub fn flacenc_hip_encode_pcm(
        h: *mut Handle, cfg: *const FrameConfig, pcm: *const u8, total_samples: u64, channels: u32,
        bytes_per_sample: u32, bits_per_sample: u32, block_size: u32, sample_rate: u32, first_frame_number: u32,
        frame_number_step: u32, out: *mut u8, out_capacity: usize, out_len: *mut u32, out_total: *mut u64,
    ) -> c_int;
    /// The ordered gather's device steps (`ParSink`, src/par.rs:67-95, across GPUs): decision records -> wire records
    /// (+ byte lengths) in one pass, and stream offsets from the all-gathered lengths (rank-major as the collective
    /// delivers them).  Device pointers; `stream` is a hipStream_t.
    pub fn flacenc_hip_frame_wire_bytes(block_size: u32) -> usize;
    pub fn flacenc_hip_stereo_frame_wire_async(
        h: *mut Handle, results: *const StereoFrameResult, n_frames: usize, block_size: u32, bits_per_sample: u32,
        sample_rate: u32, first_frame_number: u32, frame_number_step: u32, wire: *mut u8, wire_stride: usize,
        out_len: *mut u32, stream: *mut core::ffi::c_void,
    ) -> c_int;
    pub fn flacenc_hip_stream_offsets_async(
        h: *mut Handle, gathered_lengths: *const u32, n_frames_total: usize, world: u32, header_bytes: u64,
        lengths_stream: *mut u32, offsets: *mut u64, total: *mut u64, stream: *mut core::ffi::c_void,
    ) -> c_int;
    /// ParSink's ordered gather (`src/par.rs:67-95`) across processes: an RCCL communicator owned by the handle
    /// (`ncclGetUniqueId` / `ncclCommInitRank`) and the all-gather of this rank's wire records or byte lengths,
    /// rank-major and zero-padded as `flacenc_hip_stream_offsets_async` reads them.
    pub fn flacenc_hip_comm_unique_id(id: *mut u8) -> c_int;
    pub fn flacenc_hip_comm_create(h: *mut Handle, id: *const u8, rank: c_int, world: c_int) -> c_int;
    pub fn flacenc_hip_comm_destroy(h: *mut Handle) -> c_int;
    pub fn flacenc_hip_comm_info(h: *mut Handle, rank: *mut c_int, world: *mut c_int) -> c_int;
    pub fn flacenc_hip_allgather_async(
        h: *mut Handle, send: *const c_void, recv: *mut c_void, bytes_per_rank: usize, stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_allgather_records_async(
        h: *mut Handle, local: *const c_void, n_local: usize, n_total: usize, record_bytes: usize,
        gathered: *mut c_void, stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_host_alloc(bytes: usize) -> *mut core::ffi::c_void;
    pub fn flacenc_hip_host_free(p: *mut core::ffi::c_void);
    pub fn flacenc_hip_set_host_threads(h: *mut Handle, threads: i32) -> i32;
    pub fn flacenc_hip_stereo_qlpc_batch(
        h: *mut Handle, cfg: *const QlpcConfig, frames: *const i32, n_frames: usize,
        block_size: u32, stride: usize, bits_per_sample: u32, params: *mut SubframeParams,
        residual: *mut i32, residual_stride: usize, memory_kind: c_int,
    ) -> c_int;
    /// `fixed_lpc` (`src/coding.rs:298-331`) for a batch; `layout` 0 = subframes, 1 = stereo frames.
    pub fn flacenc_hip_fixed_lpc_batch(
        h: *mut Handle, cfg: *const FrameConfig, samples: *const i32, n_units: usize,
        block_size: u32, stride: usize, bps: *const u8, bits_per_sample: u32, layout: c_int,
        params: *mut SubframeParams, residual: *mut i32, residual_stride: usize,
        selector_keys: *mut u64, memory_kind: c_int,
    ) -> c_int;
    /// `encode_frame` (`src/coding.rs:530-544`) for 2-channel frames, decision on the GPU.
    pub fn flacenc_hip_encode_stereo_frames(
        h: *mut Handle, cfg: *const FrameConfig, frames: *const i32, n_frames: usize,
        block_size: u32, stride: usize, bits_per_sample: u32, results: *mut StereoFrameResult,
        residual: *mut i32, residual_stride: usize, memory_kind: c_int,
    ) -> c_int;
    /// `Frame::write` (`src/component/bitrepr.rs:289-319`) for the frames of
    /// `flacenc_hip_encode_stereo_frames`: bytes for `Frame::set_precomputed_bitstream`.
    pub fn flacenc_hip_stereo_frame_bytes_bound(block_size: u32, bits_per_sample: u32) -> usize;
    pub fn flacenc_hip_pack_stereo_frames(
        h: *mut Handle, frames: *const i32, n_frames: usize, block_size: u32, stride: usize,
        results: *const StereoFrameResult, residual: *const i32, residual_stride: usize,
        bits_per_sample: u32, sample_rate: u32, first_frame_number: u32, frame_number_step: u32,
        out: *mut u8, out_stride: usize, out_len: *mut u32, memory_kind: c_int,
    ) -> c_int;
    /// `FrameBuf::fill_le_bytes` (`src/source.rs:288-298`) for a run of frames.
    pub fn flacenc_hip_fill_le_bytes(
        h: *mut Handle, bytes: *const u8, total_samples: u64, channels: u32, bytes_per_sample: u32,
        n_frames: usize, block_size: u32, frames: *mut i32, stride: usize, memory_kind: c_int,
    ) -> c_int;
    pub fn flacenc_hip_qlpc_batch_async(
        h: *mut Handle, cfg: *const QlpcConfig, samples: *const i32, n_subframes: usize,
        block_size: u32, stride: usize, bps: *const u8, params: *mut SubframeParams,
        residual: *mut i32, residual_stride: usize, autocorr: *mut f64, lpc_coefs: *mut f64,
        stream: *mut c_void,
    ) -> c_int;
    // ---- the rest of include/flacenc_hip.h (tests/test_rust_binding.py holds this block to the header: every export,
    // ---- same arity, same argument classes)
    pub fn flacenc_hip_device_count() -> c_int;
    /// `lpc::window_weights` (`src/lpc.rs:96-120`) as the kernels use it.
    pub fn flacenc_hip_window_weights(cfg: *const QlpcConfig, block_size: u32, out: *mut f32) -> c_int;
    pub fn flacenc_hip_stereo_qlpc_batch_async(
        h: *mut Handle, cfg: *const QlpcConfig, frames: *const i32, n_frames: usize, block_size: u32, stride: usize,
        bits_per_sample: u32, params: *mut SubframeParams, residual: *mut i32, residual_stride: usize,
        stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_encode_stereo_frames_async(
        h: *mut Handle, cfg: *const FrameConfig, frames: *const i32, n_frames: usize, block_size: u32, stride: usize,
        bits_per_sample: u32, results: *mut StereoFrameResult, residual: *mut i32, residual_stride: usize,
        stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_fixed_lpc_batch_async(
        h: *mut Handle, cfg: *const FrameConfig, samples: *const i32, n_units: usize, block_size: u32, stride: usize,
        bps: *const u8, bits_per_sample: u32, layout: c_int, params: *mut SubframeParams, residual: *mut i32,
        residual_stride: usize, selector_keys: *mut u64, stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_pack_stereo_frames_async(
        h: *mut Handle, frames: *const i32, n_frames: usize, block_size: u32, stride: usize,
        results: *const StereoFrameResult, residual: *const i32, residual_stride: usize, bits_per_sample: u32,
        sample_rate: u32, first_frame_number: u32, frame_number_step: u32, out: *mut u8, out_stride: usize,
        out_len: *mut u32, stream: *mut c_void,
    ) -> c_int;
    /// `encode_frame` for 1 or 3..=8 independent channels (`src/coding.rs:537-541`), decision on the GPU.
    pub fn flacenc_hip_encode_frames(
        h: *mut Handle, cfg: *const FrameConfig, frames: *const i32, n_frames: usize, channels: u32, block_size: u32,
        stride: usize, bits_per_sample: u32, results: *mut ChannelResult, residual: *mut i32, residual_stride: usize,
        memory_kind: c_int,
    ) -> c_int;
    pub fn flacenc_hip_encode_frames_async(
        h: *mut Handle, cfg: *const FrameConfig, frames: *const i32, n_frames: usize, channels: u32, block_size: u32,
        stride: usize, bits_per_sample: u32, results: *mut ChannelResult, residual: *mut i32, residual_stride: usize,
        stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_frame_bytes_bound(channels: u32, block_size: u32, bits_per_sample: u32) -> usize;
    pub fn flacenc_hip_pack_frames(
        h: *mut Handle, frames: *const i32, n_frames: usize, channels: u32, block_size: u32, stride: usize,
        results: *const ChannelResult, residual: *const i32, residual_stride: usize, bits_per_sample: u32,
        sample_rate: u32, first_frame_number: u32, frame_number_step: u32, out: *mut u8, out_stride: usize,
        out_len: *mut u32, memory_kind: c_int,
    ) -> c_int;
    pub fn flacenc_hip_pack_frames_async(
        h: *mut Handle, frames: *const i32, n_frames: usize, channels: u32, block_size: u32, stride: usize,
        results: *const ChannelResult, residual: *const i32, residual_stride: usize, bits_per_sample: u32,
        sample_rate: u32, first_frame_number: u32, frame_number_step: u32, out: *mut u8, out_stride: usize,
        out_len: *mut u32, stream: *mut c_void,
    ) -> c_int;
    /// `encode_fixed_size_frame` + `Frame::write` for a run of frames in HBM (`src/coding.rs:581-606`).
    pub fn flacenc_hip_encode_pack_stereo_frames_async(
        h: *mut Handle, cfg: *const FrameConfig, frames: *const i32, n_frames: usize, block_size: u32, stride: usize,
        bits_per_sample: u32, sample_rate: u32, first_frame_number: u32, frame_number_step: u32,
        results: *mut StereoFrameResult, out: *mut u8, out_stride: usize, out_len: *mut u32, stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_encode_pack_frames_async(
        h: *mut Handle, cfg: *const FrameConfig, frames: *const i32, n_frames: usize, channels: u32, block_size: u32,
        stride: usize, bits_per_sample: u32, sample_rate: u32, first_frame_number: u32, frame_number_step: u32,
        results: *mut ChannelResult, out: *mut u8, out_stride: usize, out_len: *mut u32, stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_stereo_frame_lengths_async(
        h: *mut Handle, results: *const StereoFrameResult, n_frames: usize, block_size: u32, bits_per_sample: u32,
        sample_rate: u32, first_frame_number: u32, frame_number_step: u32, out_len: *mut u32, stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_place_frames_async(
        h: *mut Handle, src: *const u8, src_offsets: *const u64, lengths: *const u32, n_frames: usize, dst: *mut u8,
        dst_offsets: *const u64, stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_fill_le_bytes_async(
        h: *mut Handle, bytes: *const u8, total_samples: u64, channels: u32, bytes_per_sample: u32, n_frames: usize,
        block_size: u32, frames: *mut i32, stride: usize, stream: *mut c_void,
    ) -> c_int;
    pub fn flacenc_hip_synchronize(h: *mut Handle) -> c_int;
}

/// One handle per host thread, like the crate's `reusable!` thread-locals (`src/lib.rs:92-116`).
/// The second field is the summation order every configuration built for this handle asks for.
pub struct Gpu(*mut Handle, SumOrder);

impl Gpu {
    /// A handle that reproduces the bytes of the crate build it is compiled into (`SumOrder::CrateBuild`): the drop-in
    /// under `encode_with_fixed_block_size` must not change the encoder's output.
    pub fn new(device_id: i32) -> Result<Self, EncodeError> {
        Self::with_sum_order(device_id, SumOrder::CrateBuild)
    }

    pub fn sum_order(&self) -> SumOrder {
        self.1
    }

    /// `SumOrder::Canonical` trades byte-identity with the CPU build for the kernels' own (fastest) order.
    pub fn with_sum_order(device_id: i32, order: SumOrder) -> Result<Self, EncodeError> {
        // (QlpcConfig is embedded by value in FrameConfig: a library of another revision would read it shifted)
        if unsafe { flacenc_hip_abi_version() } != ABI_VERSION {
            return Err(EncodeError::Config(VerifyError::new("gpu", "libflacenc_hip.so has another ABI revision")));
        }
        let mut h = std::ptr::null_mut();
        match unsafe { flacenc_hip_create(&mut h, device_id) } {
            OK => Ok(Self(h, order)),
            _ => Err(EncodeError::Config(VerifyError::new("gpu", "no usable HIP device"))),
        }
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { flacenc_hip_destroy(self.0) }
    }
}

/// Which of the crate's builds the floating-point sums reproduce bit for bit (the C++ mirror's
/// `HipContext::SumOrder`).  `CrateBuild` -- the default of `Gpu::new` -- asks for the bytes of the build this file is
/// compiled into (stable: `FLAG_REFERENCE_SUM_ORDER`; `simd-nightly`: `FLAG_NIGHTLY_SUM_ORDER` up to order 15).
/// `Canonical` is the library's unflagged mode: since ABI 6 the stable build's quantised LPC parameters -- and every integer
/// that follows from them -- on EVERY shape (blocks of 4096 / 4608 samples at orders up to 12 by an order certificate, every
/// other shape by the stable build's own chains in a pass in front of the kernels).  What it does not
/// pin is the fixed-LPC selector's per-partition sums on material of more than 16 bits (exact integer sums there), which
/// `CrateBuild` does; chosen with `Gpu::with_sum_order`.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum SumOrder {
    Canonical,
    CrateBuild,
}

/// `mae_optimization_steps` above 64 (`FLACENC_HIP_MAX_MAE_STEPS`) is refused with `BAD_CONFIG`: the reference's
/// experimental build takes any value.
fn abi_config_with(c: &config::SubFrameCoding, order: SumOrder) -> QlpcConfig {
    let (window_type, tukey_alpha) = match c.qlpc.window {
        config::Window::Rectangle => (0, 0.0),
        config::Window::Tukey { alpha } => (1, alpha),
    };
    QlpcConfig {
        lpc_order: c.qlpc.lpc_order as u32,
        quant_precision: c.qlpc.quant_precision as u32,
        window_type,
        tukey_alpha,
        max_rice_parameter: c.prc.max_parameter as u32,
        flags: if order == SumOrder::Canonical {
            0
        } else if cfg!(feature = "simd-nightly") && c.qlpc.lpc_order <= 15 {
            FLAG_NIGHTLY_SUM_ORDER
        } else if cfg!(feature = "simd-nightly") {
            0 // (the nightly split above order 15 depends on the allocator: canonical order, a valid encoding)
        } else {
            FLAG_REFERENCE_SUM_ORDER | FLAG_INTEGER_PARITY_ONLY
        },
        use_direct_mse: c.qlpc.use_direct_mse as u32,
        mae_optimization_steps: c.qlpc.mae_optimization_steps as u32,
    }
}

/// Rebuilds the `SubFrame::Lpc` that `estimated_qlpc` (`src/coding.rs:360-381`) would have
/// returned from one GPU record + residual row, through the crate's own constructors
/// (`QuantizedParameters::from_parts` `datatype.rs:2214`, `Residual::from_parts` `:2315`,
/// `Lpc::from_parts` `:2115`).
pub fn lpc_from_record(p: &SubframeParams, residual: &[i32], signal: &[i32], bps: u8) -> SubFrame {
    assert_eq!(p.status, 0, "the reference panics on non-finite LPC statistics (lpc.rs:786)");
    let order = p.order as usize;
    let qlpc = QuantizedParameters::from_parts(&p.coefs[..order], order, p.shift, p.precision as usize);
    let nparts = 1usize << p.rice_order;
    let part_size = signal.len() >> p.rice_order;
    let mut quotients = vec![0u32; signal.len()];
    let mut remainders = vec![0u32; signal.len()];
    for t in order..signal.len() {
        let rice_p = p.rice_params[t / part_size];
        // quotients_and_remainders, src/coding.rs:58-62
        let err = crate::rice::encode_signbit(residual[t]);
        quotients[t] = err >> rice_p;
        remainders[t] = err & ((1u32 << rice_p) - 1);
    }
    let residual = Residual::from_parts(
        p.rice_order, signal.len(), order, p.rice_params[..nparts].to_vec(), quotients, remainders,
    );
    Lpc::from_parts(
        heapless::Vec::from_slice(&signal[..order]).expect("LPC order exceeded the maximum"),
        qlpc, residual, bps,
    )
    .into()
}

/// Rebuilds the `SubFrame::FixedLpc` that `fixed_lpc` (`src/coding.rs:298-331`) would have returned
/// from one record of `flacenc_hip_fixed_lpc_batch` (or a `kind == 2` slot of a
/// `StereoFrameResult`) + its error-signal row (`FixedLpc::from_parts`, `src/coding.rs:321-328`).
/// For the stand-alone batch the caller first checks `selector_key < baseline_bits`
/// (`src/coding.rs:262, 284`): otherwise `fixed_lpc` returned `None`.
pub fn fixed_lpc_from_record(p: &SubframeParams, residual: &[i32], signal: &[i32], bps: u8) -> SubFrame {
    let order = p.order as usize;
    let nparts = 1usize << p.rice_order;
    let part_size = signal.len() >> p.rice_order;
    let mut quotients = vec![0u32; signal.len()];
    let mut remainders = vec![0u32; signal.len()];
    for t in order..signal.len() {
        let rice_p = p.rice_params[t / part_size];
        let err = crate::rice::encode_signbit(residual[t]);
        quotients[t] = err >> rice_p;
        remainders[t] = err & ((1u32 << rice_p) - 1);
    }
    let residual = Residual::from_parts(
        p.rice_order, signal.len(), order, p.rice_params[..nparts].to_vec(), quotients, remainders,
    );
    crate::component::FixedLpc::from_parts(
        heapless::Vec::from_slice(&signal[..order]).expect("Exceeded maximum order for FixedLpc component."),
        residual, bps,
    )
    .into()
}

/// Batched replacement of the `estimated_qlpc` calls of a run of frames.  `framebufs` are the
/// crate's `FrameBuf`s (channel-major, `src/source.rs:115-127`); returns, per frame, the LPC
/// candidates in the order `encode_frame` asks for them (L, R, M, S for stereo).
pub fn estimated_qlpc_batch(
    gpu: &Gpu, config: &Verified<config::Encoder>, staged: &[i32], n_frames: usize,
    channels: usize, block_size: usize, bits_per_sample: u8,
) -> Result<(Vec<SubframeParams>, Vec<i32>), EncodeError> {
    let cfg = abi_config_with(&config.subframe_coding, gpu.sum_order());
    let per_frame = if channels == 2 { 4 } else { channels };
    let mut params = vec![unsafe { std::mem::zeroed::<SubframeParams>() }; n_frames * per_frame];
    let mut residual = vec![0i32; n_frames * per_frame * block_size];
    let rc = unsafe {
        if channels == 2 {
            flacenc_hip_stereo_qlpc_batch(
                gpu.0, &cfg, staged.as_ptr(), n_frames, block_size as u32, block_size,
                bits_per_sample as u32, params.as_mut_ptr(), residual.as_mut_ptr(), block_size, MEM_HOST,
            )
        } else {
            let bps = vec![bits_per_sample; n_frames * channels];
            flacenc_hip_qlpc_batch(
                gpu.0, &cfg, staged.as_ptr(), n_frames * channels, block_size as u32, block_size,
                bps.as_ptr(), params.as_mut_ptr(), residual.as_mut_ptr(), block_size,
                std::ptr::null_mut(), std::ptr::null_mut(), MEM_HOST,
            )
        }
    };
    match rc {
        OK => Ok((params, residual)),
        ERR_BAD_CONFIG => Err(EncodeError::Config(VerifyError::new("subframe_coding", "rejected by the GPU path"))),
        _ => panic!("flacenc_hip device error"), // the reference panics on internal errors too
    }
}

// =====================================================================================================================
// `gpu::encode_with_fixed_block_size`: the branch `coding::encode_with_fixed_block_size` (`src/coding.rs:645-676`) takes
// next to its `par` branch (`:650-655`).  Same signature, same `Stream` out.  Two shapes:
//
//   * `encode_with_fixed_block_size`            frame BYTES from the GPU (`flacenc_hip_encode_pcm`): what a file encoder
//                                               needs.  Frames carry their header and `precomputed_bitstream`, no SubFrames.
//   * `encode_with_fixed_block_size_components` `component::SubFrame`s rebuilt from the GPU's decision records through
//                                               the crate's own `from_parts` constructors: for callers that inspect them.
//
// Both follow `par::encode_with_fixed_block_size` (`src/par.rs:355-449`) step for step: `Stream::new`, a feed loop that
// drains the `Source` into buffers with the MD5 / sample-count `Context` riding along as the second half of a `Fill`
// pair (`src/par.rs:288-325`), the frames added in frame-number order (`ParSink::finalize`, `:82-94` -- here the order
// of the batch), then STREAMINFO: MD5, `set_block_sizes(max, max)` and `set_total_samples` (`src/par.rs:425-447`).
//
// The one addition the crate needs outside this module: `Frame::set_precomputed_bitstream(&mut self, Vec<u8>)` next to
// `Frame::precompute_bitstream` (`src/component/datatype.rs:1036-1045`); the field is private to that module.
// `BitRepr for Frame` already serves `count_bits` and `write` from it (`src/component/bitrepr.rs:275-293`).
// =====================================================================================================================

/// Frames per GPU call: bounds the host buffers of a run (4096 stereo frames of 4096 16-bit samples = 64 MiB of PCM);
/// `flacenc_hip_encode_pcm` cuts a run into chunks of 768..8192 frames itself and overlaps their transfers.
pub const FRAMES_PER_RUN: usize = 4096;

/// A `Fill` (`src/source.rs:42-82`) that keeps what a `Source` hands over as packed little-endian interleaved PCM --
/// the form `flacenc_hip_encode_pcm` uploads (2..3 bytes per sample over PCIe instead of the 4 of a `FrameBuf`).
pub struct PcmRun {
    bytes: Vec<u8>,
    bytes_per_sample: usize,
    channels: usize,
    frames: usize,
}

impl PcmRun {
    pub fn new(bits_per_sample: usize, channels: usize) -> Self {
        Self { bytes: Vec::new(), bytes_per_sample: bits_per_sample.div_ceil(8), channels, frames: 0 }
    }
    fn clear(&mut self) {
        self.bytes.clear();
        self.frames = 0;
    }
    /// inter-channel samples held
    fn samples(&self) -> usize {
        self.bytes.len() / self.bytes_per_sample / self.channels
    }
}

impl Fill for PcmRun {
    fn fill_interleaved(&mut self, interleaved: &[i32]) -> Result<(), SourceError> {
        for v in interleaved {
            self.bytes.extend_from_slice(&v.to_le_bytes()[..self.bytes_per_sample]); // as Context hashes them, source.rs:411
        }
        self.frames += usize::from(!interleaved.is_empty());
        Ok(())
    }
    fn fill_le_bytes(&mut self, bytes: &[u8], bytes_per_sample: usize) -> Result<(), SourceError> {
        if bytes_per_sample != self.bytes_per_sample {
            return Err(SourceError::by_reason(SourceErrorReason::InvalidBuffer)); // a sample width the run was not set up for
        }
        self.bytes.extend_from_slice(bytes);
        self.frames += usize::from(!bytes.is_empty());
        Ok(())
    }
}

fn map_rc(rc: c_int, what: &str) -> Result<(), EncodeError> {
    match rc {
        OK => Ok(()),
        ERR_BAD_CONFIG => Err(EncodeError::Config(VerifyError::new("gpu", what))),
        _ => panic!("flacenc_hip device error in {what}"), // the reference panics on internal errors too
    }
}

/// `ChannelAssignment` from the 4-bit code of a frame header (`src/component/bitrepr.rs:373-419`; byte 3, high nibble).
fn channel_assignment_from_code(code: u8) -> ChannelAssignment {
    match code {
        8 => ChannelAssignment::LeftSide,
        9 => ChannelAssignment::RightSide,
        10 => ChannelAssignment::MidSide,
        n => ChannelAssignment::Independent(n + 1),
    }
}

fn empty_frame(block: usize, ch: ChannelAssignment, bits_per_sample: usize, sample_rate: usize, number: usize) -> Frame {
    // the specs exactly as encode_frame_impl chooses them (src/coding.rs:431-436)
    let mut frame = Frame::new_empty(
        BlockSizeSpec::from_size(block as u16),
        ch,
        SampleSizeSpec::from_bits(bits_per_sample as u8).unwrap_or(SampleSizeSpec::Unspecified),
        SampleRateSpec::from_freq(sample_rate as u32).unwrap_or(SampleRateSpec::Unspecified),
    );
    frame.header_mut().set_frame_offset(FrameOffset::Frame(number as u32)); // encode_fixed_size_frame, src/coding.rs:602-604
    frame
}

fn finalize_stream(stream: &mut Stream, context: &Context, len_hint: Option<usize>) {
    // src/coding.rs:677-693 = src/par.rs:425-447
    if stream.frame_count() > 0 {
        let max_block_size = stream.stream_info().max_block_size();
        stream.stream_info_mut().set_block_sizes(max_block_size, max_block_size).unwrap();
    }
    stream.stream_info_mut().set_md5_digest(&context.md5_digest());
    stream.stream_info_mut().set_total_samples(len_hint.unwrap_or_else(|| context.total_samples()));
}

/// Shape 1, bytes: `Source` -> packed PCM runs -> `flacenc_hip_encode_pcm` -> frames with precomputed bitstreams.
///
/// Mirror of `coding::encode_with_fixed_block_size` (`src/coding.rs:645-676`) / `par::encode_with_fixed_block_size`
/// (`src/par.rs:355-449`) with the GPU as the worker pool: the whole of `encode_fixed_size_frame` + `Frame::write`
/// (candidates, `encode_subframe`, `try_stereo_coding`, bit writer, CRCs) runs on the device.
pub fn encode_with_fixed_block_size<T: Source>(
    config: &Verified<config::Encoder>, mut src: T, block_size: usize,
) -> Result<Stream, EncodeError> {
    let gpu = Gpu::new(0)?;
    let (channels, bits_per_sample, sample_rate) = (src.channels(), src.bits_per_sample(), src.sample_rate());
    let mut stream = Stream::new(sample_rate, channels, bits_per_sample)?;
    let frame_cfg = FrameConfig::from_encoder(config, gpu.sum_order());
    let mut run_and_context = (PcmRun::new(bits_per_sample, channels), Context::new(bits_per_sample, channels));
    let bound = unsafe { flacenc_hip_frame_bytes_bound(channels as u32, block_size as u32, bits_per_sample as u32) };
    let mut out = vec![0u8; bound * FRAMES_PER_RUN];
    let mut lens = vec![0u32; FRAMES_PER_RUN];
    let mut first_frame = 0usize;
    loop {
        // the feed loop (src/par.rs:288-325), a run of frames at a time
        run_and_context.0.clear();
        let mut last_read = block_size;
        while run_and_context.0.frames < FRAMES_PER_RUN && last_read == block_size {
            last_read = src.read_samples(block_size, &mut run_and_context)?;
            if last_read == 0 {
                break;
            }
        }
        let run = &run_and_context.0;
        if run.frames == 0 {
            break;
        }
        let mut written = 0u64;
        map_rc(
            unsafe {
                flacenc_hip_encode_pcm(
                    gpu.0, &frame_cfg, run.bytes.as_ptr(), run.samples() as u64, channels as u32,
                    run.bytes_per_sample as u32, bits_per_sample as u32, block_size as u32, sample_rate as u32,
                    first_frame as u32, 1, out.as_mut_ptr(), out.len(), lens.as_mut_ptr(), &mut written,
                )
            },
            "flacenc_hip_encode_pcm",
        )?;
        // ParSink::finalize (src/par.rs:82-94): frames come back in frame-number order already
        let mut off = 0usize;
        let mut remaining = run.samples();
        for f in 0..run.frames {
            let bytes = out[off..off + lens[f] as usize].to_vec();
            off += lens[f] as usize;
            let block = remaining.min(block_size);
            remaining -= block;
            let mut frame =
                empty_frame(block, channel_assignment_from_code(bytes[3] >> 4), bits_per_sample, sample_rate, first_frame + f);
            frame.set_precomputed_bitstream(bytes);
            stream.add_frame(frame);
        }
        first_frame += run.frames;
        if last_read < block_size {
            break; // a short (or empty) read ends the stream, as `read_samples == 0` does after it in the serial loop
        }
    }
    finalize_stream(&mut stream, &run_and_context.1, src.len_hint());
    Ok(stream)
}

/// The signal of output channel role `role` (0 L, 1 R, 2 M, 3 S) of a stereo frame: warm-up samples and Verbatim
/// bodies are taken from it (`try_stereo_coding`'s MSFRAMEBUF, `src/coding.rs:476-491`).
fn role_signal(l: &[i32], r: &[i32], role: u8) -> Vec<i32> {
    match role {
        0 => l.to_vec(),
        1 => r.to_vec(),
        2 => l.iter().zip(r).map(|(a, b)| (a + b) >> 1).collect(),
        _ => l.iter().zip(r).map(|(a, b)| a - b).collect(),
    }
}

/// One `SubFrame` from what `encode_subframe` (`src/coding.rs:384-418`) decided on the GPU.
fn subframe_from_decision(kind: u8, dc_offset: i32, p: &SubframeParams, residual: &[i32], signal: &[i32], bps: u8) -> SubFrame {
    match kind {
        KIND_CONSTANT => Constant::from_parts(signal.len(), dc_offset, bps).into(),
        KIND_VERBATIM => Verbatim::from_samples(signal, bps).into(),
        KIND_FIXED => fixed_lpc_from_record(p, residual, signal, bps),
        _ => lpc_from_record(p, residual, signal, bps),
    }
}

/// Shape 2, components: K `FrameBuf`s per call -> `flacenc_hip_encode_stereo_frames` (2 channels) or
/// `flacenc_hip_encode_frames` (1, 3..=8) -> `Frame`s of real `SubFrame`s, bit writer and CRCs on the host as in the
/// serial encoder.  `frames_per_call` `FrameBuf`s are alive at once (`par`'s `workers * FRAMEBUF_MULTIPLICITY`,
/// `src/par.rs:364-368`).
pub fn encode_with_fixed_block_size_components<T: Source>(
    config: &Verified<config::Encoder>, mut src: T, block_size: usize, frames_per_call: usize,
) -> Result<Stream, EncodeError> {
    let gpu = Gpu::new(0)?;
    let (channels, bits_per_sample, sample_rate) = (src.channels(), src.bits_per_sample(), src.sample_rate());
    let mut stream = Stream::new(sample_rate, channels, bits_per_sample)?;
    let frame_cfg = FrameConfig::from_encoder(config, gpu.sum_order());
    let mut context = Context::new(bits_per_sample, channels);
    let mut framebufs: Vec<FrameBuf> = Vec::new();
    for _ in 0..frames_per_call {
        framebufs.push(FrameBuf::with_size(channels, block_size)?);
    }
    let mut staged = vec![0i32; frames_per_call * channels * block_size];
    let mut residual = vec![0i32; frames_per_call * channels * block_size];
    let mut stereo = vec![unsafe { std::mem::zeroed::<StereoFrameResult>() }; if channels == 2 { frames_per_call } else { 0 }];
    let mut indep = vec![unsafe { std::mem::zeroed::<ChannelResult>() }; if channels == 2 { 0 } else { frames_per_call * channels }];
    let mut first_frame = 0usize;
    let mut done = false;
    while !done {
        // fill up to K FrameBufs (src/par.rs:288-325); a batch holds frames of ONE block size, so a short last block
        // is a batch of its own
        let mut k = 0usize;
        let mut batch_block = block_size;
        while k < frames_per_call {
            let read = src.read_samples(block_size, &mut (&mut framebufs[k], &mut context))?;
            if read == 0 {
                done = true;
                break;
            }
            framebufs[k].verify_samples(bits_per_sample)?; // encode_fixed_size_frame, src/coding.rs:593
            if read < block_size {
                done = true;
                if k > 0 {
                    encode_component_batch(
                        &gpu, &frame_cfg, &framebufs[..k], block_size, channels, bits_per_sample, sample_rate, first_frame,
                        &mut staged, &mut residual, &mut stereo, &mut indep, &mut stream,
                    )?;
                    first_frame += k;
                    framebufs.swap(0, k);
                }
                k = 1;
                batch_block = read;
                break;
            }
            k += 1;
        }
        if k > 0 {
            encode_component_batch(
                &gpu, &frame_cfg, &framebufs[..k], batch_block, channels, bits_per_sample, sample_rate, first_frame,
                &mut staged, &mut residual, &mut stereo, &mut indep, &mut stream,
            )?;
            first_frame += k;
        }
    }
    finalize_stream(&mut stream, &context, src.len_hint());
    Ok(stream)
}

#[allow(clippy::too_many_arguments)]
fn encode_component_batch(
    gpu: &Gpu, frame_cfg: &FrameConfig, framebufs: &[FrameBuf], block: usize, channels: usize, bits_per_sample: usize,
    sample_rate: usize, first_frame: usize, staged: &mut [i32], residual: &mut [i32], stereo: &mut [StereoFrameResult],
    indep: &mut [ChannelResult], stream: &mut Stream,
) -> Result<(), EncodeError> {
    let k = framebufs.len();
    // FrameBuf::channel_slice IS the ABI's layout: channel c of frame f at (f * channels + c) * stride
    for (f, fb) in framebufs.iter().enumerate() {
        for c in 0..channels {
            staged[(f * channels + c) * block..][..block].copy_from_slice(fb.channel_slice(c));
        }
    }
    if channels == 2 {
        map_rc(
            unsafe {
                flacenc_hip_encode_stereo_frames(
                    gpu.0, frame_cfg, staged.as_ptr(), k, block as u32, block, bits_per_sample as u32,
                    stereo.as_mut_ptr(), residual.as_mut_ptr(), block, MEM_HOST,
                )
            },
            "flacenc_hip_encode_stereo_frames",
        )?;
        for f in 0..k {
            let res = &stereo[f];
            assert_eq!(res.analysis_status, 0, "the reference panics on these LPC statistics (lpc.rs:646, :786-799)");
            let ch_info = channel_assignment_from_code(if res.channel_assignment == 0 { 1 } else { 7 + res.channel_assignment });
            let (l, r) = (&staged[(2 * f) * block..][..block], &staged[(2 * f + 1) * block..][..block]);
            let mut frame = empty_frame(block, ch_info.clone(), bits_per_sample, sample_rate, first_frame + f);
            for c in 0..2 {
                let signal = role_signal(l, r, res.role[c]);
                let bps = (bits_per_sample + ch_info.bits_per_sample_offset(c)) as u8; // src/coding.rs:444
                frame.add_subframe(subframe_from_decision(
                    res.kind[c], res.dc_offset[c], &res.lpc[c], &residual[(2 * f + c) * block..][..block], &signal, bps,
                ));
            }
            stream.add_frame(frame);
        }
    } else {
        map_rc(
            unsafe {
                flacenc_hip_encode_frames(
                    gpu.0, frame_cfg, staged.as_ptr(), k, channels as u32, block as u32, block, bits_per_sample as u32,
                    indep.as_mut_ptr(), residual.as_mut_ptr(), block, MEM_HOST,
                )
            },
            "flacenc_hip_encode_frames",
        )?;
        for f in 0..k {
            let mut frame =
                empty_frame(block, ChannelAssignment::Independent(channels as u8), bits_per_sample, sample_rate, first_frame + f);
            for c in 0..channels {
                let i = f * channels + c;
                assert_eq!(indep[i].analysis_status, 0, "the reference panics on these LPC statistics (lpc.rs:646, :786-799)");
                frame.add_subframe(subframe_from_decision(
                    indep[i].kind, indep[i].dc_offset, &indep[i].params, &residual[i * block..][..block],
                    &staged[i * block..][..block], bits_per_sample as u8,
                ));
            }
            stream.add_frame(frame);
        }
    }
    Ok(())
}

/// `FLACENC_HIP_COMM_ID_BYTES`
pub const COMM_ID_BYTES: usize = 128;

impl Gpu {
    /// Rank 0 of a one-process-per-GPU host calls this and hands the bytes to the other ranks (a file, a socket, MPI).
    pub fn comm_unique_id() -> Result<[u8; COMM_ID_BYTES], EncodeError> {
        let mut id = [0u8; COMM_ID_BYTES];
        match unsafe { flacenc_hip_comm_unique_id(id.as_mut_ptr()) } {
            OK => Ok(id),
            _ => Err(EncodeError::Config(VerifyError::new("gpu", "librccl is not available"))),
        }
    }

    /// Collective over all `world` ranks: frame f of the stream is analysed by rank f mod world.
    pub fn comm_create(&self, id: &[u8; COMM_ID_BYTES], rank: i32, world: i32) -> Result<(), EncodeError> {
        match unsafe { flacenc_hip_comm_create(self.0, id.as_ptr(), rank, world) } {
            OK => Ok(()),
            _ => Err(EncodeError::Config(VerifyError::new("gpu", "ncclCommInitRank failed"))),
        }
    }
}

#[cfg(test)]
mod tests {
    use super::*;

    /// the summation-order choice must be reachable and must set the flag of the crate build.
    #[test]
    fn crate_build_order_sets_the_builds_flag() {
        let enc = config::Encoder::default();
        let canonical = FrameConfig::from_encoder(&enc, SumOrder::Canonical);
        assert_eq!(canonical.qlpc.flags & (FLAG_REFERENCE_SUM_ORDER | FLAG_NIGHTLY_SUM_ORDER), 0);
        let own = FrameConfig::from_encoder(&enc, SumOrder::CrateBuild);
        let want = if cfg!(feature = "simd-nightly") { FLAG_NIGHTLY_SUM_ORDER } else { FLAG_REFERENCE_SUM_ORDER };
        assert_eq!(own.qlpc.flags & (FLAG_REFERENCE_SUM_ORDER | FLAG_NIGHTLY_SUM_ORDER), want);
        // the stable build asks for its integers, not for the floating-point intermediates
        assert_eq!(own.qlpc.flags & FLAG_INTEGER_PARITY_ONLY != 0, !cfg!(feature = "simd-nightly"));
    }
}
