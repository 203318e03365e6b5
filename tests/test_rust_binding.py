"""The Rust side of the boundary (rust/flacenc_hip.rs) against include/flacenc_hip.h.

There is no rustc in the image, so the binding cannot be compiled here; what CAN drift silently is its `extern "C"`
block and its `#[repr(C)]` structs.  These tests parse both files and fail on any export the Rust side lacks, any
arity or argument-type difference, any struct field that differs in name, type or order, and any constant that
differs in value.  They also pin the presence and signature of the drop-in SURVEY section 8(b) names:
`encode_with_fixed_block_size<T: Source>` (reference: src/coding.rs:645-676, src/par.rs:355-449).
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "flacenc_hip.h")
RUST = os.path.join(ROOT, "rust", "flacenc_hip.rs")

SCALARS = {
    "int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "float": "f32", "double": "f64",
    "int32_t": "i32", "uint8_t": "u8", "int8_t": "i8", "int16_t": "i16", "void": "c_void", "char": "c_char",
}
STRUCTS = {
    "flacenc_hip_handle": "Handle", "flacenc_hip_qlpc_config": "QlpcConfig", "flacenc_hip_frame_config": "FrameConfig",
    "flacenc_hip_subframe_params": "SubframeParams", "flacenc_hip_stereo_frame_result": "StereoFrameResult",
    "flacenc_hip_channel_result": "ChannelResult",
}


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _strip_rust_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def _c_type(decl):
    """'const int32_t* samples' -> '*const i32'; 'uint8_t id[128]' -> '*mut u8'; 'size_t n' -> 'usize'."""
    decl = decl.strip()
    array = re.search(r"\[[^\]]*\]\s*$", decl)
    if array:
        decl = decl[: array.start()].strip()
    m = re.match(r"^(const\s+)?(\w+)\s*(\*+)?\s*(\w+)?$", decl)
    assert m, decl
    const, base, stars, _name = m.groups()
    stars = (stars or "") + ("*" if array else "")
    base = SCALARS.get(base) or STRUCTS[base]
    if not stars:
        return base
    out = base
    for level in range(len(stars)):
        # only the innermost pointee can be const in this header
        out = ("*const " if (const and level == 0) else "*mut ") + out
    return out


def _rust_type(t):
    t = t.strip().replace("core::ffi::c_void", "c_void").replace("std::os::raw::", "")
    t = re.sub(r"\s+", " ", t)
    return "c_int" if t == "i32" else t


def c_prototypes():
    text = _strip_c_comments(open(HEADER).read())
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)  # preprocessor lines
    protos = {}
    for ret, name, args in re.findall(r"((?:const\s+)?\w+\s*\**)\s*\b(flacenc_hip_\w+)\s*\(([^)]*)\)\s*;", text):
        args = args.strip()
        arg_types = [] if args in ("", "void") else [_c_type(a) for a in args.split(",")]
        ret = ret.strip()
        ret_t = None if ret == "void" else _c_type(ret + " x") if "*" not in ret else _c_type(ret.replace("*", "* x"))
        protos[name] = (arg_types, ret_t)
    return protos


def rust_externs():
    text = _strip_rust_comments(open(RUST).read())
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S)
    assert block, 'no extern "C" block'
    decls = {}
    for name, args, ret in re.findall(r"pub fn (flacenc_hip_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block.group(1)):
        arg_types = []
        for a in args.split(","):
            a = a.strip()
            if a:
                arg_types.append(_rust_type(a.split(":", 1)[1]))
        decls[name] = (arg_types, _rust_type(ret) if ret else None)
    return decls


def test_every_export_of_the_header_is_declared_with_the_same_signature():
    c, r = c_prototypes(), rust_externs()
    assert len(c) >= 44
    missing = sorted(set(c) - set(r))
    assert not missing, f"exports without a Rust declaration: {missing}"
    extra = sorted(set(r) - set(c))
    assert not extra, f"Rust declarations the header does not have: {extra}"
    for name, (c_args, c_ret) in c.items():
        r_args, r_ret = r[name]
        assert len(c_args) == len(r_args), f"{name}: {len(c_args)} arguments in the header, {len(r_args)} in Rust"
        for i, (ca, ra) in enumerate(zip(c_args, r_args)):
            assert ca == ra, f"{name}: argument {i} is {ca} in the header and {ra} in Rust"
        assert c_ret == r_ret, f"{name}: returns {c_ret} in the header and {r_ret} in Rust"


def c_structs():
    text = _strip_c_comments(open(HEADER).read())
    structs = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\}\s*(\w+);", text, flags=re.S):
        fields = []
        for line in body.split(";"):
            line = line.strip()
            if not line:
                continue
            m = re.match(r"^(\w+)\s+(\w+)(?:\[(\w+)\])?$", line)
            assert m, line
            base, fname, count = m.groups()
            base = SCALARS.get(base) or STRUCTS[base]
            if count:
                count = {"FLACENC_HIP_MAX_RICE_PARTITIONS": "256"}.get(count, count)
                base = f"[{base}; {count}]"
            fields.append((fname, base))
        structs[STRUCTS[name]] = fields
    return structs


def rust_structs():
    text = _strip_rust_comments(open(RUST).read())
    structs = {}
    for name, body in re.findall(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", text, flags=re.S):
        fields = []
        for fname, ftype in re.findall(r"pub (\w+):\s*([^,\n]+),", body):
            fields.append((fname, re.sub(r"\s+", " ", ftype.strip())))
        structs[name] = fields
    return structs


def test_repr_c_structs_have_the_headers_fields():
    c, r = c_structs(), rust_structs()
    assert set(c) == {"QlpcConfig", "SubframeParams", "FrameConfig", "StereoFrameResult", "ChannelResult"}
    for name, fields in c.items():
        assert name in r, f"no #[repr(C)] struct {name}"
        want = [(f, "i32" if t == "c_int" else t) for f, t in fields]
        assert r[name] == want, f"{name}: header {want} != Rust {r[name]}"


def test_constants_agree():
    h = _strip_c_comments(open(HEADER).read())
    r = _strip_rust_comments(open(RUST).read())
    defines = {k: v for k, v in re.findall(r"#define (FLACENC_HIP_\w+)\s+\(?(-?\d+)u?\)?", h)}
    consts = {k: v for k, v in re.findall(r"pub const (\w+):\s*[\w:]+\s*=\s*(-?\d+);", r)}
    assert consts["ABI_VERSION"] == defines["FLACENC_HIP_ABI_VERSION"]
    pairs = {
        "OK": "FLACENC_HIP_OK", "ERR_BAD_CONFIG": "FLACENC_HIP_ERR_BAD_CONFIG", "MEM_HOST": "FLACENC_HIP_MEM_HOST",
        "MEM_DEVICE": "FLACENC_HIP_MEM_DEVICE", "FLAG_ALLOW_ORDER_32": "FLACENC_HIP_FLAG_ALLOW_ORDER_32",
        "FLAG_REFERENCE_SUM_ORDER": "FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER",
        "FLAG_NIGHTLY_SUM_ORDER": "FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER",
        "FLAG_CANONICAL_SUM_ORDER": "FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER",
        "FLAG_INTEGER_PARITY_ONLY": "FLACENC_HIP_FLAG_INTEGER_PARITY_ONLY", "COMM_ID_BYTES": "FLACENC_HIP_COMM_ID_BYTES",
        "KIND_CONSTANT": "FLACENC_HIP_KIND_CONSTANT", "KIND_VERBATIM": "FLACENC_HIP_KIND_VERBATIM",
        "KIND_FIXED": "FLACENC_HIP_KIND_FIXED", "KIND_LPC": "FLACENC_HIP_KIND_LPC",
    }
    for rust_name, c_name in pairs.items():
        assert consts[rust_name] == defines[c_name], (rust_name, consts[rust_name], c_name, defines[c_name])


def test_the_drop_in_survey_8b_names_is_there():
    """`fn encode_with_fixed_block_size<T: Source>(&Verified<config::Encoder>, T, usize) -> Result<Stream, EncodeError>`:
    the signature of src/coding.rs:645-649 / src/par.rs:355-359, in both shapes INTEGRATION.md describes."""
    r = re.sub(r"\s+", " ", _strip_rust_comments(open(RUST).read()))
    sig = (r"pub fn encode_with_fixed_block_size<T: Source>\( config: &Verified<config::Encoder>, mut src: T, "
           r"block_size: usize, \) -> Result<Stream, EncodeError>")
    assert re.search(sig, r), "bytes shape missing or its signature differs from src/coding.rs:645"
    assert re.search(r"pub fn encode_with_fixed_block_size_components<T: Source>\( config: &Verified<config::Encoder>, "
                     r"mut src: T, block_size: usize, frames_per_call: usize, \) -> Result<Stream, EncodeError>", r)
    # the calls each shape makes through the ABI, and the crate constructors the components shape rebuilds with
    for needle in ("flacenc_hip_encode_pcm(", "flacenc_hip_encode_stereo_frames(", "flacenc_hip_encode_frames(",
                   "lpc_from_record(", "fixed_lpc_from_record(", "Constant::from_parts(", "Verbatim::from_samples(",
                   "set_precomputed_bitstream(", "set_md5_digest(", "set_total_samples(", "set_block_sizes("):
        assert needle in r, needle


@pytest.mark.parametrize("name", ["flacenc_hip_encode_frames", "flacenc_hip_pack_frames_async", "flacenc_hip_synchronize",
                                  "flacenc_hip_window_weights"])
def test_round5_gaps_are_closed(name):
    assert name in rust_externs()
