"""profiles/headline_phases.json from the output of tools/phase_insts.sh (dynamic VALU instructions per wave of
diagnostic builds that end the headline kernel after phase k): the per-phase cost next to the floor of the formulation,
tagged with the kernel source fingerprint (bench.py puts it into the roofline object of runs of that very source).
    python tools/make_headline_phases.py gpurun_out/r04_phase/summary.txt r04 > profiles/headline_phases.json"""
import json
import os
import re
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

# fraction of the HBM roof the same source measured (third argument; bench.py's roofline.frac)
FRAC = float(sys.argv[3]) if len(sys.argv) > 3 else 0.373
cum = {}
for line in open(sys.argv[1]):
    m = re.match(r"(libflacenc_\w+)\s+VALU/wave\s+([\d.]+)", line)
    if m:
        cum[m.group(1)] = float(m.group(2))
order = ["libflacenc_exit0", "libflacenc_exit1", "libflacenc_exit2", "libflacenc_exit3", "libflacenc_hip"]
names = ["load (HBM -> LDS images, window table)", "window + autocorrelation (9 lags x 64 samples per lane)",
         "Levinson-Durbin + quantisation + order certificate (4 subframes on 4 lanes of wave 0)", "residual (compute_error)",
         "bit planes + Rice search + encode_frame decision + store of the two chosen rows"]
# what the formulation cannot do without, per wave (64 samples per lane, order 8, stereo roles averaged):
floors = [
    (30, "addresses of 8 + 8 16-byte moves per lane"),
    (576 + 216 + 62 + 54 + 36,
     "576 fma (9 lags x 64) + 216 conversions ((64 + 8) x cvt_f32_i32, mul_f32, cvt_f64_f32) + 62 for the 64 lane sums "
     "(round 5: reduce-scatter on v_permlane32/16_swap + four row shifts; one chain per lane, no in-lane tree) + 54 mid / "
     "side forming (role average) + 36 min / max"),
    (163 + 33, "650 serial instructions of the recursion + ~130 of the order certificate, on one wave of four"),
    (256 + 64 + 128 + 54, "256 v_dot2_i32_i16 (8 taps x 64 / 2) + 64 packs + 128 shift / subtract + 54 mid / side forming"),
    (128 + 120 + 33 + 21 + 40 + 98 + 70 + 60 + 40,
     "128 sign-magnitude + 120 carry-save + 33 plane adds + 21 plane sums + 40 table entries (4 parameters) + 98 merge "
     "levels + 70 level totals + 60 decision / records + 40 store addressing (two of four waves)"),
]
missing = [k for k in order if k not in cum]
if missing:
    # (round 6: a collection ran without ab/libflacenc_exit*.so and wrote a table without phases, which bench.py attached)
    sys.exit("make_headline_phases.py: no counters for %s in %s -- run tools/build_exit_variants.sh before the gpurun call" % (
        ", ".join(missing), sys.argv[1]))
phases, prev = [], 0.0
for key, name, (floor, what) in zip(order, names, floors):
    if key not in cum:
        continue
    phases.append({"phase": name, "valu_insts_per_wave": round(cum[key] - prev, 1), "floor": floor, "floor_is": what})
    prev = cum[key]
total = cum.get("libflacenc_hip")
floor_total = sum(p["floor"] for p in phases)
out = {
    "round": sys.argv[2] if len(sys.argv) > 2 else "",
    "kernel": bench.kernel_name(bench.parse_args([])),
    "kernel_source_sha": bench.kernel_source_sha(),
    "method": "tools/phase_insts.sh: SQ_INSTS_VALU / SQ_WAVES of builds that end the program after phase k (-DFLACENC_EXIT_AFTER=k)",
    "valu_insts_per_wave": total,
    "valu_floor_insts_per_wave": floor_total,
    "phases": phases,
    "reading": "at the measured %.3f of the HBM roof the 0.50 target would need at most %d instructions per wave at today's "
               "issue efficiency; the floor of this formulation is %d" % (FRAC, int(total * FRAC / 0.50) if total else 0, floor_total),
}
print(json.dumps(out, indent=1))
