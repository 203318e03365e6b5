"""profiles/headline_pmc.json from a pmc_summary.json: HBM bytes per launch (FETCH_SIZE x 2 on gfx950 for
16 B/lane streaming reads, WRITE_SIZE exact -- MI355X_MICROARCH.md, HBM) and VALU instructions per wave,
tagged with the fingerprint of the kernel source they were measured on (bench.py attaches them only to
runs of that very source and of the default configuration).
    python tools/make_headline_pmc.py pmc_summary.json r02 > profiles/headline_pmc.json"""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

summary = json.load(open(sys.argv[1]))
tag = sys.argv[2] if len(sys.argv) > 2 else ""
key = [k for k in summary if "qlpc_wave4096_kernel" in k or k.endswith("flacenc_hip::")]
c = summary[key[0]] if key else next(iter(summary.values()))
g = lambda n: c[n]["mean_per_dispatch"] if n in c else None
args = bench.parse_args([])
algo = bench.ALGO_BYTES_PER_SAMPLE * args.frames * 2 * args.block_size
rd, wr = g("FETCH_SIZE"), g("WRITE_SIZE")
out = {
    "round": tag,
    "kernel": bench.kernel_name(args),
    "kernel_source_sha": bench.kernel_source_sha(),
    "launch": f"{args.frames} stereo frames x {args.block_size} samples, LPC order {args.lpc_order} (bench.py default)",
    "FETCH_SIZE_KB_raw": rd, "WRITE_SIZE_KB_raw": wr,
    "correction": "gfx950: FETCH_SIZE x2 for 16 B/lane streaming reads, WRITE_SIZE exact (MI355X_MICROARCH.md, HBM)",
    "read_bytes_per_launch": rd * 1024 * 2 if rd else None,
    "write_bytes_per_launch": wr * 1024 if wr else None,
    "bytes_per_launch": (rd * 1024 * 2 + wr * 1024) if rd and wr else None,
    "algorithmic_bytes_per_launch": algo,
    "valu_insts_per_wave": g("SQ_INSTS_VALU") / g("SQ_WAVES") if g("SQ_INSTS_VALU") and g("SQ_WAVES") else None,
    "salu_insts_per_wave": g("SQ_INSTS_SALU") / g("SQ_WAVES") if g("SQ_INSTS_SALU") and g("SQ_WAVES") else None,
    "lds_insts_per_wave": g("SQ_INSTS_LDS") / g("SQ_WAVES") if g("SQ_INSTS_LDS") and g("SQ_WAVES") else None,
    "wave_cycles_per_wave": 4 * g("SQ_WAVE_CYCLES") / g("SQ_WAVES") if g("SQ_WAVE_CYCLES") and g("SQ_WAVES") else None,
    "wait_any_frac": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES") else None,
    "lds_bank_conflict_frac": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_BANK_CONFLICT") and g("SQ_LDS_IDX_ACTIVE") else None,
    # share of the SIMDs' cycles in which the VALU pipe is executing, from counters alone (no assumed clock):
    # SQ_ACTIVE_INST_VALU counts 4-cycle quanta over all SIMDs, GRBM_GUI_ACTIVE the kernel's cycles summed over
    # the 8 XCDs
    "valu_busy_frac": (4.0 * g("SQ_ACTIVE_INST_VALU") / (bench.N_SIMDS * g("GRBM_GUI_ACTIVE") / 8.0))
                      if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE") else None,
    "valu_busy_frac_formula": "4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)",
}
if out["bytes_per_launch"]:
    out["ratio_to_algorithmic"] = out["bytes_per_launch"] / algo
print(json.dumps(out, indent=1))
