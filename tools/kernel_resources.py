#!/usr/bin/env python3
"""VGPR / SGPR / spill / LDS / scratch of every kernel built into the library, read from the per-TU
objects (llvm-readelf --notes of each object's gfx950 code object).
    python tools/kernel_resources.py [build-dir] [name-substring]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
build = sys.argv[1] if len(sys.argv) > 1 else "flacenc_rs_amd/csrc/build"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
for obj in sorted(glob.glob(os.path.join(build, "*.o"))):
    fat, co = f"{tmp}/fat.bin", f"{tmp}/dev.co"
    for f in (fat, co):
        if os.path.exists(f):
            os.remove(f)
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=False)
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        continue
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    rows, cur = [], {}
    for line in txt.splitlines():
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur:
            rows.append(cur)
            cur = {}
        cur[k] = v
    if cur:
        rows.append(cur)
    for r in rows:
        if "vgpr_count" not in r:
            continue
        dem = subprocess.run(["c++filt", r.get("name", "?")], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("void flacenc_hip::(anonymous namespace)::", "").replace("(flacenc_hip::QlpcKernelArgs)", "")
        if flt and flt not in dem:
            continue
        print(f"{dem[:64]:64s} vgpr {r.get('vgpr_count'):>4} agpr {r.get('agpr_count'):>3} sgpr {r.get('sgpr_count'):>4} "
              f"vspill {r.get('vgpr_spill_count'):>4} lds {r.get('group_segment_fixed_size'):>6} "
              f"scratch {r.get('private_segment_fixed_size'):>5}")
