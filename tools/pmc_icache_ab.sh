#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc ${PMC:-SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES} --output-format csv -d $OUT/ic -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --canonical-order > $OUT/ic.log 2>&1
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$OUT/ic/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(float); n=0
    for r in csv.DictReader(open(f)):
        if "qlpc_wave4096" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]);
    disp=len([1 for r in csv.DictReader(open(f)) if "qlpc_wave4096" in r["Kernel_Name"] and r["Counter_Name"]=="SQ_WAVE_CYCLES"])
    print("$1", {k: round(v/max(disp,1)) for k,v in acc.items()}, "dispatches", disp)
PY
