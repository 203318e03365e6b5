#!/bin/bash
# Two quick counter passes of the bench workload (VALU instructions per wave, VALU pipe share): tools/pmc_quick.sh <outdir> [bench args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1
}
BENCH_ARGS=("$@")
run q1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY
python3 - $OUT <<'PY'
import csv,sys,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Kernel_Name']); k=(m.group(1)+(m.group(2) or '')) if m else r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    if m.get('SQ_WAVES',0)<1000: continue
    w=m['SQ_WAVES']
    print(k, 'waves', w, 'VALU/wave %.1f SALU/wave %.1f LDS/wave %.1f'%(m['SQ_INSTS_VALU']/w, m['SQ_INSTS_SALU']/w, m['SQ_INSTS_LDS']/w),
          'valu_busy %.3f'%(4*m['SQ_ACTIVE_INST_VALU']/(1024*m['GRBM_GUI_ACTIVE']/8)), 'wait_any %.2f'%(m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']))
PY
