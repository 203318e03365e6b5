#!/bin/bash
# Counter passes of an arbitrary script: tools/pmc_script.sh <outdir under gpurun_out> <script.py> [args]
# prints per-kernel VALU instructions per wave, the VALU pipe share, LDS bank-conflict share and wait share
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $SCRIPT "${ARGS[@]}" > $OUT/$name.log 2>&1
}
ARGS=("$@")
run q1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY
run q2 SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python3 - $OUT <<'PY'
import csv,sys,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Kernel_Name']); k=(m.group(1)+(m.group(2) or '')) if m else r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    if m.get('SQ_WAVES',0)<64: continue
    w=m['SQ_WAVES']
    print(k, 'waves', w, 'VALU/wave %.1f SALU/wave %.1f LDS/wave %.1f'%(m['SQ_INSTS_VALU']/w, m['SQ_INSTS_SALU']/w, m['SQ_INSTS_LDS']/w),
          'valu_busy %.3f'%(4*m['SQ_ACTIVE_INST_VALU']/(1024*m['GRBM_GUI_ACTIVE']/8)), 'wait_any %.2f'%(m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']),
          'lds_conflict %.3f'%(m.get('SQ_LDS_BANK_CONFLICT',0)/max(1.0,m.get('SQ_ACTIVE_INST_LDS',1))), 'wave_cycles/wave %.0f'%(m['SQ_WAVE_CYCLES']/w))
PY
