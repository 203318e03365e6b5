#!/bin/bash
R=$PWD; OUT=$R/gpurun_out/pmc_fixed2; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --use-fixed --lpc-order 10 --frames 24576 > $OUT/$name.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run b SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
run c SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES
python3 - $OUT <<'PY'
import csv,sys,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Kernel_Name']); k=(m.group(1)+(m.group(2) or '')) if m else r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); calls[k][r['Counter_Name']]+=1
for k,v in agg.items():
    if 'SQ_WAVES' not in v: continue
    w=v['SQ_WAVES']/calls[k]['SQ_WAVES']
    print(k, 'waves/launch', w)
    for c,val in sorted(v.items()): print('   %-28s %10.1f per wave'%(c, val/calls[k][c]/w))
PY
