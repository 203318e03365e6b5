#!/bin/bash
# Counter passes of a stereo QLPC batch shape (one group per run): tools/pmc_bigres.sh <out-subdir> [n] [order] [frames] [bps]
R=$PWD; OUT=$R/gpurun_out/$1; N=${2:-8192}; P=${3:-24}; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
fr=${4:-$((50331648 / N))}; BPS=${5:-24}
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/prof_config.py --n $N --order $P --frames $fr --bps $BPS > $OUT/$name.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA
run b SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
run c SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run d TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run e FETCH_SIZE
run f WRITE_SIZE
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
    for c,val in sorted(v.items()):
        print('   %-28s %12.1f per wave   %16.0f per launch'%(c, val/calls[k][c]/w, val/calls[k][c]))
PY
