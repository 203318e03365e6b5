#!/bin/bash
# tools/pmc_chain_kernel.sh OUT: counters of acorr_reference_mfma_kernel in the two-pass form of the headline launch
# (tools/prof_two_pass.py): how busy is the vector pipe (the f64 MFMA and its operands' conversions share it), and where the waves wait.
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pass$i -- python3 $R/tools/prof_two_pass.py 10 > $OUT/pass$i.log 2>&1
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/pass*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'acorr_reference_mfma' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
m={k:sum(v)/len(v) for k,v in agg.items()}
for k in sorted(m): print('%-28s %.5g'%(k,m[k]))
if 'SQ_ACTIVE_INST_VALU' in m and 'GRBM_GUI_ACTIVE' in m:
    print('valu issue fraction = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE / 8) = %.3f' % (4*m['SQ_ACTIVE_INST_VALU']/(1024*m['GRBM_GUI_ACTIVE']/8)))
PY
