#!/bin/bash
# counters of direct_mse_kernel at one shape: tools/pmc_direct_mse.sh <out-subdir>
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
cat > /tmp/dm_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["R"])
import torch
from flacenc_rs_amd import _capi
h = _capi.Handle(0)
F, n = 3072, 4096
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, 16, 200.0, 0.4, 0.1, seed=7)).cuda()
params = torch.empty((F * 4, 352), dtype=torch.uint8, device="cuda"); resid = torch.empty((F * 4, n), dtype=torch.int32, device="cuda")
cfg = _capi.make_config(lpc_order=8, use_direct_mse=True, window="rectangle")
for _ in range(4):
    h.stereo_qlpc_batch_device(cfg, x.data_ptr(), F, n, n, 16, params.data_ptr(), resid.data_ptr(), n, stream=0)
torch.cuda.synchronize()
PY
export R
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 /tmp/dm_one.py > $OUT/$name.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA
run b SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
run c GRBM_GUI_ACTIVE SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_LDS_IDX_ACTIVE
python3 - $OUT <<'PY'
import csv,sys,glob,collections,re
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1]+'/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(\w+_kernel\w*)(<[^>]*>)?',r['Kernel_Name']); k=(m.group(1)+(m.group(2) or '')) if m else r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); calls[k][r['Counter_Name']]+=1
for k,v in agg.items():
    if 'SQ_WAVES' not in v or 'direct' not in k: continue
    w=v['SQ_WAVES']/calls[k]['SQ_WAVES']
    print(k, 'waves/launch', w)
    for c,val in sorted(v.items()):
        print('   %-28s %12.1f per wave'%(c, val/calls[k][c]/w))
PY
