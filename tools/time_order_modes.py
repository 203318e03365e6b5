import torch, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","."))
from flacenc_rs_amd import _capi
F,n,bps=24576,4096,16
h=_capi.Handle(0)
x=torch.from_numpy(_capi.sigen_frames(F,2,n,bps,200.0,0.4,0.4,seed=1)).cuda()
res=torch.empty((F,752),dtype=torch.uint8,device="cuda"); rr=torch.empty((F*2,n),dtype=torch.int32,device="cuda")
st=torch.cuda.current_stream()
for name,flags in (("reference",_capi.FLAG_REFERENCE_SUM_ORDER),("nightly",_capi.FLAG_NIGHTLY_SUM_ORDER)):
    cfg=_capi.make_frame_config(_capi.make_config(lpc_order=10,flags=flags),use_fixed=True)
    for _ in range(8): h.encode_stereo_frames_device(cfg,x.data_ptr(),F,n,n,bps,res.data_ptr(),rr.data_ptr(),n,stream=st.cuda_stream)
    torch.cuda.synchronize()
