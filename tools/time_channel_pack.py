import sys,os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import numpy as np, torch
from flacenc_rs_amd import _capi
h=_capi.Handle(0)
for ch,F in ((8,2048),(1,8192),(3,4096)):
    n,bps=4096,16
    x=torch.from_numpy(_capi.sigen_frames(F,ch,n,bps,200.0,0.4,0.1,seed=7)).cuda()
    res=torch.empty((F*ch,368),dtype=torch.uint8,device="cuda"); resid=torch.empty((F*ch,n),dtype=torch.int32,device="cuda")
    cfg=_capi.make_frame_config(_capi.make_config(lpc_order=10),use_fixed=True)
    stride=(int(h._lib.flacenc_hip_frame_bytes_bound(ch,n,bps))+15)//16*16
    out=torch.empty((F,stride),dtype=torch.uint8,device="cuda"); lens=torch.zeros(F,dtype=torch.int32,device="cuda")
    st=None
    h._check(h._lib.flacenc_hip_encode_frames_async(h._h,cfg,x.data_ptr(),F,ch,n,n,bps,res.data_ptr(),resid.data_ptr(),n,st))
    go=lambda: h._check(h._lib.flacenc_hip_pack_frames_async(h._h,x.data_ptr(),F,ch,n,n,res.data_ptr(),resid.data_ptr(),n,bps,44100,0,1,out.data_ptr(),stride,lens.data_ptr(),st))
    for _ in range(20): go()
    torch.cuda.synchronize(); ms=[]
    for _ in range(8):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); go(); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    import zlib
    print(os.path.basename(os.environ.get("FLACENC_HIP_LIB","default")), f"channels={ch} F={F}: pack median {np.median(ms):.4f} ms  bytes {int(lens.sum())} crc {zlib.crc32(out.cpu().numpy().tobytes()[:1<<20])&0xffffffff:08x}")
