import sys,os,time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import numpy as np, torch
from flacenc_rs_amd import _capi
h=_capi.Handle(0)
F,n,bps=24576,4096,16
x=_capi.sigen_frames(F,2,n,bps,200.0,0.4,0.4,seed=1,nthreads=8)
host16=np.ascontiguousarray(x.transpose(0,2,1)).astype("<i2").view(np.uint8).reshape(-1)
cfg=_capi.make_frame_config(_capi.make_config(lpc_order=8),use_fixed=False)
dst=np.empty(F*(h.frame_bytes_bound(n,bps)+16),np.uint8)
for th in (1,2,4,6,8,12):
    h.set_host_threads(th)
    best=None
    for _ in range(3):
        t0=time.perf_counter(); out,lens=h.encode_pcm_stereo(host16,cfg,2,bps,n,44100,out=dst); dt=time.perf_counter()-t0
        best=dt if best is None else min(best,dt)
    print(th,"threads:",round(best*1e3,2),"ms",round(F*2*n/best/1e9,2),"G samples/s")
