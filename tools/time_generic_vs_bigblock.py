import sys,os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import numpy as np, torch
from flacenc_rs_amd import _capi
h=_capi.Handle(0)
for order in (16,24):
    n,F=4096,12288
    x=torch.from_numpy(_capi.sigen_frames(F,2,n,16,200.0,0.4,0.1,seed=7)).cuda()
    params=torch.empty((F*4,352),dtype=torch.uint8,device="cuda"); resid=torch.empty((F*4,n),dtype=torch.int32,device="cuda")
    for flags,name in ((0,"big-block path"),(_capi.FLAG_GENERIC_KERNEL,"generic kernel")):
        cfg=_capi.make_config(lpc_order=order,flags=flags)
        st=torch.cuda.current_stream()
        go=lambda: h.stereo_qlpc_batch_device(cfg,x.data_ptr(),F,n,n,16,params.data_ptr(),resid.data_ptr(),n,stream=st.cuda_stream)
        for _ in range(3): go()
        torch.cuda.synchronize(); ms=[]
        for _ in range(6):
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            a.record(st); go(); b.record(st); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
        print(f"n=4096 order={order} {name}: {np.median(ms):.3f} ms -> {F*2*n/np.median(ms)/1e6:.1f} G input samples/s")
