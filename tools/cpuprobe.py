import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("grep -m1 'model name' /proc/cpuinfo; nproc; free -g | head -2")
from flacenc_rs_amd import _capi
from oracle import oracle as orc
x=_capi.sigen_frames(2048,2,4096,16,200.0,0.4,0.4,seed=1)
cfg=orc.make_config(lpc_order=8)
for th in (1,4,8,16,32,64,128,256):
    s,_=orc.bench_stereo_qlpc(x,16,cfg,th,2)
    print(th,'threads',round(2048*2*4096*2/s/1e6,1),'Msamples/s input')
