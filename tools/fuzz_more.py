import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
torch.cuda.init()  # before the library creates its own HIP context
import test_gpu_parity as T
from flacenc_rs_amd import _capi
h = _capi.Handle(0)
if os.environ.get("FUZZ_FINEST") == "1":  # run every fuzzer with FLACENC_HIP_FLAG_FINEST_RICE_ORDER on both sides
    from oracle import oracle as _orc
    _g, _o = _capi.make_config, _orc.make_config
    _capi.make_config = lambda *a, **k: _g(*a, **{**k, "rice_finest_only": True})
    _orc.make_config = lambda *a, **k: _o(*a, **{**k, "rice_finest_only": True})


class _Env:  # stands in for pytest's monkeypatch
    def setenv(self, k, v):
        os.environ[k] = v


t0 = time.time(); bad = 0
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 406
for seed in range(lo, hi):
    try:
        if os.environ.get("FUZZ_KIND") == "xcand":
            T.test_extreme_candidates_fuzz(h, seed)
        elif os.environ.get("FUZZ_KIND") == "extreme":
            T.test_extreme_signals_and_layout_fuzz(h, _Env(), seed)
        elif os.environ.get("FUZZ_KIND") == "channel":
            T.test_candidate_and_channel_api_fuzz(h, seed)
        else:
            T.test_frame_pipeline_config_fuzz(h, seed)
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, str(e)[:600], flush=True)
        if bad > 5: break
print("done", seed, "failures", bad, "in", round(time.time() - t0), "s")
