"""One handle, many shapes: the handle's grow-only scratch, the two alternating marked-subframe counters and the pipelines
that mark work for a clean-up launch (sub-wave kernel, big-block kernels) must not carry state from one call into the
next.  Every call on a long-lived handle is compared with the same call on a fresh one."""
import numpy as np
import pytest

from flacenc_rs_amd import _capi

pytestmark = pytest.mark.gpu


def _loud(rng, n, bps):
    return (rng.integers(0, 2, n) * 2 - 1).astype(np.int32) * (2 ** (bps - 1) - 1)


def _calls(rng):
    """(name, callable(handle) -> tuple of arrays): shapes that take the sub-wave kernel (with and without frames it has to
    hand back), the fused 4096 kernel, the big-block pipeline, the generic kernel and the experimental estimators."""
    out = []
    for n, bps, order in ((1152, 16, 8), (256, 16, 10), (2048, 24, 12), (4096, 16, 8), (8192, 24, 24), (1000, 16, 8),
                          (576, 24, 6), (4608, 16, 10)):
        nf = int(rng.integers(1, 9))
        x = _capi.sigen_frames(nf, 2, n, bps, float(rng.uniform(20, 400)), 0.4, 0.1, seed=int(rng.integers(1, 1 << 30)))
        if bps == 24 and nf > 1:
            x[nf // 2, 0] = _loud(rng, n, bps)   # residuals beyond the exact sums: marked and redone
            x[nf // 2, 1] = -x[nf // 2, 0]
        q = _capi.make_config(lpc_order=order)
        fc = _capi.make_frame_config(q, use_fixed=True)
        out.append((f"frames {n}", lambda h, x=x, bps=bps, fc=fc: h.encode_stereo_frames(x, bps, fc)))
        out.append((f"candidates {n}", lambda h, x=x, bps=bps, q=q: h.stereo_qlpc_batch(x, bps, q)))
        out.append((f"fixed {n}", lambda h, x=x, bps=bps, fc=fc: h.fixed_lpc_batch(x, bps, fc, stereo=True)))
        xc = x.reshape(1, nf * 2, n)[:, : min(nf * 2, 8)]
        out.append((f"channels {n}", lambda h, xc=xc, bps=bps, fc=fc: h.encode_frames(xc, bps, fc)))
    x = _capi.sigen_frames(3, 2, 4096, 16, 150.0, 0.4, 0.1, seed=5)
    for steps in (0, 2):
        q = _capi.make_config(lpc_order=8, window="rectangle", use_direct_mse=True, mae_optimization_steps=steps)
        out.append((f"direct mse {steps}", lambda h, x=x, q=q: h.stereo_qlpc_batch(x, 16, q)))
    return out


def _same(a, b):
    return all(np.asarray(u).tobytes() == np.asarray(v).tobytes() for u, v in zip(a, b))


@pytest.mark.parametrize("seed", range(4))
def test_calls_on_one_handle_equal_calls_on_fresh_handles(seed):
    rng = np.random.default_rng(seed)
    calls = _calls(rng)
    want = {}
    for name, fn in calls:
        want[name] = fn(_capi.Handle(0))
    h = _capi.Handle(0)
    order = rng.permutation(len(calls) * 3) % len(calls)
    for i in order:
        name, fn = calls[int(i)]
        got = fn(h)
        assert _same(got, want[name]), (seed, name)
