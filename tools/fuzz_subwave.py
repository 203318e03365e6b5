"""Fuzz of the sub-wave kernel (blocks of 512 / 1024 / 2048 / 576 / 1152 / 2304 samples) against the generic kernel it
replaces (FLACENC_HIP_FLAG_GENERIC_KERNEL: byte-identical records, rows, keys and frame results and channel results required) and, every
`--oracle-every` seeds, against the CPU oracle: python tools/fuzz_subwave.py <seed lo> <seed hi> [--oracle-every N]"""
import os
import sys
import time

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np  # noqa: E402

from flacenc_rs_amd import _capi  # noqa: E402

SIZES = [256, 512, 1024, 2048, 288, 576, 1152, 2304]


def signals(rng, nf, n, bps):
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    x = np.empty((nf, 2, n), np.int32)
    for f in range(nf):
        for c in range(2):
            kind = int(rng.integers(0, 12))
            t = np.arange(n)
            if kind == 0:
                v = np.zeros(n)
            elif kind == 1:
                v = np.full(n, rng.integers(lo, hi + 1))
            elif kind == 2:
                v = rng.integers(lo, hi + 1, n)
            elif kind == 3:
                v = np.where(t % 2 == 0, hi, lo)
            elif kind == 4:
                v = rng.integers(-3, 4, n)
                v[rng.integers(0, n, int(rng.integers(1, 6)))] = rng.choice([lo, hi])
            elif kind == 5:
                v = np.round(hi * rng.random() * np.sin(t / rng.uniform(1.5, 200.0) + rng.random()))
            elif kind == 6:
                v = np.round(hi * 0.5 * np.sin(t / rng.uniform(2, 60)) + rng.normal(0, hi * rng.choice([1e-4, 1e-2, 0.2]), n))
            elif kind == 7:
                v = np.where(t < rng.integers(1, n), 0, rng.integers(lo // 2, hi // 2 + 1, n))
            elif kind == 8:
                v = np.cumsum(rng.integers(-40, 41, n))
            elif kind == 9:
                v = (t * int(rng.integers(-2000, 2000))) // 7
            elif kind == 10:
                v = np.repeat(rng.integers(lo, hi + 1, (n + 63) // 64), 64)[:n]
            else:
                v = rng.integers(-1, 2, n) * hi
            x[f, c] = np.clip(v, lo, hi).astype(np.int32)
        r = rng.random()
        if r < 0.15:
            x[f, 1] = x[f, 0]
        elif r < 0.3:
            x[f, 1] = -x[f, 0]
        elif r < 0.45:
            x[f, 1] = np.clip((x[f, 0].astype(np.int64) * 7) // 8 + rng.integers(-2, 3, n), lo, hi)
    return x


def _diff_channels(cr, cg, crr, cgr, chn):
    """Where two encode_frames results part ways: first differing (frame, channel) and the fields that differ there."""
    out = " (%d channels)" % chn
    a, b = cr.reshape(-1), cg.reshape(-1)
    for i in range(a.size):
        if a[i].tobytes() != b[i].tobytes() or not np.array_equal(crr.reshape(a.size, -1)[i], cgr.reshape(a.size, -1)[i]):
            out += " subframe %d:" % i
            for f in ("kind", "analysis_status", "dc_offset", "bits"):
                if a[i][f] != b[i][f]:
                    out += " %s %s != %s;" % (f, a[i][f], b[i][f])
            for f in a[i]["params"].dtype.names:
                if not np.array_equal(a[i]["params"][f], b[i]["params"][f]):
                    out += " params.%s %s != %s;" % (f, np.asarray(a[i]["params"][f]).ravel()[:14], np.asarray(b[i]["params"][f]).ravel()[:14])
            ra, rb = crr.reshape(a.size, -1)[i], cgr.reshape(a.size, -1)[i]
            if not np.array_equal(ra, rb):
                k = int(np.flatnonzero(ra != rb)[0])
                out += " residual differs from sample %d (%d of %d): %s != %s" % (k, int((ra != rb).sum()), ra.size, ra[k:k + 4], rb[k:k + 4])
            break
    return out


def main():
    lo_seed, hi_seed = int(sys.argv[1]), int(sys.argv[2])
    oracle_every = int(sys.argv[sys.argv.index("--oracle-every") + 1]) if "--oracle-every" in sys.argv else 10
    h = _capi.Handle(0)
    from oracle import oracle as orc
    import test_gpu_parity as T
    t0 = time.time()
    bad = 0
    marked = 0
    for seed in range(lo_seed, hi_seed):
        rng = np.random.default_rng(770000 + seed)
        n = int(rng.choice(SIZES))
        bps = int(rng.choice([8, 12, 16, 16, 16, 20, 24, 24]))
        order = int(rng.integers(1, 13))
        qkw = dict(lpc_order=order, quant_precision=int(rng.integers(2, 16)),
                   window=("rectangle" if rng.random() < 0.2 else ("tukey", float(np.round(rng.random(), 2)))),
                   max_rice_parameter=int(rng.choice([0, 3, 7, 14, 15, 30, 30])), rice_finest_only=bool(rng.random() < 0.1))
        fkw = dict(use_constant=bool(rng.random() < 0.85), use_fixed=bool(rng.random() < 0.75), use_lpc=bool(rng.random() < 0.9),
                   use_leftside=bool(rng.random() < 0.8), use_rightside=bool(rng.random() < 0.8),
                   use_midside=bool(rng.random() < 0.8), fixed_max_order=int(rng.integers(0, 5)),
                   fixed_partitions=int(rng.choice([1, 2, 4, 8, 16, 16, 32, 64, 3, 12])))
        nf = int(rng.integers(1, 20))
        x = signals(rng, nf, n, bps)
        tag = (seed, n, bps, qkw, fkw, nf)
        try:
            q0, q1 = _capi.make_config(**qkw), _capi.make_config(flags=_capi.FLAG_GENERIC_KERNEL, **qkw)
            gp, gr = h.stereo_qlpc_batch(x, bps, q0)
            pp, pr = h.stereo_qlpc_batch(x, bps, q1)
            assert gp.tobytes() == pp.tobytes() and np.array_equal(gr, pr), "stereo candidates"
            flat = x.reshape(nf * 2, n)
            bpsv = rng.choice([bps, bps, min(bps + 1, 25)], nf * 2).astype(np.uint8)
            gp, gr, _, _ = h.qlpc_batch(flat, bpsv, q0)
            pp, pr, _, _ = h.qlpc_batch(flat, bpsv, q1)
            assert gp.tobytes() == pp.tobytes() and np.array_equal(gr, pr), "plain candidates"
            f0, f1 = _capi.make_frame_config(q0, **fkw), _capi.make_frame_config(q1, **fkw)
            if fkw["use_fixed"]:
                a = h.fixed_lpc_batch(x, bps, f0, stereo=True)
                b = h.fixed_lpc_batch(x, bps, f1, stereo=True)
                assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), "fixed stereo"
                a = h.fixed_lpc_batch(flat, bpsv, f0)
                b = h.fixed_lpc_batch(flat, bpsv, f1)
                assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), "fixed plain"
            got, gres = h.encode_stereo_frames(x, bps, f0)
            gen, genres = h.encode_stereo_frames(x, bps, f1)
            assert got.tobytes() == gen.tobytes() and np.array_equal(gres, genres), "frames"
            chn = int(rng.choice([1, 3, 8]))
            nfc = (nf * 2) // chn
            if nfc:
                xc = flat[: nfc * chn].reshape(nfc, chn, n)
                cr, crr = h.encode_frames(xc, bps, f0)
                cg, cgr = h.encode_frames(xc, bps, f1)
                assert cr.tobytes() == cg.tobytes() and np.array_equal(crr, cgr), "independent-channel frames" + _diff_channels(cr, cg, crr, cgr, chn)
            marked += int((np.abs(x.astype(np.int64)).max(axis=(1, 2)) >= (1 << 22)).sum())
            if seed % oracle_every == 0:
                okw = {k: v for k, v in fkw.items() if not k.startswith("fixed_")}
                ocfg = orc.make_frame_config(orc.make_config(acorr=orc.ACORR_CANONICAL, **qkw),
                                             fixed=orc.make_fixed_config(max_order=fkw["fixed_max_order"],
                                                                         partitions=fkw["fixed_partitions"],
                                                                         sum_mode=orc.SUMABS_CANONICAL), **okw)
                want, wres = orc.encode_stereo_frames_cfg(x, bps, ocfg)
                T._check_frames_against_oracle(x, bps, got, gres, want, wres)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", tag, str(e)[:500], flush=True)
            if bad > 5:
                break
    print("done", seed, "failures", bad, "in", round(time.time() - t0), "s; frames with samples at a quarter of full scale and more:", marked)


main()
