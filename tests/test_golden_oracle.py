"""The oracle reproduces the committed golden vectors (guards the oracle against drift)."""
import os
import zlib

import numpy as np
import pytest

import util
from oracle import oracle as orc

GOLD = np.load(os.path.join(util.GOLDEN, "qlpc_golden.npz"))
NAMES = sorted({k.split("/")[0] for k in GOLD.files})


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_golden(name):
    n, order, bps = (int(v) for v in GOLD[f"{name}/meta"])
    x = GOLD[f"{name}/input"].astype(np.int32)
    r = orc.estimated_qlpc(x, bps, orc.make_config(lpc_order=order))
    assert np.array_equal(r["autocorr"], GOLD[f"{name}/autocorr"])
    assert np.array_equal(r["lpc_coefs"], GOLD[f"{name}/lpc_coefs"])
    assert r["coefs"].tolist() == GOLD[f"{name}/coefs"].tolist()
    sc = GOLD[f"{name}/scalars"]
    assert [r["order"], r["shift"], r["rice_order"], r["code_bits"], r["subframe_bits"],
            r["sum_quotients"], zlib.crc32(r["residual"].tobytes())] == sc.tolist()
    assert r["rice_params"].tolist() == GOLD[f"{name}/rice_params"].tolist()
