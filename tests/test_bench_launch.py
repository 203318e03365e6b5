"""bench.py's launch contract on CPU: `--gpus N` with no launcher in the environment must start N ranks
itself (missing #1), the printed line must carry the rank count the collective layer saw, and
the ordered exchange (lengths, 752-B component records, packed frame bytes) must deliver every frame once,
in stream order, on every rank.  `--dry-run --backend gloo` replaces the GPU analysis by stand-in records;
sharding (flacenc_rs_amd/shard.py) and the collectives are the real ones (src/par.rs:67-95 is the
reference's ordered gather)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("gather", ["records", "payload", "lengths"])
def test_gpus_flag_starts_that_many_ranks(gather):
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--gather", gather])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_observed"] == 2 and line["dry_run"] is True
    assert line["config"]["exchange_check"] == {"ok": True, "stream_frames": 128}
    assert line["config"]["gather"] == gather


def test_three_ranks_uneven_is_still_ordered():
    r = _run(["--gpus", "3", "--backend", "gloo", "--dry-run", "--steps", "1", "--frames", "5", "--gather", "payload"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 3 and line["ranks_observed"] == 3
    assert line["config"]["exchange_check"]["ok"] is True


def test_eight_ranks_ragged_frame_count():
    """The launch an 8-GPU node gets (`--gpus 8`), over gloo: eight ranks, 13 frames per rank-step that do not divide
    evenly into anything, every gather kind's stream checked on every rank."""
    for gather in ("records", "payload"):
        r = _run(["--gpus", "8", "--backend", "gloo", "--dry-run", "--steps", "1", "--frames", "13", "--gather", gather],
                 timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = _json_line(r.stdout)
        assert line["n_gpus"] == 8 and line["ranks_observed"] == 8
        assert line["config"]["exchange_check"]["ok"] is True


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run"], {"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_gloo_needs_dry_run():
    r = _run(["--gpus", "1", "--backend", "gloo"])
    assert r.returncode != 0 and "dry-run" in r.stderr


def test_no_gpu_fails_loudly():
    """Without --dry-run there is no CPU fallback: on a box without a GPU the bench must refuse."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = _run(["--steps", "1", "--warmup", "0", "--frames", "4"])
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["records", "payload"])
def test_two_real_ranks_on_the_visible_gpus(gather):
    """The real multi-rank path of bench.py (sharded sigen stream, GPU analysis, exchange on its own stream,
    stream-order checks across ranks) with two ranks.  On a one-GPU box the ranks share the device and the
    collectives run over gloo (`--shared-gpu-test`; marked in the JSON, value 0): everything but RCCL itself."""
    r = _run(["--gpus", "2", "--backend", "gloo", "--shared-gpu-test", "--frames", "1536", "--steps", "3",
              "--warmup", "1", "--gather", gather], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_observed"] == 2 and line["shared_gpu_test"] is True
    chk = line["config"]["exchange_check"]
    assert chk["ok"] is True and chk["stream_frames"] == 2 * 1536 and chk["stream_bytes"] > 0
    assert line["value"] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["records", "payload"])
def test_one_rank_exchange_through_the_librarys_communicator(gather):
    """`--comm capi`: the exchange's all-gathers run on the RCCL communicator the C library owns
    (flacenc_hip_comm_create; flacenc_hip_allgather_records_async for lengths and wire records, flacenc_hip_allgather_async
    for the packed runs) -- the calls a Rust / C++ host with one process per GPU makes -- and assemble the same stream."""
    r = _run(["--force-exchange", "--comm", "capi", "--frames", "1536", "--steps", "3", "--warmup", "1", "--gather", gather,
              "--no-secondary", "--no-cpu-baseline"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["ranks_observed"] == 1
    assert line["exchange_collective"].startswith("C ABI: flacenc_hip_comm_create")
    chk = line["config"]["exchange_check"]
    assert chk["ok"] is True and chk["stream_frames"] == 1536 and chk["stream_bytes"] > 0
    assert line["value"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["records", "payload"])
def test_one_rank_rccl_exchange(gather):
    """RCCL itself, on the one GPU of a test box: `--force-exchange` initialises a 1-rank `nccl` process group
    and the exchange step runs the real collectives -- all_reduce for the rank count, all_gather_into_tensor
    on the int32 lengths, the uint8 wire records and (payload) the packed frame runs -- exactly the calls an
    8-GPU run makes (ParSink::finalize, src/par.rs:82-94)."""
    r = _run(["--force-exchange", "--frames", "1536", "--steps", "3", "--warmup", "1", "--gather", gather,
              "--no-secondary", "--no-cpu-baseline"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["ranks_observed"] == 1 and line["collective_backend"] == "nccl"
    chk = line["config"]["exchange_check"]
    assert chk["ok"] is True and chk["stream_frames"] == 1536 and chk["stream_bytes"] > 0
    assert "RCCL all_gather" in line["config"]["gather"]
    assert line["value"] > 0
