"""Multi-process (world_size 2, gloo, CPU) check of the data-parallel harness bench.py uses at N > 1:
shards tile the global batch exactly, each rank's synthetic shard equals the same rows of the global batch, and the
control-plane reductions (barrier, max of times, sum of counts) agree on every rank.  No GPU, no data-path collective."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from layoutdit_amd import dp, synth


def test_shard_range_tiles_the_batch():
    for total in (0, 1, 7, 64, 65, 513):
        for world in (1, 2, 3, 8):
            edges = [dp.shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        dp.shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r = dp.init(backend="gloo")
    try:
        per_rank = 3                                    # weak scaling: fixed images per rank
        lo, hi = dp.shard_range(per_rank * world, r.rank, r.world)
        x = synth.synth_images(hi - lo, 32, 32, seed=1234, first_index=lo)
        dp.barrier(r)
        t = dp.max_over_ranks(r, 1.0 + r.rank)          # slowest rank defines the step time
        n = dp.sum_over_ranks(r, float(hi - lo))
        q.put((r.rank, lo, hi, float(x.astype(np.float64).sum()), t, n, r.is_main))
    finally:
        dp.finalize(r)


def test_two_ranks_over_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = synth.synth_images(6, 32, 32, seed=1234)
    for rank, lo, hi, s, t, n, is_main in got:
        assert (lo, hi) == (3 * rank, 3 * rank + 3)
        assert abs(s - float(full[lo:hi].astype(np.float64).sum())) < 1e-9    # shard == rows of the global batch
        assert t == 2.0 and n == 6.0
        assert is_main == (rank == 0)


def test_bench_launches_itself_at_n_greater_than_one():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's plain form): the parent must start
    two ranks with the rendezvous in their environment and relay rank 0's single JSON line.  LDIT_BENCH_DRYRUN=1 keeps
    the ranks on the CPU (gloo), everything up to the first GPU call is the real code path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LDIT_BENCH_DRYRUN"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    per = rec.pop("per_rank")
    assert rec == {"dryrun": True, "n_gpus": 2, "max_over_ranks": 2.0, "gpus_arg": 2}
    # BASELINE metric "MFMA util % at 1/2/4/8 GPU": every rank's own throughput and GEMM roofline fraction reach rank 0's line
    assert per["ranks"] == 2
    assert per["images_per_sec"] == {"min": 100.0, "mean": 150.0, "max": 200.0, "per_rank": [100.0, 200.0]}
    assert per["gemm_mfma_roofline_frac"] == {"min": 0.1, "mean": 0.15, "max": 0.2, "per_rank": [0.1, 0.2]}
    assert per["ms_per_step"]["per_rank"] == [5.0, 6.0]
    # and the torch.distributed.run form keeps working: WORLD_SIZE present -> no self-launch
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=env2, capture_output=True,
                         text=True, timeout=300)
    one = json.loads(res.stdout.strip().splitlines()[-1])
    assert res.returncode == 0 and one["n_gpus"] == 1 and one["per_rank"]["ranks"] == 1


def test_bench_launcher_stops_all_ranks_when_one_dies_in_start_up(tmp_path):
    """a rank that dies before the rendezvous must not leave rank 0 waiting for its 10-minute timeout - the
    parent polls every child, terminates the survivors and returns the failing status, with the rank's stderr kept."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(LDIT_BENCH_DRYRUN="1", LDIT_BENCH_DRYRUN_FAIL_RANK="1", LDIT_BENCH_LOGDIR=str(tmp_path))
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 3
    assert time.time() - t0 < 120
    assert "rank 1 exited with status 3" in res.stderr
    assert "fails in start-up on request" in (tmp_path / "bench_rank1.err").read_text()
