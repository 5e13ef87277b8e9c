"""Data-parallel sharding on the GPU: two processes (gloo control plane, both on the box's one card) each run the
encoder on their contiguous shard of a global batch; gathered back, the taps are BIT-IDENTICAL to one process running
the whole batch - images never mix and the arithmetic does not depend on the batch an image rides in, so bench.py's
N > 1 path (no data-path collective) computes exactly what N = 1 computes."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp          # noqa: E402

from layoutdit_amd import config as cfgs, dp, synth     # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from layoutdit_amd.modeling import DiTEncoder
    r = dp.init(backend="gloo")
    try:
        cfg = cfgs.vit_tiny()
        m = DiTEncoder(cfg).load_numpy(synth.synth_weights(cfg, 0)).to("cuda:0").eval()
        lo, hi = dp.shard_range(total, r.rank, r.world)
        x = torch.from_numpy(synth.synth_images(hi - lo, 224, 224, seed=1234, first_index=lo)).to("cuda:0")
        dp.barrier(r)
        with torch.no_grad():
            out = m(x).hidden_states
        torch.cuda.synchronize()
        n = dp.sum_over_ranks(r, float(hi - lo))
        q.put((r.rank, lo, hi, n, {t: out[t].cpu().numpy() for t in cfg.taps}))
    finally:
        dp.finalize(r)


def test_two_gpu_ranks_reproduce_the_unsharded_batch_bit_for_bit():
    world, total, port = 2, 5, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(g[1], g[2]) for g in got] == [(0, 3), (3, 5)] and all(g[3] == total for g in got)
    from layoutdit_amd.modeling import DiTEncoder
    cfg = cfgs.vit_tiny()
    m = DiTEncoder(cfg).load_numpy(synth.synth_weights(cfg, 0)).to("cuda:0").eval()
    with torch.no_grad():
        full = m(torch.from_numpy(synth.synth_images(total, 224, 224, seed=1234)).to("cuda:0")).hidden_states
    for t in cfg.taps:
        ref = full[t].cpu().numpy()
        np.testing.assert_array_equal(np.concatenate([g[4][t] for g in got]), ref)


def test_rccl_control_plane_one_rank():
    """The `nccl` (= RCCL) branch of dp.py - init with device_id, barrier(device_ids), all_reduce MAX / SUM, destroy - on
    the box's one GPU with a one-rank group, in a child process (a process group is per process)."""
    import subprocess
    import sys
    code = (
        "import os, torch\n"
        "from layoutdit_amd import dp\n"
        "r = dp.init(backend='nccl', force_group=True)\n"
        "assert r.backend == 'nccl' and torch.distributed.is_initialized()\n"
        "dp.barrier(r)\n"
        "assert dp.max_over_ranks(r, 3.5) == 3.5 and dp.sum_over_ranks(r, 2.0) == 2.0\n"
        "assert dp.gather_over_ranks(r, [1.5, 0.25, 7.0]) == [[1.5, 0.25, 7.0]]      # bench.py's per-rank report over RCCL\n"
        "g = torch.ones(1 << 20, device='cuda:0')\n"
        "torch.distributed.all_reduce(g)\n"
        "torch.cuda.synchronize(); assert float(g.sum()) == float(1 << 20)\n"
        "dp.finalize(r); print('rccl-ok')\n")
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "rccl-ok" in res.stdout, res.stderr[-2000:]


def _train_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from layoutdit_amd import training
    from layoutdit_amd.modeling import DiTEncoder
    from tests.golden.make_golden_grad import upstream
    r = dp.init(backend="gloo")
    try:
        cfg = cfgs.vit_micro()
        total = 4
        lo, hi = dp.shard_range(total, r.rank, r.world)
        m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(synth.synth_weights(cfg, 3)).to("cuda:0").train()
        x = torch.from_numpy(synth.synth_images(hi - lo, 64, 64, seed=5, kind="uniform", first_index=lo)).to("cuda:0")
        dt = [torch.from_numpy(d[lo:hi].copy()).to("cuda:0") for d in upstream(cfg, total, cfg.tokens(64, 64), 9)]
        step = training.TrainStep(m, r, lr=1e-3, weight_decay=0.0, dtaps=dt, drop_path_rate=0.0, img_size=(64, 64))
        p0 = step.flat_params.cpu().numpy().copy()
        step.step(x)
        torch.cuda.synchronize()
        q.put((r.rank, p0, step.state.grads.cpu().numpy(), step.flat_params.cpu().numpy()))
    finally:
        dp.finalize(r)


def test_train_step_two_ranks_average_gradients():
    """DP train step (BASELINE configs[2]) with two ranks on the box's one card (gloo control + data plane for the test;
    RCCL on a real node): after the per-layer bucketed all-reduce both ranks hold the SAME summed gradient, equal to a
    single process's gradient on the concatenated batch, and the SAME updated parameters = AdamW on the averaged gradient."""
    from layoutdit_amd import training
    from layoutdit_amd.modeling import DiTEncoder
    from tests.golden.make_golden_grad import upstream
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0, n0), (_, _, g1, n1) = got
    np.testing.assert_array_equal(g0, g1)                 # all-reduced gradient identical on both ranks
    np.testing.assert_array_equal(n0, n1)                 # ... hence identical replicas after the step
    cfg = cfgs.vit_micro()
    m = DiTEncoder(cfg, compute_dtype="bf16").load_numpy(synth.synth_weights(cfg, 3)).to("cuda:0").train()
    x = torch.from_numpy(synth.synth_images(4, 64, 64, seed=5, kind="uniform")).to("cuda:0")
    dt = [torch.from_numpy(d).to("cuda:0") for d in upstream(cfg, 4, cfg.tokens(64, 64), 9)]
    single = training.TrainStep(m, None, lr=1e-3, weight_decay=0.0, dtaps=dt, drop_path_rate=0.0, img_size=(64, 64))
    single.step(x)
    full = single.state.grads.cpu().numpy().astype(np.float64)
    err = np.linalg.norm(g0.astype(np.float64) - full) / np.linalg.norm(full)
    # same arithmetic up to (a) the order of the fp32 sums over the batch and (b) the GEMM tiling the row count selects
    # (34 rows per rank take the split-K skinny kernel, 68 rows the 128 x 128 tiles): a few bf16 roundings flip
    assert err < 1e-3, err
    # AdamW (step 1) on the AVERAGED gradient, float64 on the host
    g = g0.astype(np.float64) / world
    mo, vo = 0.1 * g, 0.001 * g * g
    want = p0.astype(np.float64) - (1e-3 / 0.1) * mo / (np.sqrt(vo) / np.sqrt(0.001) + 1e-8)
    assert np.abs(n0 - want).max() < 2e-6


def test_train_step_bucketed_all_reduce_over_rccl_one_rank():
    """The `nccl` branch of TrainStep - staged backward, one async all-reduce per layer bucket on RCCL's stream, wait, fused
    AdamW - with a one-rank RCCL group on the box's GPU: must equal the un-staged single-process step bit for bit."""
    import subprocess
    import sys
    code = (
        "import torch\n"
        "from layoutdit_amd import config as cfgs, dp, synth, training\n"
        "from layoutdit_amd.modeling import DiTEncoder\n"
        "from tests.golden.make_golden_grad import upstream\n"
        "r = dp.init(backend='nccl', force_group=True)\n"
        "cfg = cfgs.vit_micro()\n"
        "x = torch.from_numpy(synth.synth_images(4, 64, 64, seed=5, kind='uniform')).to('cuda:0')\n"
        "dt = [torch.from_numpy(d).to('cuda:0') for d in upstream(cfg, 4, cfg.tokens(64, 64), 9)]\n"
        "out = []\n"
        "for rank, force in ((r, True), (None, False)):\n"
        "    m = DiTEncoder(cfg, compute_dtype='bf16').load_numpy(synth.synth_weights(cfg, 3)).to('cuda:0').train()\n"
        "    st = training.TrainStep(m, rank, lr=1e-3, dtaps=dt, drop_path_rate=0.0, img_size=(64, 64), force_comm=force)\n"
        "    assert st.comm == force\n"
        "    st.step(x); st.step(x)\n"
        "    torch.cuda.synchronize(); out.append((st.state.grads.clone(), st.flat_params.clone()))\n"
        "assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])\n"
        # the overlap structure, asserted on the GPU timeline (events on the compute stream, recorded in host order): bucket l is
        # handed to RCCL BEFORE stage l-1's backward kernels are even enqueued, so the collective has that whole backward stage
        # to run under; nothing waits until every stage has been enqueued; AdamW comes after the last wait
        "m = DiTEncoder(cfg, compute_dtype='bf16').load_numpy(synth.synth_weights(cfg, 3)).to('cuda:0').train()\n"
        "st = training.TrainStep(m, r, lr=1e-3, dtaps=dt, drop_path_rate=0.0, img_size=(64, 64), force_comm=True)\n"
        "tr = []\n"
        "st.step(x, _trace=tr); torch.cuda.synchronize()\n"
        "L = cfg.num_hidden_layers\n"
        "names = [(w, s) for w, s, _ in tr]\n"
        "want = []\n"
        "for s in range(L, -1, -1): want += [('backward_done', s), ('allreduce_issued', s)]\n"
        "assert names == want + [('waited', None), ('adamw_done', None)], names\n"
        "ev = {(w, s): e for w, s, e in tr}\n"
        "for s in range(L, 0, -1):\n"
        "    under = ev[('allreduce_issued', s)].elapsed_time(ev[('backward_done', s - 1)])\n"
        "    assert under > 0.0, (s, under)      # stage s-1's backward ran AFTER bucket s was with RCCL\n"
        "assert ev[('allreduce_issued', 0)].elapsed_time(ev[('adamw_done', None)]) > 0.0\n"
        "dp.finalize(r); print('rccl-train-ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=root)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert res.returncode == 0 and "rccl-train-ok" in res.stdout, res.stderr[-3000:]


def _ddp_model(cfgs_mod, synth_mod, torch_mod):
    from layoutdit_amd.modeling import DiTWithFPN
    cfg = cfgs_mod.vit_micro()
    cfg.drop_path_rate = 0.0                    # no coin flips: the two-rank and the one-process gradients are comparable
    cfg.taps = [1, 1, 2, 3]
    torch_mod.manual_seed(0)
    m = DiTWithFPN(config=cfg, compute_dtype="bf16")
    m.backbone.dit.load_numpy(synth_mod.synth_weights(cfg, 3))
    g = torch_mod.Generator().manual_seed(7)
    with torch_mod.no_grad():
        for p in m.fpn.parameters():
            if p.dim() == 1:
                p.copy_(0.05 * torch_mod.randn(p.shape, generator=g))
    return cfg, m.to("cuda:0").train()


def _head_weights(feats, total, lo, hi):
    """Fixed synthetic upstream gradients standing in for the detection head (RPN / RoI heads are torchvision code, out of
    scope): one weight tensor per FPN level for the GLOBAL batch, sliced to this rank's images."""
    ws = {}
    for i, (k, f) in enumerate(feats.items()):
        shape = (total,) + tuple(f.shape[1:])
        n = int(np.prod(shape))
        ws[k] = torch.from_numpy((synth.normal(90 + i, 9, n) / np.sqrt(n)).astype(np.float32).reshape(shape)[lo:hi].copy()).to(f.device)
    return ws


def _ddp_worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from torch.nn.parallel import DistributedDataParallel as DDP
    r = dp.init(backend="gloo")
    try:
        cfg, m = _ddp_model(cfgs, synth, torch)
        # the wrapper a maintainer of the reference would use (ref README.md:59 "Add support for distributed training");
        # find_unused_parameters: BEiT's mask token and pooler LayerNorm never receive a gradient (with HF's BeitModel neither)
        ddp = DDP(m, device_ids=[0], find_unused_parameters=True)
        opt = torch.optim.AdamW(ddp.parameters(), lr=1e-4, weight_decay=0.0)          # ref trainer.py:62-68
        lo, hi = dp.shard_range(total, r.rank, r.world)
        x = torch.from_numpy(synth.synth_images(hi - lo, 64, 64, seed=5, kind="uniform", first_index=lo)).to("cuda:0")
        grads = None
        for it in range(2):                                                           # two iterations: the reducer must re-arm
            opt.zero_grad()
            feats = ddp(x)
            ws = _head_weights(feats, total, lo, hi)
            loss = sum((f * ws[k]).sum() for k, f in feats.items())                   # ref trainer.py:169-178
            loss.backward()
            if it == 0:
                grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}
            opt.step()
        torch.cuda.synchronize()
        params = {k: p.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
        q.put((r.rank, grads, params))
    finally:
        dp.finalize(r)


def test_dit_with_fpn_under_distributed_data_parallel_two_ranks():
    """North-star: "RCCL all-reduce of the detection-head gradients ... only when training".  DiTWithFPN (FPN parameters + the
    encoder through the autograd path, its parameters re-homed as views of a flat block at the first training forward) wrapped in
    torch's DistributedDataParallel, loss.backward() + torch.optim.AdamW as ref trainer.py:62-68,169-180, two ranks on the box's
    one card (gloo here, RCCL on a real node - same wrapper): FPN AND encoder gradients identical on both ranks and equal to
    1 / world x the single-process gradient of the concatenated batch; the replicas stay identical after two optimizer steps."""
    world, total, port = 2, 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in range(world)), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, p0), (_, g1, p1) = got
    assert set(g0) == set(g1) and any(k.startswith("fpn.") for k in g0) and any(k.startswith("backbone.dit.encoder") for k in g0)
    for k in g0:
        np.testing.assert_array_equal(g0[k], g1[k], err_msg=k)        # DDP's averaged gradient: the same bits on both ranks
    for k in p0:
        np.testing.assert_array_equal(p0[k], p1[k], err_msg=k)        # ... hence identical replicas after the optimizer steps
    assert not any(k.endswith("mask_token") or "pooler" in k for k in g0)
    # one process, the concatenated batch
    cfg, m = _ddp_model(cfgs, synth, torch)
    x = torch.from_numpy(synth.synth_images(total, 64, 64, seed=5, kind="uniform")).to("cuda:0")
    feats = m(x)
    ws = _head_weights(feats, total, 0, total)
    sum((f * ws[k]).sum() for k, f in feats.items()).backward()
    torch.cuda.synchronize()
    worst = ("", 0.0)
    for k, p in m.named_parameters():
        if p.grad is None:
            assert k not in g0
            continue
        full = p.grad.detach().cpu().numpy().astype(np.float64) / world
        err = np.linalg.norm(g0[k].astype(np.float64) - full) / max(np.linalg.norm(full), 1e-30)
        worst = max(worst, (k, err), key=lambda t: t[1])
        # same arithmetic up to the order of the fp32 sums over the batch and the GEMM tiling the row count selects (bf16 operands)
        assert err < 5e-3, (k, err)
    print("DDP two ranks vs one process, worst tensor:", worst)
