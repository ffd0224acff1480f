"""N > 1 path on CPU (gloo, world_size 2): utterance sharding + all_gather of rendered audio reproduces the
unsharded result, ragged batch sizes included.  The renderer here is the CPU oracle (tests may use it as a
stand-in; the product renderer needs a GPU), the sharding / gather code is the product's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synthetic
from oracle import synth as OS


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, Fr, q):
    import sys
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, cfg = synthetic.build_model("CombSubFast", seed=17)
    sd = model.state_dict()
    full = synthetic.make_inputs(31, B, Fr)
    mine = sharding.shard_batch(full, world, rank)
    lo, hi = sharding.shard_rows(B, world, rank)
    with torch.no_grad():
        sig = OS.combsubfast_forward(sd, cfg, mine["units"], mine["f0"], mine["volume"], mine["spk_id"],
                                     noise=mine["noise"])[0]
    T = Fr * 512
    per = -(-B // world)                                   # pad ragged shards to the largest one
    padded = torch.zeros(per, T)
    padded[:hi - lo] = sig
    g = sharding.AudioGather(world, per, T, "cpu")
    out = g.submit(padded)
    g.wait()
    rows = []
    for r in range(world):
        a, b = sharding.shard_rows(B, world, r)
        rows.append(out[r * per:r * per + (b - a)])
    q.put((rank, torch.cat(rows).clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])
def test_sharded_render_matches_unsharded(B):
    Fr, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, Fr, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model, cfg = synthetic.build_model("CombSubFast", seed=17)
    full = synthetic.make_inputs(31, B, Fr)
    with torch.no_grad():
        want = OS.combsubfast_forward(model.state_dict(), cfg, full["units"], full["f0"], full["volume"],
                                      full["spk_id"], noise=full["noise"])[0]
    for r in range(world):
        assert got[r].shape == want.shape
        assert (got[r] - want).abs().max() < 1e-5        # every rank holds every utterance after the gather


def _grad_worker(rank, world, port, q):
    import sys
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import training
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)                       # identical parameters on every rank
    params = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(11)), torch.nn.Parameter(torch.randn(2, 2))]
    torch.manual_seed(100 + rank)              # different gradients
    for p in params[:2]:
        p.grad = torch.randn_like(p)
    local = [p.grad.clone() for p in params[:2]]
    training.allreduce_gradients(params, world)
    gathered = [torch.zeros(world, *g.shape) for g in local]
    for g, l in zip(gathered, local):
        dist.all_gather_into_tensor(g, l.unsqueeze(0).contiguous())
    ok = all(torch.allclose(p.grad, g.mean(0), atol=1e-6) for p, g in zip(params[:2], gathered)) and params[2].grad is None
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_mean():
    """Data-parallel training exchange: every rank ends with the mean gradient; parameters without a
    gradient are skipped consistently."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(res.values())


# ---- ragged lengths: every utterance rendered at its own length, one gather with counts ----------------------------------
def _ragged_worker(rank, world, port, lengths, q):
    import sys
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, cfg = synthetic.build_model("CombSubFast", seed=17)
    sd = model.state_dict()
    plan = sharding.RaggedPlan(lengths, world)
    rendered = []
    for i in plan.local(rank):
        inp = synthetic.make_inputs(500 + i, 1, lengths[i])
        with torch.no_grad():
            rendered.append(OS.combsubfast_forward(sd, cfg, inp["units"], inp["f0"], inp["volume"], inp["spk_id"],
                                                   noise=inp["noise"])[0][0])
    out = plan.unpack(plan.gather(plan.pack(rank, rendered)))
    q.put((rank, [o.numpy().copy() for o in out], plan.local(rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_lengths_gather_with_counts():
    lengths = [5, 9, 3, 7, 4, 9, 1]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, lengths, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model, cfg = synthetic.build_model("CombSubFast", seed=17)
    sd = model.state_dict()
    mine = sorted(sum((r[2] for r in res), []))
    assert mine == list(range(len(lengths)))                       # every utterance rendered exactly once
    loads = [sum(lengths[i] for i in r[2]) for r in res]
    assert max(loads) - min(loads) <= max(lengths)                 # balanced
    for i, n in enumerate(lengths):
        inp = synthetic.make_inputs(500 + i, 1, n)
        with torch.no_grad():
            want = OS.combsubfast_forward(sd, cfg, inp["units"], inp["f0"], inp["volume"], inp["spk_id"],
                                          noise=inp["noise"])[0][0]
        for r in res:
            assert r[1][i].shape == (n * 512,)
            assert torch.allclose(torch.from_numpy(r[1][i]), want, atol=1e-6), i      # (ATen's CPU FFT is not bit-stable across thread counts)


# ---- training.train_step under data parallelism: shared n_fft draw, one flat gradient collective ---------------------------
class _OracleSynth(torch.nn.Module):
    """CPU stand-in with the product model's forward signature (the product model needs a GPU): the oracle's CombSubFast."""

    def __init__(self, sd, cfg):
        super().__init__()
        self.cfg = cfg
        self.buffers_sd = {k: v for k, v in sd.items() if not v.is_floating_point() or "projection_matrix" in k
                           or k == "window"}
        self.p = torch.nn.ParameterDict({k.replace(".", "|"): torch.nn.Parameter(v.clone()) for k, v in sd.items()
                                         if k not in self.buffers_sd})

    def forward(self, units, f0, volume, spk_id, infer=True, noise=None):
        sd = dict(self.buffers_sd)
        sd.update({k.replace("|", "."): v for k, v in self.p.items()})
        return OS.combsubfast_forward(sd, self.cfg, units, f0, volume, spk_id, infer=infer, noise=noise)[:3]


class _OracleLoss:
    def __init__(self):
        self.scales = None

    def set_scales(self, s):
        self.scales = [int(v) for v in s]

    def __call__(self, x_pred, x_true):
        from oracle import loss as OL
        return OL.rss_loss(x_pred, x_true, self.scales)


def _train_worker(rank, world, port, q):
    import sys
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import sharding
    import training
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    model, cfg = synthetic.build_model("CombSubFast", seed=19)
    net = _OracleSynth(model.state_dict(), cfg)
    B, Fr = 4, 10
    full = synthetic.make_inputs(41, B, Fr)
    full["audio"] = 0.1 * torch.randn(B, Fr * 512, generator=torch.Generator().manual_seed(3))
    batch = sharding.shard_batch(full, world, rank)
    opt = torch.optim.AdamW(net.parameters(), lr=5e-4, weight_decay=0.0)
    bucket = training.GradBucket(net.parameters())
    loss = training.train_step(net, opt, _OracleLoss(), batch, world=world, scales=[300, 777, 1531, 2047], bucket=bucket)
    views_ok = all(p.grad.data_ptr() >= bucket.flat.data_ptr() and
                   p.grad.data_ptr() < bucket.flat.data_ptr() + 4 * bucket.flat.numel() for p in bucket.params)
    # (numpy, not tensors: a tensor travels as a shared-memory file that is gone when this process has exited)
    q.put((rank, float(loss), {k: v.detach().numpy().copy() for k, v in net.p.items()}, bucket.flat.numpy().copy(), views_ok))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_train_step_data_parallel_equals_full_batch():
    """with the n_fft draw shared and equal shards, the mean of the rank losses is the full-batch loss and the
    averaged gradient its gradient (both loss terms are batch means, ddsp/loss.py:20-22): two gloo ranks running
    `training.train_step` end with the parameters of one process stepping on the whole batch."""
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    (_, loss1, params1, flat1, ok1), = res[1]
    (_, la, pa, fa, oka), (_, lb, pb, fb, okb) = res[2]
    assert ok1 and oka and okb                                       # .grad stayed views of the flat buffer
    assert abs(0.5 * (la + lb) - loss1) < 1e-5 * abs(loss1), (la, lb, loss1)
    import numpy as np
    assert np.array_equal(fa, fb)                                    # both ranks hold the same averaged gradient
    assert float(np.linalg.norm(fa - flat1) / np.linalg.norm(flat1)) < 1e-4
    for k in params1:
        assert np.array_equal(pa[k], pb[k]), k
        assert float(np.abs(params1[k] - pa[k]).max()) <= 2.1 * 5e-4, k   # first AdamW step ~ lr*sign(g): zero-crossing entries may flip
