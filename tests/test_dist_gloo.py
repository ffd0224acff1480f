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
