"""The GPU branch of the N > 1 path on the one GPU a test box has: a ONE-rank RCCL group (backend "nccl") drives
`sharding.AudioGather` exactly as `bench.py --gpus N` does - side stream, stream waits, record_stream, ring of output
buffers - so that what the gloo tests (CPU branch, world size 2) cannot reach is executed at least once before the
driver's multi-GPU run.  One process, one rank: nothing here needs a second GPU."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_audio_gather_gpu_branch_single_rank_rccl(dev, lib_path):
    import torch.distributed as dist
    import sharding
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        B, T = 4, 8 * 512
        g = sharding.AudioGather(1, B, T, dev)
        assert g.stream is not None                       # the GPU branch
        kept = []
        for i in range(5):                                # more submits than ring slots
            # produced on the compute stream right before the gather: the side stream must wait for it
            audio = torch.full((B, T), float(i), device=dev) + torch.arange(T, device=dev, dtype=torch.float32) * 1e-3
            out = g.submit(audio)
            kept.append((i, out))
            del audio
        last = g.wait()
        torch.cuda.synchronize()
        assert last is kept[-1][1]
        want = torch.full((B, T), 4.0, device=dev) + torch.arange(T, device=dev, dtype=torch.float32) * 1e-3
        assert torch.equal(last, want)
        # slot of submit 3 was not overwritten by submit 4 (ring of two)
        assert torch.equal(kept[3][1][:, 0], torch.full((B,), 3.0, device=dev))
        # the mean-all-reduce of gradients is the identity on one rank and leaves the tensors in place
        import training
        p = torch.nn.Parameter(torch.zeros(7, device=dev))
        p.grad = torch.arange(7, device=dev, dtype=torch.float32)
        training.allreduce_gradients([p], 1)
        assert torch.equal(p.grad, torch.arange(7, device=dev, dtype=torch.float32))
        dist.barrier()
    finally:
        dist.destroy_process_group()
