"""Utterance sharding over the GPUs of one node: one process per GPU, every op of the path is
independent per utterance, so rank r renders rows [r*B/N, (r+1)*B/N) with replicated weights and the only
exchange is an RCCL all_gather of the rendered audio over xGMI.

`AudioGather` double-buffers that collective on a side stream so that the gather of step i overlaps the
compute of step i+1 (each pair of GPUs exchanges its own shard over its own xGMI link).
"""
import torch


def shard_rows(n_rows, world, rank):
    """Contiguous, balanced [lo, hi) row range of `rank` (ragged totals: the first n_rows % world ranks get one more)."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(batch, world, rank):
    """Slices every (B, ...) tensor of a batch dict to this rank's utterances ((1, ...) tensors are replicated)."""
    B = max(v.shape[0] for v in batch.values())
    lo, hi = shard_rows(B, world, rank)
    return {k: (v[lo:hi] if v.shape[0] == B else v) for k, v in batch.items()}


class AudioGather:
    """all_gather of (B_local, T) fp32 audio into (world*B_local, T), overlapped with compute."""

    def __init__(self, world, b_local, T, device, group=None, depth=2):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = world
        self.device = device
        on_gpu = torch.device(device).type == "cuda"
        self.stream = torch.cuda.Stream(device=device) if on_gpu else None
        self.out = [torch.empty(world * b_local, T, device=device, dtype=torch.float32) for _ in range(depth)]
        self.events = [None] * depth
        self.i = 0
        self.last = None

    def submit(self, audio):
        """Enqueue the gather of `audio`; returns the buffer that will hold all ranks' audio after wait()."""
        slot = self.i % len(self.out)
        self.i += 1
        out = self.out[slot]
        if self.stream is None:
            self.dist.all_gather_into_tensor(out, audio.contiguous(), group=self.group)
        else:
            cur = torch.cuda.current_stream(self.device)
            self.stream.wait_stream(cur)                 # audio is ready when the compute stream gets here
            with torch.cuda.stream(self.stream):
                self.dist.all_gather_into_tensor(out, audio, group=self.group)
                audio.record_stream(self.stream)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            self.events[slot] = ev
        self.last = out
        return out

    def wait(self):
        """Make the compute stream (and the host, via a later synchronize) see every submitted gather."""
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return self.last
