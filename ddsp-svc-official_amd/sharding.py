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


class RaggedPlan:
    """Utterances of different lengths over the ranks (sort by length, pad to frame multiples, gather with
    counts).  An utterance cannot be padded INSIDE the network (GroupNorm statistics and the linear-attention sums run
    over all its frames), so every utterance is rendered at its own length; what is padded is the exchange: each rank
    packs its rendered utterances back to back into one flat buffer of `slot` samples (the largest rank total, the same
    on every rank because the plan is a pure function of the lengths), one `all_gather_into_tensor` moves the slots, and
    `unpack` cuts the gathered buffer by the counts into the original order.
    Assignment: utterances sorted by length (longest first) are dealt to the rank with the least work so far."""

    def __init__(self, n_frames, world, hop=512):
        self.n_frames = [int(n) for n in n_frames]
        self.world, self.hop = world, hop
        order = sorted(range(len(self.n_frames)), key=lambda i: (-self.n_frames[i], i))
        load = [0] * world
        self.assign = [[] for _ in range(world)]
        for i in order:
            r = min(range(world), key=lambda k: (load[k], k))
            self.assign[r].append(i)
            load[r] += self.n_frames[i]
        self.counts = [[self.n_frames[i] * hop for i in a] for a in self.assign]     # samples per utterance, per rank
        self.slot = max(1, max(sum(c) for c in self.counts))

    def local(self, rank):
        """Indices (into the caller's list) of the utterances `rank` renders, in packing order."""
        return list(self.assign[rank])

    def pack(self, rank, rendered, device="cpu"):
        """rendered: this rank's (T_i,) audio tensors in `local(rank)` order -> (slot,) flat buffer (zero tail)."""
        flat = torch.zeros(self.slot, device=device, dtype=torch.float32)
        off = 0
        for a, n in zip(rendered, self.counts[rank]):
            flat[off:off + n] = a.reshape(-1)
            off += n
        return flat

    def gather(self, flat, group=None):
        import torch.distributed as dist
        out = torch.empty(self.world * self.slot, device=flat.device, dtype=torch.float32)
        dist.all_gather_into_tensor(out, flat.contiguous(), group=group)
        return out

    def unpack(self, gathered):
        """(world * slot,) -> list of (T_i,) tensors in the ORIGINAL utterance order."""
        out = [None] * len(self.n_frames)
        for r in range(self.world):
            off = r * self.slot
            for i, n in zip(self.assign[r], self.counts[r]):
                out[i] = gathered[off:off + n]
                off += n
        return out
