"""Training-side glue of the reference (`train.py:41-48`, `solver.py:100-114`) on the device path: an AdamW whose
update runs as a libddsp_amd kernel (state dict compatible with `torch.optim.AdamW`, so `Saver` checkpoints resume),
the data-parallel gradient exchange, and one training step."""
import torch

import hipddsp


class AdamW(torch.optim.Optimizer):
    """Same constructor / defaults / state keys ('step', 'exp_avg', 'exp_avg_sq') as torch.optim.AdamW (no amsgrad)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            # parameters that can share one library call: same device, same step count, contiguous storage
            batches = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("AdamW kernel runs on a HIP device only (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if not p.data.is_contiguous():
                    raise ValueError("AdamW kernel needs contiguous parameters")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                batches.setdefault((p.device, int(st["step"])), []).append((p.data, g, st["exp_avg"], st["exp_avg_sq"]))
            for (device, step), items in batches.items():
                ps, gs, ms, vs = zip(*items)
                hipddsp.context_for(device).adamw_step_multi(ps, gs, ms, vs, group["lr"], b1, b2, group["eps"],
                                                             group["weight_decay"], step)
        return loss


def allreduce_gradients(params, world, group=None):
    """Data-parallel mean of the gradients: one flat 14 MB bucket, one RCCL all_reduce.  The backward of
    the control network is a single library call, so there is nothing to overlap the exchange with except the next
    step's host work; at 7 x 153 GB/s of xGMI the bucket takes tens of microseconds."""
    if world <= 1:
        return
    import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= world
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class GradBucket:
    """All gradients of a model in ONE flat fp32 buffer: every `p.grad` is a view into it, so the data-parallel exchange
    is a single collective on memory that already holds the gradients (no `cat` before, no `copy_` after - 14 MB each
    way per step otherwise).  autograd accumulates into an existing `.grad` in place, so the views survive a backward
    pass; `zero()` replaces `optimizer.zero_grad()` (whose default, set_to_none, would drop the views)."""

    def __init__(self, params, model=None):
        """`model`: the synthesiser whose parameters these are - its control network then receives its gradients directly
        in the bucket (`Unit2Control._grads_in_place`, a one-shot token that `zero()` arms) instead of through autograd's
        accumulation."""
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, device=ref.device, dtype=torch.float32)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._net = model.unit2ctrl if model is not None and hasattr(model, "unit2ctrl") else None

    def zero(self):
        """Zero every gradient and ARM the control network's direct write for the NEXT backward pass only: the library
        overwrites (`=`) the bucket's views instead of handing gradients to autograd, which is right exactly once per
        zeroed bucket.  A second backward before the next `zero()` (micro-batches, two losses) finds the token spent and
        goes through autograd's accumulation again (`Unit2Control.backward_flat`)."""
        self.flat.zero_()
        if self._net is not None:
            self._net._grads_in_place = True

    def allreduce(self, world, group=None):
        """Mean over the ranks, one all_reduce (RCCL over xGMI on GPUs)."""
        if world <= 1:
            return
        import torch.distributed as dist
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat /= world


def train_step(model, optimizer, loss_fn, batch, world=1, scales=None, bucket=None):
    """One step of `solver.train` (`solver.py:110-114`): zero_grad -> forward(infer=False) -> loss -> backward -> step.
    `scales` pins the loss's n_fft draw (ranks of a data-parallel job must share it; both terms of the loss
    are batch means, `ddsp/loss.py:20-22`, so with equal shards the mean of the rank losses is the global loss and the
    mean of the rank gradients its gradient).  `bucket`: a GradBucket over the model's parameters (one flat collective)."""
    if bucket is not None:
        bucket.zero()
    else:
        optimizer.zero_grad()
    signal, _, _ = model(batch["units"].float(), batch["f0"], batch["volume"], batch["spk_id"], infer=False,
                         **({"noise": batch["noise"]} if "noise" in batch else {}))
    if scales is not None:
        loss_fn.set_scales(scales)
    loss = loss_fn(signal, batch["audio"])
    loss.backward()
    if bucket is not None:
        bucket.allreduce(world)
    else:
        allreduce_gradients(list(model.parameters()), world)
    optimizer.step()
    return loss.detach()
