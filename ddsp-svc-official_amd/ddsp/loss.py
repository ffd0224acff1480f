"""Drop-in for the reference's `ddsp/loss.py` (`SSSLoss`, `RSSLoss`), evaluated by libddsp_amd (value AND
gradient w.r.t. the prediction come from hand-written kernels; no CPU fallback).

`RSSLoss(fft_min, fft_max, n_scale, alpha=1.0, overlap=0, eps=1e-7, device='cuda')(x_pred, x_true)` draws
`n_scale` integers in [fft_min, fft_max) with `torch.randint` per call exactly like the reference
(`ddsp/loss.py:39`), so seeding torch reproduces the reference's sequence of scales; `set_scales` pins the next
draw (tests, and data-parallel ranks that must share one draw - SURVEY 8e).  `overlap` sets the hop as the reference
does, `int(n_fft * (1 - overlap))` (`ddsp/loss.py:13`); its callers use 0 (`train.py:48`).
"""
import torch
import torch.nn as nn

import hipddsp


class _SpectralLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_pred, x_true, n_ffts, alpha, eps, overlap=0):
        c = hipddsp.context_for(x_pred.device)
        need = x_pred.requires_grad
        hops = None if overlap == 0 else [_hop(n, overlap) for n in n_ffts]
        loss, grad = c.rss_loss(x_pred, x_true, n_ffts, alpha, eps, want_grad=need, hops=hops)
        ctx.grad = grad
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        grad = ctx.grad
        ctx.grad = None
        return (grad * g if grad is not None else None), None, None, None, None, None


def _hop(n_fft, overlap):
    """The reference's hop (`ddsp/loss.py:13`), in Python floats like there."""
    return int(n_fft * (1 - overlap))


def _check(x_pred, x_true, overlap):
    if not 0 <= overlap < 1:
        raise ValueError("overlap must lie in [0, 1)")
    if not x_pred.is_cuda:
        raise RuntimeError("the spectral loss runs on a HIP device only (no CPU fallback)")


class SSSLoss(nn.Module):
    """Single-scale spectral loss (reference `ddsp/loss.py:7-25`); call order is (x_true, x_pred) as there."""

    def __init__(self, n_fft=111, alpha=1.0, overlap=0, eps=1e-7):
        super().__init__()
        self.n_fft, self.alpha, self.overlap, self.eps = int(n_fft), alpha, overlap, eps

    def forward(self, x_true, x_pred):
        _check(x_pred, x_true, self.overlap)
        return _SpectralLossFn.apply(x_pred, x_true.to(x_pred.dtype), [self.n_fft], self.alpha, self.eps, self.overlap)


class RSSLoss(nn.Module):
    """Random-scale spectral loss (reference `ddsp/loss.py:28-43`)."""

    def __init__(self, fft_min, fft_max, n_scale, alpha=1.0, overlap=0, eps=1e-7, device="cuda"):
        super().__init__()
        self.fft_min, self.fft_max, self.n_scale = fft_min, fft_max, n_scale
        self.alpha, self.overlap, self.eps = alpha, overlap, eps
        self._pinned = None
        self.last_scales = None

    def set_scales(self, n_ffts):
        """Use these scales for the next call instead of drawing (one-shot)."""
        self._pinned = [int(n) for n in n_ffts]

    def forward(self, x_pred, x_true):
        _check(x_pred, x_true, self.overlap)
        if self._pinned is not None:
            n_ffts, self._pinned = self._pinned, None
        else:
            n_ffts = [int(v) for v in torch.randint(self.fft_min, self.fft_max, (self.n_scale,))]
        self.last_scales = n_ffts
        # cached training audio may be fp16 (reference data_loaders.py:81-83): promote the target
        return _SpectralLossFn.apply(x_pred, x_true.to(torch.float32), n_ffts, self.alpha, self.eps, self.overlap)
