"""Drop-in for the names callers import from the reference's `ddsp/core.py`, backed by libddsp_amd.

`upsample` is imported directly by the reference callers (`main.py:11,116`, `gui.py:9,112`).  The
remaining functions keep the reference's names and argument meaning so code written against
`ddsp.core` keeps working; all of them need device tensors (no CPU fallback).
"""
import torch

from hipddsp import COMB_NONE, FIR_ALLPASS, FIR_DYNAMIC, FIR_STATIC, context_for


def upsample(signal, factor):
    """Frame->sample linear interpolation (reference `ddsp/core.py:7-21`).

    signal :: (B, Frame, Feat) device tensor; factor :: int or 0-dim integer tensor (the reference
    passes its `block_size` buffer).  Returns (B, Frame*factor, Feat).
    """
    hop = int(factor)
    ctx = context_for(signal.device)
    return ctx.upsample(signal, hop).to(signal.dtype)


def fo_to_rot(fo, sr, initial_phase=None, precise=False):
    """Wrapped rotation of a sample-rate fo contour (reference `ddsp/core.py:31-51`).

    The device kernel integrates from FRAME-rate f0 (it fuses the upsampler), so this sample-rate
    entry point is provided for API completeness via a frame size of 1.
    """
    ctx = context_for(fo.device)
    out = ctx.phase_scan(fo.unsqueeze(-1).reshape(fo.shape[0], -1), 1, int(sr), initial_phase, precise,
                         COMB_NONE, want_rot=True)
    return out["rot"].to(fo.dtype)


def frequency_filter(audio, magnitudes, hann_window=True, half_width_frames=None):
    """LTV-FIR from frequency responses (reference `ddsp/core.py:331-336`) is exposed at the control
    level on the device path: use `hipddsp.Context.fir_from_ctrl` + `.ltv_fir` (the exp / tanh-cumsum
    activations are fused into the filter synthesis, so the complex `magnitudes` tensor of the
    reference is never materialised)."""
    raise NotImplementedError("use hipddsp.Context.fir_from_ctrl/ltv_fir (control-level API); see INTEGRATION.md")
