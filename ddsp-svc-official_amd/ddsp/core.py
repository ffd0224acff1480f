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


def remove_above_fmax(amplitudes, pitch, fmax, level_start=1):
    """Anti-aliasing mask of the additive bank (reference `ddsp/core.py:22-26`): harmonic k of a frame keeps its
    amplitude (times 1 + 1e-7) while k * pitch < fmax and is scaled by 1e-7 above.

    amplitudes :: (B, Frame, H), pitch :: (B, Frame, 1).  Inside `Sins.forward` this mask is applied by the bank
    kernel itself (`ddsp_sins_bank`); the stand-alone function is three elementwise device ops of PyTorch, kept so
    that callers of `ddsp.core` find the name.  Device tensors only, like everything here.
    """
    if not (amplitudes.is_cuda and pitch.is_cuda):
        raise RuntimeError("ddsp.core.remove_above_fmax needs device tensors (no CPU fallback)")
    k = torch.arange(level_start, amplitudes.shape[-1] + level_start, device=pitch.device).to(pitch.dtype)
    keep = (pitch * k < fmax).to(torch.float32) + 1e-7
    return amplitudes * keep


def fo_to_rot(fo, sr, initial_phase=None, precise=False):
    """Wrapped rotation of a sample-rate fo contour (reference `ddsp/core.py:31-51`).

    The device kernel integrates from FRAME-rate f0 (it fuses the upsampler), so this sample-rate
    entry point is provided for API completeness via a frame size of 1.
    """
    ctx = context_for(fo.device)
    out = ctx.phase_scan(fo.unsqueeze(-1).reshape(fo.shape[0], -1), 1, int(sr), initial_phase, precise,
                         COMB_NONE, want_rot=True)
    return out["rot"].to(fo.dtype)


_IRDFT_TABLES = {}


def _irfft(spectra):
    """Inverse real DFT of (..., n_mag) complex spectra, n = 2 (n_mag - 1), as ONE product on the library's fp32 MFMA GEMM
    (`ddsp_gemm_f32`; round 3 - until then `torch.fft.irfft`, i.e. rocFFT, was the one vendor-library call behind this API):
        h[t] = (1/n) [ Re X_0 + (-1)^t Re X_{n/2} + 2 sum_{0<k<n/2} (Re X_k cos(2 pi k t / n) - Im X_k sin(2 pi k t / n)) ]
    (the imaginary parts of the first and last bin do not enter, as in `irfft`).  The table is made once per size in fp64."""
    n_mag = spectra.shape[-1]
    n = 2 * (n_mag - 1)
    dev = spectra.device
    key = (n_mag, dev)
    tab = _IRDFT_TABLES.get(key)
    if tab is None:
        t = torch.arange(n, device=dev, dtype=torch.float64)[:, None]
        k = torch.arange(n_mag, device=dev, dtype=torch.float64)[None, :]
        c = torch.full((1, n_mag), 2.0, device=dev, dtype=torch.float64)
        c[0, 0] = c[0, -1] = 1.0
        ang = (t * k % n) * (2.0 * torch.pi / n)              # exact argument reduction: t * k is an integer below 2^53
        tab = torch.cat([c * torch.cos(ang), -c * torch.sin(ang)], dim=1) / n
        tab[:, n_mag] = 0.0                                   # sin terms of the first and last bin
        tab[:, -1] = 0.0
        tab = _IRDFT_TABLES[key] = tab.float().contiguous()  # (n, 2 n_mag): the GEMM's [N][K] operand
    a = torch.cat([spectra.real, spectra.imag], dim=-1).reshape(-1, 2 * n_mag).float().contiguous()
    h = context_for(dev).gemm(a, tab)                         # (`ddsp_gemm_f32`: fp32 products whatever the context's mode)
    return h.reshape(*spectra.shape[:-1], n)


def _impulse_response(magnitudes, hann_window=True, half_width_frames=None):
    """Frequency responses (B, Frame, n_mag) complex -> causal impulse responses (B, Frame, n = 2*(n_mag-1)).

    Restates reference `ddsp/core.py:296-327` (and its two window helpers, `:241-293`) as index arithmetic on the
    zero-phase response h = irfft(magnitudes):  g[i] = h[(i - n/2) mod n] * w[i]  with
      * no window:       w = 1
      * static window:   w = periodic Hann of length n  (window and response are both rotated by n/2 there)
      * dynamic window:  x = (i - n/2) / half_width;  x > 1 is set to 0 BEFORE the raised cosine (so the weight is 1,
                         not 0) and x < -1 is not clamped - both as the reference does it.
    The inverse DFT is a product on the library's GEMM (`_irfft`), the rest elementwise device ops; the models never call this -
    their filters come from `ddsp_fir_from_ctrl`, which fuses the activations and the inverse DFT into one MFMA GEMM.
    """
    h = _irfft(magnitudes)
    n = h.shape[-1]
    i = torch.arange(n, device=h.device)
    g = h.index_select(-1, (i - n // 2) % n)
    if not hann_window:
        return g
    if half_width_frames is None:
        return g * (0.5 - 0.5 * torch.cos(i.to(h.dtype) * (2.0 * torch.pi / n)))
    x = (i - n // 2).to(h.dtype) / half_width_frames.to(h.dtype)
    x = torch.where(x > 1, torch.zeros_like(x), x)
    return g * ((1 + torch.cos(torch.pi * x)) / 2)


def frequency_filter(audio, magnitudes, hann_window=True, half_width_frames=None):
    """Linear time-varying FIR filtering from per-frame frequency responses (reference `ddsp/core.py:331-336`).

    audio :: (B, T) device tensor, magnitudes :: (B, Frame, n_mag) complex, half_width_frames :: (B, Frame, 1) or None.
    The impulse responses are formed by `_impulse_response` (inverse DFT on the library's GEMM + elementwise ops), the filtering itself -
    `_fft_convolve`, `core.py:190-238`: 50 %-overlapped Bartlett frames, one filter per frame, output delayed by
    n/2 and cropped to T - is the hand-written `ddsp_ltv_fir` kernel.  That kernel supports what the models use:
    T = Frame * 512 and even filter lengths 32..2046 (n_mag 17..1024); other shapes raise ValueError.  Forward only:
    the models differentiate through their own fused path, so tensors that require grad are refused here.
    """
    if audio.requires_grad or magnitudes.requires_grad or (half_width_frames is not None and half_width_frames.requires_grad):
        raise NotImplementedError("ddsp.core.frequency_filter is forward-only on the device path; train through the model classes")
    if not (audio.is_cuda and magnitudes.is_cuda):
        raise RuntimeError("ddsp.core.frequency_filter needs device tensors (no CPU fallback)")
    if audio.dim() != 2 or magnitudes.dim() != 3 or audio.shape[0] != magnitudes.shape[0]:
        raise ValueError("frequency_filter: audio (B, T) and magnitudes (B, Frame, n_mag) expected")
    B, T = audio.shape
    Fr = magnitudes.shape[1]
    if Fr == 0 or T % Fr != 0:
        raise ValueError("frequency_filter: T must be a whole number of frames on the device path")
    ir = _impulse_response(magnitudes, hann_window, half_width_frames).float().contiguous()
    ctx = context_for(audio.device)
    out, _ = ctx.ltv_fir(audio.float().contiguous(), ir, B, Fr, T // Fr)
    return out.to(audio.dtype)
