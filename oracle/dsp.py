"""Oracle: DSP primitives of the synth path (TEST INFRASTRUCTURE, see oracle/__init__.py).

CPU restatement in PyTorch ops of reference `ddsp/core.py` (file:line cited per
function).  Pinned by tests/golden/core_*.npz, which were produced by importing
the unmodified reference module (tests/golden/make_golden.py) and by the
reference's own known-answer tests (`ddsp/core.py:54-97`).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

TWO_PI = 2.0 * math.pi


def frames_to_samples(x, hop):
    """Frame-rate -> sample-rate linear interpolation.  ref: ddsp/core.py:7-21.

    x :: (B, Fr, C) -> (B, Fr*hop, C).  The last frame is repeated once so that
    the final hop interpolates towards itself; the grid is align_corners so the
    step is exactly 1/hop.
    """
    hop = int(hop)
    ch_first = x.transpose(1, 2)
    ext = torch.cat([ch_first, ch_first[..., -1:]], dim=-1)
    n_out = ch_first.shape[-1] * hop + 1
    up = F.interpolate(ext, size=n_out, mode="linear", align_corners=True)
    return up[..., :-1].transpose(1, 2)


def frames_to_samples_closed_form(x, hop):
    """Same as frames_to_samples, written as the closed form the HIP kernel uses:
    out[t] = fma(1-a, x[i], a*x[min(i+1, Fr-1)]), i = t//hop, a = (t%hop)/hop.
    Bit-exact with ATen's CPU kernel on this image (tests/test_oracle_dsp.py)."""
    hop = int(hop)
    B, Fr, C = x.shape
    t = torch.arange(Fr * hop)
    i0 = t // hop
    i1 = torch.clamp(i0 + 1, max=Fr - 1)
    a = ((t % hop).to(torch.float32) / hop)[None, :, None]
    hi = (a * x[:, i1, :]).to(torch.float64)           # rounded product
    lo = (1.0 - a).to(torch.float64) * x[:, i0, :].to(torch.float64)  # exact in f64
    return (lo + hi).to(torch.float32)                 # single rounding == fmaf


def rotation_from_f0(f0, sr, initial_phase=None, precise=False):
    """Wrapped rotation (cycles) of an instantaneous-frequency series.  ref: ddsp/core.py:31-51.

    f0 :: (B, T) Hz.  `precise` integrates in fp64; otherwise the increments are
    fp32 and the running sum is an fp32 tensor (ATen's CPU cumsum accumulates it
    in double and rounds every output element to fp32).  Wrap is x - rint(x)
    (half-to-even) so the range is [-0.5, 0.5].
    """
    work = f0.to(torch.float64) if precise else f0
    acc = torch.cumsum(work / sr, dim=1)
    if initial_phase is not None:
        acc = acc + initial_phase.reshape(-1, 1).to(acc) / 2 / np.pi
    wrapped = acc - torch.round(acc)
    return wrapped.to(f0.dtype)


def sinc_comb(rot, f0, sr, zero_unvoiced=False):
    """Band-limited pulse train.  ref: ddsp/vocoder.py:539 (CombSub), :459-460 (CombSubFast).

    `sr` enters as fp32 (the reference multiplies by a 0-dim int64 buffer, which
    type-promotes to the fp32 of `rot`).
    """
    x = torch.tensor(float(sr), dtype=rot.dtype) * rot / (f0 + 1e-3)
    comb = torch.sinc(x)
    if zero_unvoiced:
        comb = torch.where(f0 <= 0.0, torch.zeros_like(comb), comb)
    return comb


def fir_from_response(resp, hann=True, half_width=None):
    """Frequency response frames -> linear-phase FIR frames.  ref: ddsp/core.py:306-328.

    resp :: (B, Fr, M) real or complex -> (B, Fr, n=2(M-1)).
      hann=False            : irfft, rotate by n/2                       (:324-326)
      hann=True, hw=None    : static periodic Hann, zero-phase -> causal (:242-289, window_size=0 branch)
      hann=True, hw=(B,Fr,1): "dynamic" raised cosine of half width hw   (:292-303)
    The dynamic window reproduces the reference's order of operations: taps with
    w > 1 are zeroed BEFORE the cosine, so they end up with weight 1, and taps
    with w < -1 are not clamped at all.
    """
    ir = torch.fft.irfft(resp)
    n = ir.shape[-1]
    if not hann:
        return torch.roll(ir, n // 2, dims=-1)
    if half_width is None:
        win = torch.roll(torch.hann_window(n, dtype=ir.dtype), n // 2, dims=-1)
        return torch.roll(ir * win, n // 2, dims=-1)
    pos = torch.arange(-(n // 2), (n + 1) // 2).to(ir) / half_width
    pos = torch.where(pos > 1, torch.zeros_like(pos), pos)
    win = (1 + torch.cos(np.pi * pos)) / 2
    return torch.roll(ir, n // 2, dims=-1) * win


def ltv_fir_fft(audio, ir):
    """Frame-varying FIR via 50%-overlap triangular framing + FFT products.  ref: ddsp/core.py:185-239.

    audio :: (B, T), ir :: (B, Fr, n) (or (B, n)) -> (B, T).  FFT length is
    2*hop + n - 1 (not a power of two, :226); the last IR frame is used twice
    (:228); output is cropped to start at n//2 (:147-182, :238).
    """
    if ir.dim() == 2:
        ir = ir[:, None, :]
    B_ir, Fr, n = ir.shape
    B, T = audio.shape
    if B != B_ir:
        raise ValueError(f"Batch size of audio ({B}) and impulse response ({B_ir}) must be the same.")
    hop = int(T / Fr)
    flen = 2 * hop
    frames = F.pad(audio, (hop, hop)).unfold(1, flen, hop)
    frames = frames * torch.bartlett_window(flen).to(frames)
    nfft = flen + n - 1
    spec = torch.fft.rfft(frames, nfft) * torch.fft.rfft(torch.cat([ir, ir[:, -1:, :]], dim=1), nfft)
    seg = torch.fft.irfft(spec, nfft)                          # (B, Fr+1, nfft)
    total = Fr * hop + nfft
    ola = F.fold(seg.transpose(1, 2), output_size=(1, total), kernel_size=(1, nfft), stride=(1, hop))
    ola = ola[:, 0, 0, hop:]
    start = n // 2
    return ola[:, start:start + T]


def ltv_fir_direct(audio, ir, dtype=torch.float64):
    """Independent time-domain evaluation of the same operator (scatter form), for
    small cases: input sample t = m*hop + j is filtered by
    (1 - j/hop) * ir[m] + (j/hop) * ir[min(m+1, Fr-1)]; output is delayed by n//2.
    Mathematically identical to ltv_fir_fft (any zero-padded FFT length gives the
    same linear convolution)."""
    B, T = audio.shape
    _, Fr, n = ir.shape
    hop = T // Fr
    x = audio.to(dtype)
    h = ir.to(dtype)
    h = torch.cat([h, h[:, -1:, :]], dim=1)
    out = torch.zeros(B, T + n + 2 * hop, dtype=dtype)
    ramp = torch.arange(hop, dtype=dtype) / hop
    for m in range(Fr):
        seg = x[:, m * hop:(m + 1) * hop]
        for which, wgt in ((m, 1.0 - ramp), (m + 1, ramp)):
            xs = seg * wgt                                      # (B, hop)
            # full convolution of xs with h[:, which]
            full = torch.zeros(B, hop + n - 1, dtype=dtype)
            for b in range(B):
                full[b] = torch.from_numpy(np.convolve(xs[b].numpy(), h[b, which].numpy()))
            out[:, m * hop:m * hop + hop + n - 1] += full
    start = n // 2
    return out[:, start:start + T]


def frequency_filter(audio, resp, hann=True, half_width=None):
    """ref: ddsp/core.py:331-336."""
    return ltv_fir_fft(audio, fir_from_response(resp, hann, half_width))


def mask_above_nyquist(amps, f0_frames, fmax, level_start=1):
    """ref: ddsp/core.py:24-28.  Harmonics at or above fmax are scaled by 1e-7, others by 1+1e-7."""
    H = amps.shape[-1]
    k = torch.arange(level_start, H + level_start).to(f0_frames)
    keep = (f0_frames * k < fmax).float() + 1e-7
    return amps * keep


def harmonic_bank(amps_frames, phase, hop, chunk=32):
    """Additive sinusoid bank.  ref: ddsp/vocoder.py:403-412.

    amps_frames :: (B, Fr, H) already masked, phase :: (B, T) radians (fp32,
    wrapped) -> (B, T).  Accumulated in chunks of `chunk` harmonics like the
    reference (sum over the chunk, then added to the running total).
    """
    H = amps_frames.shape[-1]
    k = torch.arange(1, H + 1).to(phase)
    total = 0.0
    for lo in range(0, H, chunk):
        ph = phase.unsqueeze(-1) * k[lo:lo + chunk]
        a = frames_to_samples(amps_frames[:, :, lo:lo + chunk], hop)
        total = total + (a * torch.sin(ph)).sum(-1)
    return total


def windowed_spectral_ola(comb, noise, src_filter, noise_filter, hop):
    """CombSubFast DSP stage.  ref: ddsp/vocoder.py:462-490.

    comb, noise :: (B, T); src_filter (complex), noise_filter (real) :: (B, Fr, hop+1).
    sqrt-Hann analysis and synthesis windows, circular 2*hop FFT per frame, the
    last filter frame reused for frame Fr, overlap-add, drop `hop` from each end.
    """
    flen = 2 * hop
    win = torch.sqrt(torch.hann_window(flen))
    cf = F.pad(comb, (hop, hop)).unfold(1, flen, hop) * win
    nf = F.pad(noise, (hop, hop)).unfold(1, flen, hop) * win
    Hs = torch.cat([src_filter, src_filter[:, -1:, :]], dim=1)
    Hn = torch.cat([noise_filter, noise_filter[:, -1:, :]], dim=1)
    spec = torch.fft.rfft(cf, flen) * Hs + torch.fft.rfft(nf, flen) * Hn
    seg = torch.fft.irfft(spec, flen) * win
    total = (seg.shape[1] + 1) * hop
    ola = F.fold(seg.transpose(1, 2), output_size=(1, total), kernel_size=(1, flen), stride=(1, hop))
    return ola[:, 0, 0, hop:-hop]
