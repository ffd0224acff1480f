"""Oracle: caller-side glue around the synth call (TEST INFRASTRUCTURE, see oracle/__init__.py).

CPU restatement of the SOLA splice of the real-time path (reference
`gui.py:405-430`, windows `gui.py:349-351`), the offline slice cross-fade
(`main.py:50-57`) and the volume gate (`main.py:111-116`, `gui.py:108-112`).
The reference has no test of its own for these.  PINNED by tests/golden/ref_gui_stream.npz and
ref_offline_glue.npz, which tests/golden/make_golden.py (tier e) produced by EXECUTING
`gui.GUI.audio_callback` (eight blocks, with and without the phase-vocoder splice),
`gui.SvcDDSP.infer`'s gate and `main.cross_fade`; the older glue_*.npz hold the same
expressions evaluated inline by the generator.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import dsp


def fade_windows(xfade):
    """ref: gui.py:349-351."""
    fade_in = torch.sin(np.pi * torch.arange(0, 1, 1 / xfade) / 2) ** 2
    return fade_in, 1 - fade_in


def sola_step(audio, sola_buffer, block, xfade, search, delay):
    """One block of the SOLA splice.  ref: gui.py:405-430.

    audio :: (N,) model output for the sliding window; sola_buffer :: (xfade,)
    tail kept from the previous block.  Returns (emitted (block,), new_buffer, shift).
    """
    tmp = audio[-block - xfade - search - delay: -delay].clone()
    head = tmp[None, None, : xfade + search]
    num = F.conv1d(head, sola_buffer[None, None, :])
    den = torch.sqrt(F.conv1d(head ** 2, torch.ones(1, 1, xfade)) + 1e-8)
    shift = int(torch.argmax(num[0, 0] / den[0, 0]))
    tmp = tmp[shift: shift + block + xfade].clone()
    fade_in, fade_out = fade_windows(xfade)
    tmp[:xfade] = tmp[:xfade] * fade_in + sola_buffer * fade_out
    return tmp[:-xfade], tmp[-xfade:].clone(), shift


def sola_step_phase_vocoder(audio, sola_buffer, block, xfade, search, delay):
    """The same block with `use_phase_vocoder` (gui.py:417-423): the head is `phase_vocoder(kept tail, new head)`."""
    tmp = audio[-block - xfade - search - delay: -delay].clone()
    head = tmp[None, None, : xfade + search]
    num = F.conv1d(head, sola_buffer[None, None, :])
    den = torch.sqrt(F.conv1d(head ** 2, torch.ones(1, 1, xfade)) + 1e-8)
    shift = int(torch.argmax(num[0, 0] / den[0, 0]))
    tmp = tmp[shift: shift + block + xfade].clone()
    fade_in, fade_out = fade_windows(xfade)
    tmp[:xfade] = phase_vocoder(sola_buffer, tmp[:xfade], fade_out, fade_in)
    return tmp[:-xfade], tmp[-xfade:].clone(), shift


def slide_window(window, indata):
    """ref: gui.py:373-374.  window (N,) float32 numpy, indata (block, channels): shift left by one block and put the
    mono mix (librosa.to_mono = channel mean) at the end."""
    block = indata.shape[0]
    out = np.roll(window, -block)
    out[-block:] = np.mean(indata.T, axis=0)
    return out


def slice_cross_fade(a, b, idx):
    """ref: main.py:50-57 (numpy float64)."""
    out = np.zeros(idx + b.shape[0])
    fade = a.shape[0] - idx
    out[:idx] = a[:idx]
    k = np.linspace(0, 1.0, num=fade, endpoint=True)
    out[idx:a.shape[0]] = (1 - k) * a[idx:] + k * b[:fade]
    out[a.shape[0]:] = b[fade:]
    return out


def volume_gate(volume, threshold_db, hop):
    """9-frame max-dilated gate, upsampled to sample rate.  ref: main.py:111-116.

    volume :: (Fr,) numpy -> (1, Fr*hop) torch fp32.
    """
    m = (volume > 10 ** (float(threshold_db) / 20)).astype("float")
    m = np.pad(m, (4, 4), constant_values=(m[0], m[-1]))
    m = np.array([np.max(m[n: n + 9]) for n in range(len(m) - 8)])
    m = torch.from_numpy(m).float()[None, :, None]
    return dsp.frames_to_samples(m, hop).squeeze(-1)


def phase_vocoder(a, b, fade_out, fade_in):
    """ref: gui.py:14-31 (the optional cross-fade of gui.py:417-423).  Spectra of the kept tail `a` and the new head
    `b`; per bin the summed magnitudes (doubled inside the spectrum), the phase of `a`, and the phase advance to `b`
    wrapped into [-pi, pi) and added to the bin's own 2*pi*f; an oscillator bank with those per-bin frequencies is
    mixed in with weight fade_out*fade_in/n next to the squared-window cross-fade of the two signals."""
    import numpy as np
    n = a.shape[0]
    Fa, Fb = torch.fft.rfft(a), torch.fft.rfft(b)
    amp = Fa.abs() + Fb.abs()
    if n % 2 == 0:
        amp[1:-1] *= 2
    else:
        amp[1:] *= 2
    pa, pb = torch.angle(Fa), torch.angle(Fb)
    d = pb - pa
    d = d - 2 * np.pi * torch.floor(d / 2 / np.pi + 0.5)
    w = 2 * np.pi * torch.arange(n // 2 + 1).to(a) + d
    t = torch.arange(n).unsqueeze(-1).to(a) / n
    osc = torch.sum(amp * torch.cos(w * t + pa), -1)
    return a * (fade_out ** 2) + b * (fade_in ** 2) + osc * fade_out * fade_in / n
