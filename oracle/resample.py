"""ORACLE (test infrastructure only): windowed-sinc polyphase resampling as torchaudio publishes it
(`torchaudio.functional.resample`, method `sinc_interp_hann`, rolloff 0.99; used by the reference through
`torchaudio.transforms.Resample(orig, new, lowpass_filter_width=128)`, gui.py:399-404, enhancer.py:50-53,69-73).
PARITY UNPINNED: torchaudio is not installed and the reference holds no fixture of a resampled signal; this file
restates the algorithm from its published source (kernel in fp64 rounded to fp32, conv1d with stride `orig`)."""
import math

import torch


def sinc_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, dtype=torch.float64):
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=dtype)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=dtype)[:, None, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    kern = torch.where(t == 0, torch.tensor(1.0, dtype=dtype), t.sin() / t) * window * scale
    return kern, width, orig, new


def resample(x, orig_freq, new_freq, lowpass_filter_width=6, dtype=torch.float32):
    """x (B,T) -> (B, ceil(T*new/orig)); dtype float64 gives the reference evaluation the kernels are tested against."""
    kern, width, orig, new = sinc_kernel(orig_freq, new_freq, lowpass_filter_width)
    kern = kern.to(dtype)
    B, T = x.shape
    xp = torch.nn.functional.pad(x.to(dtype), (width, width + orig))
    y = torch.nn.functional.conv1d(xp[:, None], kern, stride=orig)
    y = y.transpose(1, 2).reshape(B, -1)
    return y[:, : math.ceil(new * T / orig)]
