"""The synthesis half of the reference's offline CLI (`main.py:103-174`) restated around the device path.

The reference's `main.py` is broken as shipped (`.astype` on a tensor at :112, key shift applied twice at :105/:120,
SURVEY 0.3), and its analysis front-end (f0 / units extraction) is outside this build's scope, so this module
takes the analysed features and reproduces what happens from there: per-slice `model(...)[0]`, the volume gate
multiplied into the returned signal in place, optional enhancer, silence padding / cross-fade of slices.
"""
import numpy as np
import torch

import hipddsp


def cross_fade(a: np.ndarray, b: np.ndarray, idx: int):
    """Linear cross-fade of two host slices (reference `main.py:50-57`); numpy float64 like the reference."""
    out = np.zeros(idx + b.shape[0])
    fade = a.shape[0] - idx
    out[:idx] = a[:idx]
    k = np.linspace(0, 1.0, num=fade, endpoint=True)
    out[idx:a.shape[0]] = (1 - k) * a[idx:] + k * b[:fade]
    out[a.shape[0]:] = b[fade:]
    return out


def volume_mask(volume, threshold_db, block_size):
    """(1, Fr) device volume -> (1, Fr*block) gate, the reference's mask (`main.py:111-116`) as one device kernel."""
    ctx = hipddsp.context_for(volume.device)
    ones = torch.ones(volume.shape[0], volume.shape[1] * block_size, device=volume.device)
    return ctx.volume_gate_(ones, volume, threshold_db, block_size)


@torch.no_grad()
def render(model, args, segments, f0, volume, spk_id, spk_mix_dict=None, threshold_db=-60, enhancer=None,
           enhancer_adaptive_key=0, noise_seed=None):
    """segments: list of (start_frame, units (1, Fr_seg, n_unit)) as `main.py:143-151` produces them;
    f0 (1, Fr, 1), volume (1, Fr) cover the whole file.  Returns (float64 numpy waveform, sample rate)."""
    block = int(args.data.block_size)
    sr = int(args.data.sampling_rate)
    ctx = hipddsp.context_for(f0.device)
    result = np.zeros(0)
    current = 0
    sr_o = sr
    for start, units in segments:
        n = units.size(1)
        seg_f0 = f0[:, start:start + n, :]
        seg_vol = volume[:, start:start + n]
        kw = {} if noise_seed is None else {"noise_seed": noise_seed + start}
        out = model(units, seg_f0, seg_vol, spk_id=spk_id, spk_mix_dict=spk_mix_dict, **kw)[0]
        # the gate of the WHOLE file sliced to this segment (main.py:159): dilation sees the neighbours
        gate = volume_mask(volume, threshold_db, block)[:, start * block:(start + n) * block]
        out *= gate
        if enhancer is not None:
            out, sr_o = enhancer.enhance(out, sr, seg_f0, block, adaptive_key=enhancer_adaptive_key)
        out = out.squeeze().cpu().numpy()
        silent = round(start * block * sr_o / sr) - current
        if silent >= 0:
            result = np.append(result, np.zeros(silent))
            result = np.append(result, out)
        else:
            result = cross_fade(result, out, current + silent)
        current = current + silent + len(out)
    return result, sr_o
