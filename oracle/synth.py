"""Oracle: the three synthesiser forwards (TEST INFRASTRUCTURE, see oracle/__init__.py).

CPU restatement of reference `ddsp/vocoder.py:372-550` (`Sins`, `CombSubFast`,
`CombSub`).  Each function takes the model's full state dict (reference key
names), the inputs of `forward(...)` and - because the reference draws its noise
from the CPU mt19937 stream (`torch.rand_like`, :418,:461,:545), which no GPU
kernel can reproduce - an explicit `noise` tensor U[0,1) of shape (B, T) that
stands where `rand_like` stood.  With `noise=None` it is drawn here.
"""
import numpy as np
import torch

from . import dsp
from .ctrlnet import unit2control

SPLITS = {
    "CombSub": lambda c: {"group_delay": c["n_mag_allpass"], "harmonic_magnitude": c["n_mag_harmonic"],
                          "noise_magnitude": c["n_mag_noise"]},
    "Sins": lambda c: {"amplitudes": c["n_harmonics"], "group_delay": c["n_mag_allpass"],
                       "noise_magnitude": c["n_mag_noise"]},
    "CombSubFast": lambda c: {"harmonic_magnitude": c["block_size"] + 1, "harmonic_phase": c["block_size"] + 1,
                              "noise_magnitude": c["block_size"] + 1},
}


def _ctrl_state(sd):
    pre = "unit2ctrl."
    return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}


def _front(cfg, f0_frames, initial_phase, infer):
    """Shared head of all three forwards.  ref: ddsp/vocoder.py:391-393, :449-451, :515-517."""
    hop, sr = int(cfg["block_size"]), int(cfg["sampling_rate"])
    f0 = dsp.frames_to_samples(f0_frames, hop).squeeze(-1)
    rot = dsp.rotation_from_f0(f0, sr, initial_phase, infer)
    return hop, sr, f0, rot


def combsub_forward(sd, cfg, units, f0_frames, volume, spk_id, spk_mix_dict=None, initial_phase=None,
                    infer=True, noise=None, ctrl_override=None):
    """ref: ddsp/vocoder.py:504-550.  Returns (signal, phase_frames(B,Fr,1), (harmonic, noise), aux)."""
    hop, sr, f0, rot = _front(cfg, f0_frames, initial_phase, infer)
    phase_frames = 2 * np.pi * rot[:, ::hop]
    if ctrl_override is None:
        ctrl = unit2control(_ctrl_state(sd), units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict,
                            SPLITS["CombSub"](cfg), causal=bool(cfg.get("c", False)))
    else:
        ctrl = ctrl_override
    group_delay = np.pi * torch.tanh(ctrl["group_delay"])
    src = torch.exp(ctrl["harmonic_magnitude"])
    nse = torch.exp(ctrl["noise_magnitude"]) / 128

    comb = dsp.sinc_comb(rot, f0, sr)
    harmonic = dsp.frequency_filter(comb, torch.exp(1.j * torch.cumsum(group_delay, dim=-1)), hann=False)
    harmonic = dsp.frequency_filter(harmonic, torch.complex(src, torch.zeros_like(src)), hann=True,
                                    half_width=1.5 * sr / (f0_frames + 1e-3))
    if noise is None:
        noise = torch.rand_like(harmonic)
    exc = noise * 2 - 1
    noise_out = dsp.frequency_filter(exc, torch.complex(nse, torch.zeros_like(nse)), hann=True)
    signal = harmonic + noise_out
    aux = {"rot": rot, "comb": comb, "ctrl": ctrl}
    return signal, phase_frames.unsqueeze(-1), (harmonic, noise_out), aux


def sins_forward(sd, cfg, units, f0_frames, volume, spk_id, spk_mix_dict=None, initial_phase=None,
                 infer=True, noise=None, ctrl_override=None, max_upsample_dim=32):
    """ref: ddsp/vocoder.py:381-423.  Note the returned phase is sample-rate (B,T,1)."""
    hop, sr, f0, rot = _front(cfg, f0_frames, initial_phase, infer)
    phase = 2 * np.pi * rot
    phase_frames = phase[:, ::hop]
    if ctrl_override is None:
        ctrl = unit2control(_ctrl_state(sd), units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict,
                            SPLITS["Sins"](cfg), causal=bool(cfg.get("c", False)))
    else:
        ctrl = ctrl_override
    amps = torch.exp(ctrl["amplitudes"]) / 128
    group_delay = np.pi * torch.tanh(ctrl["group_delay"])
    nse = torch.exp(ctrl["noise_magnitude"]) / 128

    amps = dsp.mask_above_nyquist(amps, f0_frames, sr / 2, level_start=1)
    sinusoids = dsp.harmonic_bank(amps, phase, hop, chunk=max_upsample_dim)
    harmonic = dsp.frequency_filter(sinusoids, torch.exp(1.j * torch.cumsum(group_delay, dim=-1)), hann=False)
    if noise is None:
        noise = torch.rand_like(harmonic)
    exc = noise * 2 - 1
    noise_out = dsp.frequency_filter(exc, torch.complex(nse, torch.zeros_like(nse)), hann=True)
    signal = harmonic + noise_out
    aux = {"rot": rot, "sinusoids": sinusoids, "ctrl": ctrl}
    return signal, phase.unsqueeze(-1), (harmonic, noise_out), aux


def combsubfast_forward(sd, cfg, units, f0_frames, volume, spk_id, spk_mix_dict=None, initial_phase=None,
                        infer=True, noise=None, ctrl_override=None):
    """ref: ddsp/vocoder.py:437-492.  Returns the same tensor three times like the reference."""
    hop, sr, f0, rot = _front(cfg, f0_frames, initial_phase, infer)
    phase_frames = 2 * np.pi * rot[:, ::hop]
    if ctrl_override is None:
        ctrl = unit2control(_ctrl_state(sd), units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict,
                            SPLITS["CombSubFast"](cfg), causal=bool(cfg.get("c", False)))
    else:
        ctrl = ctrl_override
    comb = dsp.sinc_comb(rot, f0, sr, zero_unvoiced=True)
    if noise is None:
        noise = torch.rand_like(comb)
    exc = noise * 2 - 1
    src_filter = torch.exp(ctrl["harmonic_magnitude"] + 1.j * np.pi * ctrl["harmonic_phase"])
    noise_filter = torch.exp(ctrl["noise_magnitude"]) / 128
    signal = dsp.windowed_spectral_ola(comb, exc, src_filter, noise_filter, hop)
    aux = {"rot": rot, "comb": comb, "ctrl": ctrl}
    return signal, phase_frames.unsqueeze(-1), (signal, signal), aux


FORWARD = {"CombSub": combsub_forward, "Sins": sins_forward, "CombSubFast": combsubfast_forward}
