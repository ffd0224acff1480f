"""Synthetic inputs and seeded weights for tests and benches (SURVEY.md 8(d) "Synthetic inputs").

Everything is drawn from numpy.random.Generator(PCG64(seed)) so that the generating script of the
golden vectors, the CPU oracle and the GPU tests see bit-identical inputs without storing them.
"""
import numpy as np
import torch

BASE_SEED = 20240117
SR = 44100
HOP = 512
FRAME_RATE = SR / HOP


def make_inputs(seed, B, Fr, n_unit=256, n_spk=100, with_noise=True, hop=HOP):
    """units ~ N(0,1) (B,Fr,n_unit); f0: per-utterance base 220*2^U(-1,1) Hz with 3 % / 5.5 Hz vibrato and a
    slow +-2 semitone random walk, clipped to [65, 800] Hz (B,Fr,1); volume ~ U(0,0.3) (B,Fr);
    spk_id ~ randint[1, n_spk] (B,1); noise ~ U[0,1) (B,Fr*hop) stands where the reference calls rand_like."""
    rng = np.random.Generator(np.random.PCG64(seed))
    units = rng.standard_normal((B, Fr, n_unit), dtype=np.float32)
    base = 220.0 * 2.0 ** rng.uniform(-1.0, 1.0, size=(B, 1))
    t = np.arange(Fr)[None, :] / FRAME_RATE
    vib = 1.0 + 0.03 * np.sin(2 * np.pi * 5.5 * t + rng.uniform(0, 2 * np.pi, size=(B, 1)))
    walk = np.cumsum(rng.standard_normal((B, Fr)) * 0.08, axis=1)
    walk = np.clip(walk, -2.0, 2.0)
    f0 = np.clip(base * vib * 2.0 ** (walk / 12.0), 65.0, 800.0).astype(np.float32)[..., None]
    volume = rng.uniform(0.0, 0.3, size=(B, Fr)).astype(np.float32)
    spk_id = rng.integers(1, n_spk + 1, size=(B, 1)).astype(np.int64)
    out = {
        "units": torch.from_numpy(units),
        "f0": torch.from_numpy(f0),
        "volume": torch.from_numpy(volume),
        "spk_id": torch.from_numpy(spk_id),
    }
    if with_noise:
        out["noise"] = torch.from_numpy(rng.random((B, Fr * hop), dtype=np.float32))
    return out


def make_ctrl(seed, B, Fr, width, std=0.5):
    """Control values as the network would emit them, ctrl ~ N(0, std^2) (B,Fr,width)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((rng.standard_normal((B, Fr, width)) * std).astype(np.float32))


MODEL_CFG = {
    "CombSub": dict(type="CombSub", sampling_rate=SR, block_size=HOP, n_mag_allpass=256, n_mag_harmonic=512,
                    n_mag_noise=256, n_unit=256, n_spk=100),
    "Sins": dict(type="Sins", sampling_rate=SR, block_size=HOP, n_harmonics=128, n_mag_allpass=256, n_mag_noise=256,
                 n_unit=256, n_spk=100),
    "Sins256": dict(type="Sins", sampling_rate=SR, block_size=HOP, n_harmonics=256, n_mag_allpass=256,
                    n_mag_noise=256, n_unit=256, n_spk=100),
    "CombSubFast": dict(type="CombSubFast", sampling_rate=SR, block_size=HOP, n_unit=256, n_spk=100),
}


def build_model(name, seed=BASE_SEED, device="cpu", n_unit=None):
    """Constructs the product module with seeded random weights (reference state-dict shapes).  `n_unit` overrides the
    encoder width (configs/combsub_xunit.yaml: 4, combsub_yunit.yaml: 512, reference `data.encoder_out_channels`)."""
    from ddsp.vocoder import CombSub, CombSubFast, Sins
    cfg = dict(MODEL_CFG[name])
    if n_unit is not None:
        cfg["n_unit"] = int(n_unit)
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        if cfg["type"] == "CombSub":
            m = CombSub(cfg["sampling_rate"], cfg["block_size"], cfg["n_mag_allpass"], cfg["n_mag_harmonic"],
                        cfg["n_mag_noise"], cfg["n_unit"], cfg["n_spk"])
        elif cfg["type"] == "Sins":
            m = Sins(cfg["sampling_rate"], cfg["block_size"], cfg["n_harmonics"], cfg["n_mag_allpass"],
                     cfg["n_mag_noise"], cfg["n_unit"], cfg["n_spk"])
        else:
            m = CombSubFast(cfg["sampling_rate"], cfg["block_size"], cfg["n_unit"], cfg["n_spk"])
    finally:
        torch.random.set_rng_state(gen_state)
    return m.to(device).eval(), cfg
