"""Pins the CPU oracle against fixtures produced by the reference (tests/golden/make_golden.py) and against
the reference's own known-answer tests (ddsp/core.py:54-97).  CPU only."""
import os

import numpy as np
import pytest
import torch

import synthetic
from conftest import GOLDEN, rms
from oracle import dsp as O
from oracle import realtime as RT
from oracle import synth as S

SR, HOP = 44100, 512


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def close(a, b, rel=2e-6):
    """FFT-based outputs: ATen's CPU FFT is not bit-reproducible across thread counts / ISAs."""
    scale = max(float(b.abs().max()), 1e-6)
    return float((a - b).abs().max()) <= rel * scale * 8 and rms(a - b) <= rel * max(rms(b), 1e-6)


def wrap_close(a, b, tol):
    d = (a - b).double()
    d = d - torch.round(d)
    return float(d.abs().max()) <= tol


def rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def t32(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


# ---- the reference's own known-answer tests ---------------------------------------------------------
def test_rotation_known_answers():
    f = torch.tensor([[1.0, 1.0, 1.0]])
    assert O.rotation_from_f0(f, 1, precise=False).dtype == f.dtype
    assert O.rotation_from_f0(f, 1, precise=True).dtype == f.dtype
    assert torch.allclose(O.rotation_from_f0(f, 4), torch.tensor([[0.25, 0.50, -0.25]]))
    assert torch.allclose(O.rotation_from_f0(torch.tensor([[1.0, 2.0, 3.0]]), 4), torch.tensor([[0.25, -0.25, -0.50]]))
    assert torch.allclose(O.rotation_from_f0(f, 4, torch.tensor([np.pi])), torch.tensor([[-0.25, 0.0, 0.25]]))
    f2 = torch.tensor([[1.0, 1.0, 1.0], [1.0, 2.0, 3.0]])
    got = O.rotation_from_f0(f2, 4, torch.tensor([np.pi, 0.0]), precise=True)
    assert torch.allclose(got, torch.tensor([[-0.25, 0.0, 0.25], [0.25, -0.25, -0.50]]), atol=1e-5)


# ---- tier A: ddsp/core.py ----------------------------------------------------------------------------
def test_upsample_golden():
    g = load("core_upsample.npz")
    for Fr, Ch in [(1, 1), (2, 3), (7, 2), (172, 1)]:
        y = O.frames_to_samples(g[f"x_{Fr}_{Ch}"], HOP)
        y = y if Fr < 100 else y[:, ::61]
        assert (y - g[f"y_{Fr}_{Ch}"]).abs().max() <= 1e-6 * 1500   # <= 1 ulp of the largest value
        yc = O.frames_to_samples_closed_form(g[f"x_{Fr}_{Ch}"], HOP)   # the HIP kernel's formula
        yc = yc if Fr < 100 else yc[:, ::61]
        assert torch.equal(yc, g[f"y_{Fr}_{Ch}"])


def test_phase_golden():
    g = load("core_phase.npz")
    f0 = O.frames_to_samples(g["f0_frames"], HOP).squeeze(-1)
    for precise in (True, False):
        for use_init in (False, True):
            tag = f"p{int(precise)}_i{int(use_init)}"
            rot = O.rotation_from_f0(f0, SR, g["init"] if use_init else None, precise)
            assert wrap_close(rot[:, ::37], g["rot_" + tag], 1e-7 if precise else 2.5e-4)
            assert wrap_close(rot[:, ::HOP], g["phase_frames_" + tag] / (2 * np.pi), 1e-6 if precise else 2.5e-4)
            if not use_init:
                assert rms(O.sinc_comb(rot, f0, SR)[:, ::37] - g["comb_" + tag]) < 1e-6


def _filters(ctrl, f0f, audio):
    gd = np.pi * torch.tanh(ctrl[..., :256])
    src = torch.exp(ctrl[..., 256:768])
    nse = torch.exp(ctrl[..., 768:]) / 128
    resp_ap = torch.exp(1.j * torch.cumsum(gd, dim=-1))
    hw = 1.5 * SR / (f0f + 1e-3)
    cz = lambda m: torch.complex(m, torch.zeros_like(m))
    irs = (O.fir_from_response(resp_ap, hann=False), O.fir_from_response(cz(src), True, hw),
           O.fir_from_response(cz(nse), True))
    y_ap = O.frequency_filter(audio, resp_ap, hann=False)
    y_h = O.frequency_filter(y_ap, cz(src), True, hw)
    y_n = O.frequency_filter(audio, cz(nse), True)
    return irs, (y_ap, y_h, y_n)


def test_filter_golden_short():
    g = load("core_filter_short.npz")
    irs, ys = _filters(g["ctrl"], g["f0_frames"], g["audio"])
    for got, key in zip(irs, ("ir_ap", "ir_h", "ir_n")):
        assert close(got, g[key]), key
    for got, key in zip(ys, ("y_ap", "y_h", "y_n")):
        assert close(got, g[key]), key
    # the independent time-domain evaluation agrees with the FFT form (fp64 direct vs fp32 FFT)
    for ir, x, key in ((irs[0], g["audio"], "y_ap"), (irs[2], g["audio"], "y_n")):
        direct = O.ltv_fir_direct(x, ir)
        assert rms(direct - g[key].double()) < 2e-6 * max(1.0, rms(g[key]))
    # the dynamic-window quirk is present in the fixture: frame (0,1) has half width 82 < 511
    hw = float(1.5 * SR / (800.0 + 1e-3))
    n = 1022
    raw = torch.fft.irfft(torch.exp(g["ctrl"][0, 1, 256:768])).roll(n // 2)
    k_hi = n // 2 + int(hw) + 5      # w > 1  -> weight 1 (not 0)
    assert abs(float(g["ir_h"][0, 1, k_hi]) - float(raw[k_hi])) < 1e-9


def test_filter_golden_long():
    g = load("core_filter_long.npz")
    ctrl = t32(rng(310).standard_normal((1, 172, 1024)) * 0.5)
    f0f = t32(rng(311).uniform(65, 800, size=(1, 172, 1)))
    audio = t32(rng(312).uniform(-1, 1, size=(1, 172 * HOP)))
    _, ys = _filters(ctrl, f0f, audio)
    for got, key in zip(ys, ("y_ap", "y_h", "y_n")):
        assert close(got[:, ::29], g[key]), key


def test_fmax_golden():
    g = load("core_fmax.npz")
    assert torch.equal(O.mask_above_nyquist(g["amps"], g["pitch"], SR / 2), g["out"])


# ---- tier B: the three models (extorch boundary unpinned, see oracle/ctrlnet.py) --------------------
CASES = [("infer", dict(infer=True)), ("train", dict(infer=False)),
         ("mix", dict(infer=True, spk_mix_dict={1: 0.25, 7: 0.75})),
         ("init", dict(infer=True, initial_phase=torch.tensor([1.0, -2.0])))]


@pytest.mark.parametrize("name", ["CombSub", "Sins", "CombSubFast"])
def test_model_golden(name):
    g = load(f"model_{name}.npz")
    model, cfg = synthetic.build_model(name, seed=g["seed_weights"])
    sd = model.state_dict()
    inp = synthetic.make_inputs(g["seed_inputs"], 2, 12)
    if name == "Sins":
        inp["f0"][0, 3, 0] = 700.0
    for tag, kw in CASES:
        with torch.no_grad():
            sig, ph, (hm, nz), aux = S.FORWARD[name](sd, cfg, inp["units"], inp["f0"], inp["volume"], inp["spk_id"],
                                                     noise=inp["noise"], **kw)
        ctrl = torch.cat(list(aux["ctrl"].values()), dim=-1)
        # same torch ops in the same order on the same machine: expect (near) bit equality
        assert (ctrl - g[f"ctrl_{tag}"]).abs().max() < 1e-5, tag
        ph = ph if ph.shape[1] == 12 else ph[:, ::HOP]
        assert wrap_close(ph / (2 * np.pi), g[f"phase_{tag}"] / (2 * np.pi), 1e-6 if kw["infer"] else 2.5e-4), tag
        scale = max(rms(g[f"signal_{tag}"]), 1e-3)
        assert rms(sig - g[f"signal_{tag}"]) < 1e-5 * scale, (tag, rms(sig - g[f"signal_{tag}"]))
        if name != "CombSubFast":
            assert rms(hm - g[f"harmonic_{tag}"]) < 1e-5 * scale
            assert rms(nz - g[f"noise_{tag}"]) < 1e-5 * scale


# ---- tier C: caller glue ------------------------------------------------------------------------------
def test_sola_golden():
    g = load("glue_sola.npz")
    block, xfade, search, delay = [int(v) for v in g["sizes"]]
    emitted, buf, shift = RT.sola_step(g["audio"], g["prev"], block, xfade, search, delay)
    assert shift == g["shift"]
    assert (emitted - g["emitted"]).abs().max() < 1e-6 and (buf - g["new_buffer"]).abs().max() < 1e-6


def test_offline_glue_golden():
    z = np.load(os.path.join(GOLDEN, "glue_offline.npz"))
    out = RT.slice_cross_fade(z["a"], z["b"], int(z["idx"]))
    assert np.array_equal(out, z["crossfaded"])
    gate = RT.volume_gate(z["volume"], -60, HOP)
    assert (gate[:, ::16] - torch.from_numpy(z["mask_up"])).abs().max() < 1e-6


# ---- tier D: front-end steps before the path against the reference's own outputs --------------
def test_frontend_golden():
    from frontend_cases import FRONTEND_VOLUME, FRONTEND_ALIGN, volume_audio, align_units_input
    from oracle import frontend as F
    g = load("glue_frontend.npz")
    for i in range(len(FRONTEND_VOLUME)):
        audio, hop = volume_audio(i)
        got = F.volume_extract(audio, hop)
        want = g[f"vol_{i}"].numpy()
        assert got.shape == want.shape == (len(audio) // hop + 1,)
        assert np.allclose(got, want, rtol=1e-6, atol=0), i
    for i in range(len(FRONTEND_ALIGN)):
        units, n, sr, hop = align_units_input(i)
        got = F.align_units(units, n, sr, hop)
        assert torch.equal(got, g[f"align_{i}"]), i
    # the fixture contains the cases it is meant to: a clamped tail and a tie rounded to even
    u, n, sr, hop = align_units_input(3)
    assert torch.equal(g["align_3"][0, -1], u[0, -1]) and torch.equal(g["align_3"][0, -5], u[0, -1])
    u, n, sr, hop = align_units_input(2)                      # ratio 1.5: frame 1 -> unit 2 (1.5), frame 3 -> unit 4 (4.5)
    assert torch.equal(g["align_2"][0, 1], u[0, 2]) and torch.equal(g["align_2"][0, 3], u[0, 4])


def test_phase_vocoder_golden():
    """oracle/realtime.phase_vocoder against the reference's gui.phase_vocoder (tests/golden/glue_phase_vocoder.npz)."""
    from frontend_cases import PV_SIZES, pv_inputs
    from oracle import realtime as RT
    g = load("glue_phase_vocoder.npz")
    for i, n in enumerate(PV_SIZES):
        a, b, fo, fi = pv_inputs(i)
        got = RT.phase_vocoder(a, b, fo, fi)
        assert got.shape == (n,)
        assert close(got, g[f"pv_{i}"]), i
