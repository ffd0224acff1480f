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
    return {k: (z[k] if z[k].dtype.kind in "US" else torch.from_numpy(z[k]) if z[k].ndim else z[k].item())
            for k in z.files}


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


# ---- tier E: G8 / G9 / G10 - fixtures made by EXECUTING the reference's loss, training step and caller glue -------------
def test_loss_oracle_against_reference_run():
    """oracle/loss.py against `ddsp/loss.py` itself (tests/golden/ref_loss.npz; Spectrogram stood on torch.stft in the
    generator: that boundary stays unpinned, the loss arithmetic is pinned)."""
    import glue_cases as GC
    from oracle import loss as OL
    g = load("ref_loss.npz")
    xp, xt = GC.loss_signals()
    for N in GC.LOSS_SCALES:
        x = xp.clone().requires_grad_(True)
        v = OL.sss_loss(xt, x, N)
        v.backward()
        v = v.detach()
        assert abs(float(v) - g[f"sss_{N}"]) < 2e-6 * g[f"sss_{N}"], N
        assert abs(float(v) - g[f"sss64_{N}"]) < 5e-6 * g[f"sss64_{N}"], N
        # the fp32 gradient is ill-conditioned (1/S): the reference's own fp32 run is this far from its fp64 run
        ref_err = float((g[f"sss_grad_{N}"].double() - g[f"sss64_grad_{N}"]).norm() / g[f"sss64_grad_{N}"].norm())
        err = float((x.grad[:, ::97].double() - g[f"sss64_grad_{N}"]).norm() / g[f"sss64_grad_{N}"].norm())
        assert err < max(3 * ref_err, 1e-5), (N, err, ref_err)
        assert abs(float(x.grad.norm()) - g[f"sss64_gradnorm_{N}"]) < max(3 * ref_err, 1e-5) * g[f"sss64_gradnorm_{N}"], N
    torch.manual_seed(GC.RSS_SEED)
    scales = OL.draw_scales(256, 2048, 4)
    assert scales == [int(v) for v in g["rss_scales"]]
    x = xp.clone().requires_grad_(True)
    v = OL.rss_loss(x, xt, scales)
    v.backward()
    assert abs(float(v.detach()) - g["rss"]) < 2e-6 * g["rss"]
    assert float((x.grad[:, ::97] - g["rss_grad"]).norm() / g["rss_grad"].norm()) < 2e-3
    assert abs(float(x.grad.norm()) - g["rss_gradnorm"]) < 2e-3 * g["rss_gradnorm"]


def test_loss_overlap_oracle_against_reference_run():
    """overlap > 0 (hop = int(n_fft * (1 - overlap)), ddsp/loss.py:13): oracle/loss.py against the reference's SSSLoss /
    RSSLoss run with overlap (tests/golden/ref_loss_overlap.npz, make_golden.py tier h)."""
    import glue_cases as GC
    from oracle import loss as OL
    g = load("ref_loss_overlap.npz")
    xp, xt = GC.loss_signals()
    for N, ov in GC.LOSS_OVERLAP_CASES:
        assert OL.hop_length(N, ov) == int(g[f"hop_{N}"])
        x = xp.clone().requires_grad_(True)
        v = OL.sss_loss(xt, x, N, overlap=ov)
        v.backward()
        assert abs(float(v.detach()) - g[f"sss_{N}"]) < 2e-6 * g[f"sss_{N}"], N
        ref_err = float((g[f"sss_grad_{N}"].double() - g[f"sss64_grad_{N}"]).norm() / g[f"sss64_grad_{N}"].norm())
        err = float((x.grad[:, ::97].double() - g[f"sss64_grad_{N}"]).norm() / g[f"sss64_grad_{N}"].norm())
        assert err < max(3 * ref_err, 1e-5), (N, err, ref_err)
    scales = [int(v) for v in g["rss_scales"]]
    x = xp.clone().requires_grad_(True)
    v = OL.rss_loss(x, xt, scales, overlap=0.75)
    v.backward()
    assert abs(float(v.detach()) - g["rss"]) < 2e-6 * g["rss"]
    assert float((x.grad[:, ::97] - g["rss_grad"]).norm() / g["rss_grad"].norm()) < 2e-3


def oracle_train_steps(steps=None):
    """Three iterations of solver.py:110-114 through the oracle forward, the oracle loss, PyTorch autograd and
    torch.optim.AdamW; shared by the CPU pin below and the GPU test."""
    import glue_cases as GC
    from oracle import loss as OL
    g = load("ref_train_step.npz")
    model, cfg = synthetic.build_model("CombSub", seed=GC.TRAIN_WEIGHT_SEED)
    sd0 = model.state_dict()
    names = [n for n, _ in model.named_parameters()]
    params = {k: sd0[k].clone().requires_grad_(True) for k in names}
    sd = dict(sd0)
    sd.update(params)
    inp = synthetic.make_inputs(GC.TRAIN_INPUT_SEED, GC.TRAIN_B, GC.TRAIN_FR)
    target = GC.train_target()
    opt = torch.optim.AdamW(list(params.values()), lr=GC.TRAIN_LR, weight_decay=GC.TRAIN_WD)
    out = {"losses": [], "names": names}
    for step in range(steps or GC.TRAIN_STEPS):
        opt.zero_grad()
        sig = S.FORWARD["CombSub"](sd, cfg, inp["units"], inp["f0"], inp["volume"], inp["spk_id"], infer=False,
                                   noise=inp["noise"])[0]
        loss = OL.rss_loss(sig, target, [int(v) for v in g["scales"][step]])
        loss.backward()
        if step == 0:
            out["signal0"] = sig.detach()
            out["gradnorm0"] = np.array([float(params[n].grad.norm()) for n in names])
        opt.step()
        if step == 0:
            out["deltanorm0"] = np.array([float((params[n].detach() - sd0[n]).norm()) for n in names])
        out["losses"].append(float(loss.detach()))
    out["deltanorm_all"] = np.array([float((params[n].detach() - sd0[n]).norm()) for n in names])
    return out, g


def test_training_step_oracle_against_reference_run():
    """G9: the oracle's forward + loss under autograd and AdamW reproduce the reference's three training iterations
    (losses, per-parameter gradient norms and update norms; tests/golden/ref_train_step.npz)."""
    out, g = oracle_train_steps()
    assert out["names"] == [str(n) for n in g["param_names"]]
    assert rms(out["signal0"][:, ::61] - g["signal0"]) < 1e-5 * float(g["signal0_rms"])
    # Iteration 0 is the tight check.  From iteration 1 on the comparison is loose by necessity: the first AdamW update
    # is ~lr*sign(g), so gradient entries near zero (313 of 3.5 M here) move by +lr in one run and -lr in the other, and
    # the loss gradient is so ill-conditioned (1/S in near-empty bins) that the ORACLE's own gradient at those two
    # parameter sets differs by a relative 1.6 (norm 2.2 against 3.5, measured when the fixture was made).
    want = g["losses"].numpy()
    assert abs(out["losses"][0] - want[0]) < 1e-5 * want[0], (out["losses"], want)
    assert abs(out["losses"][1] - want[1]) < 1e-3 * want[1], (out["losses"], want)
    assert abs(out["losses"][2] - want[2]) < 3e-2 * want[2], (out["losses"], want)
    assert want[2] < want[1] < want[0] and out["losses"][2] < out["losses"][1] < out["losses"][0]
    gn, gw = out["gradnorm0"], g["gradnorm0"].numpy()
    assert np.allclose(gn, gw, rtol=2e-2, atol=1e-6), np.abs(gn / gw - 1).max()
    assert np.allclose(out["deltanorm0"], g["deltanorm0"].numpy(), rtol=1e-2, atol=1e-7)
    assert np.allclose(out["deltanorm_all"], g["deltanorm_all"].numpy(), rtol=0.5, atol=1e-7)


def test_gui_stream_oracle_against_reference_run():
    """G10: oracle/realtime.sola_step (and phase_vocoder) against eight blocks of the reference's own
    `gui.GUI.audio_callback` (tests/golden/ref_gui_stream.npz)."""
    import glue_cases as GC
    g = load("ref_gui_stream.npz")
    block, xfade, search, delay, n_in = [int(v) for v in g["sizes"]]
    assert n_in == max(block + xfade + search + 2 * delay, (1 + GC.GUI_BUFFER_NUM) * block)
    for tag in ("plain", "pv"):
        buf = torch.zeros(xfade)
        for k in range(GC.GUI_BLOCKS):
            audio = GC.gui_model_output(k)
            if tag == "plain":
                emitted, buf, shift = RT.sola_step(audio, buf, block, xfade, search, delay)
            else:
                emitted, buf, shift = RT.sola_step_phase_vocoder(audio, buf, block, xfade, search, delay)
            assert shift == int(g[f"shift_{tag}"][k]), (tag, k)
            assert (emitted[xfade:] - g["out_plain"][k][xfade:]).abs().max() == 0, (tag, k)
            if tag == "plain":
                assert (emitted - g["out_plain"][k]).abs().max() < 2e-6, k
            else:
                assert (emitted[:xfade] - g["head_pv"][k]).abs().max() < 2e-5, k
        assert (buf - g[f"buffer_{tag}"]).abs().max() < 1e-6
    # the sliding input window of the callback (gui.py:373-374): roll by one block, mono mix at the end
    win = np.zeros(n_in, dtype=np.float32)
    for k in range(GC.GUI_BLOCKS):
        win = RT.slide_window(win, GC.gui_indata(k))
    assert np.array_equal(win[-3 * block::37], g["input_wav_plain"].numpy())


def test_offline_glue_against_reference_run():
    """Volume gate as `gui.SvcDDSP.infer` computes and applies it, and `main.cross_fade`, both executed by the generator
    (tests/golden/ref_offline_glue.npz): the oracle AND the product's host function (`infer_offline.cross_fade`)."""
    import glue_cases as GC
    import infer_offline
    from oracle import frontend as F
    z = np.load(os.path.join(GOLDEN, "ref_offline_glue.npz"))
    vol = F.volume_extract(GC.gate_audio(), HOP)
    assert np.allclose(vol, z["volume"], rtol=1e-6, atol=0)
    gated = GC.gate_model_output() * RT.volume_gate(vol, GC.GATE_THRESHOLD, HOP)
    assert torch.equal(gated[0], torch.from_numpy(z["gated"]))
    assert 0.05 < float((gated == 0).float().mean()) < 0.5          # the fixture does close the gate somewhere
    for i, (a, b, idx) in enumerate(GC.crossfade_cases()):
        assert np.array_equal(RT.slice_cross_fade(a, b, idx), z[f"xfade_{i}"]), i
        assert np.array_equal(infer_offline.cross_fade(a, b, idx), z[f"xfade_{i}"]), i


# ---- tier F: the NSF-HiFiGAN post-net against the reference's own nsf_hifigan/models.py ------------------------------------------
def test_enhancer_oracle_against_reference_run():
    import glue_cases as GC
    from oracle import enhancer as OE
    g = load("ref_enhancer.npz")
    sd = GC.nsf_state_dict()
    mel, f0, ri = GC.nsf_inputs()
    upp = int(np.prod(GC.NSF_CONFIG["upsample_rates"]))
    src = OE.sine_source(sd, f0, upp, GC.NSF_CONFIG["sampling_rate"], ri)
    assert src.shape == (1, GC.NSF_L * upp, 1)
    assert (src[0, :, 0] - g["source"]).abs().max() < 1e-6
    audio = OE.generator(sd, GC.NSF_CONFIG, mel, f0, ri)
    assert audio.shape == (1, 1, GC.NSF_L * upp)
    assert (audio[0, 0] - g["audio"]).abs().max() < 1e-5
    assert 0.05 < rms(g["audio"]) < 0.5                      # the fixture is not saturated by the final tanh
    # the unvoiced stretch: f0 = 0 gives a constant phase (only the initial offsets), not silence - as in the reference
    assert rms(g["source"][3 * upp:6 * upp]) > 0


# ---- tier G: causal mode (c: true) -------------------------------------------------------------------------------------------------
def test_causal_oracle_against_reference_run():
    """oracle/ctrlnet.py's causal path against the reference's CombSub built with c=True (stand-ins for the two third-party
    causal primitives, see make_golden.py tier g): the reference's own wiring and normaliser are pinned."""
    z = np.load(os.path.join(GOLDEN, "model_CombSub_causal.npz"))
    model, cfg = synthetic.build_model("CombSub", seed=int(z["seed_weights"]))
    cfg = dict(cfg, c=True)
    inp = synthetic.make_inputs(int(z["seed_inputs"]), 2, 24)
    with torch.no_grad():
        sig, ph, (hm, nz), aux = S.combsub_forward(model.state_dict(), cfg, inp["units"], inp["f0"], inp["volume"],
                                                   inp["spk_id"], noise=inp["noise"])
    want = torch.from_numpy(z["signal"])
    assert rms(sig - want) < 1e-5 * rms(want)
    assert rms(hm[:, ::7] - torch.from_numpy(z["harmonic"])) < 1e-5 * rms(want)
    # causality of the network itself: changing the LAST frames' units leaves the earlier control frames unchanged,
    # except through GroupNorm's utterance-wide statistics (the reference keeps GroupNorm in causal mode)
    from oracle import ctrlnet as C
    sd = {k[len("unit2ctrl."):]: v for k, v in model.state_dict().items() if k.startswith("unit2ctrl.")}
    phase = torch.zeros(2, 24)
    a = C.unit2control(sd, inp["units"], inp["f0"], phase, inp["volume"], inp["spk_id"], None, {"x": 1024}, True, causal=True)
    assert (a - torch.from_numpy(z["ctrl"])).abs().max() > 0        # (phase differs from the fixture's: a different input)
