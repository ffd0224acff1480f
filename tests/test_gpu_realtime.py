"""GPU parity of the caller-side glue kernels (a14 SOLA splice, a15 volume gate) against reference-derived fixtures."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import realtime as RT

pytestmark = pytest.mark.gpu
HOP = 512


def test_sola_against_golden(ctx, dev):
    z = np.load(os.path.join(GOLDEN, "glue_sola.npz"))
    block, xfade, search, delay = [int(v) for v in z["sizes"]]
    buf = torch.from_numpy(z["prev"]).to(dev).clone()
    emitted, shift = ctx.sola(torch.from_numpy(z["audio"]).to(dev), buf, block, xfade, search, delay)
    assert int(shift.item()) == int(z["shift"])
    assert (emitted.cpu() - torch.from_numpy(z["emitted"])).abs().max() < 2e-6
    assert (buf.cpu() - torch.from_numpy(z["new_buffer"])).abs().max() < 1e-7


def test_sola_stream_against_oracle(ctx, dev):
    """Eight consecutive blocks of a drifting sine: every shift and every emitted block match the oracle."""
    block, xfade, search, delay = 8820, 1764, 441, 882
    n = 44544
    rng = np.random.Generator(np.random.PCG64(3))
    buf_o = torch.zeros(xfade)
    buf_g = torch.zeros(xfade, device=dev)
    for k in range(8):
        t = (np.arange(n) + k * block + rng.integers(-200, 200)) / 44100
        audio = torch.from_numpy((0.3 * np.sin(2 * np.pi * 196.0 * t) + 0.01 * rng.standard_normal(n)).astype(np.float32))
        em_o, buf_o, sh_o = RT.sola_step(audio, buf_o, block, xfade, search, delay)
        em_g, sh_g = ctx.sola(audio.to(dev), buf_g, block, xfade, search, delay)
        assert int(sh_g.item()) == sh_o, k
        assert (em_g.cpu() - em_o).abs().max() < 2e-6
        assert (buf_g.cpu() - buf_o).abs().max() < 1e-7


def test_volume_gate(ctx, dev):
    z = np.load(os.path.join(GOLDEN, "glue_offline.npz"))
    vol = z["volume"].astype(np.float32)
    Fr = len(vol)
    sig = torch.ones(1, Fr * HOP, device=dev)
    ctx.volume_gate_(sig, torch.from_numpy(vol)[None].to(dev), -60, HOP)
    want = RT.volume_gate(vol, -60, HOP)
    assert torch.equal(sig.cpu(), want)
    assert (sig.cpu()[:, ::16] - torch.from_numpy(z["mask_up"])).abs().max() < 1e-6
    # batched + in place on a real signal
    rng = np.random.Generator(np.random.PCG64(4))
    v2 = (np.abs(rng.standard_normal((3, 50))) * 0.002).astype(np.float32)
    s2 = torch.from_numpy(rng.standard_normal((3, 50 * HOP)).astype(np.float32))
    want2 = torch.cat([s2[i:i + 1] * RT.volume_gate(v2[i], -55, HOP) for i in range(3)])
    got2 = ctx.volume_gate_(s2.to(dev).clone(), torch.from_numpy(v2).to(dev), -55, HOP)
    assert torch.equal(got2.cpu(), want2)


def test_offline_render_plumbing(dev, lib_path):
    """BASELINE config #0 plumbing (main.py:143-174): per-slice model call, whole-file gate sliced per segment and
    multiplied in place, silence padding and cross-fade of overlapping slices - device path vs the CPU oracle."""
    import infer_offline
    import synthetic
    from ddsp.vocoder import DotDict
    from oracle import synth as OS
    model, cfg = synthetic.build_model("CombSubFast", seed=8)
    sd = model.state_dict()
    Fr = 173                                            # a 2 s file at 44.1 kHz / 512 (main.py:41-46)
    inp = synthetic.make_inputs(91, 1, Fr)
    inp["volume"][0, 60:75] = 1e-5                      # a silent stretch: the gate closes there
    args = DotDict({"data": {"block_size": 512, "sampling_rate": 44100}})
    # three slices: a gap before the second (silence padding) and an overlap before the third (cross-fade)
    spans = [(0, 50), (55, 120), (118, 173)]
    segments = [(a, inp["units"][:, a:b]) for a, b in spans]
    # oracle pipeline
    gate = RT.volume_gate(inp["volume"][0].numpy(), -60, 512)
    result, current = np.zeros(0), 0
    for a, b in spans:
        with torch.no_grad():
            out = OS.combsubfast_forward(sd, cfg, inp["units"][:, a:b], inp["f0"][:, a:b], inp["volume"][:, a:b],
                                         inp["spk_id"], noise=inp["noise"][:, a * 512:b * 512])[0]
        out = (out * gate[:, a * 512:b * 512]).squeeze().numpy()
        silent = a * 512 - current
        if silent >= 0:
            result = np.append(np.append(result, np.zeros(silent)), out)
        else:
            result = RT.slice_cross_fade(result, out, current + silent)
        current = current + silent + len(out)
    # device pipeline with the same injected noise per slice
    model = model.to(dev)

    class Injected(torch.nn.Module):
        def forward(self, units, f0, vol, spk_id=None, spk_mix_dict=None, **kw):
            a = self.start.pop(0)
            n = units.shape[1]
            return model(units, f0, vol, spk_id, spk_mix_dict, noise=inp["noise"][:, a * 512:(a + n) * 512].to(dev))
    inj = Injected()
    inj.start = [a for a, _ in spans]
    seg_dev = [(a, u.to(dev)) for a, u in segments]
    got, sr_o = infer_offline.render(inj, args, seg_dev, inp["f0"].to(dev), inp["volume"].to(dev), inp["spk_id"].to(dev),
                                     threshold_db=-60)
    assert sr_o == 44100 and got.shape == result.shape
    assert np.sqrt(np.mean((got - result) ** 2)) < 1e-4
    assert np.abs(got[64 * 512:70 * 512]).max() == 0.0    # 9-frame dilation: frames 64..70 of the 60..74 stretch are closed


def test_graphed_forward_equals_eager(dev, lib_path):
    """graphed.GraphedSynth (HIP-graph replay of the inference forward on a context of its own) against the eager
    call: same inputs and the same injected noise give the same bits, across several replays with new inputs, after
    an in-place weight update, and the captured context refuses eager use."""
    import synthetic, graphed
    model, cfg = synthetic.build_model("CombSub", seed=11, device=dev)
    model.eval()
    B, Fr = 1, 87
    g = graphed.GraphedSynth(model, B, Fr)
    T = Fr * 512
    for rep in range(3):
        inp = {k: v.to(dev) for k, v in synthetic.make_inputs(500 + rep, B, Fr, with_noise=False).items()}
        noise = torch.rand(B, T, device=dev, generator=torch.Generator(device=dev).manual_seed(rep))
        with torch.no_grad():
            want = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=noise)
        got = g(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=noise)
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert torch.equal(got[2][0], want[2][0]) and torch.equal(got[2][1], want[2][1])
        assert float(want[0].abs().max()) > 0
    # fresh noise per replay when none is injected: two replays of the same inputs differ in the noise branch only
    a = [t.clone() for t in (g(inp["units"], inp["f0"], inp["volume"], inp["spk_id"])[2])]
    b = g(inp["units"], inp["f0"], inp["volume"], inp["spk_id"])[2]
    assert torch.equal(a[0], b[0]) and not torch.equal(a[1], b[1])
    # weights are read at replay time
    with torch.no_grad():
        list(model.unit2ctrl.parameters())[-1].add_(0.05)
        want = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=noise)
    got = g(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=noise)
    assert torch.equal(got[0], want[0])
    with pytest.raises(RuntimeError):
        g.ctx.upsample(torch.zeros(1, 2, 1, device=dev), 512)       # frozen: owned by the graph
    with pytest.raises(ValueError):
        g(inp["units"][:, :5], inp["f0"][:, :5], inp["volume"][:, :5], inp["spk_id"])


def test_phase_vocoder_matches_reference(dev, lib_path):
    """`ddsp_phase_vocoder` against the reference's own `gui.phase_vocoder` outputs for the GUI's
    1764-sample cross-fade, an odd length and a short even one.  The kernels repeat the reference's fp32 operation
    order after the DFTs (its oscillator arguments reach 5.5e3 rad, so that order is part of the result); measured
    error 3e-7 max / 7e-8 RMS on a 0.22-RMS signal (tools/pv_check.py), asserted at 5e-6 / 1e-6."""
    import realtime
    from frontend_cases import PV_SIZES, pv_inputs
    g = np.load(os.path.join(GOLDEN, "glue_phase_vocoder.npz"))
    for i, n in enumerate(PV_SIZES):
        a, b, fo, fi = pv_inputs(i)
        got = realtime.phase_vocoder(a.to(dev), b.to(dev), fo.to(dev), fi.to(dev)).cpu()
        want = torch.from_numpy(g[f"pv_{i}"])
        assert got.shape == want.shape == (n,)
        err = (got - want).abs()
        assert float(err.max()) < 5e-6 and float(err.pow(2).mean().sqrt()) < 1e-6, (n, float(err.max()))
        # ends: pure `a` at t = 0 (fade_out = 1), and the oscillator term vanishes where a window is 0
        assert abs(float(got[0]) - float(a[0])) < 1e-6
    with pytest.raises(ValueError):
        realtime.phase_vocoder(a.to(dev), b[:10].to(dev), fo.to(dev), fi.to(dev))


def test_splicer_with_phase_vocoder(dev, lib_path):
    """Splicer(use_phase_vocoder=True): same SOLA shift and same samples after the cross-fade region as the linear
    splice; the head equals phase_vocoder(kept tail, new head at the shift)."""
    import realtime
    sr = 44100
    n = 44544
    t = torch.arange(n) / sr
    mk = lambda ph: (0.3 * torch.sin(2 * torch.pi * 220.0 * t + ph)).to(dev)
    lin = realtime.Splicer(sr, 0.2, 0.04, dev)
    pv = realtime.Splicer(sr, 0.2, 0.04, dev, use_phase_vocoder=True)
    for ph in (0.0, 0.7, 1.9):
        x = mk(ph)
        kept = pv.buffer.clone()
        e_lin, e_pv = lin.push(x), pv.push(x)
        assert int(lin.last_shift) == int(pv.last_shift)
        assert torch.equal(e_lin[pv.xfade:], e_pv[pv.xfade:])
        assert torch.equal(lin.buffer, pv.buffer)
        start = n - pv.block - pv.xfade - pv.search - pv.delay + int(pv.last_shift)
        want = realtime.phase_vocoder(kept, x[start:start + pv.xfade], pv.fade_out, pv.fade_in)
        assert torch.equal(e_pv[:pv.xfade], want)


@pytest.mark.parametrize("use_pv", [False, True])
def test_splicer_against_reference_callback(dev, lib_path, use_pv):
    """G10: `realtime.Splicer` over eight blocks against the reference's own `gui.GUI.audio_callback`
    (tests/golden/ref_gui_stream.npz: make_golden.py tier e ran the callback on an object made without its window)."""
    import glue_cases as GC
    import realtime
    z = np.load(os.path.join(GOLDEN, "ref_gui_stream.npz"))
    block, xfade, search, delay, n_in = [int(v) for v in z["sizes"]]
    sp = realtime.Splicer(GC.GUI_SR, GC.GUI_BLOCK_TIME, GC.GUI_XFADE_TIME, dev, use_phase_vocoder=use_pv)
    assert (sp.block, sp.xfade, sp.search, sp.delay) == (block, xfade, search, delay)
    assert sp.input_frames(GC.GUI_BUFFER_NUM) == n_in
    tag = "pv" if use_pv else "plain"
    for k in range(GC.GUI_BLOCKS):
        emitted = sp.push(GC.gui_model_output(k).to(dev))
        assert int(sp.last_shift.item()) == int(z[f"shift_{tag}"][k]), k
        e = emitted.cpu()
        assert (e[xfade:] - torch.from_numpy(z["out_plain"][k][xfade:])).abs().max() == 0, k
        if use_pv:
            assert (e[:xfade] - torch.from_numpy(z["head_pv"][k])).abs().max() < 2e-5, k
        else:
            assert (e - torch.from_numpy(z["out_plain"][k])).abs().max() < 2e-6, k
    assert (sp.buffer.cpu() - torch.from_numpy(z[f"buffer_{tag}"])).abs().max() < 1e-6


def test_gate_against_reference_infer(ctx, dev):
    """The volume gate as `gui.SvcDDSP.infer` computes and applies it (gui.py:103-112,125-127; executed by the fixture
    generator with stand-ins for the extractors and the model): device volume extraction + device gate."""
    import glue_cases as GC
    z = np.load(os.path.join(GOLDEN, "ref_offline_glue.npz"))
    vol = ctx.volume_extract(torch.from_numpy(GC.gate_audio())[None].to(dev), HOP)
    assert np.allclose(vol.cpu().numpy()[0], z["volume"], rtol=2e-6, atol=0)
    out = ctx.volume_gate_(GC.gate_model_output().to(dev).clone(), vol, GC.GATE_THRESHOLD, HOP)
    assert torch.equal(out.cpu()[0], torch.from_numpy(z["gated"]))
