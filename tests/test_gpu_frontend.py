"""SURVEY 8(f) rank 2 on the GPU: `ddsp_volume_extract` / `ddsp_align_units` (and their `ddsp.vocoder` mirror) against
tests/golden/glue_frontend.npz, the outputs of the reference's own `Volume_Extractor.extract` and
`Units_Encoder.encode` alignment."""
import os

import numpy as np
import pytest
import torch

from frontend_cases import FRONTEND_ALIGN, FRONTEND_VOLUME, align_units_input, volume_audio

pytestmark = pytest.mark.gpu


def _golden():
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "glue_frontend.npz"))
    return {k: d[k] for k in d.files}


def test_volume_extract_matches_reference(dev, lib_path):
    from ddsp.vocoder import Volume_Extractor
    g = _golden()
    for i in range(len(FRONTEND_VOLUME)):
        audio, hop = volume_audio(i)
        want = g[f"vol_{i}"]
        ve = Volume_Extractor(hop, device=dev)
        got_np = ve.extract(audio)                               # numpy in -> numpy out, as main.py calls it
        assert isinstance(got_np, np.ndarray) and got_np.shape == want.shape and got_np.dtype == np.float32
        # fp32 block means: the reference sums pairwise in fp32, the kernel in fp64 -> a few fp32 ulps
        assert np.allclose(got_np, want, rtol=2e-6, atol=0), (i, np.abs(got_np - want).max())
        got_t = ve.extract(torch.from_numpy(audio).to(dev))      # device tensor in -> device tensor out
        assert got_t.is_cuda and np.array_equal(got_t.cpu().numpy(), got_np)
    # batched: rows are independent
    audio, hop = volume_audio(1)
    x = torch.from_numpy(np.stack([audio, audio[::-1].copy()])).to(dev)
    both = Volume_Extractor(hop, device=dev).extract(x)
    assert both.shape == (2, len(audio) // hop + 1)
    assert np.allclose(both[0].cpu().numpy(), g["vol_1"], rtol=2e-6, atol=0)


def test_align_units_matches_reference_exactly(dev, lib_path):
    from ddsp.vocoder import align_units
    g = _golden()
    for i in range(len(FRONTEND_ALIGN)):
        units, n, sr, hop = align_units_input(i)
        got = align_units(units.to(dev), n, sr, hop).cpu()
        want = torch.from_numpy(g[f"align_{i}"])
        assert got.shape == want.shape
        assert torch.equal(got, want), i                         # an index gather: bit-exact
    # batch of two utterances, odd feature width (scalar copy path)
    u = torch.randn(2, 33, 7, generator=torch.Generator().manual_seed(1))
    got = align_units(u.to(dev), 16000, 16000, 480).cpu()
    idx = torch.clamp(torch.round(1.5 * torch.arange(16000 // 480 + 1)).long(), max=32)
    assert torch.equal(got, u[:, idx])


def test_frontend_refusals(ctx, dev, lib_path):
    from ddsp.vocoder import Volume_Extractor
    with pytest.raises(ValueError):
        ctx.volume_extract(torch.zeros(1, 200, device=dev), 512)     # shorter than the reflect padding
    with pytest.raises(RuntimeError):
        Volume_Extractor(512, device=dev).extract(torch.zeros(2048))  # CPU tensor: no CPU path
    assert ctx.align_units(torch.zeros(0, 4, 8, device=dev), 5, 1.0).shape == (0, 5, 8)
