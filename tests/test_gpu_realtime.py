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
