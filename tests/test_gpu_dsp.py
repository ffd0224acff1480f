"""GPU parity of the DSP stage kernels (a1-a3, a5-a8) against the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import rms
from oracle import dsp as O
import hipddsp

pytestmark = pytest.mark.gpu

SR, HOP = 44100, 512


def _f0(seed, B, Fr, lo=65.0, hi=800.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.uniform(lo, hi, size=(B, Fr, 1)).astype(np.float32))


@pytest.mark.parametrize("Fr,C", [(1, 1), (2, 3), (7, 32), (172, 1), (173, 5)])
def test_upsample_bit_exact(ctx, dev, Fr, C):
    rng = np.random.Generator(np.random.PCG64(11 + Fr))
    x = torch.from_numpy(rng.standard_normal((3, Fr, C)).astype(np.float32) * 300)
    want = O.frames_to_samples(x, HOP)
    got = ctx.upsample(x.to(dev), HOP).cpu()
    assert got.shape == want.shape
    assert torch.equal(got, want)


@pytest.mark.parametrize("precise", [True, False])
@pytest.mark.parametrize("with_init", [False, True])
def test_phase_scan_matches_oracle(ctx, dev, precise, with_init):
    B, Fr = 5, 172
    f0f = _f0(3, B, Fr)
    init = torch.tensor([0.3, -2.0, 3.1, 0.0, 6.0]) if with_init else None
    f0 = O.frames_to_samples(f0f, HOP).squeeze(-1)
    rot = O.rotation_from_f0(f0, SR, init, precise)
    comb = O.sinc_comb(rot, f0, SR)
    out = ctx.phase_scan(f0f.to(dev), HOP, SR, None if init is None else init.to(dev), precise, 1,
                         want_rot=True, want_phase=True, want_f0=True)
    assert torch.equal(out["f0_up"].cpu(), f0)
    d = (out["rot"].cpu() - rot)
    d = d - torch.round(d)                      # a wrap at +-0.5 is the same phase
    # fp64 running sums differ only by summation order; in train mode a 1-ulp(fp32 of S) flip is possible
    tol = 1e-6 if precise else 2.5e-4
    assert d.abs().max() < tol
    assert (d.abs() > 1e-6).float().mean() < 1e-4
    pf = 2 * np.pi * rot[:, ::HOP]
    dp = out["phase_frames"].cpu() - pf
    assert (dp - 2 * np.pi * torch.round(dp / (2 * np.pi))).abs().max() < 2e-3
    # combtooth: sin(pi*x)/(pi*x) with x up to ~340 -> compare where the wrap did not flip
    same = d.abs() < 1e-7
    dc = (out["comb"].cpu() - comb)[same]
    assert dc.abs().max() < 2e-5
    assert rms(out["comb"].cpu() - comb) < 1e-5


def test_phase_scan_known_answers(ctx, dev):
    """The reference's own known-answer tests (ddsp/core.py:62-97) with hop = 1 (sample-rate f0)."""
    f = torch.tensor([[1.0, 1.0, 1.0]])
    out = ctx.phase_scan(f.to(dev), 1, 4, None, False, 0, want_rot=True)["rot"].cpu()
    assert torch.allclose(out, torch.tensor([[0.25, 0.50, -0.25]]))
    f = torch.tensor([[1.0, 2.0, 3.0]])
    out = ctx.phase_scan(f.to(dev), 1, 4, None, False, 0, want_rot=True)["rot"].cpu()
    assert torch.allclose(out, torch.tensor([[0.25, -0.25, -0.50]]))
    f = torch.tensor([[1.0, 1.0, 1.0]])
    out = ctx.phase_scan(f.to(dev), 1, 4, torch.tensor([np.pi]).float().to(dev), False, 0, want_rot=True)["rot"].cpu()
    assert torch.allclose(out, torch.tensor([[-0.25, 0.0, 0.25]]), atol=1e-6)
    f = torch.tensor([[1.0, 1.0, 1.0], [1.0, 2.0, 3.0]])
    ip = torch.tensor([np.pi, 0.0]).float()
    out = ctx.phase_scan(f.to(dev), 1, 4, ip.to(dev), True, 0, want_rot=True)["rot"].cpu()
    assert torch.allclose(out, torch.tensor([[-0.25, 0.0, 0.25], [0.25, -0.25, -0.50]]), atol=1e-5)


def _ctrl(seed, B, Fr, W, std=0.5):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((rng.standard_normal((B, Fr, W)) * std).astype(np.float32))


def _resp(mode, c):
    if mode == 0:
        return torch.exp(1.j * torch.cumsum(np.pi * torch.tanh(c), dim=-1))
    m = torch.exp(c) if mode == 1 else torch.exp(c) / 128
    return torch.complex(m, torch.zeros_like(m))


@pytest.mark.parametrize("mode,n_mag", [(0, 256), (1, 512), (2, 256), (1, 256), (2, 513), (0, 65)])
def test_fir_from_ctrl(ctx, dev, mode, n_mag):
    B, Fr = 3, 9
    W = n_mag + 24
    c = _ctrl(5 + mode, B, Fr, W)
    f0f = _f0(9, B, Fr, 65.0, 800.0)
    f0f[0, 0, 0] = 65.0      # half width 1017 > 511: plain raised cosine everywhere
    f0f[0, 1, 0] = 800.0     # half width 82: the w>1 -> 1 quirk and the periodic w<-1 side both show
    sub = c[..., 8:8 + n_mag]
    hw = 1.5 * SR / (f0f + 1e-3) if mode == 1 else None
    want = O.fir_from_response(_resp(mode, sub), hann=(mode != 0), half_width=hw)
    got = ctx.fir_from_ctrl(mode, c.reshape(B * Fr, W).to(dev), 8, n_mag, B * Fr, SR,
                            f0f.to(dev) if mode == 1 else None).cpu().reshape(B, Fr, -1)
    assert got.shape == want.shape
    scale = float(want.abs().max())
    assert (got - want).abs().max() < 2e-5 * scale + 1e-7, (got - want).abs().max()
    assert rms(got - want) < 2e-6 * scale


@pytest.mark.parametrize("n,Fr,B", [(510, 6, 2), (1022, 6, 2), (510, 7, 1), (1022, 1, 2), (64, 3, 1), (2046, 5, 1)])
def test_ltv_fir_small_vs_direct(ctx, dev, n, Fr, B):
    rng = np.random.Generator(np.random.PCG64(n + Fr))
    x = torch.from_numpy(rng.uniform(-1, 1, size=(B, Fr * HOP)).astype(np.float32))
    ir = torch.from_numpy((rng.standard_normal((B, Fr, n)) / np.sqrt(n)).astype(np.float32))
    want64 = O.ltv_fir_direct(x, ir)
    want_fft = O.ltv_fir_fft(x, ir)
    got, _ = ctx.ltv_fir(x.to(dev), ir.to(dev), B, Fr, HOP)
    got = got.cpu()
    assert rms(want_fft.double() - want64) < 1e-6           # the oracle's two forms agree
    assert rms(got.double() - want64) < 1e-6
    assert (got.double() - want64).abs().max() < 1e-5


@pytest.mark.parametrize("n,Fr,B", [(510, 6, 2), (1022, 6, 2), (510, 7, 1), (1022, 1, 2), (64, 3, 1), (32, 2, 1), (254, 13, 3),
                                    (1022, 25, 2), (1534, 9, 1), (2046, 5, 1)])
def test_ltv_fir_split_bf16_vs_direct(ctx, dev, n, Fr, B):
    """The inference arithmetic (three bf16 matrix products per fp32 product, include/ddsp_amd.h DDSP_FIR_SPLIT_BF16)
    against the fp64 direct form.  Tolerance: 1e-5 of the output's rms / 3e-5 of its peak, i.e. an order of magnitude
    inside the 1e-4 waveform gate; n = 1534 and 2046 exceed that kernel's staging and must fall back to fp32 products."""
    rng = np.random.Generator(np.random.PCG64(n + Fr))
    x = torch.from_numpy(rng.uniform(-1, 1, size=(B, Fr * HOP)).astype(np.float32))
    ir = torch.from_numpy((rng.standard_normal((B, Fr, n)) / np.sqrt(n)).astype(np.float32))
    add = torch.from_numpy(rng.standard_normal((B, Fr * HOP)).astype(np.float32))
    want64 = O.ltv_fir_direct(x, ir)
    got, got_sum = ctx.ltv_fir(x.to(dev), ir.to(dev), B, Fr, HOP, add_in=add.to(dev), math=hipddsp.FIR_SPLIT_BF16)
    got = got.cpu()
    scale, peak = rms(want64), float(want64.abs().max())
    assert rms(got.double() - want64) < 1e-5 * scale
    assert (got.double() - want64).abs().max() < 3e-5 * peak
    assert torch.equal(got_sum.cpu(), add + got)
    if n >= 1534:
        ref, _ = ctx.ltv_fir(x.to(dev), ir.to(dev), B, Fr, HOP)
        assert torch.equal(ref.cpu(), got)
    # both excitation modes feed the same samples to either arithmetic
    for exc, src in ((1, torch.from_numpy(rng.random((B, Fr * HOP), dtype=np.float32)).to(dev)), (2, None)):
        a, _ = ctx.ltv_fir(src, ir.to(dev), B, Fr, HOP, excitation=exc, noise_seed=5)
        b_, _ = ctx.ltv_fir(src, ir.to(dev), B, Fr, HOP, excitation=exc, noise_seed=5, math=hipddsp.FIR_SPLIT_BF16)
        assert (a - b_).abs().max() < 3e-5 * float(a.abs().max())


@pytest.mark.parametrize("B,Fr,n", [(130, 17, 64), (64, 40, 254), (128, 13, 1022), (43, 61, 510)])
def test_ltv_fir_split_bf16_marching_blocks(ctx, dev, B, Fr, n):
    """Large batches take the kernel whose blocks walk along a batch row with the frames in an LDS ring (one or several
    runs per row, ragged last run / last step): same result as the fp32-product kernel within the split-bf16 error, in
    all three excitation modes and with the fused sum."""
    g = torch.Generator().manual_seed(B + Fr)
    x = (torch.rand(B, Fr * HOP, generator=g) * 2 - 1).to(dev)
    ir = (torch.randn(B, Fr, n, generator=g) / n ** 0.5).to(dev)
    add = torch.randn(B, Fr * HOP, generator=g).to(dev)
    ref, _ = ctx.ltv_fir(x, ir, B, Fr, HOP)
    got, got_sum = ctx.ltv_fir(x, ir, B, Fr, HOP, add_in=add, math=hipddsp.FIR_SPLIT_BF16)
    peak = float(ref.abs().max())
    assert (got - ref).abs().max() < 3e-5 * peak
    assert rms((got - ref).cpu()) < 1e-5 * rms(ref.cpu())
    assert torch.equal(got_sum, add + got)
    for exc, src in ((1, torch.rand(B, Fr * HOP, generator=g).to(dev)), (2, None)):
        a, _ = ctx.ltv_fir(src, ir, B, Fr, HOP, excitation=exc, noise_seed=11)
        b_, _ = ctx.ltv_fir(src, ir, B, Fr, HOP, excitation=exc, noise_seed=11, math=hipddsp.FIR_SPLIT_BF16)
        assert (a - b_).abs().max() < 3e-5 * float(a.abs().max())


def test_ltv_fir_full_size_and_fusions(ctx, dev):
    B, Fr, n = 4, 172, 1022
    rng = np.random.Generator(np.random.PCG64(77))
    u = torch.from_numpy(rng.random((B, Fr * HOP), dtype=np.float32))
    ir = torch.from_numpy((rng.standard_normal((B, Fr, n)) / np.sqrt(n)).astype(np.float32))
    add = torch.from_numpy(rng.standard_normal((B, Fr * HOP)).astype(np.float32))
    want = O.ltv_fir_fft(u * 2 - 1, ir)
    got, got_sum = ctx.ltv_fir(u.to(dev), ir.to(dev), B, Fr, HOP, excitation=1, add_in=add.to(dev))
    assert rms(got.cpu() - want) < 1e-6
    assert torch.equal(got_sum.cpu(), add + got.cpu())
    # linearity: filter(a*x) == a*filter(x) exactly for a power of two
    got2, _ = ctx.ltv_fir((u * 2 - 1).mul(0.5).to(dev), ir.to(dev), B, Fr, HOP)
    assert torch.equal(got2.cpu() * 2, got.cpu())
    # in-kernel excitation: deterministic per seed, different across seeds, U[-1,1)-like after an identity filter
    ident = torch.zeros(1, 4, 64)
    ident[:, :, 32] = 1.0
    g1, _ = ctx.ltv_fir(None, ident.to(dev), 1, 4, HOP, excitation=2, noise_seed=123)
    g2, _ = ctx.ltv_fir(None, ident.to(dev), 1, 4, HOP, excitation=2, noise_seed=123)
    g3, _ = ctx.ltv_fir(None, ident.to(dev), 1, 4, HOP, excitation=2, noise_seed=124)
    assert torch.equal(g1, g2) and not torch.equal(g1, g3)
    g = g1.cpu()
    assert g.min() >= -1 and g.max() < 1 and abs(float(g.mean())) < 0.05 and abs(float(g.var()) - 1 / 3) < 0.03


def test_argument_errors(ctx, dev):
    x = torch.zeros(1, 2 * HOP, device=dev)
    ir = torch.zeros(1, 2, 511, device=dev)
    with pytest.raises(ValueError):
        ctx.ltv_fir(x, ir, 1, 2, HOP)                         # odd filter length
    with pytest.raises(ValueError):
        ctx.ltv_fir(x, torch.zeros(1, 2, 510, device=dev), 1, 2, 256)   # hop not built
    with pytest.raises(RuntimeError):
        ctx.upsample(torch.zeros(1, 2, 1), HOP)               # CPU tensor: no fallback


# ---- ddsp.core.frequency_filter (the reference's public filter entry point) against the reference's own outputs ----
def _golden(name):
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    return {k: torch.from_numpy(d[k]) for k in d.files}


def _responses(ctrl, f0f):
    """The three complex frequency responses the reference models feed to frequency_filter (as in make_golden.py)."""
    cz = lambda x: torch.complex(x, torch.zeros_like(x))
    gd = np.pi * torch.tanh(ctrl[..., :256])
    return (torch.exp(1.j * torch.cumsum(gd, axis=-1)), cz(torch.exp(ctrl[..., 256:768])),
            cz(torch.exp(ctrl[..., 768:]) / 128), 1.5 * SR / (f0f + 1e-3))


def test_frequency_filter_matches_reference_short(dev, lib_path):
    """`ddsp.core.frequency_filter` (device impulse responses + the HIP LTV-FIR kernel) on the inputs of
    tests/golden/core_filter_short.npz, whose y_* were produced by the unmodified reference `ddsp/core.py:331`:
    all-pass (no window), dynamic window (half widths 1017 and 82: both sides of the window quirk) and static Hann."""
    from ddsp import core as C
    g = _golden("core_filter_short.npz")
    ap, src, nse, hw = _responses(g["ctrl"], g["f0_frames"])
    d = lambda t: t.to(dev)
    cases = (("y_ap", g["audio"], ap, False, None), ("y_h", g["y_ap"], src, True, hw), ("y_n", g["audio"], nse, True, None))
    for key, x, resp, hann, half in cases:
        got = C.frequency_filter(d(x), d(resp), hann_window=hann, half_width_frames=None if half is None else d(half)).cpu()
        want = g[key]
        assert got.shape == want.shape and got.dtype == want.dtype
        assert rms(got - want) < 2e-5 * max(1.0, rms(want)), (key, rms(got - want), rms(want))
        assert (got - want).abs().max() < 2e-4 * max(1.0, float(want.abs().max())), key


def test_frequency_filter_matches_reference_long(dev, lib_path):
    """Fr = 172 (one 2 s clip): inputs regenerated from the fixture's seeds, reference outputs stored every 29th sample."""
    from ddsp import core as C
    g = _golden("core_filter_long.npz")
    r = lambda s: np.random.Generator(np.random.PCG64(s))
    t32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    ctrl = t32(r(310).standard_normal((1, 172, 1024)) * 0.5)
    f0f = t32(r(311).uniform(65, 800, size=(1, 172, 1)))
    audio = t32(r(312).uniform(-1, 1, size=(1, 172 * HOP)))
    ap, src, nse, hw = _responses(ctrl, f0f)
    d = lambda t: t.to(dev)
    y_ap = C.frequency_filter(d(audio), d(ap), hann_window=False)
    y_h = C.frequency_filter(y_ap, d(src), hann_window=True, half_width_frames=d(hw))
    y_n = C.frequency_filter(d(audio), d(nse), hann_window=True)
    for got, key, ref_rms in ((y_ap, "y_ap", g["rms"][0]), (y_h, "y_h", g["rms"][1]), (y_n, "y_n", g["rms"][2])):
        got = got.cpu()
        assert abs(rms(got) - float(ref_rms)) < 1e-4 * max(1.0, float(ref_rms)), key
        assert rms(got[:, ::29] - g[key]) < 3e-5 * max(1.0, float(ref_rms)), (key, rms(got[:, ::29] - g[key]))


def test_frequency_filter_refusals(dev, lib_path):
    from ddsp import core as C
    x = torch.zeros(1, 4 * HOP, device=dev)
    resp = torch.ones(1, 4, 65, dtype=torch.complex64, device=dev)
    assert C.frequency_filter(x, resp, hann_window=True).shape == (1, 4 * HOP)
    with pytest.raises(ValueError):
        C.frequency_filter(x[:, :-3], resp)                       # not a whole number of frames
    with pytest.raises(ValueError):
        C.frequency_filter(torch.zeros(1, 4 * 256, device=dev), resp)   # hop 256: outside what the kernel supports
    with pytest.raises(NotImplementedError):
        C.frequency_filter(x.clone().requires_grad_(), resp)     # forward only
    with pytest.raises(RuntimeError):
        C.frequency_filter(x.cpu(), resp.cpu())                   # no CPU fallback


def test_remove_above_fmax_matches_reference(dev, lib_path):
    """tests/golden/core_fmax.npz holds the unmodified reference's output (pitches on both sides of Nyquist / 16)."""
    from ddsp import core as C
    g = _golden("core_fmax.npz")
    got = C.remove_above_fmax(g["amps"].to(dev), g["pitch"].to(dev), SR / 2, level_start=1).cpu()
    assert torch.equal(got, g["out"])
    with pytest.raises(RuntimeError):
        C.remove_above_fmax(g["amps"], g["pitch"], SR / 2)
