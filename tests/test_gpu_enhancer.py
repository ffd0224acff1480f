"""SURVEY 8(f) rank 1 on the device: the NSF-HiFiGAN post-net (source module, generator, log-mel front end, Enhancer.enhance)
against fixtures produced by the reference's own nsf_hifigan/models.py (tests/golden/ref_enhancer.npz) and against the CPU
oracle.  Unpinned third-party boundaries: librosa's mel filter bank, torchaudio's resampler (absent from the image)."""
import json
import os

import numpy as np
import pytest
import torch

import glue_cases as GC
from conftest import GOLDEN, rms
from oracle import enhancer as OE
from oracle import resample as OR

pytestmark = pytest.mark.gpu
UPP = int(np.prod(GC.NSF_CONFIG["upsample_rates"]))


def _generator(dev):
    from enhancer import AttrDict, Generator
    return Generator(AttrDict(GC.NSF_CONFIG), GC.nsf_state_dict()).to(dev)


def test_source_module_against_reference(ctx, dev):
    z = np.load(os.path.join(GOLDEN, "ref_enhancer.npz"))
    sd = GC.nsf_state_dict()
    mel, f0, ri = GC.nsf_inputs()
    r = ri[0].clone()
    r[0] = 0
    src = ctx.nsf_source(f0[0].to(dev), r.to(dev), sd["m_source.l_linear.weight"].reshape(-1).to(dev),
                         sd["m_source.l_linear.bias"].to(dev), UPP, 44100, 0.1)
    assert src.shape == (GC.NSF_L * UPP,)
    assert float((src.cpu() - torch.from_numpy(z["source"])).abs().max()) < 2e-6
    # a long track: the fp64 phase stays exact over a minute of audio
    L, upp = 5000, 512
    f0l = torch.full((1, L), 440.0)
    want = OE.sine_source(sd, f0l, upp, 44100, ri)[0, :, 0]
    got = ctx.nsf_source(f0l[0].to(dev), r.to(dev), sd["m_source.l_linear.weight"].reshape(-1).to(dev),
                         sd["m_source.l_linear.bias"].to(dev), upp, 44100, 0.1)
    assert float((got.cpu() - want).abs().max()) < 5e-6


def test_generator_against_reference(dev, lib_path):
    z = np.load(os.path.join(GOLDEN, "ref_enhancer.npz"))
    gen = _generator(dev)
    mel, f0, ri = GC.nsf_inputs()
    audio = gen(mel.to(dev), f0.to(dev), rand_ini=ri[0])
    assert audio.shape == (1, 1, GC.NSF_L * UPP)
    want = torch.from_numpy(z["audio"])
    err = rms(audio[0, 0].cpu() - want)
    assert err < 2e-5 * max(rms(want), 1e-3) + 1e-6, (err, rms(want))
    assert float((audio[0, 0].cpu() - want).abs().max()) < 1e-4


@pytest.mark.parametrize("L", [1, 2, 64, 173])
def test_generator_against_oracle_lengths(dev, lib_path, L):
    gen = _generator(dev)
    mel, f0, ri = GC.nsf_inputs(L=L, seed=77 + L)
    want = OE.generator(GC.nsf_state_dict(), GC.NSF_CONFIG, mel, f0, ri)
    got = gen(mel.to(dev), f0.to(dev), rand_ini=ri[0])
    assert got.shape == want.shape
    assert float((got.cpu() - want).abs().max()) < 1e-4, L


@pytest.mark.parametrize("math", ["split_bf16", "fp32"])
def test_generator_wide_stages(ctx, dev, lib_path, math):
    """A generator whose first stages have 128 and 64 channels (the shipped one has 256 / 128 / 64): with split-bf16
    products those stages pass their activations in the split operand layout from convolution to convolution (weights
    converted at load) - against the oracle generator (= the reference's, tests/golden/ref_enhancer.npz pins it)."""
    import hipddsp
    from enhancer import AttrDict, Generator
    cfg = dict(GC.NSF_CONFIG, upsample_rates=[4, 2, 2], upsample_kernel_sizes=[8, 4, 4], upsample_initial_channel=256, num_mels=32)
    sd = GC.nsf_state_dict(cfg, seed=77)
    mel, f0, ri = GC.nsf_inputs(cfg, L=37, seed=78)
    want = OE.generator(sd, cfg, mel, f0, ri)
    gen = Generator(AttrDict(cfg), sd)
    ctx.set_math(hipddsp.MATH_SPLIT_BF16 if math == "split_bf16" else hipddsp.MATH_FP32)
    try:
        got = gen(mel.to(dev), f0.to(dev), rand_ini=ri[0])
    finally:
        ctx.set_math(hipddsp.MATH_SPLIT_BF16)
    assert got.shape == want.shape
    err = float((got.cpu() - want).abs().max())
    assert err < (3e-4 if math == "split_bf16" else 1e-4), err


def test_conv1d_building_block(ctx, dev):
    """`ddsp_conv1d` alone against torch's conv1d (fp64): kernel sizes 1..11, dilations, leaky-ReLU on load, residual."""
    g = torch.Generator().manual_seed(5)
    for (T, Cin, Cout, k, d, slope, res) in [(50, 16, 24, 3, 1, 1.0, False), (333, 32, 32, 7, 3, 0.1, True),
                                             (129, 64, 8, 11, 5, 0.1, False), (7, 4, 4, 1, 1, 0.5, True),
                                             (2000, 128, 64, 7, 1, 1.0, False),
                                             # the narrow stages' own kernel (Cin == Cout in {16, 32}): windows of 64 frames
                                             (4099, 16, 16, 11, 5, 0.1, True), (100, 16, 16, 3, 1, 1.0, False), (64, 16, 16, 1, 1, 0.1, False),
                                             (1000, 32, 32, 11, 3, 1.0, True), (31, 32, 32, 3, 5, 0.1, False)]:
        x = torch.randn(T, Cin, generator=g)
        w = torch.randn(Cout, Cin, k, generator=g) / np.sqrt(Cin * k)
        b = torch.randn(Cout, generator=g)
        r = torch.randn(T, Cout, generator=g) if res else None
        want = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x.double(), slope).t()[None], w.double(), b.double(),
                                          dilation=d, padding=(k * d - d) // 2)[0].t()
        if res:
            want = want + r.double()
        wp = w.permute(0, 2, 1).reshape(Cout, -1).contiguous()
        got = ctx.conv1d(x.to(dev), wp.to(dev), b.to(dev), k, d, slope, residual=None if r is None else r.to(dev))
        # (split-bf16 products, the context's default: the LDS-DMA kernel when no activation on load is asked and
        # Cin % 32 == 0, the narrow kernel with bf16 fragments for 32 -> 32 channels; fp32 products elsewhere)
        split = (slope == 1.0 and Cin % 32 == 0) or Cin == Cout == 32
        assert float((got.cpu().double() - want).abs().max()) < (6e-5 if split else 2e-5), (T, Cin, Cout, k, d)
    with pytest.raises(ValueError):
        ctx.conv1d(x.to(dev), wp.to(dev), b.to(dev), 2, 1, 1.0)          # even tap counts are not "same"-paddable


@pytest.mark.parametrize("math,C", [("split_bf16", 16), ("fp32", 16), ("split_bf16", 32)])
def test_conv1d_residual_pair(ctx, dev, math, C):
    """`ddsp_conv1d_pair` (one ResBlock1 pair of a narrow stage in one launch: x in, x out, activations on load, c1's output
    kept in the LDS) against torch's conv1d in fp64 and against the two-launch path: every kernel size / dilation the shipped
    generator has, signal lengths around the window sizes (c2's zero padding of c1's OUTPUT at both ends of the signal is the
    case a fused kernel gets wrong), 16 channels in both arithmetics (fp32 MFMA / three bf16 products per fp32 product), 32
    channels in split arithmetic (with fp32 products the 32-channel stage stays on two GEMM launches)."""
    import hipddsp
    F = torch.nn.functional
    g = torch.Generator().manual_seed(11)
    ctx.set_math(hipddsp.MATH_SPLIT_BF16 if math == "split_bf16" else hipddsp.MATH_FP32)
    try:
        for (T, k, d) in [(1, 3, 1), (5, 11, 5), (63, 3, 3), (64, 7, 5), (65, 11, 1), (86, 11, 3), (87, 11, 3), (130, 11, 3),
                          (4099, 11, 5), (20000, 7, 3), (777, 3, 5), (90, 3, 1), (91, 1, 1), (22, 11, 5), (23, 11, 5), (54, 11, 1),
                          (55, 11, 1), (3000, 7, 1), (3001, 3, 3)]:
            assert ctx.conv1d_pair_supported(C, k, d)
            x = torch.randn(T, C, generator=g)
            w1, w2 = (torch.randn(C, C, k, generator=g) / np.sqrt(C * k) for _ in range(2))
            b1, b2 = torch.randn(C, generator=g), torch.randn(C, generator=g)
            xa = F.leaky_relu(x.double(), 0.1).t()[None]
            xt = F.conv1d(xa, w1.double(), b1.double(), dilation=d, padding=(k * d - d) // 2)
            want = (F.conv1d(F.leaky_relu(xt, 0.1), w2.double(), b2.double(), padding=(k - 1) // 2)[0].t() + x.double())
            pk = lambda w: w.permute(0, 2, 1).reshape(C, -1).contiguous().to(dev)
            xd = x.to(dev)
            out, act = ctx.conv1d_pair(xd, pk(w1), b1.to(dev), pk(w2), b2.to(dev), k, d, 0.1, want_act=True)
            assert float((out.cpu().double() - want).abs().max()) < (4e-5 if math == "split_bf16" else 2e-5), (T, k, d)
            assert torch.equal(act, F.leaky_relu(out, 0.1))
            _, mid = ctx.conv1d(xd, pk(w1), b1.to(dev), k, d, 0.1, want_out=False, act_slope=0.1)
            two, _ = ctx.conv1d(mid, pk(w2), b2.to(dev), k, 1, 1.0, residual=xd, act_slope=0.1)
            assert float((out - two).abs().max()) < (6e-5 if math == "split_bf16" else 2e-6), (T, k, d)
            only_act = ctx.conv1d_pair(xd, pk(w1), b1.to(dev), pk(w2), b2.to(dev), k, d, 0.1, want_out=False, want_act=True)
            assert only_act[0] is None and torch.equal(only_act[1], act)
        assert not ctx.conv1d_pair_supported(C, 4, 1) and not ctx.conv1d_pair_supported(C, 13, 1) and not ctx.conv1d_pair_supported(64, 3, 1)
        with pytest.raises(ValueError):
            ctx.conv1d_pair(xd, pk(w1), b1.to(dev), pk(w2), b2.to(dev), 4, 1, 0.1)
        if C == 32:
            ctx.set_math(hipddsp.MATH_FP32)
            assert not ctx.conv1d_pair_supported(32, 3, 1)
    finally:
        ctx.set_math(hipddsp.MATH_SPLIT_BF16)


@pytest.mark.parametrize("math", ["split_bf16", "fp32"])
def test_conv1d_on_the_dma_kernel(ctx, dev, math):
    """Inputs that need no activation on load (in_slope = 1, Cin % 32 == 0) run on the LDS-DMA GEMM with one row pointer per
    tap: dilations, taps that fall off both ends, row counts that are not tile multiples, wide outputs (the transposed
    convolutions' 3-tap form), raw and activated outputs - in both product arithmetics of the context."""
    import hipddsp
    g = torch.Generator().manual_seed(6)
    tol = 6e-5 if math == "split_bf16" else 2e-5
    ctx.set_math(hipddsp.MATH_SPLIT_BF16 if math == "split_bf16" else hipddsp.MATH_FP32)
    try:
        for (T, Cin, Cout, k, d, res) in [(2000, 128, 64, 7, 1, False), (777, 32, 32, 11, 5, True), (130, 64, 64, 3, 3, True),
                                          (5000, 256, 256, 11, 5, True), (61, 256, 1024, 3, 1, True), (9, 32, 16, 7, 5, False)]:
            x = torch.randn(T, Cin, generator=g)
            w = torch.randn(Cout, Cin, k, generator=g) / np.sqrt(Cin * k)
            b = torch.randn(Cout, generator=g)
            r = torch.randn(T, Cout, generator=g) if res else None
            want = torch.nn.functional.conv1d(x.double().t()[None], w.double(), b.double(), dilation=d, padding=(k * d - d) // 2)[0].t()
            if res:
                want = want + r.double()
            wp = w.permute(0, 2, 1).reshape(Cout, -1).contiguous()
            rr = None if r is None else r.to(dev)
            got, act = ctx.conv1d(x.to(dev), wp.to(dev), b.to(dev), k, d, 1.0, residual=rr, act_slope=0.1)
            assert float((got.cpu().double() - want).abs().max()) < tol, (T, Cin, Cout, k, d)
            assert torch.equal(act, torch.nn.functional.leaky_relu(got, 0.1))
            none, act2 = ctx.conv1d(x.to(dev), wp.to(dev), b.to(dev), k, d, 1.0, residual=rr, want_out=False, act_slope=0.1)
            assert none is None and torch.equal(act2, act)
            assert torch.equal(ctx.conv1d(x.to(dev), wp.to(dev), b.to(dev), k, d, 1.0, residual=rr), got)
    finally:
        ctx.set_math(hipddsp.MATH_SPLIT_BF16)


def test_log_mel_against_oracle(dev, lib_path):
    """`STFT.get_mel` (device) against the oracle's torch.stft formulation with the same (restated) mel filter bank."""
    from enhancer import STFT, mel_filterbank
    h = GC.NSF_CONFIG
    st = STFT(h["sampling_rate"], h["num_mels"], h["n_fft"], h["win_size"], h["hop_size"], h["fmin"], h["fmax"])
    basis = torch.from_numpy(mel_filterbank(h["sampling_rate"], h["n_fft"], h["num_mels"], h["fmin"], h["fmax"]))
    # the filter bank has the properties librosa's has: non-negative triangles, one peak per band, Slaney area normalisation
    assert basis.shape == (h["num_mels"], h["n_fft"] // 2 + 1) and float(basis.min()) >= 0
    assert all(int((basis[i] > 0).sum()) >= 1 for i in range(h["num_mels"]))
    for T in (2000, 4096, 100):
        y = GC.nsf_audio(T)
        want = OE.log_mel(y, h, basis)
        got = st.get_mel(y.to(dev))
        assert got.shape == want.shape == (1, h["num_mels"], want.shape[-1])
        assert float((got.cpu() - want).abs().max()) < 2e-4, T


def test_enhancer_end_to_end(dev, lib_path, tmp_path):
    """`Enhancer('nsf-hifigan', ckpt, device).enhance(...)` from a checkpoint + config.json on disk (the reference's file
    layout), with and without the adaptive-key resampling and the silent front, against the same pipeline assembled from the
    oracle's parts on the CPU."""
    from enhancer import Enhancer, mel_filterbank
    h = dict(GC.NSF_CONFIG)
    with open(tmp_path / "config.json", "w") as fh:
        json.dump(h, fh)
    torch.save({"generator": GC.nsf_state_dict()}, tmp_path / "model")
    enh = Enhancer("nsf-hifigan", str(tmp_path / "model"), device=dev)
    assert enh.enhancer_sample_rate == 44100 and enh.enhancer_hop_size == 32
    sd = GC.nsf_state_dict()
    basis = torch.from_numpy(mel_filterbank(h["sampling_rate"], h["n_fft"], h["num_mels"], h["fmin"], h["fmax"]))
    ri = torch.tensor([0.0, 0.3, 0.7, 0.1, 0.9, 0.5, 0.2, 0.8, 0.4])

    def cpu_pipeline(audio, sr, f0, hop, key, silence_front):
        start = int(silence_front * sr / hop)
        rsf = start * hop / sr
        audio = audio[:, int(np.round(rsf * sr)):]
        f0 = f0[:, start:, :]
        if key == "auto":
            key = max(0, np.ceil(12 * np.log2(float(torch.max(f0) / 760))))
        fac = 2 ** (-float(key) / 12)
        asr = 100 * int(np.round(44100 / fac / 100))
        rf = 44100 / asr
        a = audio if sr == asr else OR.resample(audio, sr, asr, 128)
        n_frames = int(a.size(-1) // 32 + 1)
        f = f0.squeeze(0).squeeze(-1).numpy().copy() * rf
        t0 = (hop / sr) * np.arange(len(f)) / rf
        t1 = (32 / 44100) * np.arange(n_frames)
        fr = torch.from_numpy(np.interp(t1, t0, f, left=f[0], right=f[-1])).unsqueeze(0).float()
        mel = OE.log_mel(a, h, basis)
        out = OE.generator(sd, h, mel, fr[:, :mel.size(-1)], ri[None]).reshape(1, -1)
        out = OR.resample(out, asr, 44100, 128) if fac != 0 and asr != 44100 else out
        if start > 0:
            out = torch.nn.functional.pad(out, (int(np.round(44100 * rsf)), 0))
        return out

    T, hop = 4096, 512
    audio = GC.nsf_audio(T)
    f0 = torch.full((1, T // hop, 1), 300.0)
    # a track that is not constant (the re-timing interpolates) and peaks at 1000 Hz: adaptive_key='auto' (enhancer.py:34-38)
    # then picks ceil(12 log2(1000 / 760)) = 5 semitones; the flat 300 Hz track leaves 'auto' at key 0
    f0_var = (300.0 + 700.0 * torch.sin(torch.arange(T // hop) / 2.5) ** 2).reshape(1, -1, 1)
    assert float(f0_var.max()) > 990
    for key, sf, track in ((0, 0, f0), (4, 0, f0), (0, 0.03, f0), ("auto", 0, f0), ("auto", 0, f0_var), (2, 0.03, f0_var)):
        got, sr_o = enh.enhance(audio.to(dev), 44100, track.to(dev), hop, adaptive_key=key, silence_front=sf, rand_ini=ri)
        want = cpu_pipeline(audio, 44100, track, hop, key, sf)
        assert sr_o == 44100 and got.shape == want.shape, (key, sf, got.shape, want.shape)
        assert float((got.cpu() - want).abs().max()) < 5e-4, (key, sf)
    with pytest.raises(ValueError):
        enh.enhance(audio.to(dev), 44100, f0.to(dev), hop, adaptive_key="automatic")
    with pytest.raises(ValueError):
        Enhancer("other", str(tmp_path / "model"))


def test_retime_f0_against_numpy(ctx, dev):
    """`ddsp_retime_f0` against the reference's host expression (enhancer.py:56-62): scaled fp32 values, fp64 knots, numpy.interp
    with held ends - for up-, down- and same-rate grids, a single-frame track and targets beyond both ends."""
    rng = np.random.Generator(np.random.PCG64(31))
    for n, hop, sr, factor, hop_e, sr_e, n_dst in [(173, 512, 44100, 1.0, 512, 44100, 173), (87, 512, 44100, 44100 / 55600, 512, 44100, 120),
                                                   (40, 441, 44100, 1.26, 32, 44100, 700), (1, 512, 44100, 0.9, 512, 44100, 5),
                                                   (500, 160, 16000, 0.7071, 512, 44100, 431)]:
        f0 = (rng.uniform(60, 900, n)).astype(np.float32)
        f0[n // 3: n // 3 + 2] = 0.0
        vals = f0.copy()
        vals *= factor
        t_org = (hop / sr) * np.arange(n) / factor
        t_dst = (hop_e / sr_e) * np.arange(n_dst)
        want = np.interp(t_dst, t_org, vals, left=vals[0], right=vals[-1]).astype(np.float32)
        got = ctx.retime_f0(torch.from_numpy(f0).to(dev), hop / sr, factor, factor, hop_e / sr_e, n_dst).cpu().numpy()
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (n, np.abs(got - want).max())


@pytest.mark.parametrize("math", ["split_bf16", "fp32"])
def test_generator_shipped_geometry(ctx, dev, lib_path, math):
    """The geometry that is benchmarked and shipped with the reference's pretrained enhancer (nsf_hifigan config: 512 initial
    channels, upsample rates 8-8-2-2-2 with kernels 16-16-4-4-4, 128 mels, resblock kernels 3 / 7 / 11, dilations 1 / 3 / 5;
    nsf_hifigan/models.py:219-276) on 24 frames against the oracle generator, in both product arithmetics.  This is the
    configuration that selects the 128x128 and 64x128 tiles of the wide stages and the 32-channel bf16-fragment kernel."""
    import hipddsp
    from enhancer import AttrDict, Generator
    cfg = dict(GC.NSF_CONFIG, upsample_rates=[8, 8, 2, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4, 4], upsample_initial_channel=512,
               num_mels=128, hop_size=512, n_fft=2048, win_size=2048)
    sd = GC.nsf_state_dict(cfg, seed=91)
    mel, f0, ri = GC.nsf_inputs(cfg, L=24, seed=92)
    want = OE.generator(sd, cfg, mel, f0, ri)
    gen = Generator(AttrDict(cfg), sd)
    ctx.set_math(hipddsp.MATH_SPLIT_BF16 if math == "split_bf16" else hipddsp.MATH_FP32)
    try:
        got = gen(mel.to(dev), f0.to(dev), rand_ini=ri[0])
    finally:
        ctx.set_math(hipddsp.MATH_SPLIT_BF16)
    assert got.shape == want.shape == (1, 1, 24 * 512)
    scale = float(want.abs().max())
    err = float((got.cpu() - want).abs().max())
    assert scale > 1e-3 and err < (3e-4 if math == "split_bf16" else 1e-4) * max(1.0, scale), (err, scale)
    assert rms(got.cpu() - want) < (3e-5 if math == "split_bf16" else 1e-5) * max(1.0, rms(want) / 0.1), (rms(got.cpu() - want), rms(want))


def test_noise_conv_building_block(ctx, dev):
    """`ddsp_nsf_noise_conv` (the strided 1-channel convolutions that bring the source signal to each stage's rate) against
    torch's conv1d in fp64: the shipped geometries (K = 2 s, stride s, padding s / 2; K = 1 at the last stage), channel counts
    that do and do not divide 256, outputs that end inside a block."""
    g = torch.Generator().manual_seed(9)
    for (C, K, s, pad, T_out) in [(256, 128, 64, 32, 77), (128, 16, 8, 4, 1000), (16, 1, 1, 0, 4099), (24, 4, 2, 1, 130), (300, 8, 4, 2, 50)]:
        T_src = T_out * s
        src = torch.randn(T_src, generator=g)
        w = torch.randn(C, K, generator=g) / np.sqrt(K)
        b = torch.randn(C, generator=g)
        want = torch.nn.functional.conv1d(src.double()[None, None], w.double()[:, None], b.double(), stride=s, padding=pad)[0].t()[:T_out]
        got = ctx.nsf_noise_conv(src.to(dev), w.to(dev), b.to(dev), K, s, pad, T_out)
        assert got.shape == (T_out, C)
        assert float((got.cpu().double() - want).abs().max()) < 2e-5, (C, K, s)


def _unsplit(t):
    """Decode the split operand layout (groups of 8 floats = 8 bf16 high parts | 8 bf16 remainders) back to fp32."""
    b = t.contiguous().view(torch.bfloat16).reshape(*t.shape[:-1], t.shape[-1] // 8, 16)
    return (b[..., :8].float() + b[..., 8:].float()).reshape(t.shape)


def test_conv1d_split_operands_and_large_tiles(ctx, dev):
    """The generator's wide stages at sizes where the convolution picks its larger tiles (128x128 for 128-channel multiples
    from 256 workgroups, 128x64 from 512), with the operands as the generator passes them: input and weight in the split
    layout, raw output in fp32, activated output in the split layout (decoded here)."""
    import hipddsp
    g = torch.Generator().manual_seed(12)
    for (T, C, k, d) in [(40000, 128, 7, 3), (70000, 64, 3, 1), (3000, 256, 11, 5), (33000, 128, 3, 5)]:
        x = torch.randn(T, C, generator=g)
        w = torch.randn(C, C, k, generator=g) / np.sqrt(C * k)
        b = torch.randn(C, generator=g)
        r = torch.randn(T, C, generator=g)
        wp = w.permute(0, 2, 1).reshape(C, -1).contiguous()
        xs, ws = hipddsp.presplit(x), hipddsp.presplit(wp)
        # what the kernel multiplies: the two-piece values of x and w (their fp32 originals differ by ~4e-6 relative)
        want = torch.nn.functional.conv1d(_unsplit(xs).double().t()[None], _unsplit(ws).reshape(C, k, C).permute(0, 2, 1).double(),
                                          b.double(), dilation=d, padding=(k * d - d) // 2)[0].t() + r.double()
        out, act = ctx.conv1d(xs.to(dev), wp.to(dev), b.to(dev), k, d, 1.0, residual=r.to(dev), act_slope=0.1, w_split=ws.to(dev),
                              x_split=True, act_split=True)
        assert float((out.cpu().double() - want).abs().max()) < 3e-5, (T, C, k, d)
        act_want = torch.nn.functional.leaky_relu(out.cpu(), 0.1)
        assert float((_unsplit(act.cpu()) - act_want).abs().max()) < 2e-5 * float(act_want.abs().max())
        assert torch.equal(act.cpu(), hipddsp.presplit(act_want))          # the same bits as the host-side conversion
