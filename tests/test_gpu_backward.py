"""Adjoints of the DSP stage (training path) against PyTorch autograd through the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import dsp as O

pytestmark = pytest.mark.gpu
SR, HOP = 44100, 512


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("n,Fr,B", [(510, 6, 2), (1022, 7, 2), (1022, 1, 1), (64, 3, 1), (510, 172, 1), (2046, 5, 1)])
def test_ltv_fir_adjoints(ctx, dev, n, Fr, B):
    r = _rng(n + Fr)
    x = torch.from_numpy(r.uniform(-1, 1, (B, Fr * HOP)).astype(np.float32)).requires_grad_(True)
    ir = torch.from_numpy((r.standard_normal((B, Fr, n)) / np.sqrt(n)).astype(np.float32)).requires_grad_(True)
    g = torch.from_numpy(r.standard_normal((B, Fr * HOP)).astype(np.float32))
    y = O.ltv_fir_fft(x, ir)
    (y * g).sum().backward()
    dx, dir_ = ctx.ltv_fir_bwd(x.detach().to(dev), ir.detach().to(dev), g.to(dev), B, Fr, HOP)
    assert _rel(dx.cpu(), x.grad) < 2e-5, _rel(dx.cpu(), x.grad)
    assert _rel(dir_.cpu().reshape(B, Fr, n), ir.grad) < 2e-5, _rel(dir_.cpu().reshape(B, Fr, n), ir.grad)
    # adjoint identity on the device alone: <F(x), g> == <x, F^T(g)>
    yg, _ = ctx.ltv_fir(x.detach().to(dev), ir.detach().to(dev), B, Fr, HOP)
    lhs = float((yg.double() * g.to(dev).double()).sum())
    rhs = float((x.detach().to(dev).double() * dx.double()).sum())
    assert abs(lhs - rhs) < 1e-5 * float(yg.double().norm() * g.double().norm())


def test_ltv_fir_filter_grad_with_noise_excitation(ctx, dev):
    B, Fr, n = 2, 5, 510
    r = _rng(3)
    u = torch.from_numpy(r.random((B, Fr * HOP), dtype=np.float32))
    ir = torch.from_numpy((r.standard_normal((B, Fr, n)) / np.sqrt(n)).astype(np.float32)).requires_grad_(True)
    g = torch.from_numpy(r.standard_normal((B, Fr * HOP)).astype(np.float32))
    (O.ltv_fir_fft(u * 2 - 1, ir) * g).sum().backward()
    _, d1 = ctx.ltv_fir_bwd(u.to(dev), ir.detach().to(dev), g.to(dev), B, Fr, HOP, excitation=1, want_d_audio=False)
    assert _rel(d1.cpu().reshape(B, Fr, n), ir.grad) < 2e-5
    # generated excitation: gradient is consistent with the forward of the same seed (finite-difference free check:
    # d/d ir <F_seed(ir), g> is linear in ir, so F(ir1) - F(ir0) paired with g equals <d_ir, ir1 - ir0>)
    ir0 = ir.detach().to(dev)
    ir1 = ir0 + 0.1 * torch.randn_like(ir0)
    y0, _ = ctx.ltv_fir(None, ir0, B, Fr, HOP, excitation=2, noise_seed=9)
    y1, _ = ctx.ltv_fir(None, ir1, B, Fr, HOP, excitation=2, noise_seed=9)
    _, d2 = ctx.ltv_fir_bwd(None, ir0, g.to(dev), B, Fr, HOP, excitation=2, noise_seed=9, want_d_audio=False)
    lhs = float(((y1 - y0).double() * g.to(dev).double()).sum())
    rhs = float((d2.reshape(B, Fr, n).double() * (ir1 - ir0).double()).sum())
    assert abs(lhs - rhs) < 1e-4 * (abs(lhs) + 1.0)


def _resp(mode, c):
    if mode == 0:
        return torch.exp(1.j * torch.cumsum(np.pi * torch.tanh(c), dim=-1))
    m = torch.exp(c) if mode == 1 else torch.exp(c) / 128
    return torch.complex(m, torch.zeros_like(m))


@pytest.mark.parametrize("mode,n_mag", [(0, 256), (1, 512), (2, 256), (0, 65), (2, 513)])
def test_fir_from_ctrl_adjoint(ctx, dev, mode, n_mag):
    B, Fr = 2, 7
    r = _rng(10 + mode)
    W = n_mag + 16
    c = torch.from_numpy((r.standard_normal((B, Fr, W)) * 0.5).astype(np.float32))
    f0f = torch.from_numpy(r.uniform(65, 800, (B, Fr, 1)).astype(np.float32))
    sub = c[..., 8:8 + n_mag].clone().requires_grad_(True)
    hw = 1.5 * SR / (f0f + 1e-3) if mode == 1 else None
    ir = O.fir_from_response(_resp(mode, sub), hann=(mode != 0), half_width=hw)
    g = torch.from_numpy(r.standard_normal(tuple(ir.shape)).astype(np.float32))
    (ir * g).sum().backward()
    d_ctrl = torch.full((B * Fr, W), 7.0, device=dev)
    ctx.fir_from_ctrl_bwd(mode, c.reshape(B * Fr, W).to(dev), 8, n_mag, B * Fr, SR, g.reshape(B * Fr, -1).to(dev).clone(),
                          d_ctrl, f0f.to(dev) if mode == 1 else None)
    got = d_ctrl.cpu().reshape(B, Fr, W)
    assert torch.all(got[..., :8] == 7.0) and torch.all(got[..., 8 + n_mag:] == 7.0)    # only its column block is written
    assert _rel(got[..., 8:8 + n_mag], sub.grad) < (2e-4 if mode == 0 else 2e-5), _rel(got[..., 8:8 + n_mag], sub.grad)


# ---- control network backward -------------------------------------------------------------------------
# (24, 172): 4128 rows - from 4033 rows on the prenet convs (forward recompute AND the d_t2 adjoint) run the LDS-DMA
# implicit-im2col GEMM instead of the register-staged one, and the N <= 256 layers the 4-wave 64x64 tile
# causal = True: the `c: true` network (causal taps in the three convolutions, causal linear attention): its backward is three
# sequential scans per (utterance, head) and shifted taps in the conv adjoints; the oracle restates the two third-party causal
# primitives from their definitions (oracle/ctrlnet.py), autograd differentiates that restatement
@pytest.mark.parametrize("B,Fr,spk_mode,causal", [(2, 12, "per_row", False), (3, 40, "broadcast", False), (2, 172, "mix", False),
                                                  (24, 172, "per_row", False), (2, 12, "per_row", True), (3, 47, "mix", True),
                                                  (24, 172, "broadcast", True)])
def test_unit2ctrl_parameter_gradients(dev, lib_path, B, Fr, spk_mode, causal):
    import synthetic
    from oracle import ctrlnet as OC
    model, cfg = synthetic.build_model("CombSub", seed=31)
    if causal:
        from ddsp.vocoder import CombSub
        cm = CombSub(44100, 512, cfg["n_mag_allpass"], cfg["n_mag_harmonic"], cfg["n_mag_noise"], cfg["n_unit"], cfg["n_spk"], c=True)
        cm.load_state_dict(model.state_dict(), strict=True)
        model = cm
    u2c = model.unit2ctrl
    inp = synthetic.make_inputs(100 + Fr, B, Fr, with_noise=False)
    r = _rng(Fr)
    phase = torch.from_numpy(r.uniform(-np.pi, np.pi, (B, Fr)).astype(np.float32))
    d_ctrl = torch.from_numpy(r.standard_normal((B, Fr, u2c.n_out)).astype(np.float32)) / (B * Fr)
    spk = inp["spk_id"] if spk_mode == "per_row" else inp["spk_id"][:1]
    if spk_mode == "per_row" and B > 1:
        spk[1] = spk[0]                                  # two utterances of one speaker: gradients must add up
    mix = {3: 0.5, 10: 0.2, 99: 0.3} if spk_mode == "mix" else None
    # oracle gradients by autograd
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "projection_matrix" not in k)
          for k, v in u2c.state_dict().items()}
    out = OC.unit2control(sd, inp["units"], inp["f0"], phase, inp["volume"], spk, mix, u2c.output_splits, return_flat=True,
                          causal=causal)
    (out * d_ctrl).sum().backward()
    model = model.to(dev)
    grads = model.unit2ctrl.backward_flat(inp["units"].to(dev), inp["f0"].to(dev), phase.to(dev), inp["volume"].to(dev),
                                          spk.to(dev), mix, d_ctrl.to(dev))
    by_name = {n: grads[p] for n, p in model.unit2ctrl.named_parameters()}
    assert len(by_name) == len([k for k, v in sd.items() if v.requires_grad])
    worst = []
    for name, g in by_name.items():
        want = sd[name].grad
        assert want is not None and g.shape == want.shape, name
        err = _rel(g.cpu(), want)
        worst.append((err, name))
        if name == "spk_embed.weight":
            used = want.abs().sum(1) > 0
            assert torch.all(g.cpu()[~used] == 0)
    worst.sort(reverse=True)
    assert worst[0][0] < 2e-3, worst[:5]
    assert sum(e for e, _ in worst) / len(worst) < 3e-4, worst[:5]
    # the training pair (ddsp_unit2ctrl_fwd_keep / ddsp_unit2ctrl_bwd_kept): the same forward with its activations left in a
    # caller-owned region, the same backward started from them - same bits as the call that recomputes (every
    # reduction of the backward pass has a fixed order)
    import hipddsp
    dargs = (inp["units"].to(dev), inp["f0"].to(dev), phase.to(dev), inp["volume"].to(dev), spk.to(dev), mix)
    ctrl, kept = model.unit2ctrl.forward_flat_keep(*dargs)
    assert _rel(ctrl.cpu(), out.detach()) < 2e-5
    grads_k = model.unit2ctrl.backward_flat(*dargs, d_ctrl.to(dev), kept=kept)
    for n, p in model.unit2ctrl.named_parameters():
        assert torch.equal(grads_k[p], grads[p]), n
    with pytest.raises(ValueError):                      # a region of the wrong size is refused, not overrun
        model.unit2ctrl.backward_flat(*dargs, d_ctrl.to(dev), kept=kept[: kept.numel() // 2])
    # fp32 products in the backward as well (ddsp_ctx_set_math(FP32): the round-1 kernels)
    c = hipddsp.context_for(dev)
    c.set_math(hipddsp.MATH_FP32)
    try:
        grads32 = model.unit2ctrl.backward_flat(*dargs, d_ctrl.to(dev))
    finally:
        c.set_math(hipddsp.MATH_SPLIT_BF16)
    errs = sorted(((_rel(grads32[p].cpu(), sd[n].grad), n) for n, p in model.unit2ctrl.named_parameters()), reverse=True)
    assert errs[0][0] < 2e-3 and sum(e for e, _ in errs) / len(errs) < 3e-4, errs[:5]
    # the two backward paths against each other: split arithmetic (batched weight-gradient kernel and the fused feature-map
    # adjoint of the attention, round 3) and fp32 products on the unfused chain share the forward, so they differ by the product
    # rounding of a linear map only
    cross = sorted(((_rel(grads[p].cpu(), grads32[p].cpu()), n) for n, p in model.unit2ctrl.named_parameters()), reverse=True)
    assert cross[0][0] < 1e-3, cross[:5]   # (measured: 2.1e-4 on to_q of the (2, 12) case, below 2e-4 elsewhere)
