"""The fused Performer attention kernels (a4, inference) against an fp64 evaluation of ddsp/pcmer.py:69-77,123-159 on the
same device tensors: the fp32-MFMA kernels and the split-bf16 kernels (six piece products for the feature projections,
which enter an exponential; three for the context products)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H, DH, NF = 8, 64, 266


def reference(q, k, v, P, B, Fr):
    """fp64 restatement of `softmax_kernel` (query / key branches) + `linear_attention` (non-causal)."""
    q, k, v, P = (t.double() for t in (q, k, v, P))
    sp = lambda x: x.reshape(B, Fr, H, DH).permute(0, 2, 1, 3)          # (B, H, Fr, 64)
    q, k, v = sp(q), sp(k), sp(v)
    dn, ratio, eps = DH ** -0.25, NF ** -0.5, 1e-4
    dq, dk = dn * q @ P.t(), dn * k @ P.t()                               # (B, H, Fr, 266)
    diag_q = (q ** 2).sum(-1, keepdim=True) * 0.5 * dn ** 2
    diag_k = (k ** 2).sum(-1, keepdim=True) * 0.5 * dn ** 2
    qf = ratio * (torch.exp(dq - diag_q - dq.amax(dim=-1, keepdim=True)) + eps)
    kf = ratio * torch.exp(dk - diag_k + eps)
    ks = kf.sum(dim=-2)
    dinv = 1.0 / (torch.einsum("bhnd,bhd->bhn", qf, ks) + 1e-8)
    ctx = torch.einsum("bhnd,bhne->bhde", kf, v)
    out = torch.einsum("bhde,bhnd,bhn->bhne", ctx, qf, dinv)
    return out.permute(0, 2, 1, 3).reshape(B * Fr, H * DH), kf


def make(seed, B, Fr, scale, dev):
    g = torch.Generator().manual_seed(seed)
    mk = lambda s: (torch.randn(B * Fr, H * DH, generator=g) * s).to(dev)
    # a Gaussian orthogonal random matrix like pcmer.gaussian_orthogonal_random_matrix: rows of norm ~ 8
    blocks = []
    for _ in range(5):
        qm, _ = torch.linalg.qr(torch.randn(DH, DH, generator=g))
        blocks.append(qm.t())
    P = torch.cat(blocks)[:NF] * torch.randn(NF, DH, generator=g).norm(dim=1, keepdim=True)
    return mk(scale), mk(scale), mk(1.0), P.contiguous().to(dev)


# (32, 400) and (32, 193): 13 and 7 frame tiles - the fused split kernel's query phase then runs 3 and 2 passes of six frame
# tiles over the context it keeps in the LDS, the last pass with idle waves; (64, 200): 7 tiles, ragged
# (1, 87), (8, 172), (2, 33), (3, 1): few work items - the fp32 query kernel then splits the feature range over the four waves
# of a workgroup (performer_q_split_kernel) and combines them at their common row maximum
@pytest.mark.parametrize("B,Fr", [(32, 172), (33, 87), (40, 1), (32, 33), (64, 200), (32, 400), (32, 193), (1, 87), (8, 172),
                                  (2, 33), (3, 1)])
@pytest.mark.parametrize("math", [0, 3])
def test_attention_against_fp64(ctx, dev, B, Fr, math):
    q, k, v, P = make(B + Fr, B, Fr, 1.0, dev)
    want, _ = reference(q, k, v, P, B, Fr)
    got = ctx.performer_attention(q, k, v, P, B, Fr, math=math)
    err = float((got.double() - want).norm() / want.norm())
    assert torch.isfinite(got).all()
    assert err < (2e-6 if math == 0 else 1e-5), err


def causal_reference(q, k, v, P, B, Fr):
    """fp64 restatement of `softmax_kernel` + `causal_linear_attention` (pcmer.py:170-188; `CausalDotProduct` by its
    definition: out_n = q'_n . sum_{m<=n} k'_m (x) v_m)."""
    q, k, v, P = (t.double() for t in (q, k, v, P))
    sp = lambda x: x.reshape(B, Fr, H, DH).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    dn, ratio, eps = DH ** -0.25, NF ** -0.5, 1e-4
    dq, dk = dn * q @ P.t(), dn * k @ P.t()
    qf = ratio * (torch.exp(dq - (q ** 2).sum(-1, keepdim=True) * 0.5 * dn ** 2 - dq.amax(dim=-1, keepdim=True)) + eps)
    kf = ratio * torch.exp(dk - (k ** 2).sum(-1, keepdim=True) * 0.5 * dn ** 2 + eps)
    dinv = 1.0 / torch.einsum("bhnd,bhnd->bhn", qf, kf.cumsum(dim=-2) + 1e-6)
    A = torch.tril(qf @ kf.transpose(-1, -2))                             # (B, H, Fr, Fr)
    out = (A @ v) * dinv.unsqueeze(-1)
    return out.permute(0, 2, 1, 3).reshape(B * Fr, H * DH)


# frame counts around the 16-frame chunks of the kernel: one partial chunk, exact multiples, the bench's 172, long clips
@pytest.mark.parametrize("B,Fr", [(1, 1), (2, 15), (3, 16), (2, 17), (1, 87), (8, 172), (4, 400), (64, 32)])
def test_causal_attention_against_fp64(ctx, dev, B, Fr):
    import hipddsp
    q, k, v, P = make(3 * B + Fr, B, Fr, 1.0, dev)
    want = causal_reference(q, k, v, P, B, Fr)
    got = ctx.performer_attention(q, k, v, P, B, Fr, math=hipddsp.ATTENTION_CAUSAL)
    assert torch.isfinite(got).all()
    err = float((got.double() - want).norm() / want.norm())
    assert err < 3e-6, err
    # causality itself: changing the last frames leaves every earlier output bit for bit as it was
    if Fr > 20:
        q2, k2, v2 = q.clone(), k.clone(), v.clone()
        rows = torch.arange(B, device=dev)[:, None] * Fr + torch.arange(Fr - 5, Fr, device=dev)[None]
        for t in (q2, k2, v2):
            t[rows.reshape(-1)] += 1.0
        got2 = ctx.performer_attention(q2, k2, v2, P, B, Fr, math=hipddsp.ATTENTION_CAUSAL)
        keep = torch.ones(B * Fr, dtype=torch.bool, device=dev)
        keep[rows.reshape(-1)] = False
        assert torch.equal(got[keep], got2[keep])


@pytest.mark.parametrize("math", [0, 3])
def test_attention_large_exponents(ctx, dev, math):
    """Keys and queries scaled so that the projected exponents reach |dn x.P| ~ 20: the error of the projection product
    enters exp() undamped.  With six piece products the split kernels stay in the fp32 class (a three-product split
    would be ~1e-4 here)."""
    B, Fr = 32, 96
    q, k, v, P = make(7, B, Fr, 6.0, dev)
    want, kf = reference(q, k, v, P, B, Fr)
    big = float((DH ** -0.25 * k.double().reshape(B, Fr, H, DH) @ P.double().t()).abs().max())
    assert big > 15.0, big
    got = ctx.performer_attention(q, k, v, P, B, Fr, math=math)
    err = float((got.double() - want).norm() / want.norm())
    assert err < (5e-6 if math == 0 else 2e-5), (err, big)


def test_attention_modes_agree_in_the_model(dev, lib_path):
    """ddsp_unit2ctrl_fwd picks the split kernels from 32 utterances on: the control matrix with them equals the one with
    the fp32 kernels to the accuracy of the split GEMMs around them."""
    import hipddsp, synthetic
    model, cfg = synthetic.build_model("CombSub", seed=4, device=dev)
    B, Fr = 32, 100
    inp = {k: v.to(dev) for k, v in synthetic.make_inputs(17, B, Fr, with_noise=False).items()}
    c = hipddsp.context_for(dev)
    outs = {}
    try:
        for mode in (hipddsp.MATH_FP32, hipddsp.MATH_SPLIT_BF16):
            c.set_math(mode)
            ps = c.phase_scan(inp["f0"], 512, 44100)
            with torch.no_grad():
                outs[mode] = model.unit2ctrl.forward_flat(inp["units"], inp["f0"], ps["phase_frames"], inp["volume"],
                                                          inp["spk_id"], None)
    finally:
        c.set_math(hipddsp.MATH_SPLIT_BF16)
    a, b = outs[hipddsp.MATH_FP32], outs[hipddsp.MATH_SPLIT_BF16]
    assert float((a - b).norm() / a.norm()) < 5e-5


def test_fused_kernel_against_the_kernel_pair(ctx, dev, monkeypatch):
    """Round 3: key and query side of the split attention in one kernel per (utterance, head).  Its feature tiles are summed
    in another order (even / odd tiles on two waves, combined at their common row maximum), so it equals the round-2 kernel
    pair (ablation code 100 = the pair, nothing switched off) to rounding, not to the bit."""
    B, Fr = 48, 172
    q, k, v, P = make(5, B, Fr, 1.5, dev)
    fused = ctx.performer_attention(q, k, v, P, B, Fr, math=3)
    pair = ctx.performer_attention(q, k, v, P, B, Fr, math=100)
    assert float((fused - pair).norm() / pair.norm()) < 2e-6
    for _ in range(3):
        assert torch.equal(ctx.performer_attention(q, k, v, P, B, Fr, math=3), fused)      # run-to-run bit stability
