"""The fp32 MFMA GEMM building block (all operand layouts, edge shapes, both staging pipelines) against fp64 matmul."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, seed, dev):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dev)


@pytest.mark.parametrize("M,N,K", [(1, 1, 4), (37, 29, 12), (200, 130, 100), (129, 257, 64), (1000, 266, 64), (300, 510, 257)])
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("tile", [0, 1, 5])
def test_register_staged_layouts(ctx, dev, M, N, K, a_kc, b_kc, tile):
    pad = lambda n: (n + 3) // 4 * 4
    # leading dimensions are padded to multiples of 4 floats (16-byte aligned rows), as every caller guarantees
    A_full = _mk((M, pad(K)) if a_kc else (K, pad(M)), 1, dev)
    B_full = _mk((N, pad(K)) if b_kc else (K, pad(N)), 2, dev)
    A = A_full[:, :K] if a_kc else A_full[:, :M]
    B = B_full[:, :K] if b_kc else B_full[:, :N]
    bias = _mk((N,), 3, dev)
    C = ctx.gemm(A, B, bias, a_k_contig=a_kc, b_k_contig=b_kc, tile=tile)
    Am = (A if a_kc else A.t()).double().cpu()
    Bm = (B.t() if b_kc else B).double().cpu()
    want = Am @ Bm + bias.double().cpu()
    scale = float(want.abs().max()) + 1e-6
    assert float((C.double().cpu() - want).abs().max()) < 2e-6 * scale * max(1.0, K ** 0.5 / 4)


# tiles: 0 auto, 10/14 128x64 (3/2 stages), 11/13 128x128 (3/2 stages), 12 256x128, 15/16 64x64 and 64x128 on 4 waves.
# (4100, 250, 96) and (8200, 1024, 64) make the AUTO choice pick the 4-wave 64x64 tile with ragged M and N, and the
# 2-stage 128x128 tile with a ragged last row block - the two configurations the Linear layers run at the bench size.
@pytest.mark.parametrize("M,N,K", [(128, 64, 32), (130, 70, 64), (11008 // 8, 512, 256), (1000, 250, 96), (64, 1536, 1024),
                                   (4100, 250, 96), (8200, 1024, 64)])
@pytest.mark.parametrize("tile", [0, 10, 11, 12, 13, 14, 15, 16])
def test_lds_dma_pipeline(ctx, dev, M, N, K, tile):
    A, B, bias = _mk((M, K), 4, dev), _mk((N, K), 5, dev), _mk((N,), 6, dev)
    C = ctx.gemm(A, B, bias, tile=tile)
    want = A.double().cpu() @ B.double().cpu().t() + bias.double().cpu()
    scale = float(want.abs().max())
    assert float((C.double().cpu() - want).abs().max()) < 2e-6 * scale * max(1.0, K ** 0.5 / 4)
    # run-to-run bit stability (no race between the DMA ring and the operand reads)
    for _ in range(3):
        assert torch.equal(ctx.gemm(A, B, bias, tile=tile), C)


def test_dma_requires_alignment(ctx, dev):
    A, B = _mk((64, 36), 7, dev), _mk((64, 36), 8, dev)
    with pytest.raises(ValueError):
        ctx.gemm(A, B, tile=10)        # K % 32 != 0


# ---- the wave-specialised kernel (csrc/gemm_ws.h): tiles 70+ ---------------------------------------------------------
def _presplit(X):
    """(rows, K) fp32 -> the (8 bf16 hi | 8 bf16 lo) operand layout with lo = 0, and the bf16-rounded values it encodes."""
    Xb = X.to(torch.bfloat16)
    hi = Xb.view(torch.int16).view(-1, X.shape[1] // 8, 8)
    return torch.cat([hi, torch.zeros_like(hi)], dim=2).reshape(X.shape[0], -1).view(torch.float32).contiguous(), Xb.float()


@pytest.mark.parametrize("M,N,K", [(11008, 1536, 256), (11008 // 4, 1024, 256), (1000, 256, 256), (300, 384, 512),
                                   (128 * 5 + 7, 1024, 768), (129, 128, 1024)])
def test_wave_specialised_kernel(ctx, dev, M, N, K):
    """Loader / product waves on an LDS ring, epilogue pieces riding in the next tile: fp32 and split-bf16 products against
    fp64 and BIT-IDENTICAL to the round-2 LDS-DMA kernel (same order of partial sums), the gated-pair epilogue, the residual
    epilogue whose operand the loader waves fetch into the LDS; ragged row counts, one to many tiles per workgroup."""
    A, B, bias = _mk((M, K), 14, dev), _mk((N, K), 15, dev), _mk((N,), 16, dev)
    ref = A.double() @ B.double().t() + bias.double()
    scale = float(ref.abs().max())
    c0 = ctx.gemm(A, B, bias, tile=70, variant=0)
    c3 = ctx.gemm(A, B, bias, tile=70, variant=3)
    assert float((c0.double() - ref).abs().max()) < 2e-6 * scale * max(1.0, K ** 0.5 / 4)
    assert float((c3.double() - ref).abs().max()) < 3e-5 * scale
    assert torch.equal(c0, ctx.gemm(A, B, bias, tile=13)) and torch.equal(c3, ctx.gemm(A, B, bias, tile=30))
    for _ in range(3):
        assert torch.equal(ctx.gemm(A, B, bias, tile=70, variant=3), c3)
    if N % 128 == 0:
        cg = ctx.gemm(A, B, bias, tile=70, variant=3 + 32)[:, : N // 2]
        full = ref.view(M, N // 64, 2, 32)
        want = (full[:, :, 0] * torch.sigmoid(full[:, :, 1])).reshape(M, N // 2)
        # d(a sigmoid(g)) <= |da| + |a| |dg| / 4 with |da|, |dg| ~ 5e-6 * scale for three-product split-bf16
        assert float((cg.double() - want).abs().max()) < 1e-5 * scale * (1 + 0.25 * scale)
    if K >= 32 * 13:
        As, Ab = _presplit(A)
        Bs, Bb = _presplit(B)
        out = _mk((M, N), 17, dev)
        want = out.double() + Ab.double() @ Bb.double().t() + bias.double()
        got = ctx.gemm(As, Bs, bias, tile=75, variant=8 + 16, out=out.clone())
        assert float((got.double() - want).abs().max()) < 2e-6 * float(want.abs().max())


def test_wave_specialised_kernel_rejects_short_k(ctx, dev):
    A, B = _mk((256, 128), 7, dev), _mk((128, 128), 8, dev)
    with pytest.raises(ValueError):
        ctx.gemm(A, B, tile=70)        # K < 256: a tile's epilogue rides in eight k-steps of the next


# ---- residual Linear layer + LayerNorm in one kernel (csrc/gemm_ln.h) ---------------------------------------------------
def _split_layout(X):
    """(rows, K) fp32 -> the pre-split operand layout (per 8 k: 8 bf16 hi | 8 bf16 lo, lo = bf16(x - hi)) as a (rows, K) float
    tensor, and the values hi + lo it encodes."""
    hi = X.to(torch.bfloat16)
    lo = (X - hi.float()).to(torch.bfloat16)
    g = lambda t: t.view(torch.int16).view(X.shape[0], X.shape[1] // 8, 8)
    packed = torch.cat([g(hi), g(lo)], dim=2).reshape(X.shape[0], -1).view(torch.float32).contiguous()
    return packed, hi.double() + lo.double()


def _unsplit(Y):
    """the inverse for an output written in that layout: (rows, N) floats -> hi + lo as fp64."""
    w = Y.contiguous().view(torch.int16).view(Y.shape[0], Y.shape[1] // 8, 16)
    hi, lo = w[:, :, :8].contiguous().view(torch.bfloat16), w[:, :, 8:].contiguous().view(torch.bfloat16)
    return (hi.double() + lo.double()).reshape(Y.shape[0], -1)


@pytest.mark.parametrize("M,K", [(11008, 512), (1, 64), (63, 512), (64, 96), (65, 512), (1000, 256), (4100, 1024)])
def test_residual_layernorm_kernel(ctx, dev, M, K):
    """x = res + A W^T + b and LayerNorm(x) from ONE launch: against fp64 on the values the split operands encode, ragged row
    counts around the 64-row workgroup tile, both output forms; x bit-identical to the LDS-DMA GEMM on the same operands."""
    A, W = _mk((M, K), 21, dev), _mk((256, K), 22, dev) / K ** 0.5
    bias, res, gamma, beta = _mk((256,), 23, dev), _mk((M, 256), 24, dev), _mk((256,), 25, dev), _mk((256,), 26, dev)
    As, Av = _split_layout(A)
    Ws, Wv = _split_layout(W)
    X, Y = ctx.gemm_res_ln(As, Ws, bias, res, gamma, beta, y_split=True)
    X2, Y2 = ctx.gemm_res_ln(As, Ws, bias, res, gamma, beta, y_split=False)
    want = res.double() + Av @ Wv.t() + bias.double()
    # three of the four piece products are kept: |error| <= 2^-16 |a||w| summed over K
    assert float((X.double() - want).abs().max()) < 3e-5 * max(1.0, float(want.abs().max()))
    assert torch.equal(X, X2)
    mu = X.double().mean(dim=1, keepdim=True)
    var = ((X.double() - mu) ** 2).mean(dim=1, keepdim=True)
    ln = (X.double() - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    assert float((Y2.double() - ln).abs().max()) < 2e-5 * max(1.0, float(ln.abs().max()))
    # the split form encodes the fp32 result to 16 mantissa bits
    assert float((_unsplit(Y) - Y2.double()).abs().max()) < 2.0 ** -15 * max(1.0, float(Y2.abs().max()))
    for _ in range(2):
        X3, Y3 = ctx.gemm_res_ln(As, Ws, bias, res, gamma, beta, y_split=True)
        assert torch.equal(X3, X) and torch.equal(Y3, Y)
    # A as fp32 rows, split in the kernel (the form mid-size batches run): the hi + lo pieces it forms are the ones _split_layout made
    # (same values; where x lies exactly between two bf16 numbers the in-kernel split may pick the other high part, so sums agree to
    # rounding, not to the bit)
    X4, Y4 = ctx.gemm_res_ln(Av.float().contiguous(), Ws, bias, res, gamma, beta, y_split=False, a_fp32=True)
    assert float((X4 - X).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))
    assert float((Y4 - Y2).abs().max()) < 1e-5 * max(1.0, float(ln.abs().max()))
    if K >= 32 * 13:     # (the wave-specialised kernel's residual form, itself bit-identical to kernel_dma: needs 13 k-steps)
        assert torch.equal(ctx.gemm(As, Ws, bias, tile=75, variant=8 + 16, out=res.clone()), X)


def test_residual_layernorm_in_the_model_is_bit_identical(dev, lib_path):
    """The control matrix of a B = 32 forward with the fused kernel equals, bit for bit, the one with the GEMM and LayerNorm
    launched apart (DDSP_GEMM_LN=0; child processes, the switch is read once)."""
    import subprocess, sys, os, hashlib
    code = (
        "import sys, os, hashlib; sys.path.insert(0, os.path.join(%r, 'ddsp-svc-official_amd'));"
        "import torch, hipddsp, synthetic;"
        "dev = torch.device('cuda:0');"
        "model, cfg = synthetic.build_model('CombSub', seed=5, device=dev);"
        "inp = {k: v.to(dev) for k, v in synthetic.make_inputs(11, int(os.environ.get('LN_TEST_B', '32')), 172, with_noise=False).items()};"
        "ctx = hipddsp.context_for(dev);"
        "ps = ctx.phase_scan(inp['f0'], 512, 44100);"
        "ctrl = model.unit2ctrl.forward_flat(inp['units'], inp['f0'], ps['phase_frames'], inp['volume'], inp['spk_id'], None);"
        "print(hashlib.sha256(ctrl.cpu().numpy().tobytes()).hexdigest())"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for nb in ("32", "8"):      # pre-split activations / fp32 activations split in the kernel
        digests = []
        for flag in ("1", "0"):
            env = dict(os.environ, DDSP_GEMM_LN=flag, LN_TEST_B=nb, DDSP_GEMM_LN_MIN="512")   # (default: from 8192 rows)
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, out.stderr[-2000:]
            digests.append(out.stdout.strip().splitlines()[-1])
        assert digests[0] == digests[1], (nb, digests)
