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
