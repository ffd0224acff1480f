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
