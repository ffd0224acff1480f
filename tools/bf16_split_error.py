"""CPU estimate of what a split-bf16 GEMM (3 or 6 bf16 products per fp32 product, fp32 accumulation) would cost in
accuracy at the control network's GEMM shapes - input for the 'bf16x3 path' listed as next step in DESIGN.md section 9.
No GPU, no product or oracle code: random operands with the layers' statistics (LayerNorm-ed activations ~N(0,1),
weights ~U(-1/sqrt(K), 1/sqrt(K)))."""
import torch

torch.manual_seed(0)
bf = lambda x: x.to(torch.bfloat16).to(torch.float32)


def split(x, parts):
    out, r = [], x
    for _ in range(parts):
        h = bf(r)
        out.append(h)
        r = r - h
    return out


def gemm_split(A, W, terms):
    a, w = split(A, 3), split(W, 3)
    pairs = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)][:terms]
    C = torch.zeros(A.shape[0], W.shape[0])
    for i, j in pairs:
        C += a[i] @ w[j].t()          # each product of two bf16 values is exact in fp32; accumulation in fp32
    return C


for K, N in [(256, 1536), (512, 256), (256, 1024), (768, 256)]:
    A = torch.randn(2048, K)
    W = (torch.rand(N, K) * 2 - 1) / K ** 0.5
    ref = (A.double() @ W.double().t())
    rel = lambda C: float((C.double() - ref).norm() / ref.norm())
    print(f"K={K:4d} N={N:4d}: fp32 {rel(A @ W.t()):.1e}   bf16x1 {rel(gemm_split(A, W, 1)):.1e}   "
          f"bf16x3 {rel(gemm_split(A, W, 3)):.1e}   bf16x4 {rel(gemm_split(A, W, 4)):.1e}   bf16x6 {rel(gemm_split(A, W, 6)):.1e}")
