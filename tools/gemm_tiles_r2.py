"""Round-2 tile experiment for the split-bf16 DMA GEMM: LDS read bandwidth is co-critical with the matrix pipe when a wave
owns a 32x64 accumulator (12 fragment reads per 12 MFMAs per k-step); 64x64 per wave needs 16 reads per 24 MFMAs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp
dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
for (M, N, K) in [(11008, 1536, 256), (11008, 1024, 256), (11008, 256, 512), (11008, 512, 256)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
    ref = A.double() @ B.double().t()
    res = {}
    A = A.abs() * 0.01; B = B.abs() * 0.01        # (as bf16 bit patterns: small finite numbers)
    for tile in (50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60):
        C = ctx.gemm(A, B, tile=tile)
        res[tile] = []
    for rnd in range(5):
        for key in res:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ctx.gemm(A, B, tile=key)
            e.record(); torch.cuda.synchronize()
            res[key].append(s.elapsed_time(e) / 10)
    print(f"M={M} N={N} K={K}: " + "  ".join(f"t{k}={min(v)*1e3:.1f}us" for k, v in res.items()), flush=True)
