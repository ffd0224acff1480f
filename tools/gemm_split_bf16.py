"""Speed and error of the experimental split-bf16 product mode of the DMA GEMM (ddsp_gemm_f32 tiles 30-32) next to the
fp32 MFMA tiles (13: 128x128, 15: 64x64 on 4 waves), at the control network's shapes.  Error = ||C - C64|| / ||C64||
against an fp64 matmul on the device."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp

dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
torch.manual_seed(0)
for (M, N, K) in [(11008, 1536, 256), (11008, 1024, 256), (11008, 256, 512), (11008, 256, 768), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device=dev)
    B = (torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5
    ref = A.double() @ B.double().t()
    line = []
    for tile, name in ((13, "fp32 128x128"), (30, "bf16x3 128x128"), (31, "bf16x6 128x128"), (15, "fp32 64x64"), (32, "bf16x3 64x64")):
        C = ctx.gemm(A, B, tile=tile)
        err = float((C.double() - ref).norm() / ref.norm())
        best = 1e9
        for rnd in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ctx.gemm(A, B, tile=tile)
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 10)
        line.append(f"{name}: {best * 1e3:.1f}us/{2.0 * M * N * K / best / 1e9:.0f}TF err {err:.1e}")
    print(f"M={M} N={N} K={K}:  " + "   ".join(line), flush=True)
