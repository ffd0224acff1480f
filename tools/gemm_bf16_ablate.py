"""Where the time of the split-bf16 DMA GEMM goes at the control network's shapes: the 128x128 kernel (tile 30) next to
other tile shapes (33: 128x64, 34: 64x128 on 4 waves, 32: 64x64 on 4 waves) and to its ablations (35: no split + MFMA,
36: no DMA, 37: no DMA and no barrier, 38: split + MFMA + LDS reads only).  Ablated kernels compute garbage: timing only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp

dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
torch.manual_seed(0)
names = {13: "fp32 128x128", 30: "bf16x3 128x128", 33: "bf16x3 128x64", 34: "bf16x3 64x128/4w", 32: "bf16x3 64x64/4w",
         35: "no split+MFMA", 36: "no DMA", 37: "no DMA, no barrier", 38: "split+MFMA+LDS only"}
for (M, N, K) in [(11008, 1536, 256), (11008, 1024, 256), (11008, 512, 256), (11008, 1024, 512)]:
    A = torch.randn(M, K, device=dev)
    B = (torch.rand(N, K, device=dev) * 2 - 1) / K ** 0.5
    line = []
    for tile in (13, 30, 33, 34, 32, 35, 36, 37, 38):
        best = 1e9
        for rnd in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ctx.gemm(A, B, tile=tile)
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 10)
        line.append(f"{names[tile]}: {best * 1e3:.1f}")
    print(f"M={M} N={N} K={K} (us):  " + "   ".join(line), flush=True)
