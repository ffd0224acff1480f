"""Per-shape time of the NSF-HiFiGAN generator's convolutions (ddsp_conv1d) at 860 frames (10 s): where the 12.6 ms go."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp
dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
L = 860
tot = 0.0
for stage, (T, C) in enumerate([(8 * L, 256), (64 * L, 128), (128 * L, 64), (256 * L, 32), (512 * L, 16)]):
    x = torch.randn(T, C, device=dev)
    for k in (3, 7, 11):
        w = torch.randn(C, k * C, device=dev) * 0.05
        b = torch.zeros(C, device=dev)
        for d in (1, 5):
            for _ in range(2):
                ctx.conv1d(x, w, b, k, d, 0.1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                ctx.conv1d(x, w, b, k, d, 0.1, residual=x)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 5 * 1e3
            gf = 2.0 * T * C * C * k / 1e9
            print(f"stage {stage} T={T} C={C} k={k} d={d}: {us:8.1f} us  {gf / us * 1e3:6.1f} TFLOP/s  traffic {8.0 * T * C / us / 1e3:6.1f} GB/s alg")
            if d == 1:
                tot += us * (2 if True else 1) * 1.0   # d=1 runs 3 (convs2) + 1 (convs1 d=1) = 4 of 6; rough
