"""Per-shape time of the NSF-HiFiGAN generator's convolutions (ddsp_conv1d) at 860 frames (10 s), as the generator calls them:
activated input (in_slope = 1), raw + activated outputs, split operand layout where the stage is a multiple of 64 channels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp
dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
L = 860
for stage, (T, C) in enumerate([(8 * L, 256), (64 * L, 128), (128 * L, 64), (256 * L, 32), (512 * L, 16)]):
    x = torch.randn(T, C, device=dev).abs() * 0.01
    split = C % 64 == 0
    tot = 0.0
    for k in (3, 7, 11):
        w = torch.randn(C, k * C, device=dev) * 0.05
        ws = hipddsp.presplit(w.cpu()).to(dev) if split else None
        b = torch.zeros(C, device=dev)
        for d in (1, 5):
            kw = dict(residual=x, act_slope=0.1, w_split=ws, x_split=split, act_split=split)
            for _ in range(2):
                ctx.conv1d(x, w, b, k, d, 1.0, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                ctx.conv1d(x, w, b, k, d, 1.0, **kw)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 5 * 1e3
            gf = 2.0 * T * C * C * k / 1e9
            tot += us * (2 if d == 1 else 1) * 2   # per stage: 3 blocks x (3 convs1 with d = 1, 3, 5 + 3 convs2 with d = 1)
            print(f"stage {stage} T={T} C={C} k={k} d={d}: {us:8.1f} us  {gf / us * 1e3:6.1f} TFLOP/s  {16.0 * T * C / us / 1e3:6.0f} GB/s (x, res in; raw, act out)")
    print(f"stage {stage}: ~{tot / 1e3:.2f} ms for its 18 convolutions")
