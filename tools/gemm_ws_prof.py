"""Launches a few GEMM variants back to back (for rocprofv3 passes): python tools/gemm_ws_prof.py [M N K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp
dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
shapes = [(11008, 1536, 256), (11008, 256, 512)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).abs() * 0.01; B = torch.randn(N, K, device=dev).abs() * 0.01
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev)
    cases = [dict(tile=50), dict(tile=53), dict(tile=70, variant=8), dict(tile=72, variant=8), dict(tile=70, variant=8 + 512),
             dict(tile=70, variant=8 + 1024), dict(tile=75, variant=8)]
    for kw in cases:
        for _ in range(12):
            ctx.gemm(A, B, bias, out=out, **kw)
    torch.cuda.synchronize()
