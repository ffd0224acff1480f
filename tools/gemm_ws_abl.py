"""Ablations of the wave-specialised GEMM at the QKV shape (timing only): python tools/gemm_ws_abl.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp
dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
for (M, N, K) in [(11008, 1536, 256)]:
    A = torch.randn(M, K, device=dev).abs() * 0.01; B = torch.randn(N, K, device=dev).abs() * 0.01
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev)
    cases = {"r2": dict(tile=50), "ws": dict(tile=70, variant=8)}
    for a in (2, 8, 34, 50, 64, 178, 306, 434):
        cases[f"abl{a}"] = dict(tile=70, variant=8 + 256 * a)
    res = {k: [] for k in cases}
    for k, kw in cases.items():
        ctx.gemm(A, B, bias, out=out, **kw)
    torch.cuda.synchronize()
    for rnd in range(5):
        for k, kw in cases.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ctx.gemm(A, B, bias, out=out, **kw)
            e.record(); torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / 10)
    print(f"M={M} N={N} K={K}: " + "  ".join(f"{k}={min(v)*1e3:.1f}" for k, v in res.items()) + " us", flush=True)
print("abl bits: 2 no DMA, 4 no stores, 8 product waves barrier only, 16 no lgkmcnt(0), 32 no epilogue pieces, 64 half the DMA pieces")
