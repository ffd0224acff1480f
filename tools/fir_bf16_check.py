"""Split-bf16 LTV-FIR kernel against the fp32-MFMA kernel: error on odd shapes, then timing at the bench shape.

Usage: python tools/fir_bf16_check.py [reps]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp
dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
hop = 512
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)
worst = 0.0
for (B, Fr, n) in [(2, 6, 510), (2, 6, 1022), (1, 7, 510), (2, 1, 1022), (1, 3, 64), (1, 5, 2046), (3, 13, 254), (2, 25, 1022),
                   (1, 2, 32), (2, 9, 1534), (2, 173, 1022), (3, 40, 510), (1, 64, 254), (5, 50, 1022)]:
    x = torch.rand(B, Fr * hop, device=dev) * 2 - 1
    ir = torch.randn(B, Fr, n, device=dev) / n ** 0.5
    add = torch.randn(B, Fr * hop, device=dev)
    ref, ref_sum = ctx.ltv_fir(x, ir, B, Fr, hop, add_in=add, math=0)
    for math in (3, 31, 32, 33, 34, 41, 42, 43, 44, 45, 46, 47, 48):
        got, got_sum = ctx.ltv_fir(x, ir, B, Fr, hop, add_in=add, math=math)
        e = (got - ref).abs().max().item() / ref.abs().max().item()
        es = (got_sum - ref_sum).abs().max().item() / ref.abs().max().item()
        worst = max(worst, e, es)
        flag = "" if max(e, es) < 3e-5 else "   <-- BAD"
        print(f"B={B} Fr={Fr} n={n} math={math}: max err / max|y| = {e:.2e} (sum output {es:.2e}){flag}")
    # generated excitation: same counter stream in both kernels
    r2, _ = ctx.ltv_fir(None, ir, B, Fr, hop, excitation=2, noise_seed=7, math=0)
    g2, _ = ctx.ltv_fir(None, ir, B, Fr, hop, excitation=2, noise_seed=7, math=3)
    u = torch.rand(B, Fr * hop, device=dev)
    r1, _ = ctx.ltv_fir(u, ir, B, Fr, hop, excitation=1, math=0)
    g1, _ = ctx.ltv_fir(u, ir, B, Fr, hop, excitation=1, math=3)
    e2 = (g2 - r2).abs().max().item() / r2.abs().max().item()
    e1 = (g1 - r1).abs().max().item() / r1.abs().max().item()
    worst = max(worst, e1, e2)
    print(f"   generated noise {e2:.2e}, unit noise {e1:.2e}")
print(f"worst relative-to-peak error {worst:.2e}")

B, Fr = 64, 172
x = torch.rand(B, Fr * hop, device=dev) * 2 - 1
for n in (1022, 510):
    ir = torch.randn(B, Fr, n, device=dev) / n ** 0.5
    for math in (0, 31, 42, 44, 45, 46, 47, 3, 52, 54, 55, 56, 57):
        for _ in range(3):
            ctx.ltv_fir(x, ir, B, Fr, hop, math=math)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            ctx.ltv_fir(x, ir, B, Fr, hop, math=math)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / reps
        print(f"n={n} math={math}: {t*1e3:.1f} us  {B*Fr*hop*n*4/t/1e9:.1f} TFLOP/s algorithmic", flush=True)
