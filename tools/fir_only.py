"""Runs only the LTV-FIR kernel at the bench shape (for rocprofv3 --pmc passes and quick timing).
Usage: python tools/fir_only.py [taps] [reps] [math: 0 fp32, 3 split-bf16, 31.. forced block shapes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp
dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
B, Fr, hop = 64, 172, 512
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1022
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
math = int(sys.argv[3]) if len(sys.argv) > 3 else 0
x = torch.rand(B, Fr * hop, device=dev) * 2 - 1
ir = torch.randn(B, Fr, n, device=dev) / n ** 0.5
for _ in range(2):
    ctx.ltv_fir(x, ir, B, Fr, hop, math=math)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    ctx.ltv_fir(x, ir, B, Fr, hop, math=math)
e.record()
torch.cuda.synchronize()
t = s.elapsed_time(e) / reps
print(f"n={n} math={math}: {t*1e3:.1f} us  {B*Fr*hop*n*4/t/1e9:.1f} TFLOP/s algorithmic")
