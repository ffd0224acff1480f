"""Phase scan (frame sums + in-frame scan + combtooth) at the bench shape: B=64, Fr=172, hop=512 -> 22.5 MB of comb.
Device time per call from the library's HIP-event family timer, and GB/s of the algorithmic bytes.   python tools/phase_scan_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp

dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
B, Fr, hop, sr = 64, 172, 512, 44100
g = torch.Generator().manual_seed(0)
f0 = (torch.rand(B, Fr, generator=g) * 600 + 80).to(dev)
for precise in (True, False):
    for _ in range(5):
        out = ctx.phase_scan(f0, hop, sr, precise=precise, comb_mode=hipddsp.COMB_SINC)
    torch.cuda.synchronize()
    ctx.profile_begin(["phase_scan"])
    N = 50
    for _ in range(N):
        out = ctx.phase_scan(f0, hop, sr, precise=precise, comb_mode=hipddsp.COMB_SINC)
    prof = ctx.profile_end()
    us = prof["phase_scan"]["ms_total"] * 1e3 / N
    mb = B * Fr * hop * 4 / 1e6
    print(f"precise={precise}: {us:.2f} us per call, {mb:.1f} MB -> {mb / us * 1e-3 * 1e3:.0f} GB/s = {mb / us / 8e3 * 1e3:.3f} of 8 TB/s")
