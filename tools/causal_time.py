"""The causal network (`c: true`, pcmer.py:170-188) at the bench batch: forward time and per-family device time.
python tools/causal_time.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp, synthetic
from ddsp.vocoder import CombSub

dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
B, Fr = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 172
model64, cfg = synthetic.build_model("CombSub", seed=1, device=dev)
mc = CombSub(44100, 512, cfg["n_mag_allpass"], cfg["n_mag_harmonic"], cfg["n_mag_noise"], 256, cfg["n_spk"], c=True)
mc.load_state_dict(model64.state_dict(), strict=True)
mc = mc.to(dev).eval()
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(3, B, Fr, with_noise=False).items()}
fn = lambda: mc(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=1)
with torch.no_grad():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    N = 10
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / N * 1e3
    print(f"CombSub causal forward B={B}: {ms:.3f} ms")
    ctx.profile_begin()
    for _ in range(N):
        fn()
    prof = ctx.profile_end()
tot = 0.0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms_total"]):
    if v["launches"]:
        print(f"  {k:20s} {v['launches'] // N:4d} launches  {v['ms_total'] / N * 1e3:8.1f} us")
        tot += v["ms_total"] / N
print(f"  family sum {tot * 1e3:.1f} us")
