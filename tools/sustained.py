"""Headline forward (CombSub B=64) back to back for ~12 s: ms per step in one-second windows (clock / power behaviour of the
box under a sustained load).   python tools/sustained.py [seconds]"""
import os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp, synthetic
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    model, cfg = synthetic.build_model("CombSub", seed=3, device=dev)
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(9, 64, 172, with_noise=False).items()}
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
for i in range(5):
    with torch.no_grad():
        model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=i)
torch.cuda.synchronize()
t_start = time.perf_counter()
while time.perf_counter() - t_start < secs:
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 1.0:
        for i in range(50):
            with torch.no_grad():
                model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=i)
        torch.cuda.synchronize()
        n += 50
    print(f"t = {time.perf_counter() - t_start:5.1f} s: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step", flush=True)
