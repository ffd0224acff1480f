"""Error of ddsp_phase_vocoder against the reference fixture and its time (one call per real-time block)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import realtime
from frontend_cases import PV_SIZES, pv_inputs
dev = torch.device("cuda:0")
g = np.load(os.path.join(ROOT, "tests", "golden", "glue_phase_vocoder.npz"))
for i, n in enumerate(PV_SIZES):
    a, b, fo, fi = [x.to(dev) for x in pv_inputs(i)]
    got = realtime.phase_vocoder(a, b, fo, fi).cpu()
    want = torch.from_numpy(g[f"pv_{i}"])
    e = (got - want).abs()
    print(f"n={n}: max err {float(e.max()):.2e} rms err {float(e.pow(2).mean().sqrt()):.2e} (signal rms {float(want.pow(2).mean().sqrt()):.3f})")
a, b, fo, fi = [x.to(dev) for x in pv_inputs(0)]
for _ in range(5): realtime.phase_vocoder(a, b, fo, fi)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): realtime.phase_vocoder(a, b, fo, fi)
torch.cuda.synchronize(); print("n=1764: %.1f us per call" % ((time.perf_counter() - t0) / 100 * 1e6))
