"""Where the training step's wall time goes: host enqueue time (no synchronisation between steps) against the device
time of the same steps, and a cProfile of the host side of one step.  Run on the GPU box:
    python tools/train_host_time.py [B]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import synthetic  # noqa: E402
import training  # noqa: E402
from ddsp.loss import RSSLoss  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
model, cfg = synthetic.build_model("CombSub", seed=1, device=dev)
model.train()
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(5, B, 172, with_noise=False).items()}
inp["audio"] = 0.1 * torch.randn(B, 172 * 512, device=dev)
opt = training.AdamW(model.parameters(), lr=5e-4, weight_decay=0.0)
bucket = training.GradBucket(model.parameters(), model)
loss_fn = RSSLoss(256, 2048, 4, device=dev)
scales = [300, 777, 1200, 2000]


def step():
    return training.train_step(model, opt, loss_fn, inp, scales=scales, bucket=bucket)


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, wall incl. drain {1e3 * (t2 - t0) / N:.3f} ms/step")
# host alone, device idle between steps
hs = []
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    step()
    hs.append(time.perf_counter() - a)
    torch.cuda.synchronize()
print("host time of one step on an idle device (ms):", [round(1e3 * h, 3) for h in hs])
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue()[:6000])
