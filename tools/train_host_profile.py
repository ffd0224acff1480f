"""Host-side time of the pieces of one training step (no synchronisation inside a step: what the Python thread spends
before the GPU can start each phase).  Usage: python tools/train_host_profile.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import synthetic, training
from ddsp.loss import RSSLoss

dev = torch.device("cuda:0")
model, cfg = synthetic.build_model("CombSub", seed=1, device=dev)
model.train()
B, Fr = 32, 172
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(3, B, Fr, with_noise=False).items()}
audio = 0.1 * torch.randn(B, Fr * 512, device=dev)
opt = training.AdamW(model.parameters(), lr=5e-4, weight_decay=0.0)
crit = RSSLoss(256, 2048, 4, device=dev)
acc = {}
def tick(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t
N = 20
for it in range(N + 5):
    if it == 5:
        torch.cuda.synchronize(); acc.clear(); wall0 = time.perf_counter()
    t = time.perf_counter()
    opt.zero_grad(); t = tick("zero_grad", t)
    sig, _, _ = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], infer=False); t = tick("forward", t)
    crit.set_scales([300, 700, 1100, 1900]); loss = crit(sig, audio); t = tick("loss", t)
    loss.backward(); t = tick("backward", t)
    training.allreduce_gradients(list(model.parameters()), 1); t = tick("allreduce(world=1)", t)
    opt.step(); t = tick("optimizer.step", t)
torch.cuda.synchronize()
wall = (time.perf_counter() - wall0) / N * 1e3
print("wall ms/step %.3f" % wall)
for k, v in acc.items():
    print("  host %-20s %.3f ms" % (k, v / N * 1e3))
print("  host total %.3f ms" % (sum(acc.values()) / N * 1e3))
