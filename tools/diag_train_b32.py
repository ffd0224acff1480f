"""Diagnostic: where the B=32 x 172 training gradients of the device path and of autograd through the CPU oracle part
ways (signal, loss gradient w.r.t. the signal in fp32 and fp64, d_ctrl, parameter gradients)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import numpy as np, torch
import synthetic
from oracle import loss as OL, synth as OS
from ddsp.loss import RSSLoss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Fr = 172
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
model, cfg = synthetic.build_model("CombSub", seed=17)
sd0 = model.state_dict()
inp = synthetic.make_inputs(777, B, Fr)
rng = np.random.Generator(np.random.PCG64(12))
target = torch.from_numpy((0.1 * rng.standard_normal((B, Fr * 512))).astype(np.float32))
scales = [300, 777, 1531, 2047]
torch.set_num_threads(16)
names = [n for n, _ in model.named_parameters()]
params = {k: sd0[k].clone().requires_grad_(True) for k in names}
sd = dict(sd0); sd.update(params)
sig_o = OS.combsub_forward(sd, cfg, inp["units"], inp["f0"], inp["volume"], inp["spk_id"], infer=True, noise=inp["noise"])[0]
sig_o.retain_grad()
loss_o = OL.rss_loss(sig_o, target, scales)
loss_o.backward()
x64 = sig_o.detach().double().requires_grad_(True)
OL.rss_loss(x64, target.double(), scales).backward()
print("oracle: loss", float(loss_o), "dL/dsig fp32 vs fp64", rel(sig_o.grad, x64.grad), "norm", float(x64.grad.norm()))

m = model.to(dev).train()
d = {k: v.to(dev) for k, v in inp.items()}
sig = m(d["units"], d["f0"], d["volume"], d["spk_id"], infer=True, noise=d["noise"])[0]
sig.retain_grad()
crit = RSSLoss(256, 2048, 4, device=dev); crit.set_scales(scales)
loss = crit(sig, target.to(dev)); loss.backward()
print("gpu: loss", float(loss), "sig rel", rel(sig.detach().cpu(), sig_o.detach()), "dL/dsig vs fp64", rel(sig.grad.cpu(), x64.grad))
# loss gradient of the GPU kernel AT THE ORACLE'S SIGNAL (same input): separates the loss kernel from the synthesis
import hipddsp
ctx = hipddsp.context_for(dev)
_, g_at_o = ctx.rss_loss(sig_o.detach().to(dev), target.to(dev), scales, want_grad=True)
print("gpu loss grad at the oracle's signal vs fp64:", rel(g_at_o.cpu(), x64.grad))
for n in scales:
    _, g1 = ctx.rss_loss(sig_o.detach().to(dev), target.to(dev), [n], want_grad=True)
    y = sig_o.detach().double().requires_grad_(True); OL.rss_loss(y, target.double(), [n]).backward()
    print("   scale", n, rel(g1.cpu(), y.grad))
errs = sorted(((rel(p.grad.cpu(), params[n].grad), n) for n, p in m.named_parameters()), reverse=True)
print("param grads worst:", errs[:4], "best:", errs[-2:])
# push the ORACLE's loss gradient through the device backward: separates the adjoints from the loss
m.zero_grad()
sig2 = m(d["units"], d["f0"], d["volume"], d["spk_id"], infer=True, noise=d["noise"])[0]
sig2.backward(sig_o.grad.to(dev))
errs = sorted(((rel(p.grad.cpu(), params[n].grad), n) for n, p in m.named_parameters()), reverse=True)
print("device backward fed with the oracle's dL/dsig: worst", errs[:4], "mean", sum(e for e, _ in errs) / len(errs))
