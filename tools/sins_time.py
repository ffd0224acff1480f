"""Sinusoid bank alone (H = 256, 64 x 172 frames) and the Sins-256 forward."""
import os, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch, hipddsp, synthetic
dev = torch.device("cuda:0"); ctx = hipddsp.context_for(dev)
B, Fr, H = 64, 172, 256
ctrl = torch.randn(B * Fr, H, device=dev) * 0.5
f0 = torch.rand(B, Fr, 1, device=dev) * 700 + 65
ps = ctx.phase_scan(f0, 512, 44100, None, True, 0, want_phase=True)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
print("bank H=256 ms", t(lambda: ctx.sins_bank(ctrl, 0, H, f0, ps["phase"], B, Fr, 512, 44100)))
with contextlib.redirect_stdout(sys.stderr):
    model, cfg = synthetic.build_model("Sins256", seed=1, device=dev)
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(5, B, Fr, with_noise=False).items()}
with torch.no_grad():
    print("Sins256 forward ms", t(lambda: model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=1)))
