"""CombSub B=64 forward loop for profiler passes: python tools/fwd_loop.py [steps]"""
import os, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp, synthetic
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    model, cfg = synthetic.build_model(os.environ.get("FWD_MODEL", "CombSub"), seed=3, device=dev)
B = int(os.environ.get("AB_B", "64"))
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(9, B, 172, with_noise=False).items()}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for i in range(n):
    with torch.no_grad():
        model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=i)
torch.cuda.synchronize()
