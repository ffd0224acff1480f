import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import synthetic
from oracle import ctrlnet as OC
dev = torch.device("cuda:0")
model, cfg = synthetic.build_model("CombSub", seed=99)
sd = {k[len("unit2ctrl."):]: v for k, v in model.state_dict().items() if k.startswith("unit2ctrl.")}
B, Fr = 1, 87
inp = synthetic.make_inputs(1234 + B, B, Fr, with_noise=False)
phase = torch.zeros(B, Fr)
with torch.no_grad():
    want = OC.unit2control(sd, inp["units"], inp["f0"], phase, inp["volume"], inp["spk_id"], None, model.unit2ctrl.output_splits, return_flat=True)
model = model.to(dev)
got = model.unit2ctrl.forward_flat(inp["units"].to(dev), inp["f0"].to(dev), phase.to(dev), inp["volume"].to(dev), inp["spk_id"].to(dev), None).cpu()
err = (got - want).abs()
print("max", float(err.max()), "mean", float(err.mean()))
print("per-frame max:", [round(float(x), 5) for x in err.reshape(Fr, -1).max(1).values])
