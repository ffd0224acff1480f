"""Waveform error of the full bench batch (CombSub 64 x 172) against the fp32 small-batch path of the same library:
the 64-clip call runs the split-bf16 GEMMs, shards of 8 run fp32 MFMA.  (The comparison against the CPU oracle with the
1e-4 gate is tests/test_gpu_models.py::test_full_bench_batch_against_oracle.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import synthetic
dev = torch.device("cuda:0")
rms = lambda x: float(torch.sqrt(torch.mean(x.double() ** 2)))
for name in ("CombSub", "Sins", "CombSubFast"):
    model, cfg = synthetic.build_model(name, seed=5, device=dev)
    model.eval()
    B, Fr = 64, 172
    inp = {k: v.to(dev) for k, v in synthetic.make_inputs(900, B, Fr, with_noise=False).items()}
    noise = torch.rand(B, Fr * 512, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    with torch.no_grad():
        full = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=noise)[0]
        parts = torch.cat([model(inp["units"][i:i + 8], inp["f0"][i:i + 8], inp["volume"][i:i + 8], inp["spk_id"][i:i + 8],
                                 noise=noise[i:i + 8])[0] for i in range(0, B, 8)])
    print(f"{name}: signal rms {rms(full):.4f}, split-bf16 vs fp32 path: rms diff {rms(full - parts):.2e}, max {float((full - parts).abs().max()):.2e}")
