"""CombSub forward time against the batch size (2 s clips): where the small-batch kernel choices cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch, contextlib
import synthetic
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    model, cfg = synthetic.build_model("CombSub", seed=1, device=dev)
for B in (1, 2, 4, 8, 16, 24, 32, 47, 48, 64, 96, 128):
    inp = {k: v.to(dev) for k, v in synthetic.make_inputs(3, B, 172, with_noise=False).items()}
    def f():
        with torch.no_grad():
            return model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=1)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:4d}: {ms:7.3f} ms/step  {ms / B * 1e3:7.1f} us per clip  {B * 172 * 512 / ms / 44.1:9.0f}x real time")
