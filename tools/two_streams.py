"""Experiment: the 64-clip forward as S concurrent sub-batches on S streams (one hipddsp.Context each): do the kernels of
independent sub-batches fill each other's partly filled tile rounds and launch ramps?   python tools/two_streams.py"""
import os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp, synthetic
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    model, cfg = synthetic.build_model("CombSub", seed=3, device=dev)
B = 64
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(9, B, 172, with_noise=False).items()}


def run(S, n):
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    ctxs = [hipddsp.Context(dev) for _ in range(S)]
    parts = [{k: v[i * (B // S):(i + 1) * (B // S)].contiguous() for k, v in inp.items()} for i in range(S)]
    cur = torch.cuda.current_stream(dev)

    def step(i):
        outs = []
        for s, c, p in zip(streams, ctxs, parts):
            s.wait_stream(cur)
            with torch.cuda.stream(s), hipddsp.use_context(c), torch.no_grad():
                outs.append(model(p["units"], p["f0"], p["volume"], p["spk_id"], noise_seed=i)[0])
        for s in streams:
            cur.wait_stream(s)
        return outs
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    best = 1e9
    for r in range(4):
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


for S in (1, 2, 4, 1, 2):
    print(f"{S} stream(s) x {B // S} clips: {run(S, 30):.4f} ms per 64 clips", flush=True)
