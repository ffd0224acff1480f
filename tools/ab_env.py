"""A/B of one environment switch on the CombSub B=64 forward, interleaved rounds in ONE process is impossible for a getenv-once
switch, so: two child processes per round, alternating.  python tools/ab_env.py DDSP_GEMM_WS 0 1"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%r, "ddsp-svc-official_amd"))
import torch, contextlib
import hipddsp, synthetic
dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    model, cfg = synthetic.build_model("CombSub", seed=3, device=dev)
B = int(os.environ.get("AB_B", "64"))
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(9, B, 172, with_noise=False).items()}
def step(i):
    with torch.no_grad():
        return model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=i)[0]
for i in range(10): step(i)
torch.cuda.synchronize()
best = 1e9
for r in range(5):
    t0 = time.perf_counter()
    for i in range(30): step(i)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
ctx = hipddsp.context_for(dev)
ctx.profile_begin()
for i in range(3): step(i)
torch.cuda.synchronize()
fam = ctx.profile_end()
print("%%.4f" %% best, " ".join("%%s=%%.3f" %% (k, v["ms_total"] / 3) for k, v in fam.items()))
''' % ROOT
name, vals = sys.argv[1], sys.argv[2:]
for rnd in range(2):
    for v in vals:
        env = dict(os.environ, **{name: v})
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"{name}={v}: ms/step {r.stdout.strip()}" if r.returncode == 0 else r.stderr[-2000:], flush=True)
