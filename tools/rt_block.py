"""Real-time block (BASELINE config #5: B = 1, 44 100-sample window = 87 frames, model + gate + SOLA): per-kernel anatomy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp, synthetic, realtime
dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    model, cfg = synthetic.build_model("CombSub", seed=1, device=dev)
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(5, 1, 87, with_noise=False).items()}
sp = realtime.Splicer(44100, 0.2, 0.04, dev)
def rt():
    with torch.no_grad():
        sig = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=2)[0]
        ctx.volume_gate_(sig, inp["volume"], -60, 512)
        return sp.push(sig[0])
for _ in range(5):
    rt()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    rt()
torch.cuda.synchronize()
print("ms per block", (time.perf_counter() - t0) / 50 * 1e3)

# the same block with the model forward replayed as a HIP graph (graphed.GraphedSynth), gate + splice eager
import graphed
gs = graphed.GraphedSynth(model, 1, 87)
def rt_graph():
    with torch.no_grad():
        sig = gs(inp["units"], inp["f0"], inp["volume"], inp["spk_id"])[0]
        ctx.volume_gate_(sig, inp["volume"], -60, 512)
        return sp.push(sig[0])
for _ in range(5):
    rt_graph()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    rt_graph()
torch.cuda.synchronize()
print("ms per block, graphed forward", (time.perf_counter() - t0) / 200 * 1e3)
# replay alone (device time of the captured forward)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    gs.graph.replay()
e1.record()
torch.cuda.synchronize()
print("ms per replay (forward only, back to back)", e0.elapsed_time(e1) / 200)
# latency of ONE block on an idle device (what the audio callback sees)
lat = []
for _ in range(20):
    torch.cuda.synchronize()
    a = time.perf_counter()
    rt_graph()
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - a) * 1e3)
print("latency of one graphed block on an idle device (ms): median", sorted(lat)[len(lat) // 2])
lat = []
for _ in range(20):
    torch.cuda.synchronize()
    a = time.perf_counter()
    rt()
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - a) * 1e3)
print("latency of one eager block on an idle device (ms): median", sorted(lat)[len(lat) // 2])
