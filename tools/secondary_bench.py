"""Secondary measurements for DESIGN.md: BASELINE configs #3 (Sins-256 additive bank), CombSubFast, #4 (train step),
#5 (one real-time block: model forward on a 44 100-sample window + SOLA splice, B=1)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp, synthetic, realtime

dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
out = {}

def host_ms(fn, n=50):
    """Host time to ENQUEUE one call (GPU idle at the start, no synchronisation inside the loop)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e3


def timeit(fn, n=20, warm=3, rounds=3):
    """Seconds per call: the MEDIAN of `rounds` timed rounds of n calls each.  (One round of 20 calls is at the mercy of a single
    stall - an allocator call, a first touch: two evidence sessions of round 3 each had one entry 2-3x off that a rerun did not show.)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return sorted(ts)[len(ts) // 2]

B, Fr = 64, 172
for name in ("CombSub", "Sins256", "CombSubFast", "Sins"):
    model, cfg = synthetic.build_model(name, seed=1, device=dev)
    inp = {k: v.to(dev) for k, v in synthetic.make_inputs(3, B, Fr, with_noise=False).items()}
    with torch.no_grad():
        t = timeit(lambda: model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=1))
    out[f"{name}_forward_B64"] = {"ms": t * 1e3, "samples_per_s": B * Fr * 512 / t, "x_realtime": B * Fr * 512 / t / 44100}

# additive-only bank of config #3 (H = 256)
H = 256
ctrl = torch.randn(B * Fr, H, device=dev) * 0.5
f0 = torch.rand(B, Fr, 1, device=dev) * 700 + 65
ps = ctx.phase_scan(f0, 512, 44100, None, True, 0, want_phase=True)
t = timeit(lambda: ctx.sins_bank(ctrl, 0, H, f0, ps["phase"], B, Fr, 512, 44100))
out["sins256_bank_only_B64"] = {"ms": t * 1e3, "samples_per_s": B * Fr * 512 / t, "hbm_GBps_algorithmic": 6.01 * B * Fr * 512 / t / 1e9}

# the front-end steps before the path, at the bench batch (HBM-bound: algorithmic GB/s)
audio = torch.rand(B, Fr * 512, device=dev) * 2 - 1
t = timeit(lambda: ctx.volume_extract(audio, 512), n=50)
out["volume_extract_B64"] = {"ms": t * 1e3, "GBps_algorithmic": 4.0 * B * (Fr * 512 + Fr + 1) / t / 1e9}
units_raw = torch.randn(B, 101, 256, device=dev)
t = timeit(lambda: ctx.align_units(units_raw, Fr + 1, (512 / 44100) / (320 / 16000)), n=50)
out["align_units_B64"] = {"ms": t * 1e3, "GBps_algorithmic": 8.0 * B * (Fr + 1) * 256 / t / 1e9}

# real-time block: window of 44 100 samples at 44.1 kHz -> Fr = 87, B = 1, then SOLA splice
model, cfg = synthetic.build_model("CombSub", seed=1, device=dev)
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(5, 1, 87, with_noise=False).items()}
sp = realtime.Splicer(44100, 0.2, 0.04, dev)
def rt():
    with torch.no_grad():
        sig = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=2)[0]
        ctx.volume_gate_(sig, inp["volume"], -60, 512)
        return sp.push(sig[0])
t = timeit(rt, n=50, warm=5)
out["realtime_block_B1"] = {"ms_per_block": t * 1e3, "block_ms_of_audio": 200.0, "x_realtime": 0.2 / t,
                            "host_enqueue_ms": host_ms(rt)}

# the same block with the model forward replayed from a HIP graph (graphed.GraphedSynth); gate and SOLA stay eager
import graphed
gs = graphed.GraphedSynth(model, 1, 87)
def rt_graph():
    sig = gs(inp["units"], inp["f0"], inp["volume"], inp["spk_id"])[0]
    ctx.volume_gate_(sig, inp["volume"], -60, 512)
    return sp.push(sig[0])
t = timeit(rt_graph, n=50, warm=5)
out["realtime_block_B1_hip_graph"] = {"ms_per_block": t * 1e3, "block_ms_of_audio": 200.0, "x_realtime": 0.2 / t,
                                      "host_enqueue_ms": host_ms(rt_graph)}
# the bench batch replayed from a HIP graph: same kernels as CombSub_forward_B64 minus the host's launch gaps, plus what
# the replay adds (input copies into static tensors, a torch uniform_ draw for the noise instead of the in-kernel one)
model64, _ = synthetic.build_model("CombSub", seed=1, device=dev)
inp64 = {k: v.to(dev) for k, v in synthetic.make_inputs(3, B, Fr, with_noise=False).items()}
gs64 = graphed.GraphedSynth(model64, B, Fr)
t = timeit(lambda: gs64(inp64["units"], inp64["f0"], inp64["volume"], inp64["spk_id"]))
t_replay = timeit(lambda: gs64.graph.replay())
out["CombSub_forward_B64_hip_graph"] = {"ms": t * 1e3, "ms_replay_only": t_replay * 1e3, "samples_per_s": B * Fr * 512 / t,
                                        "x_realtime": B * Fr * 512 / t / 44100}
# the enhancer's generator at the shipped geometry (seeded weights), 10 s and 2 s of audio
sys.path.insert(0, os.path.join(ROOT, "tests"))
import enhancer
import glue_cases as GC
ecfg = {"resblock": "1", "upsample_rates": [8, 8, 2, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4, 4],
        "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
        "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "num_mels": 128, "sampling_rate": 44100, "hop_size": 512,
        "n_fft": 2048, "win_size": 2048, "fmin": 40, "fmax": 16000}
gen = enhancer.Generator(enhancer.AttrDict(ecfg), GC.nsf_state_dict(ecfg, seed=7))
for L in (860, 172):
    mel, ef0, ri = GC.nsf_inputs(ecfg, L=L, seed=8)
    mel, ef0 = mel.to(dev), ef0.to(dev)
    t = timeit(lambda: gen(mel, ef0, rand_ini=ri[0]), n=10)
    out[f"enhancer_generator_{L}_frames"] = {"ms": t * 1e3, "x_realtime": L * 512 / 44100 / t, "tflops_algorithmic": 0.627e-3 * L / t}
# the causal network (c: true) at the bench batch - chunked causal linear attention (performer_causal_kernel)
from ddsp.vocoder import CombSub
mc = CombSub(44100, 512, cfg["n_mag_allpass"], cfg["n_mag_harmonic"], cfg["n_mag_noise"], 256, cfg["n_spk"], c=True)
mc.load_state_dict(model64.state_dict(), strict=True)
mc = mc.to(dev).eval()
with torch.no_grad():
    t = timeit(lambda: mc(inp64["units"], inp64["f0"], inp64["volume"], inp64["spk_id"], noise_seed=1), n=20, warm=3)
out["CombSub_causal_forward_B64"] = {"ms": t * 1e3, "samples_per_s": B * Fr * 512 / t, "x_realtime": B * Fr * 512 / t / 44100}
# (the model constructors print a banner line each; the JSON goes to its own file when a path is given)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
