"""NSF-HiFiGAN generator (enhancer post-net) at the shipped geometry (44.1 kHz, hop 512: upsample rates
8-8-2-2-2, 512 initial channels, residual kernels 3/7/11 x dilations 1/3/5, 128 mels), seeded random weights:
time per call and per-family device time.   python tools/enhancer_time.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hipddsp, enhancer
import glue_cases as GC

L = int(sys.argv[1]) if len(sys.argv) > 1 else 172
cfg = {"resblock": "1", "upsample_rates": [8, 8, 2, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4, 4],
       "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
       "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "num_mels": 128, "sampling_rate": 44100, "hop_size": 512,
       "n_fft": 2048, "win_size": 2048, "fmin": 40, "fmax": 16000}
dev = torch.device("cuda:0")
sd = GC.nsf_state_dict(cfg, seed=7)
gen = enhancer.Generator(enhancer.AttrDict(cfg), sd)
mel, f0, rand_ini = GC.nsf_inputs(cfg, L=L, seed=8)
mel, f0 = mel.to(dev), f0.to(dev)
for _ in range(3):
    out = gen(mel, f0, rand_ini=rand_ini[0])
torch.cuda.synchronize()
assert torch.isfinite(out).all()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    out = gen(mel, f0, rand_ini=rand_ini[0])
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / N * 1e3
print(f"frames {L} ({L * 512 / 44100:.2f} s of audio): {ms:.3f} ms per call = {L * 512 / 44.1 / ms:.0f}x real time")
# sustained: the same call for ~1 s (the clocks of the pool's boxes settle lower under a sustained matrix load)
N = 250
t0 = time.perf_counter()
for _ in range(N):
    out = gen(mel, f0, rand_ini=rand_ini[0])
torch.cuda.synchronize()
ms2 = (time.perf_counter() - t0) / N * 1e3
print(f"sustained ({N} calls back to back): {ms2:.3f} ms per call = {L * 512 / 44.1 / ms2:.0f}x real time")
