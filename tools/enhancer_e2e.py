"""`Enhancer.enhance` end to end at the shipped geometry (seeded weights, checkpoint + config.json written to a temp directory):
10 s of 44.1 kHz audio, with and without the adaptive-key resampling; per-part device time from the kernel families."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import enhancer
import glue_cases as GC

cfg = {"resblock": "1", "upsample_rates": [8, 8, 2, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4, 4],
       "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
       "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "num_mels": 128, "sampling_rate": 44100, "hop_size": 512,
       "n_fft": 2048, "win_size": 2048, "fmin": 40, "fmax": 16000}
dev = torch.device("cuda:0")
d = tempfile.mkdtemp()
json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
torch.save({"generator": GC.nsf_state_dict(cfg, seed=7)}, os.path.join(d, "model"))
enh = enhancer.Enhancer("nsf-hifigan", os.path.join(d, "model"), device=dev)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
T = int(secs * 44100) // 512 * 512
audio = (0.3 * torch.sin(2 * torch.pi * 220 * torch.arange(T) / 44100) + 0.02 * torch.randn(T))[None].to(dev)
f0 = torch.full((1, T // 512, 1), 220.0, device=dev)
for key in (0, 3):
    for _ in range(2):
        out, sr = enh.enhance(audio, 44100, f0, 512, adaptive_key=key)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out, sr = enh.enhance(audio, 44100, f0, 512, adaptive_key=key)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"enhance {secs:.0f} s, adaptive_key={key}: {ms:.2f} ms = {secs * 1e3 / ms:.0f}x real time, out {tuple(out.shape)} at {sr} Hz")
