"""Evidence for the choice of FIR formulation: time per launch of
  * the LDS radix-4 FFT kernel the path already has (spectral_frame_kernel: one 1024-point forward + one 1024-point inverse
    complex FFT per frame, spectrum products in between, frames to scratch, then the overlap-add pass), and
  * the Toeplitz-MFMA kernels (split-bf16) at 510 / 1022 taps plus their filter-synthesis GEMMs,
all at the bench shape (64 x 172 frames).  An FFT-route LTV-FIR needs per frame one 2048-point forward FFT (input frame and
filter packed) and one 2048-point inverse: 2.2x the butterflies and 2x the LDS bytes of the measured kernel per frame, with
half as many workgroups resident per CU."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp
from hipddsp import FIR_ALLPASS, FIR_DYNAMIC, FIR_STATIC, FIR_SPLIT_BF16, FIR_FP32, EXC_GENERATE

dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
B, Fr, hop = 64, 172, 512
g = torch.Generator(device=dev).manual_seed(0)
ctrl_fast = 0.5 * torch.randn(B * Fr, 1539, device=dev, generator=g)
comb = torch.randn(B, Fr * hop, device=dev, generator=g)
ctrl = 0.5 * torch.randn(B * Fr, 1024, device=dev, generator=g)
f0 = 200 + 100 * torch.rand(B * Fr, device=dev, generator=g)
out = {}

def timed(name, fams, fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ctx.profile_begin(fams)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    r = ctx.profile_end()
    out[name] = {k: round(v["ms_total"] / reps * 1e3, 1) for k, v in r.items()}
    print(name, out[name], "us per call", flush=True)

timed("spectral_ola_1024pt (2 FFTs per frame + overlap-add pass)", ["spectral_ola"],
      lambda: ctx.spectral_ola(ctrl_fast, comb, None, EXC_GENERATE, 1, B, Fr, hop))
for mode, col, nmag, name in ((FIR_ALLPASS, 0, 256, "allpass_510"), (FIR_DYNAMIC, 256, 512, "dynamic_1022"),
                              (FIR_STATIC, 768, 256, "static_510")):
    ir = ctx.fir_from_ctrl(mode, ctrl, col, nmag, B * Fr, 44100, f0 if mode == FIR_DYNAMIC else None)
    timed(f"toeplitz_split_bf16_{name}", ["ltv_fir"], lambda: ctx.ltv_fir(comb, ir, B, Fr, hop, math=FIR_SPLIT_BF16))
    timed(f"toeplitz_fp32_{name}", ["ltv_fir"], lambda: ctx.ltv_fir(comb, ir, B, Fr, hop, math=FIR_FP32))
    timed(f"filter_synthesis_{name}", ["fir_act", "fir_dft_gemm"],
          lambda: ctx.fir_from_ctrl(mode, ctrl, col, nmag, B * Fr, 44100, f0 if mode == FIR_DYNAMIC else None))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fir_formulation.json"), "w"), indent=1)
