"""Attention kernels at the bench shape (64 x 172 frames): average launch time of the K-side and Q-side kernels by HIP
events.  Run once plain and once with DDSP_ATTN_PERSIST=1 (persistent waves pulling items from a counter) to see what the
partly filled last dispatch round costs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import contextlib
import torch
import hipddsp, synthetic

dev = torch.device("cuda:0")
with contextlib.redirect_stdout(sys.stderr):
    model, cfg = synthetic.build_model("CombSub", seed=3, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
inp = {k: v.to(dev) for k, v in synthetic.make_inputs(5, B, 172, with_noise=False).items()}
ctx = hipddsp.context_for(dev)
fams = ["u2c_gemm_ctx", "u2c_gemm_attnout", "u2c_gemm_linear", "u2c_rowwise", "u2c_gemm_conv3"]
with torch.no_grad():
    for _ in range(5):
        out = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=1)[0]
    torch.cuda.synchronize()
    ctx.profile_begin(fams)
    for _ in range(20):
        model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=1)
    torch.cuda.synchronize()
    r = ctx.profile_end()
tag = os.environ.get("DDSP_ATTN_TAG", "persist" if os.environ.get("DDSP_ATTN_PERSIST") == "1" else "grid")
print(tag, {k: round(v["ms_total"] / v["launches"] * 1e3, 2) for k, v in r.items()}, "us per launch;",
      {k: round(v["ms_total"] / 20, 4) for k, v in r.items()}, "ms per step; checksum", float(out.double().abs().sum()))
