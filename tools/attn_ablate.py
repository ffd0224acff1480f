"""Where the split-bf16 attention kernels spend their time: the full kernels against builds with parts switched off
(ddsp_performer_attention math = 100 + mask: 1 no staging, 2 no first product, 4 no second product), bench shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp

dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
B, Fr = 64, 172
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (torch.randn(B * Fr, 512, device=dev, generator=g) for _ in range(3))
P = torch.randn(266, 64, device=dev, generator=g)
for math, name in [(0, "fp32 kernels"), (3, "split-bf16"), (101, "no staging"), (102, "no 1st product"), (104, "no 2nd product"),
                   (106, "staging only (+exp)"), (107, "barriers + exp only"), (114, "K: staging loads only"),
                   (122, "K: staging convert+store only")]:
    for _ in range(3):
        ctx.performer_attention(q, k, v, P, B, Fr, math=math)
    torch.cuda.synchronize()
    ctx.profile_begin(["u2c_gemm_ctx", "u2c_gemm_attnout"])
    for _ in range(20):
        ctx.performer_attention(q, k, v, P, B, Fr, math=math)
    torch.cuda.synchronize()
    r = ctx.profile_end()
    print(f"{name:24s}", {kk: round(vv["ms_total"] / vv["launches"] * 1e3, 1) for kk, vv in r.items()}, "us", flush=True)
