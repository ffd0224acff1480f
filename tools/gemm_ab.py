"""A/B timing of GEMM tiles / schedule variants in one process (interleaved rounds, HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
import torch
import hipddsp

dev = torch.device("cuda:0")
ctx = hipddsp.context_for(dev)
shapes = [(11008, 512, 256), (11008, 1024, 256), (11008, 1536, 256), (11008, 256, 512), (11008, 256, 768), (88064, 266, 64), (11000, 250, 512), (4096, 4096, 4096)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    ref = A @ B.t()
    res = {}
    for tile in (10, 13, 15, 16):
        for var in (0,):
            C = ctx.gemm(A, B, tile=tile, variant=var)
            err = float((C - ref).abs().max())
            assert tile >= 20 or err < 1e-2, (tile, var, err)
            res[(tile, var)] = []
    for rnd in range(5):
        for key in res:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ctx.gemm(A, B, tile=key[0], variant=key[1])
            e.record()
            torch.cuda.synchronize()
            res[key].append(s.elapsed_time(e) / 10)
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: " + "  ".join(f"t{k[0]}v{k[1]}={min(v)*1e3:.1f}us/{fl/min(v)/1e9:.0f}TF" for k, v in res.items()), flush=True)
