#!/usr/bin/env python
"""Throughput bench of the hot path: CombSub 44.1 kHz synthesis, batch = 64 x 2 s clips per GPU
(BASELINE.json configs[1]), units/f0/volume/spk_id -> rendered audio, inputs resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]                      # N = 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full forward (`CombSub.forward`) over the per-GPU batch.  With N > 1 utterances are sharded
over the ranks (weak scaling: 64 per GPU) and the rendered audio is all-gathered over RCCL/xGMI on a side
stream, overlapped with the next step's compute (the gather of the last step is inside the timed region).
Rank 0 prints ONE JSON line; `roofline` is measured live with HIP events around the dominant kernel,
`cpu_baseline` times the CPU oracle (a port of the reference's PyTorch CPU path) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ddsp-svc-official_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

B_PER_GPU = 64
FRAMES = 172
HOP = 512
SR = 44100
BYTES_PER_SAMPLE_API = 6.02   # units 256*4 + f0 4 + volume 4 in, 512*4 + 4 out per frame
PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_MFMA_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
# The kernel families with the largest shares of the step (profiles/); all are bracketed by HIP events in the timed
# region; the one that took longest is reported as `roofline`, the others under `roofline_others`.
#  * ltv_fir: frame-varying FIR as Toeplitz-block products (v_mfma_f32_16x16x32_bf16, operands split into bf16 hi/lo
#    planes while they are staged into the LDS).
#  * u2c_gemm_linear: the control network's Linear layers, the SPLIT-bf16 mode of the DMA GEMM at the bench batch.
#  * u2c_gemm_ctx / u2c_gemm_attnout: the fused Performer attention kernels (key side / query side).
#  `frac` = ALGORITHMIC FLOP / time / dense peak of the matrix pipe the kernel issues on (bf16 2.5 PFLOP/s for the split
#  kernels, fp32 157.3 TFLOP/s otherwise).  A split-bf16 kernel forms every fp32 product from 3 bf16 MFMAs (hi*hi + hi*lo +
#  lo*hi, fp32 accumulation), i.e. ISSUES three times its algorithmic FLOP: that utilisation figure is `mfma_issue_frac`.
FAMILIES_TIMED = ("ltv_fir", "u2c_gemm_linear", "u2c_gemm_ctx", "u2c_gemm_attnout")
# Families whose products follow the context's math mode (ddsp_ctx_set_math): with the default (split-bf16) they issue on the
# bf16 matrix pipe, every fp32 product formed from ISSUE_MULT bf16 MFMAs; with FP32 (the second leg) on the fp32 matrix pipe.
# The attention kernels are among them at this batch (B * 8 heads >= 256: performer_attn_bf16.hip; below that, and in the FP32
# leg, the fp32 kernels of performer_attn.hip run): their feature projections enter an exponential and are formed from three
# bf16 pieces per operand = 6 MFMAs per fp32 product, their context products from two pieces = 3 MFMAs - half the FLOP each.
SPLIT_FAMILIES = ("ltv_fir", "u2c_gemm_linear", "u2c_gemm_ctx", "u2c_gemm_attnout")
ISSUE_MULT = {"ltv_fir": 3.0, "u2c_gemm_linear": 3.0, "u2c_gemm_ctx": 4.5, "u2c_gemm_attnout": 4.5}
SPLIT_ARITH = {"ltv_fir": "split-bf16: 3 bf16 MFMAs per fp32 product, fp32 accumulation",
               "u2c_gemm_linear": "split-bf16: 3 bf16 MFMAs per fp32 product, fp32 accumulation",
               "u2c_gemm_ctx": "split-bf16, fp32 accumulation: k P^T from three bf16 pieces per operand (6 MFMAs per fp32 "
                               "product, it enters an exponential), k'^T v from two (3 MFMAs): 4.5 on average",
               "u2c_gemm_attnout": "split-bf16, fp32 accumulation: q P^T from three bf16 pieces per operand (6 MFMAs per fp32 "
                                   "product), q' ctx from two (3 MFMAs): 4.5 on average"}
KERNEL_LABEL = {   # (default leg: split-bf16 products, FP32 leg)
    "u2c_gemm_ctx": ("performer_fused_bf16_kernel (one workgroup of 12 waves per (utterance, head): key feature map + k'^T v "
                     "context, then query feature map + q' ctx / (q' ks) with ctx and ks kept in the LDS; "
                     "v_mfma_f32_32x32x16_bf16; three launches per step)",
                     "performer_kv_kernel (fused key feature map + k'^T v context, fp32 MFMA 16x16x4, one wavefront per "
                     "(utterance, head, feature tile); three launches per step)"),
    "u2c_gemm_attnout": ("performer_q_bf16_kernel (one workgroup per (utterance, head): query feature map + q' ctx / (q' ks), "
                         "v_mfma_f32_32x32x16_bf16, ctx pieces streamed global -> LDS; three launches per step)",
                         "performer_q_kernel (fused query feature map + q' ctx / (q' ks), fp32 MFMA 16x16x4, one wavefront "
                         "per (utterance, head, frame tile); three launches per step)"),
    "ltv_fir": ("ltv_fir_bf16_march_kernel (frame-varying FIR as Toeplitz-block products, 3 bf16 MFMA 16x16x32 per fp32 "
                "product, operands split while staged into the LDS; three launches per step: all-pass 510 taps, source 1022 "
                "taps, noise 510 taps)",
                "ltv_fir_kernel: the same launches with products on the fp32 matrix pipe (ddsp_ctx_set_math FP32)"),
    "u2c_gemm_linear": ("gemm::kernel_dma (persistent LDS-DMA GEMM, 3 bf16 MFMA 32x32x16 per fp32 product, operands pre-split): "
                        "Linear / 1x1-conv layers of unit2ctrl - 128x128 tiles on 8 waves for QKV and pw1+GLU, 64x64 tiles on 4 "
                        "waves for the N=256 layers; the head on gemm::kernel_ws (loader / product waves, gemm_ws.h); 13 "
                        "launches per step",
                        "the same launches with products on the fp32 matrix pipe (ddsp_ctx_set_math FP32): gemm::kernel_dma"),
}


def measured_traffic(family, split=True):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, FETCH doubled per MI355X_MICROARCH 'HBM'; tools/pmc_traffic.py).  PMC collection needs the profiler
    around the process, so bench.py reads the last committed measurement instead of taking it live; None when absent.
    The profiled command runs both legs; `split` picks the GEMM instantiations of one of them (last template argument of
    gemm::kernel_dma: 3 / 7 / 8 = split-bf16 products, 0 = fp32 products)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_per_launch.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    tot = n = 0.0
    for k, v in d.items():
        if family == "ltv_fir":
            if ("ltv_fir_bf16" in k) if split else k.startswith("ltv_fir_kernel"):
                tot += v["hbm_bytes_per_launch"] * v["launches_sampled"]
                n += v["launches_sampled"]
        # the Linear layers' instantiations at the bench shape (QKV, pw1+GLU, out-projection / pw2, head); the same
        # kernel template also runs the prenet convs and the filter-synthesis DFTs, which are other families
        elif family != "u2c_gemm_linear":
            return None          # no PMC pass committed for this family
        elif "kernel_ws<" in k:                        # the head (wave-specialised kernel): split-bf16 products only
            if not split:
                continue
            tot += v["hbm_bytes_per_launch"] * v["launches_sampled"]
            n += v["launches_sampled"]
        elif "kernel_dma" in k and any(t in k for t in ("EpiSplit3", "EpiGlu", "EpiResidual",
                                                        "kernel_dma<128, 128, gemm::EpiStore")):
            m = re.search(r", (\d+)>\(", k)
            if m and (int(m.group(1)) in (3, 7, 8)) != bool(split):
                continue
            tot += v["hbm_bytes_per_launch"] * v["launches_sampled"]
            # per GEMM CALL: the QKV call is two kernels (its 128x128 tiles and the 64x64 remainder of the last round)
            if "kernel_dma<64, 64, EpiSplit3" not in k:
                n += v["launches_sampled"]
    return tot / n if n else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-every", type=int, default=5,
                    help="the roofline's HIP events bracket every n-th step of the timed region (1 = every step: ~6 %% slower steps)")
    ap.add_argument("--model", default="CombSub", choices=["CombSub", "Sins256", "CombSubFast"])
    ap.add_argument("--mode", default="synth", choices=["synth", "train", "realtime"],
                    help="synth (default, BASELINE configs[1]): forward only; train (configs[3]): one full training "
                         "step per iteration, 32 clips per GPU, gradient all-reduce over RCCL; realtime (configs[4]): one "
                         "stream per GPU, one step = one 0.2 s block through realtime.StreamRenderer (replicas, no collective)")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seed, inp_cpu):
    """Times the oracle's CombSub forward (a port of the reference's PyTorch CPU path) on the host cores, on the SAME 64
    clips the GPU renders (same inputs, same B): median of 5 on the job's core share, plus a one-thread figure
    on the first 8 of those clips (median of 3) - about 15-25 s in total."""
    import synthetic
    from oracle import synth as OS
    # the GPU box gives one GPU's job a 16-core share; more threads than that only oversubscribes ATen's pools
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(16, avail))
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):       # (the constructor prints the reference's banner line)
        model, cfg = synthetic.build_model("CombSub", seed=seed)
    sd = model.state_dict()
    g = torch.Generator().manual_seed(seed)
    noise = torch.rand(B_PER_GPU, FRAMES * HOP, generator=g)

    def run(nb, reps):
        a = {k: v[:nb] for k, v in inp_cpu.items()}
        times = []
        with torch.no_grad():
            OS.combsub_forward(sd, cfg, a["units"], a["f0"], a["volume"], a["spk_id"], noise=noise[:nb])
            for _ in range(reps):
                t0 = time.perf_counter()
                OS.combsub_forward(sd, cfg, a["units"], a["f0"], a["volume"], a["spk_id"], noise=noise[:nb])
                times.append(time.perf_counter() - t0)
        return sorted(times)[reps // 2]

    torch.set_num_threads(threads)
    t_all = run(B_PER_GPU, 5)
    torch.set_num_threads(1)
    t_one = run(8, 3)
    torch.set_num_threads(threads)
    return {"value": B_PER_GPU * FRAMES * HOP / t_all, "unit": "samples/s", "cores": threads, "kind": "port",
            "cpu_model": _cpu_model(), "value_1_thread": 8 * FRAMES * HOP / t_one,
            "sample": f"oracle CombSub.forward on the bench inputs: B={B_PER_GPU} x 2 s clips, median of 5 on {threads} ATen "
                      f"threads ({t_all * 1e3:.0f} ms each); 1 thread: the first 8 clips, median of 3 ({t_one * 1e3:.0f} ms each)"}


def parity_sample(model_cpu_sd, cfg, model, inp, inp_cpu, seed, ctx, dev):
    """RMS error of the BENCHED arithmetic against the CPU oracle: the whole 64-clip batch is rendered once more with an
    injected noise draw (same kernels and batch-dependent kernel choices as the timed steps), 8 of the 64 clips are
    compared with the oracle's rendering of the same clips.  Once per product-arithmetic mode."""
    from oracle import synth as OS
    import hipddsp
    g = torch.Generator().manual_seed(seed + 5)
    noise = torch.rand(B_PER_GPU, FRAMES * HOP, generator=g)
    pick = list(range(0, B_PER_GPU, B_PER_GPU // 8))
    a = {k: v[pick] for k, v in inp_cpu.items()}
    with torch.no_grad():
        want = OS.combsub_forward(model_cpu_sd, cfg, a["units"], a["f0"], a["volume"], a["spk_id"], noise=noise[pick])[0]
    out = {"clips_compared": len(pick), "signal_rms": float(want.double().pow(2).mean().sqrt()), "gate": 1e-4}
    keep = ctx.math
    nd = noise.to(dev)
    for name, mode in (("split_bf16x3", hipddsp.MATH_SPLIT_BF16), ("fp32_mfma", hipddsp.MATH_FP32)):
        ctx.set_math(mode)
        with torch.no_grad():
            sig = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=nd)[0]
        out["rms_error_" + name] = float((sig[pick].cpu().double() - want.double()).pow(2).mean().sqrt())
    ctx.set_math(keep)
    return out


def realtime_mode(args, model, dev, dist, world, rank):
    """BASELINE configs[4]: one real-time stream per GPU (replicas only): a step is ONE 0.2 s block through
    `realtime.StreamRenderer` - sliding 44 100-sample window (87 frames), frame volume, model forward replayed from a HIP graph,
    volume gate, SOLA splice - with synthetic analysis features.  Every block is timed on its own from an idle stream to its
    last kernel (what the audio callback waits for); the line reports the mean and the p99 over the K blocks, the slowest
    rank's figures, and `value` = streams x block samples / mean block time."""
    import realtime
    import synthetic
    sr, block_time = SR, 0.2
    use_graph = os.environ.get("DDSP_RT_GRAPH", "1") != "0"     # measurement aid: 0 = eager launches instead of the graph replay
    r = realtime.StreamRenderer(model, sr, block_time, 0.04, dev, buffer_num=4, threshold_db=-60.0, spk_id=1, use_graph=use_graph)
    feats = [{k: v.to(dev) for k, v in synthetic.make_inputs(9000 + 17 * rank + i, 1, r.frames, with_noise=False).items()}
             for i in range(8)]
    g = torch.Generator(device=dev).manual_seed(11 + rank)
    blocks = [0.2 * torch.randn(r.block, device=dev, generator=g) for _ in range(8)]

    def one(i):
        f = feats[i % 8]
        return r.push_block(blocks[i % 8], units=f["units"], f0=f["f0"])

    for i in range(args.warmup):
        one(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    lat = []
    t_all = time.perf_counter()
    for i in range(args.steps):
        t0 = time.perf_counter()
        one(args.warmup + i)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t_all
    lat.sort()
    mean, p99 = sum(lat) / len(lat), lat[min(len(lat) - 1, int(0.99 * len(lat)))]
    # back-to-back blocks without the per-block host synchronisation: the device-side pace of the chain
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.steps):
        one(i)
    e1.record()
    torch.cuda.synchronize()
    pace = e0.elapsed_time(e1) / args.steps
    stats = torch.tensor([dt, mean, p99, pace], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    dt, mean, p99, pace = [float(v) for v in stats]
    if rank == 0:
        print(json.dumps({
            "metric": "audio samples/sec @44.1kHz CombSub real-time streams (0.2 s blocks, SOLA splice, one stream per GPU)",
            "value": world * r.block / (mean * 1e-3), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": mean, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 storage and accumulation; at this batch (87 rows) the Linear / conv products are split-bf16x3 with "
                     "pre-split weights, attention and FIR products f32",
            "data": "synthetic",
            "config": {"workload": f"CombSub 44.1 kHz real-time: one stream per GPU, {r.n_in}-sample sliding window "
                                   f"({r.frames} frames), block {r.block}, cross-fade {r.splicer.xfade}, search {r.splicer.search}; "
                                   "window -> volume -> model (HIP-graph replay) -> gate -> SOLA per block",
                       "sharding": "replicas only (one stream per GPU, no collective)"},
            "ms_per_block_mean": mean, "ms_per_block_p99": p99, "ms_per_block_back_to_back": pace,
            "x_realtime_per_stream": 200.0 / mean,
            "timed_region_s": dt,
        }), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the synthesis path has no CPU fallback")
    # Rehearsal switch (never set by the driver): DDSP_BENCH_REHEARSAL=gloo runs the N > 1 control flow - rendezvous,
    # barriers, the overlapped all_gather, the max-over-ranks timing, rank-0 reporting, teardown - with several ranks
    # sharing the ONE GPU of a test box (ranks map to cuda:(local % device_count), collectives go through gloo).  The
    # numbers of such a run mean nothing; it exists so that the multi-rank path is executed before the 8-GPU run.
    rehearsal = os.environ.get("DDSP_BENCH_REHEARSAL", "")
    if rehearsal:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group(rehearsal)
        else:
            dist.init_process_group("nccl", device_id=dev)

    import hipddsp
    import synthetic
    import sharding

    seed = synthetic.BASE_SEED + 2
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # the constructors print the reference's banner line; stdout is for the JSON
        model, cfg = synthetic.build_model(args.model, seed=seed, device=dev)
    inp_cpu = synthetic.make_inputs(seed + 100 + rank, B_PER_GPU, FRAMES, with_noise=False)
    inp = {k: v.to(dev) for k, v in inp_cpu.items()}
    T = FRAMES * HOP
    ctx = hipddsp.context_for(dev)

    if args.mode == "realtime":
        return realtime_mode(args, model, dev, dist, world, rank)

    gather = sharding.AudioGather(world, B_PER_GPU, T, dev) if world > 1 else None

    if args.mode == "train":
        import training
        from ddsp.loss import RSSLoss
        Bt = 32
        inp = {k: v[:Bt].contiguous() for k, v in inp.items()}
        inp["audio"] = 0.1 * torch.randn(Bt, T, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + rank))
        model.train()
        opt = training.AdamW(model.parameters(), lr=5e-4, weight_decay=0.0)
        crit = RSSLoss(256, 2048, 4, device=dev)
        sg = torch.Generator().manual_seed(1234)      # the same n_fft draws on every rank
        gather = None
        bucket = training.GradBucket(model.parameters(), model)   # every .grad a view of one flat buffer: one all_reduce per step

        def step(i):
            scales = [int(v) for v in torch.randint(256, 2048, (4,), generator=sg)]
            return training.train_step(model, opt, crit, inp, world=world, scales=scales, bucket=bucket)
    else:
        Bt = B_PER_GPU

        def step(i):
            with torch.no_grad():
                sig, _, _ = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise_seed=1000 + i)
            if gather is not None:
                gather.submit(sig)
            return sig

    def timed_leg(base):
        """W warm-up steps, then exactly K steps between barrier + synchronize pairs; max over ranks."""
        for i in range(args.warmup):
            step(base + i)
        if gather is not None:
            gather.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        # HIP events around the largest families only (22 brackets = 44 records per step; created without the system-scope
        # fence: with default events 16 brackets cost 0.095 ms per step, without it 0.035 ms - measured against a run
        # without them)
        # ... and around a SAMPLE of the timed steps only (every `--profile-every`-th, default 5): an event record costs ~2 us of
        # stream time and serialises the kernels around it - 38 records per step took 6 % of the step (1.27 against 1.19 ms for
        # the same loop without them, r03_c), which is a distortion of `value`, not a property of the path
        ctx.profile_begin(list(FAMILIES_TIMED))
        every = max(1, args.profile_every)
        t0 = time.perf_counter()
        for i in range(args.steps):
            if every > 1:
                if i % every == 0:
                    ctx.profile_mask(list(FAMILIES_TIMED))
                elif i % every == 1:
                    ctx.profile_mask([])
            step(base + args.warmup + i)
        if gather is not None:
            gather.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timed = ctx.profile_end()
        if dist is not None:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, timed

    # leg 1 (`value`): the library's default product arithmetic (split-bf16x3 at this batch)
    dt, timed = timed_leg(0)
    # leg 2 (`value_fp32_mfma`): the same steps with every contraction on the fp32 matrix pipe (the reference's precision
    # class); the training step's backward is fp32 in both, so the second leg is a synth-mode figure
    dt32 = timed32 = None
    if args.mode == "synth":
        ctx.set_math(hipddsp.MATH_FP32)
        dt32, timed32 = timed_leg(100_000)
        ctx.set_math(hipddsp.MATH_SPLIT_BF16)

    # strong scaling (a FIXED 64-clip job, what a main.py / gui.py batch looks like): at N = 1 the per-GPU share of an 8-GPU
    # run of that job - 8 clips - is timed with the same protocol (the one-GPU job itself is leg 1); at N > 1 the 64 clips are
    # sharded over the ranks (64 / N each) and gathered like the weak-scaled leg
    strong = None
    if args.mode == "synth":
        shard = 8 if world == 1 else max(1, B_PER_GPU // world)
        sub = {k: v[:shard].contiguous() for k, v in inp.items()}
        g2 = sharding.AudioGather(world, shard, T, dev) if world > 1 else None
        keep_step, keep_gather = step, gather

        def step(i):   # noqa: F811
            with torch.no_grad():
                sig, _, _ = model(sub["units"], sub["f0"], sub["volume"], sub["spk_id"], noise_seed=3000 + i)
            if g2 is not None:
                g2.submit(sig)
            return sig
        gather = g2
        dts, _ = timed_leg(200_000)
        step, gather = keep_step, keep_gather
        strong = {"job": f"{B_PER_GPU if world > 1 else 64} clips x 2 s in total", "clips_per_gpu": shard,
                  "ms_per_step": dts / args.steps * 1e3}
        if world == 1:
            strong["note"] = ("N = 1: the per-GPU work of an 8-GPU strong-scaled run (8 of the 64 clips); projected speed-up = "
                              "ms_per_step of the 64-clip leg / this")
            strong["projected_speedup_8_gpus"] = (dt / args.steps * 1e3) / strong["ms_per_step"]
        else:
            strong["value"] = world * shard * T * args.steps / dts

    total_samples = world * Bt * T * args.steps
    value = total_samples / dt
    out = {
        "metric": ("audio samples/sec @44.1kHz CombSub synth" if args.model == "CombSub" else
                   f"audio samples/sec @44.1kHz {args.model} synth") if args.mode == "synth" else
                  f"audio samples/sec @44.1kHz {args.model} training step (fwd + RSS loss + bwd + AdamW)",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f32 storage and accumulation; products of the GEMM / FIR contractions split-bf16x3 (hi*hi+hi*lo+lo*hi), of the "
                  "attention split-bf16 from three pieces (6 products) for the feature projections and two (3 products) for "
                  "the context sums; everything else f32; `value_fp32_mfma` is the leg with every product in f32"
                  ) if args.mode == "synth"
        else "f32 storage and accumulation; forward, spectral loss, attention / FIR adjoints and optimizer in f32 products; "
             "weight and input gradients of the Linear / conv layers split-bf16x3 (hi*hi+hi*lo+lo*hi)",
        "data": "synthetic",
        "config": {"workload": f"{args.model} 44.1 kHz, batch={Bt}x2 s per GPU (Fr={FRAMES}, T={T}), "
                               + ("units/f0/volume/spk_id -> audio, seeded random weights, in-kernel noise"
                                  if args.mode == "synth" else
                                  "training step: forward(infer=False) + 4-scale RSS loss + backward + AdamW"),
                   "sharding": "utterances over ranks; all_gather of rendered audio overlapped on a side stream"
                   if world > 1 else "single GPU"},
        "x_realtime": value / SR,
        "hbm_frac_api_bytes": value * BYTES_PER_SAMPLE_API / 8e12,
    }
    if dt32 is not None:
        out["value_fp32_mfma"] = total_samples / dt32
        out["ms_per_step_fp32_mfma"] = dt32 / args.steps * 1e3
    if strong is not None:
        out["strong_scaling"] = strong
    # untimed breakdown pass: every kernel family bracketed by events (adds launch gaps, so not part of `value`)
    ctx.profile_begin()
    nb = 3
    for i in range(nb):
        step(10_000 + i)
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    fam = ctx.profile_end()
    if rank == 0:
        sampled_steps = len(range(0, args.steps, max(1, args.profile_every)))

        def roofline_of(name, timed, split):
            d = timed[name]
            avg_ms = d["ms_total"] / d["launches"]
            alg = d["flops_total"] / d["launches"] / (avg_ms * 1e-3) / 1e12
            on_bf16 = split and name in SPLIT_FAMILIES and (args.mode == "synth" or name == "ltv_fir")
            r = {"kernel": KERNEL_LABEL[name][0 if on_bf16 else 1], "bound": "mfma", "unit": "TFLOP/s",
                 "traffic": measured_traffic(name, split), "avg_launch_ms": avg_ms, "launches": d["launches"],
                 "ms_per_step": d["ms_total"] / sampled_steps,
                 "sampling": f"HIP events bracket every {max(1, args.profile_every)}th of the {args.steps} timed steps ({sampled_steps} steps)",
                 "algorithmic_bytes_per_launch": d["bytes_total"] / d["launches"],
                 "algorithmic_flops_per_launch": d["flops_total"] / d["launches"]}
            if on_bf16:
                r.update({"arithmetic": SPLIT_ARITH[name], "achieved": alg, "peak": PEAK_MFMA_BF16_TFLOPS,
                          "frac": alg / PEAK_MFMA_BF16_TFLOPS, "mfma_issue_frac": ISSUE_MULT[name] * alg / PEAK_MFMA_BF16_TFLOPS})
            else:
                r.update({"arithmetic": "fp32 MFMA", "achieved": alg, "peak": PEAK_MFMA_F32_TFLOPS,
                          "frac": alg / PEAK_MFMA_F32_TFLOPS})
            return r
        present = [f for f in FAMILIES_TIMED if f in timed]
        order = sorted(present, key=lambda f: -timed[f]["ms_total"])
        out["roofline"] = roofline_of(order[0], timed, True)
        out["roofline_others"] = [roofline_of(f, timed, True) for f in order[1:]]
        if timed32 is not None:
            out["roofline_fp32_mfma_leg"] = [roofline_of(f, timed32, False)
                                             for f in sorted(present, key=lambda f: -timed32[f]["ms_total"])]
        out["kernel_families_ms_per_step"] = {k: round(v["ms_total"] / nb, 4) for k, v in fam.items()}
        out["kernel_families_tflops"] = {k: round(v["flops_total"] / (v["ms_total"] * 1e-3) / 1e12, 2)
                                         for k, v in fam.items() if v["flops_total"] > 0}
        # algorithmic bytes (what each launch must read + write once) over the family's event time: the figure to hold
        # against the 8 TB/s HBM roofline for the families that are HBM-bound (scan, activations, row kernels, prep)
        out["kernel_families_gbps_algorithmic"] = {k: round(v["bytes_total"] / (v["ms_total"] * 1e-3) / 1e9, 1)
                                                   for k, v in fam.items() if v["bytes_total"] > 0 and v["ms_total"] > 0}
        if world == 1 and not args.no_cpu_baseline and args.mode == "synth" and args.model == "CombSub":
            cpu_sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            out["parity_vs_oracle"] = parity_sample(cpu_sd, cfg, model, inp, inp_cpu, seed, ctx, dev)
            out["cpu_baseline"] = cpu_baseline(seed, inp_cpu)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
