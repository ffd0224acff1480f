"""GPU parity of the control network (a4) and of the full model forwards (a12) through the drop-in classes.

Gate (BASELINE.json north_star): waveform within 1e-4 RMS of the reference PyTorch CPU path on identical
f0 / units / volume / spk_id inputs (noise draw injected)."""
import os

import numpy as np
import pytest
import torch

import synthetic
from conftest import GOLDEN, rms
from oracle import ctrlnet as OC
from oracle import synth as OS

pytestmark = pytest.mark.gpu
HOP = 512
GATE = 1e-4


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def _to(d, dev):
    return {k: v.to(dev) for k, v in d.items()}


def test_unit2ctrl_large_batch_fused_glu_matches_oracle(dev, lib_path):
    """Inference forms the conformer's GLU inside the pw1 GEMM (gated-pair epilogue on re-ordered weights, unit2ctrl.hip):
    from 8065 rows on with 128x128 tiles (and pre-split activations), below with 64x128 tiles.  B*Fr = 8256 rows against
    the oracle at the same tolerances as the small cases, plus a check (launch counts of the row-kernel family) that the
    fused path is the one that ran at both sizes."""
    import hipddsp
    model, cfg = synthetic.build_model("CombSub", seed=99)
    sd = {k[len("unit2ctrl."):]: v for k, v in model.state_dict().items() if k.startswith("unit2ctrl.")}
    B, Fr = 48, 172
    inp = synthetic.make_inputs(77, B, Fr, with_noise=False)
    phase = torch.from_numpy(np.random.Generator(np.random.PCG64(6)).uniform(-np.pi, np.pi, (B, Fr)).astype(np.float32))
    with torch.no_grad():
        want = OC.unit2control(sd, inp["units"], inp["f0"], phase, inp["volume"], inp["spk_id"], None,
                               model.unit2ctrl.output_splits, return_flat=True)
    model = model.to(dev).eval()
    ctx = hipddsp.context_for(dev)

    def run(nb):
        args = [inp[k][:nb].to(dev) for k in ("units", "f0")] + [phase[:nb].to(dev), inp["volume"][:nb].to(dev),
                                                                 inp["spk_id"][:nb].to(dev)]
        ctx.profile_begin(["u2c_rowwise"])
        with torch.no_grad():
            out = model.unit2ctrl.forward_flat(*args, None)
        return out.cpu(), ctx.profile_end()["u2c_rowwise"]["launches"]

    got, launches_big = run(B)
    small, launches_small = run(2)
    # (round 3: small batches form the GLU inside the pw1 GEMM too, on 64x128 tiles - no separate glu kernel at any size; and
    # from 8192 rows on all seven LayerNorms ride in the GEMMs in front of them - the six behind the out-projection / pw2 layers
    # and the first one, behind the second prenet convolution: csrc/gemm_ln.h)
    assert launches_small == 12 and launches_big == launches_small - 7, (launches_big, launches_small)
    assert got.shape == want.shape
    assert (got - want).abs().max() < 2e-4
    assert rms(got - want) < 2e-5
    # the unfused small batch agrees with the fused large one on the rows they share
    assert (small - got[:2]).abs().max() < 2e-5


# (1, 87), (2, 12), (2, 128) and (1, 250): at most 256 rows - the residual GEMMs and the prenet convolutions run as K-split
# launches whose partial products the following LayerNorm / GroupNorm pass sums (256 rows exactly, and a ragged last tile);
# (3, 172) and (1, 257): just above, whole-K launches
@pytest.mark.parametrize("B,Fr", [(2, 12), (3, 172), (1, 87), (2, 128), (1, 250), (1, 257)])
@pytest.mark.parametrize("spk_mode", ["per_row", "broadcast", "mix"])
def test_unit2ctrl_matches_oracle(dev, lib_path, B, Fr, spk_mode):
    model, cfg = synthetic.build_model("CombSub", seed=99)
    sd = {k[len("unit2ctrl."):]: v for k, v in model.state_dict().items() if k.startswith("unit2ctrl.")}
    inp = synthetic.make_inputs(1234 + B, B, Fr, with_noise=False)
    phase = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).uniform(-np.pi, np.pi, (B, Fr)).astype(np.float32))
    spk = inp["spk_id"] if spk_mode == "per_row" else inp["spk_id"][:1]
    mix = {3: 0.5, 10: 0.2, 99: 0.3} if spk_mode == "mix" else None
    with torch.no_grad():
        want = OC.unit2control(sd, inp["units"], inp["f0"], phase, inp["volume"], spk, mix,
                               model.unit2ctrl.output_splits, return_flat=True)
    model = model.to(dev)
    got = model.unit2ctrl.forward_flat(inp["units"].to(dev), inp["f0"].to(dev), phase.to(dev), inp["volume"].to(dev),
                                       spk.to(dev), mix).cpu()
    assert got.shape == want.shape
    err = (got - want).abs().max()
    assert err < 2e-4, err               # ctrl ~ N(0, 0.6^2); fp32 chains of depth 256..768
    assert rms(got - want) < 2e-5
    d = model.unit2ctrl(inp["units"].to(dev), inp["f0"].to(dev), phase.to(dev), inp["volume"].to(dev), spk.to(dev), mix)
    assert list(d.keys()) == ["group_delay", "harmonic_magnitude", "noise_magnitude"]
    assert [v.shape[-1] for v in d.values()] == [256, 512, 256]


CASES = [("infer", dict(infer=True)), ("train", dict(infer=False)),
         ("mix", dict(infer=True, spk_mix_dict={1: 0.25, 7: 0.75})),
         ("init", dict(infer=True, initial_phase=torch.tensor([1.0, -2.0])))]


def test_combsub_against_reference_golden(dev, lib_path):
    """Fixture produced by the reference's own CombSub.forward (tests/golden/make_golden.py, tier B)."""
    g = load("model_CombSub.npz")
    model, cfg = synthetic.build_model("CombSub", seed=g["seed_weights"], device=dev)
    inp = _to(synthetic.make_inputs(g["seed_inputs"], 2, 12), dev)
    for tag, kw in CASES:
        kw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
        with torch.no_grad():
            sig, ph, (hm, nz) = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=inp["noise"], **kw)
        assert sig.shape == (2, 12 * HOP) and ph.shape == (2, 12, 1)
        # every mode at the north_star gate, `infer=False` included: the train-mode running phase is the fp64 sum of the
        # same fp32 increments rounded to fp32 per sample (ATen's CPU cumsum), so it is reproduced, not approximated
        e_sig, e_hm = rms(sig.cpu() - g[f"signal_{tag}"]), rms(hm.cpu() - g[f"harmonic_{tag}"])
        assert e_sig < GATE and e_hm < GATE, (tag, e_sig, e_hm)
        assert rms(nz.cpu() - g[f"noise_{tag}"]) < GATE
        dp = (ph.cpu() - g[f"phase_{tag}"]) / (2 * np.pi)
        assert (dp - torch.round(dp)).abs().max() < 1e-6, (tag, float((dp - torch.round(dp)).abs().max()))


@pytest.mark.parametrize("B,Fr,infer", [(4, 172, True), (2, 173, True), (2, 87, False), (3, 172, False)])
def test_combsub_against_oracle(dev, lib_path, B, Fr, infer):
    model, cfg = synthetic.build_model("CombSub", seed=7)
    sd = model.state_dict()
    inp = synthetic.make_inputs(4242 + Fr, B, Fr)
    with torch.no_grad():
        sig_o, ph_o, (hm_o, nz_o), aux = OS.combsub_forward(sd, cfg, inp["units"], inp["f0"], inp["volume"],
                                                            inp["spk_id"], infer=infer, noise=inp["noise"])
    model = model.to(dev)
    d = _to(inp, dev)
    with torch.no_grad():
        sig, ph, (hm, nz) = model(d["units"], d["f0"], d["volume"], d["spk_id"], infer=infer, noise=d["noise"])
    assert rms(nz.cpu() - nz_o) < GATE
    # train mode (`infer=False`, solver.py:111) rounds the running phase to fp32 per sample like ATen's CPU cumsum
    # (core.py:31-51): held to the same gate as inference, and the frame phases to the wrap tolerance
    e_hm, e_sig = rms(hm.cpu() - hm_o), rms(sig.cpu() - sig_o)
    assert e_hm < GATE and e_sig < GATE, (infer, e_hm, e_sig)
    dp = (ph.cpu() - ph_o) / (2 * np.pi)
    flipped = float(((dp - torch.round(dp)).abs() > 1e-6).float().mean())
    assert flipped == 0.0, (infer, flipped, float((dp - torch.round(dp)).abs().max()))
    # returned signal is a fresh writable tensor (callers multiply a mask in place, main.py:159)
    sig *= 0.5
    # in-kernel noise path: repeatable under torch.manual_seed, finite, and harmonic part unchanged
    with torch.no_grad():       # (with grad mode on the call is recorded for training and its control network runs fp32 products)
        torch.manual_seed(3)
        s1, _, (h1, n1) = model(d["units"], d["f0"], d["volume"], d["spk_id"], infer=infer)
        torch.manual_seed(3)
        s2, _, (h2, n2) = model(d["units"], d["f0"], d["volume"], d["spk_id"], infer=infer)
    assert torch.equal(s1, s2) and torch.equal(h1, hm) and torch.isfinite(s1).all()
    assert 0.5 < rms(n1) / rms(nz) < 2.0


def test_no_cpu_fallback(lib_path):
    model, cfg = synthetic.build_model("CombSub", seed=7)
    inp = synthetic.make_inputs(1, 1, 4)
    with pytest.raises(RuntimeError):
        model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"])


# ---- Sins and CombSubFast ------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["Sins", "CombSubFast"])
def test_other_models_against_reference_golden(dev, lib_path, name):
    g = load(f"model_{name}.npz")
    model, cfg = synthetic.build_model(name, seed=g["seed_weights"], device=dev)
    inp = synthetic.make_inputs(g["seed_inputs"], 2, 12)
    if name == "Sins":
        inp["f0"][0, 3, 0] = 700.0
    inp = _to(inp, dev)
    for tag, kw in CASES:
        kw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in kw.items()}
        with torch.no_grad():
            sig, ph, (hm, nz) = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=inp["noise"], **kw)
        assert sig.shape == (2, 12 * HOP)
        if name == "Sins":
            assert ph.shape == (2, 12 * HOP, 1)           # sample-rate phase (reference vocoder.py:423)
            ph = ph[:, ::HOP]
        else:
            assert ph.shape == (2, 12, 1) and hm is sig and nz is sig    # same tensor three times (:492)
        dp = (ph.cpu() - g[f"phase_{tag}"]) / (2 * np.pi)
        assert (dp - torch.round(dp)).abs().max() < 1e-6, (name, tag, float((dp - torch.round(dp)).abs().max()))
        err = rms(sig.cpu() - g[f"signal_{tag}"])            # `infer=False` included (train-mode phase rounding reproduced)
        assert err < GATE, (name, tag, err)
        if name == "Sins":
            assert rms(hm.cpu() - g[f"harmonic_{tag}"]) < GATE and rms(nz.cpu() - g[f"noise_{tag}"]) < GATE


@pytest.mark.parametrize("name,B,Fr", [("Sins", 3, 172), ("Sins256", 2, 87), ("CombSubFast", 3, 172),
                                       ("CombSubFast", 2, 173)])
def test_other_models_against_oracle(dev, lib_path, name, B, Fr):
    model, cfg = synthetic.build_model(name, seed=21)
    sd = model.state_dict()
    inp = synthetic.make_inputs(777 + Fr, B, Fr)
    with torch.no_grad():
        sig_o, ph_o, _, aux = OS.FORWARD[cfg["type"]](sd, cfg, inp["units"], inp["f0"], inp["volume"], inp["spk_id"],
                                                      noise=inp["noise"])
    model = model.to(dev)
    d = _to(inp, dev)
    with torch.no_grad():
        sig, ph, _ = model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])
    err = rms(sig.cpu() - sig_o)
    assert err < GATE, (name, err, rms(sig_o))
    torch.manual_seed(1)
    s1 = model(d["units"], d["f0"], d["volume"], d["spk_id"])[0]
    torch.manual_seed(1)
    s2 = model(d["units"], d["f0"], d["volume"], d["spk_id"])[0]
    assert torch.equal(s1, s2) and torch.isfinite(s1).all()


def test_sins_bank_stage(ctx, dev):
    """Bank alone (BASELINE config #3 'additive-only', H=256) against the oracle, incl. harmonics crossing sr/2."""
    from oracle import dsp as O
    B, Fr, H = 2, 40, 256
    rng = np.random.Generator(np.random.PCG64(8))
    ctrl = torch.from_numpy((rng.standard_normal((B, Fr, H + 8)) * 0.5).astype(np.float32))
    f0f = torch.from_numpy(rng.uniform(65, 800, (B, Fr, 1)).astype(np.float32))
    f0 = O.frames_to_samples(f0f, HOP).squeeze(-1)
    phase = 2 * np.pi * O.rotation_from_f0(f0, 44100, None, True)
    amps = O.mask_above_nyquist(torch.exp(ctrl[..., 4:4 + H]) / 128, f0f, 44100 / 2)
    want = O.harmonic_bank(amps, phase, HOP)
    got = ctx.sins_bank(ctrl.reshape(B * Fr, -1).to(dev), 4, H, f0f.to(dev), phase.to(dev), B, Fr, HOP, 44100).cpu()
    assert rms(got - want) < 2e-6 and (got - want).abs().max() < 2e-5, rms(got - want)


# ---- edge shapes ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["CombSub", "Sins", "CombSubFast"])
@pytest.mark.parametrize("B,Fr", [(1, 1), (1, 3), (2, 33)])
def test_tiny_and_ragged_shapes(dev, lib_path, name, B, Fr):
    model, cfg = synthetic.build_model(name, seed=3)
    sd = model.state_dict()
    inp = synthetic.make_inputs(9000 + Fr, B, Fr)
    with torch.no_grad():
        want = OS.FORWARD[cfg["type"]](sd, cfg, inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=inp["noise"])[0]
    model = model.to(dev)
    d = _to(inp, dev)
    with torch.no_grad():
        got = model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])[0]
    assert got.shape == (B, Fr * HOP)
    assert rms(got.cpu() - want) < GATE, (name, B, Fr, rms(got.cpu() - want))


def test_empty_batch_and_wide_units(dev, lib_path):
    from ddsp.vocoder import CombSubFast
    model, cfg = synthetic.build_model("CombSubFast", seed=3, device=dev)
    with torch.no_grad():
        out, ph, _ = model(torch.zeros(0, 5, 256, device=dev), torch.zeros(0, 5, 1, device=dev),
                           torch.zeros(0, 5, device=dev), torch.ones(1, 1, dtype=torch.long, device=dev))
    assert out.shape == (0, 5 * HOP) and ph.shape == (0, 5, 1)
    # 768-wide units (contentvec768 / hubertbase768 encoders of the reference configs), single speaker
    torch.manual_seed(4)
    wide = CombSubFast(44100, 512, n_unit=768, n_spk=1)
    sd = wide.state_dict()
    cfg768 = dict(type="CombSubFast", sampling_rate=44100, block_size=512, n_unit=768, n_spk=1)
    inp = synthetic.make_inputs(77, 2, 20, n_unit=768, n_spk=1)
    with torch.no_grad():
        want = OS.combsubfast_forward(sd, cfg768, inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=inp["noise"])[0]
        d = _to(inp, dev)
        got = wide.to(dev)(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])[0]
    assert rms(got.cpu() - want) < GATE


@pytest.mark.parametrize("name", ["CombSub", "Sins", "CombSubFast"])
def test_full_bench_batch_equals_its_shards(dev, lib_path, name):
    """Size-independent property at the bench size (BASELINE configs[1]: 64 clips x 172 frames): every utterance is
    independent, so rendering the whole batch and rendering it in shards of 8 (what `--gpus 8` does, and small enough
    to take the small-batch kernel choices: register-staged convs, separate GLU kernel, 128x64 / 64x64 tiles) must
    give the same audio.  Also: the same call twice gives the same bits (no atomics, fixed reduction orders)."""
    model, cfg = synthetic.build_model(name, seed=5, device=dev)
    model.eval()
    B, Fr = 64, 172
    inp = {k: v.to(dev) for k, v in synthetic.make_inputs(900, B, Fr, with_noise=False).items()}
    noise = torch.rand(B, Fr * 512, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    with torch.no_grad():
        full = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=noise)[0]
        again = model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=noise)[0]
        assert torch.equal(full, again)
        scale = max(1e-3, rms(full))
        for lo in range(0, B, 8):
            sl = slice(lo, lo + 8)
            part = model(inp["units"][sl], inp["f0"][sl], inp["volume"][sl], inp["spk_id"][sl], noise=noise[sl])[0]
            assert part.shape == (8, Fr * 512)
            err = rms(part - full[sl])
            # The 64-clip batch runs the control network's GEMMs in the split-bf16 mode (3 bf16 MFMAs per fp32 product,
            # ~4e-6 relative error per GEMM); shards of 8 are below the DMA kernels' size thresholds and run fp32 MFMA.
            # Measured difference: ~5e-6 RMS on a 0.07-RMS waveform; asserted at a fifth of the 1e-4 north-star gate.
            assert err < 2e-5 and err < 4e-4 * scale, (name, lo, err, scale)
    assert float(full.abs().max()) > 0 and torch.isfinite(full).all()


def test_full_bench_batch_against_oracle(dev, lib_path):
    """The north-star gate at the bench size itself: CombSub, 64 clips x 172 frames, GPU (split-bf16 GEMMs, fused GLU,
    DMA convs, 128x128 / 64x64 tiles) against the CPU oracle on identical inputs and injected noise: <= 1e-4 RMS."""
    model, cfg = synthetic.build_model("CombSub", seed=7)
    sd = model.state_dict()
    B, Fr = 64, 172
    inp = synthetic.make_inputs(901, B, Fr)
    with torch.no_grad():
        sig_o, ph_o, (hm_o, nz_o), aux = OS.combsub_forward(sd, cfg, inp["units"], inp["f0"], inp["volume"],
                                                            inp["spk_id"], infer=True, noise=inp["noise"])
    model = model.to(dev).eval()
    d = _to(inp, dev)
    with torch.no_grad():
        sig, ph, (hm, nz) = model(d["units"], d["f0"], d["volume"], d["spk_id"], infer=True, noise=d["noise"])
    errs = (rms(sig.cpu() - sig_o), rms(hm.cpu() - hm_o), rms(nz.cpu() - nz_o))
    assert max(errs) < GATE, errs
    assert rms(sig_o) > 1e-3


def test_full_bench_batch_both_math_modes(dev, lib_path):
    """`ddsp_ctx_set_math`: the 64 x 172 CombSub batch with every contraction on the fp32 matrix pipe (the reference's
    precision class) and with split-bf16x3 products (the default), both against the CPU oracle on 8 of the 64 clips.  The
    fp32 mode must be clearly tighter than the gate; the two modes differ from each other by what the split costs."""
    import hipddsp
    model, cfg = synthetic.build_model("CombSub", seed=9)
    sd = model.state_dict()
    B, Fr = 64, 172
    inp = synthetic.make_inputs(902, B, Fr)
    pick = list(range(3, B, 8))
    with torch.no_grad():
        want = OS.combsub_forward(sd, cfg, inp["units"][pick], inp["f0"][pick], inp["volume"][pick],
                                  inp["spk_id"][pick], infer=True, noise=inp["noise"][pick])[0]
    model = model.to(dev).eval()
    d = _to(inp, dev)
    ctx = hipddsp.context_for(dev)
    assert ctx.math == hipddsp.MATH_SPLIT_BF16                     # the library's default
    got = {}
    try:
        for mode in (hipddsp.MATH_FP32, hipddsp.MATH_SPLIT_BF16):
            ctx.set_math(mode)
            assert ctx.math == mode
            with torch.no_grad():
                got[mode] = model(d["units"], d["f0"], d["volume"], d["spk_id"], infer=True, noise=d["noise"])[0]
    finally:
        ctx.set_math(hipddsp.MATH_SPLIT_BF16)
    e32 = rms(got[hipddsp.MATH_FP32][pick].cpu() - want)
    e16 = rms(got[hipddsp.MATH_SPLIT_BF16][pick].cpu() - want)
    assert e32 < 2e-5 and e16 < GATE, (e32, e16)
    assert e32 <= e16 * 1.05, (e32, e16)
    diff = rms(got[hipddsp.MATH_FP32] - got[hipddsp.MATH_SPLIT_BF16])
    assert 0 < diff < 5e-5, diff                                    # different kernels did run, and agree
    with pytest.raises(ValueError):
        ctx.set_math(2)


def test_sins256_bench_batch_against_oracle(dev, lib_path):
    """BASELINE config #3 at its full size: Sins with 256 harmonics, 64 clips x 172 frames (the large-batch kernel
    choices of the control network, the bank and the FIRs), 8 of the 64 clips against the CPU oracle."""
    model, cfg = synthetic.build_model("Sins256", seed=11)
    sd = model.state_dict()
    B, Fr = 64, 172
    inp = synthetic.make_inputs(903, B, Fr)
    inp["f0"][5, 40:60, 0] = 700.0                                  # harmonics crossing sr/2 inside the bank
    pick = list(range(5, B, 8))
    with torch.no_grad():
        sig_o, ph_o, (hm_o, nz_o), _ = OS.sins_forward(sd, cfg, inp["units"][pick], inp["f0"][pick],
                                                       inp["volume"][pick], inp["spk_id"][pick], infer=True,
                                                       noise=inp["noise"][pick])
    model = model.to(dev).eval()
    d = _to(inp, dev)
    with torch.no_grad():
        sig, ph, (hm, nz) = model(d["units"], d["f0"], d["volume"], d["spk_id"], infer=True, noise=d["noise"])
    assert sig.shape == (B, Fr * 512) and ph.shape == (B, Fr * 512, 1)
    errs = (rms(sig[pick].cpu() - sig_o), rms(hm[pick].cpu() - hm_o), rms(nz[pick].cpu() - nz_o))
    assert max(errs) < GATE, errs
    assert rms(sig_o) > 1e-3


def test_speaker_id_out_of_range_is_reported(dev, lib_path):
    """a speaker id outside [1, n_spk] must neither read outside the table (forward) nor write outside the
    gradient table (backward); the reference's nn.Embedding raises.  The kernels skip the access and raise a device-side
    flag that the next call (or Context.poll_error) turns into ValueError."""
    import hipddsp
    model, cfg = synthetic.build_model("CombSub", seed=5, device=dev)
    B, Fr = 3, 12
    d = _to(synthetic.make_inputs(6, B, Fr), dev)
    ctx = hipddsp.context_for(dev)
    with torch.no_grad():
        good = model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])[0]
    for bad in (0, cfg["n_spk"] + 1, -7, 10 ** 12):
        ids = d["spk_id"].clone()
        ids[1, 0] = bad
        with torch.no_grad():
            out = model(d["units"], d["f0"], d["volume"], ids, noise=d["noise"])[0]
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        assert torch.equal(out[0], good[0]) and torch.equal(out[2], good[2])      # the other rows are untouched
        with pytest.raises(ValueError, match="spk_id"):
            ctx.poll_error()
        ctx.poll_error()                                                            # reported once, then clear
    # the next unit2ctrl call reports a flag nobody polled
    ids = d["spk_id"].clone()
    ids[0, 0] = 0
    with torch.no_grad():
        model(d["units"], d["f0"], d["volume"], ids, noise=d["noise"])
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="spk_id"):
        model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])
    # backward: no write outside the gradient table.  The id is changed between forward and backward (the node keeps the
    # caller's tensor), so that the table-gradient kernel meets it
    model.train()
    ids = d["spk_id"].clone()
    sig = model(d["units"], d["f0"], d["volume"], ids, infer=False, noise=d["noise"])[0]
    keep = int(ids[2, 0])
    ids[2, 0] = cfg["n_spk"] + 1
    sig.square().mean().backward()
    torch.cuda.synchronize()
    g = model.unit2ctrl.spk_embed.weight.grad
    assert torch.isfinite(g).all()
    rows_hit = (g.abs().sum(dim=1) > 0).nonzero().flatten().tolist()
    assert sorted(rows_hit) == sorted({int(ids[0, 0]) - 1, int(ids[1, 0]) - 1} - {keep - 1} | ({keep - 1} & {int(ids[0, 0]) - 1, int(ids[1, 0]) - 1})), rows_hit
    with pytest.raises(ValueError, match="spk_id"):
        ctx.poll_error()
    # a forward that raised the flag makes the backward of the same step fail loudly
    ids[2, 0] = 0
    sig = model(d["units"], d["f0"], d["volume"], ids, infer=False, noise=d["noise"])[0]
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="spk_id"):
        sig.square().mean().backward()


def test_presplit_operands_give_the_same_bits(dev, lib_path):
    """Round 2: at large batches the producers write the GEMM A operands as bf16 hi/lo groups and the preparation launch
    the weights, so the split-bf16 GEMM loops split nothing.  The rounding is the same as the in-kernel split (mode 4 of
    ddsp_ctx_set_math keeps that path as a measurement aid), so the control matrix must be IDENTICAL bit for bit."""
    import hipddsp
    for name, B, Fr in (("CombSub", 64, 172), ("CombSubFast", 48, 172), ("Sins", 33, 250)):
        model, cfg = synthetic.build_model(name, seed=21, device=dev)
        inp = {k: v.to(dev) for k, v in synthetic.make_inputs(31, B, Fr, with_noise=False).items()}
        c = hipddsp.context_for(dev)
        ps = c.phase_scan(inp["f0"], 512, 44100)
        outs = {}
        try:
            for mode in (4, hipddsp.MATH_SPLIT_BF16):
                c.set_math(mode)
                with torch.no_grad():
                    outs[mode] = model.unit2ctrl.forward_flat(inp["units"], inp["f0"], ps["phase_frames"], inp["volume"],
                                                              inp["spk_id"], None)
        finally:
            c.set_math(hipddsp.MATH_SPLIT_BF16)
        assert torch.equal(outs[4], outs[hipddsp.MATH_SPLIT_BF16]), name
        assert torch.isfinite(outs[4]).all()


@pytest.mark.parametrize("name,Fr", [("CombSub", 2600), ("Sins", 2600), ("CombSubFast", 2600), ("CombSub", 10000)])
def test_long_utterance_against_oracle(dev, lib_path, name, Fr):
    """Validation renders whole files at batch 1 (solver.py:9-82): 30 s (2 600 frames) for all three models and 116 s
    (10 000 frames) for CombSub against the CPU oracle.  GroupNorm statistics, the Performer key sums, the frame prefix of
    the phase scan and the FIR's marching runs all grow with the number of frames."""
    import os
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    model, cfg = synthetic.build_model(name, seed=23)
    sd = model.state_dict()
    inp = synthetic.make_inputs(600 + Fr, 1, Fr)
    with torch.no_grad():
        sig_o, ph_o, (hm_o, nz_o), _ = OS.FORWARD[cfg["type"]](sd, cfg, inp["units"], inp["f0"], inp["volume"], inp["spk_id"],
                                                               infer=True, noise=inp["noise"])
    model = model.to(dev).eval()
    d = _to(inp, dev)
    with torch.no_grad():
        sig, ph, (hm, nz) = model(d["units"], d["f0"], d["volume"], d["spk_id"], infer=True, noise=d["noise"])
    assert sig.shape == (1, Fr * 512)
    assert rms(sig.cpu() - sig_o) < GATE, (name, Fr, rms(sig.cpu() - sig_o), rms(sig_o))
    assert rms(hm.cpu() - hm_o) < GATE and rms(nz.cpu() - nz_o) < GATE
    # the phase keeps its precision over the whole file (fp64 running sum): compare the wrapped difference
    dph = (ph.cpu() - ph_o).double() / (2 * np.pi)
    dph = dph - torch.round(dph)
    assert float(dph.abs().max()) < 2e-5, float(dph.abs().max())
    assert rms(sig_o) > 1e-3


def test_causal_mode_against_reference_fixture(dev, lib_path):
    """`c: true`, inference: causal conv taps, causal depthwise conv and the causal linear attention on
    the device against the reference's CombSub(c=True) run (fixture; the two third-party causal primitives were stand-ins
    there) and against the oracle at a full-size batch (the pre-split / DMA kernel choices) and at batch 1."""
    from ddsp.vocoder import CombSub
    z = np.load(os.path.join(GOLDEN, "model_CombSub_causal.npz"))
    ref_model, cfg = synthetic.build_model("CombSub", seed=int(z["seed_weights"]))
    model = CombSub(44100, 512, cfg["n_mag_allpass"], cfg["n_mag_harmonic"], cfg["n_mag_noise"], 256, cfg["n_spk"], c=True)
    model.load_state_dict(ref_model.state_dict(), strict=True)
    model = model.to(dev).eval()
    inp = synthetic.make_inputs(int(z["seed_inputs"]), 2, 24)
    d = _to(inp, dev)
    with torch.no_grad():
        sig, ph, (hm, nz) = model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])
    want = torch.from_numpy(z["signal"])
    assert rms(sig.cpu() - want) < GATE, rms(sig.cpu() - want)
    assert rms(sig.cpu() - want) < 2e-4 * rms(want)
    ccfg = dict(cfg, c=True)
    for B, Fr in ((48, 172), (1, 300)):
        inp = synthetic.make_inputs(950 + B, B, Fr)
        pick = list(range(0, B, max(1, B // 4)))
        with torch.no_grad():
            sig_o = OS.combsub_forward(ref_model.state_dict(), ccfg, inp["units"][pick], inp["f0"][pick], inp["volume"][pick],
                                       inp["spk_id"][pick], noise=inp["noise"][pick])[0]
        d = _to(inp, dev)
        with torch.no_grad():
            sig = model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])[0]
        assert rms(sig[pick].cpu() - sig_o) < GATE, (B, Fr, rms(sig[pick].cpu() - sig_o))
    # training: the whole model's gradients against autograd through the oracle (exact-phase mode, like the non-causal test
    # of tests/test_gpu_training.py; the network's own backward is checked parameter by parameter in test_gpu_backward.py)
    model.train()
    inp = synthetic.make_inputs(71, 2, 40)
    d = _to(inp, dev)
    sig = model(d["units"], d["f0"], d["volume"], d["spk_id"], infer=True, noise=d["noise"])[0]
    gsig = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(sig.shape)).astype(np.float32)) / sig.numel()
    (sig * gsig.to(dev)).sum().backward()
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "projection_matrix" not in k and v.dim() > 0)
          for k, v in ref_model.state_dict().items()}
    sig_o = OS.combsub_forward(sd, ccfg, inp["units"], inp["f0"], inp["volume"], inp["spk_id"], noise=inp["noise"])[0]
    assert rms(sig.detach().cpu() - sig_o.detach()) < GATE
    (sig_o * gsig).sum().backward()
    errs = []
    for n, p in model.named_parameters():
        want = sd[n].grad
        errs.append((float((p.grad.cpu().double() - want.double()).norm() / (want.double().norm() + 1e-30)), n))
    errs.sort(reverse=True)
    assert errs[0][0] < 5e-3 and sum(e for e, _ in errs) / len(errs) < 1e-3, errs[:5]


def test_stream_switch_under_a_cached_context(dev, lib_path):
    """`context_for` caches one context per (device, thread) while calls launch on the CURRENT torch stream.  A
    caller that switches streams must not meet a half-built table or a scratch arena another stream still reads: the
    context makes the new stream wait for what it queued on the old one.  Fresh shapes (so that tables and the arena are
    (re)built right before the switch), alternating streams, results equal to the single-stream run."""
    model, cfg = synthetic.build_model("CombSub", seed=9, device=dev)
    d = _to(synthetic.make_inputs(77, 5, 61), dev)
    with torch.no_grad():
        want = model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])[0].clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    outs = []
    for i in range(4):
        if i % 2 == 0:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                outs.append(model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])[0])
        else:
            with torch.no_grad():
                outs.append(model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])[0])
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)


# ---- the shipped encoder widths (configs/combsub_xunit.yaml:11 encoder_out_channels 4, combsub_yunit.yaml:11: 512) ----
@pytest.mark.parametrize("n_unit", [4, 512])
@pytest.mark.parametrize("B", [1, 48])
def test_combsubfast_shipped_encoder_widths(dev, lib_path, n_unit, B):
    """`configs/combsub_xunit.yaml` and `combsub_yunit.yaml` (both `type: CombSubFast`): n_unit = 4 takes the prenet's
    register-staged conv1 (K = 12, no LDS-DMA, no pre-split weights), n_unit = 512 the DMA path with K = 1536; B = 1 runs
    the small-batch K-split launches, B = 48 (8256 rows) the large-batch tilings with the fused GLU."""
    Fr = 172
    model, cfg = synthetic.build_model("CombSubFast", seed=31 + n_unit, n_unit=n_unit)
    sd = model.state_dict()
    assert sd["unit2ctrl.unit_prenet.1.weight"].shape == (256, n_unit, 3)
    inp = synthetic.make_inputs(900 + n_unit + B, B, Fr, n_unit=n_unit)
    nb = min(B, 4)                                   # the oracle renders the first clips (they are independent)
    with torch.no_grad():
        sig_o = OS.combsubfast_forward(sd, cfg, inp["units"][:nb], inp["f0"][:nb], inp["volume"][:nb], inp["spk_id"][:nb],
                                       noise=inp["noise"][:nb])[0]
    model = model.to(dev)
    d = _to(inp, dev)
    with torch.no_grad():
        sig = model(d["units"], d["f0"], d["volume"], d["spk_id"], noise=d["noise"])[0]
    assert sig.shape == (B, Fr * HOP) and torch.isfinite(sig).all()
    err = rms(sig[:nb].cpu() - sig_o)
    assert err < GATE, (n_unit, B, err, rms(sig_o))


def test_prepared_weight_cache_follows_the_weights(dev, lib_path):
    """Inference keeps the control network's prepared weights (weight norm, re-ordered / bf16 copies) in the context between
    calls (`ddsp_u2c_weights::version`): an in-place change of a parameter, a load_state_dict and a second model on the same
    context must each be seen by the next forward - bit for bit what a fresh context computes."""
    import hipddsp
    B, Fr = 4, 40
    inp = _to(synthetic.make_inputs(5, B, Fr, with_noise=False), dev)
    phase = torch.zeros(B, Fr, device=dev)

    def run(model):
        with torch.no_grad():
            return model.unit2ctrl.forward_flat(inp["units"], inp["f0"], phase, inp["volume"], inp["spk_id"], None).clone()

    def fresh(model):   # the same forward with preparation forced (grad mode on: version 0)
        with torch.enable_grad():
            w, keep = model.unit2ctrl._weights_struct()
            assert w.version == 0
            ctx = hipddsp.context_for(dev)
            return ctx.unit2ctrl(w, inp["units"], inp["f0"], phase, inp["volume"], inp["spk_id"], None, model.unit2ctrl.n_out).clone()

    m1, _ = synthetic.build_model("CombSub", seed=21, device=dev)
    m2, _ = synthetic.build_model("CombSub", seed=22, device=dev)
    a1 = run(m1)
    assert torch.equal(a1, run(m1)) and torch.equal(a1, fresh(m1))            # second call: cached copies
    a2 = run(m2)
    assert torch.equal(a2, fresh(m2)) and not torch.equal(a1, a2)
    assert torch.equal(run(m1), a1)                                            # back to the first model
    with torch.no_grad():
        m1.unit2ctrl.dec_post[2].weight_g.mul_(1.5)                            # in place: `_version` advances
    b1 = run(m1)
    assert not torch.equal(b1, a1) and torch.equal(b1, fresh(m1))
    m1.load_state_dict(m2.state_dict())
    assert torch.equal(run(m1), a2)


@pytest.mark.parametrize("causal", [0, 1])
def test_tiled_depthwise_convolution_matches_the_register_kernel(dev, lib_path, causal):
    """The LDS-tiled depthwise convolution of the large-batch forward (64 frames x 64 channels per workgroup) adds the same taps in
    the same order as the register-window kernel it replaced: the control matrices agree bit for bit, centred and causal taps
    (DDSP_DW_PAIR 3 / 1, read once per process: child processes)."""
    import subprocess, sys, os
    code = (
        "import sys, os, hashlib; sys.path.insert(0, os.path.join(%r, 'ddsp-svc-official_amd'));"
        "import torch, hipddsp, synthetic;"
        "from ddsp.vocoder import CombSub;"
        "dev = torch.device('cuda:0');"
        "model, cfg = synthetic.build_model('CombSub', seed=5, device=dev);"
        "mc = CombSub(44100, 512, cfg['n_mag_allpass'], cfg['n_mag_harmonic'], cfg['n_mag_noise'], 256, cfg['n_spk'], c=bool(%d));"
        "mc.load_state_dict(model.state_dict(), strict=True); mc = mc.to(dev).eval();"
        "inp = {k: v.to(dev) for k, v in synthetic.make_inputs(11, 48, 172, with_noise=False).items()};"
        "ps = hipddsp.context_for(dev).phase_scan(inp['f0'], 512, 44100);"
        "ctrl = mc.unit2ctrl.forward_flat(inp['units'], inp['f0'], ps['phase_frames'], inp['volume'], inp['spk_id'], None);"
        "assert torch.isfinite(ctrl).all();"
        "print(hashlib.sha256(ctrl.cpu().numpy().tobytes()).hexdigest())"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), causal)
    digests = []
    for flag in ("3", "1"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DDSP_DW_PAIR=flag), capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append(out.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1], digests
