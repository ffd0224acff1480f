"""Drop-in for the synthesiser half of the reference's `ddsp/vocoder.py` (:335-550): `load_model`, `Sins`,
`CombSub`, `CombSubFast`, `DotDict`, with the same constructors, `forward(...)` signature, return tuple and
state-dict keys, executed by hand-written gfx950 kernels through libddsp_amd (no CPU fallback).

Differences a caller can observe, all additive:
  * `forward(..., noise=None, noise_seed=None)`: the reference draws its noise excitation from the CPU
    mt19937 stream (`torch.rand_like`), which a GPU cannot reproduce.  `noise` (B,T) U[0,1) injects the
    draw (parity runs); otherwise a counter-based in-kernel generator is used, seeded from torch's default
    generator (so `torch.manual_seed` still makes a run repeatable) or from `noise_seed`.
  * `c=True` (causal convolutions + causal linear attention): forward and training; the two third-party primitives
    behind it are restated from their definitions.
"""
import os

import torch
import yaml

import hipddsp
from hipddsp import (COMB_SINC, COMB_SINC_GATED, COMB_NONE, EXC_AUDIO, EXC_GENERATE, EXC_UNIT_NOISE, FIR_ALLPASS,
                     FIR_DYNAMIC, FIR_STATIC, FIR_SPLIT_BF16)

# Product arithmetic of the frame-varying FIR in the model forwards, inference and training alike (the adjoint kernels
# work from the saved inputs in fp32 products): the context's mode (`hipddsp.Context.set_math`, default split-bf16);
# `ctx.ltv_fir` itself defaults to fp32 products.

from .unit2control import Unit2Control


class Volume_Extractor:
    """Frame RMS of an audio signal on the device (reference `ddsp/vocoder.py:116-137`, same constructor and method).

    `extract(audio)`: a numpy array (T,) as the reference's callers pass (`main.py`, `preprocess.py`) comes back as a
    numpy array (Frame,) - it is copied to `device`, reduced there and copied back; a device tensor (T,) or (B,T)
    comes back as a device tensor of the same rank.  There is no CPU computation path."""

    def __init__(self, hop_size=512, device="cuda"):
        self.hop_size = hop_size
        self.device = device

    def extract(self, audio):
        import numpy as np
        is_np = isinstance(audio, np.ndarray)
        x = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).to(self.device) if is_np else audio
        if not x.is_cuda:
            raise RuntimeError("Volume_Extractor runs on a HIP device only (no CPU fallback)")
        flat = x.dim() == 1
        vol = hipddsp.context_for(x.device).volume_extract(x.reshape(1, -1) if flat else x, int(self.hop_size))
        vol = vol[0] if flat else vol
        return vol.cpu().numpy() if is_np else vol


def align_units(units, n_samples, sample_rate, hop_size, encoder_sample_rate=16000, encoder_hop_size=320):
    """Nearest-frame alignment of encoder units (B, Lu, C) to the synthesiser's frames - the tail of the reference's
    `Units_Encoder.encode` (`ddsp/vocoder.py:201-211`), for callers that keep the reference's encoders and want the
    gather on the device: n_frames = n_samples // hop_size + 1, row i takes unit min(round(ratio * i), Lu - 1)."""
    n_frames = int(n_samples) // int(hop_size) + 1
    ratio = (hop_size / sample_rate) / (encoder_hop_size / encoder_sample_rate)
    return hipddsp.context_for(units.device).align_units(units, n_frames, ratio)


class DotDict(dict):
    """Attribute access to nested config dicts (reference `ddsp/vocoder.py:335-341`)."""

    def __getattr__(self, key):
        val = self.get(key)
        return DotDict(val) if type(val) is dict else val

    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


def load_model(model_path, device="cpu"):
    """Reads `<dir>/config.yaml`, builds the model it names, loads `ckpt['model']`
    (reference `ddsp/vocoder.py:343-369`).  Returns (model.eval(), args)."""
    with open(os.path.join(os.path.split(model_path)[0], "config.yaml"), "r") as fh:
        args = DotDict(yaml.safe_load(fh))
    kind = args.model.type
    if kind == "Sins":
        model = Sins(sampling_rate=args.data.sampling_rate, block_size=args.data.block_size,
                     n_harmonics=args.model.n_harmonics, n_mag_allpass=args.model.n_mag_allpass,
                     n_mag_noise=args.model.n_mag_noise, n_unit=args.data.encoder_out_channels,
                     n_spk=args.model.n_spk, c=args.model.c)
    elif kind == "CombSub":
        model = CombSub(sampling_rate=args.data.sampling_rate, block_size=args.data.block_size,
                        n_mag_allpass=args.model.n_mag_allpass, n_mag_harmonic=args.model.n_mag_harmonic,
                        n_mag_noise=args.model.n_mag_noise, n_unit=args.data.encoder_out_channels,
                        n_spk=args.model.n_spk, c=args.model.c)
    elif kind == "CombSubFast":
        model = CombSubFast(sampling_rate=args.data.sampling_rate, block_size=args.data.block_size,
                            n_unit=args.data.encoder_out_channels, n_spk=args.model.n_spk, c=args.model.c)
    else:
        raise ValueError(f" [x] Unknown Model: {kind}")
    print(" [Loading] " + model_path)
    # weights_only: a checkpoint is {'global_step', 'model', 'optimizer'} of tensors (reference logger/saver.py:83-87)
    ckpt = torch.load(model_path, map_location=torch.device(device), weights_only=True)
    model.to(device)
    model.load_state_dict(ckpt["model"])
    model.eval()
    return model, args


def _seed_from_torch():
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


class _SynthBase(torch.nn.Module):
    def __init__(self, sampling_rate, block_size):
        super().__init__()
        self.register_buffer("sampling_rate", torch.tensor(sampling_rate))
        self.register_buffer("block_size", torch.tensor(block_size))
        self._sr = int(sampling_rate)
        self._hop = int(block_size)

    def _front(self, f0_frames, initial_phase, infer, comb_mode, **want):
        if not f0_frames.is_cuda:
            raise RuntimeError("the synthesiser runs on a HIP device only (no CPU fallback): move the model and its "
                               "inputs to 'cuda'")
        ctx = hipddsp.context_for(f0_frames.device)
        return ctx, ctx.phase_scan(f0_frames, self._hop, self._sr, initial_phase, bool(infer), comb_mode, **want)

    def _empty_result(self, f0_frames, sample_rate_phase=False, shared=False):
        """The (signal, phase, (harmonic, noise)) tuple of an empty batch (nothing is launched)."""
        Fr = f0_frames.shape[1]
        T = Fr * self._hop
        z = lambda *shape: torch.zeros(*shape, device=f0_frames.device)
        sig = z(0, T)
        ph = z(0, T, 1) if sample_rate_phase else z(0, Fr, 1)
        return sig, ph, ((sig, sig) if shared else (z(0, T), z(0, T)))

    def _training_graph(self):
        """True when the call must be recorded for autograd (grad mode on and some parameter wants a gradient)."""
        return torch.is_grad_enabled() and any(p.requires_grad for p in self.unit2ctrl.parameters())

    @staticmethod
    def _noise_args(noise, noise_seed):
        if noise is not None:
            return noise.contiguous().float(), EXC_UNIT_NOISE, 0
        return None, EXC_GENERATE, (_seed_from_torch() if noise_seed is None else int(noise_seed))


class _SynthTrainFn(torch.autograd.Function):
    """Autograd node of one synthesiser forward: forward and backward are libddsp_amd calls end to end; torch only
    routes the parameter gradients (reference: autograd through `*.forward`, solver.py:111-113).  The model supplies
    `_train_forward(ctx, ctrl, ps, f0_frames, noise_args) -> (outputs, saved)` and
    `_train_backward(ctx, saved, f0_frames, noise_args, grads) -> d_ctrl (rows, sum)`."""

    @staticmethod
    def forward(fctx, model, units, f0_frames, volume, spk_id, spk_mix_dict, initial_phase, infer, noise, noise_seed,
                *params):
        ctx, ps = model._front(f0_frames, initial_phase, infer, model._comb_mode, **model._front_wants)
        # the training forward runs fp32 products throughout (control network, filter synthesis, FIR): the loss gradient
        # amplifies a 4e-6 error of the signal a thousandfold (tools/diag_train_b32.py).  The BACKWARD runs on the
        # context's own mode: its weight / input gradient GEMMs use split-bf16 products by default (a 4e-6 product error
        # inside the backward is not amplified); `ctx.set_math(MATH_FP32)` around `loss.backward()` makes them fp32 too
        keep_math = ctx.math
        ctx.set_math(hipddsp.MATH_FP32)
        try:
            ctrl, kept = model.unit2ctrl.forward_flat_keep(units, f0_frames, ps["phase_frames"], volume, spk_id,
                                                           spk_mix_dict, ctx=ctx)
            nargs = model._noise_args(noise, noise_seed)
            outs, saved = model._train_forward(ctx, ctrl, ps, f0_frames, nargs)
        finally:
            ctx.set_math(keep_math)
        fctx.model, fctx.dsp = model, ctx
        fctx.set_materialize_grads(False)   # outputs the loss does not use arrive as None in backward, not as zero tensors to add
        fctx.args = (units, f0_frames, volume, spk_id, spk_mix_dict, nargs, ps["phase_frames"])
        fctx.saved = (ctrl, saved, kept)
        phase_out = ps["phase"] if model._front_wants.get("want_phase") else ps["phase_frames"]
        fctx.mark_non_differentiable(phase_out)
        return (phase_out,) + tuple(outs)

    @staticmethod
    def backward(fctx, d_phase, *d_outs):
        model, ctx = fctx.model, fctx.dsp
        units, f0_frames, volume, spk_id, spk_mix_dict, nargs, phase_frames = fctx.args
        if fctx.saved is None:
            raise RuntimeError("this synthesiser forward was already back-propagated (its kept activations are released)")
        ctrl, saved, kept = fctx.saved
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        d_ctrl = model._train_backward(ctx, ctrl, saved, f0_frames, nargs, d_outs)
        grads = model.unit2ctrl.backward_flat(units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict,
                                              d_ctrl.reshape(B, Fr, -1), ctx=ctx, kept=kept)
        fctx.saved = None
        return (None,) * 10 + tuple(grads.get(p) for p in model.unit2ctrl.parameters())


def _sum_grads(ref, *gs):
    """Sum of the upstream gradients that exist (zeros when none does)."""
    acc = None
    for g in gs:
        if g is not None:
            acc = g.contiguous() if acc is None else acc + g
    return torch.zeros_like(ref) if acc is None else acc


class CombSub(_SynthBase):
    """Combtooth subtractive synthesiser, classic variant (reference `ddsp/vocoder.py:495-550`)."""

    def __init__(self, sampling_rate, block_size, n_mag_allpass, n_mag_harmonic, n_mag_noise, n_unit=256, n_spk=1,
                 c=False):
        super().__init__(sampling_rate, block_size)
        print(" [DDSP Model] Combtooth Subtractive Synthesiser (Old Version)")
        self.n_mags = (int(n_mag_allpass), int(n_mag_harmonic), int(n_mag_noise))
        self.unit2ctrl = Unit2Control(n_unit, n_spk, {"group_delay": n_mag_allpass,
                                                      "harmonic_magnitude": n_mag_harmonic,
                                                      "noise_magnitude": n_mag_noise}, c)

    _comb_mode = COMB_SINC
    _front_wants = {}

    def _train_forward(self, ctx, ctrl, ps, f0_frames, nargs):
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        rows, sr, hop = B * Fr, self._sr, self._hop
        na, nh, nn_ = self.n_mags
        c2 = ctrl.reshape(rows, -1)
        ir_ap = ctx.fir_from_ctrl(FIR_ALLPASS, c2, 0, na, rows, sr)
        h1, _ = ctx.ltv_fir(ps["comb"], ir_ap, B, Fr, hop, math=ctx.fir_math)
        ir_h = ctx.fir_from_ctrl(FIR_DYNAMIC, c2, na, nh, rows, sr, f0_frames)
        harmonic, _ = ctx.ltv_fir(h1, ir_h, B, Fr, hop, math=ctx.fir_math)
        ir_n = ctx.fir_from_ctrl(FIR_STATIC, c2, na + nh, nn_, rows, sr)
        nz, exc, seed = nargs
        noise_out, signal = ctx.ltv_fir(nz, ir_n, B, Fr, hop, excitation=exc, noise_seed=seed, add_in=harmonic,
                                        math=ctx.fir_math)
        return (signal, harmonic, noise_out), (ps["comb"], h1, ir_ap, ir_h, ir_n)

    def _train_backward(self, ctx, ctrl, saved, f0_frames, nargs, d_outs):
        comb, h1, ir_ap, ir_h, ir_n = saved
        d_signal, d_harm, d_noise = d_outs
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        rows, sr, hop = B * Fr, self._sr, self._hop
        na, nh, nn_ = self.n_mags
        nz, exc, seed = nargs
        d_h = _sum_grads(comb, d_signal, d_harm)
        d_n = _sum_grads(comb, d_signal, d_noise)
        c2 = ctrl.reshape(rows, -1)
        d_ctrl = torch.empty_like(c2)
        _, d_ir = ctx.ltv_fir_bwd(nz, ir_n, d_n, B, Fr, hop, excitation=exc, noise_seed=seed, want_d_audio=False)
        ctx.fir_from_ctrl_bwd(FIR_STATIC, c2, na + nh, nn_, rows, sr, d_ir, d_ctrl)
        d_h1, d_ir = ctx.ltv_fir_bwd(h1, ir_h, d_h, B, Fr, hop)
        ctx.fir_from_ctrl_bwd(FIR_DYNAMIC, c2, na, nh, rows, sr, d_ir, d_ctrl, f0_frames)
        _, d_ir = ctx.ltv_fir_bwd(comb, ir_ap, d_h1, B, Fr, hop, want_d_audio=False)
        ctx.fir_from_ctrl_bwd(FIR_ALLPASS, c2, 0, na, rows, sr, d_ir, d_ctrl)
        return d_ctrl

    def synth_from_ctrl(self, ctx, ctrl, f0_frames, comb, noise=None, noise_seed=None):
        """DSP stage: fused control matrix (B,Fr,sum) + combtooth -> (signal, harmonic, noise)."""
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        rows, sr, hop = B * Fr, self._sr, self._hop
        na, nh, nn_ = self.n_mags
        c2 = ctrl.reshape(rows, -1)
        ir = ctx.fir_from_ctrl(FIR_ALLPASS, c2, 0, na, rows, sr)
        h, _ = ctx.ltv_fir(comb, ir, B, Fr, hop, math=ctx.fir_math)
        ir = ctx.fir_from_ctrl(FIR_DYNAMIC, c2, na, nh, rows, sr, f0_frames)
        harmonic, _ = ctx.ltv_fir(h, ir, B, Fr, hop, math=ctx.fir_math)
        ir = ctx.fir_from_ctrl(FIR_STATIC, c2, na + nh, nn_, rows, sr)
        nz, exc, seed = self._noise_args(noise, noise_seed)
        noise_out, signal = ctx.ltv_fir(nz, ir, B, Fr, hop, excitation=exc, noise_seed=seed, add_in=harmonic,
                                        math=ctx.fir_math)
        return signal, harmonic, noise_out

    def forward(self, units_frames, f0_frames, volume_frames, spk_id, spk_mix_dict=None, initial_phase=None,
                infer=True, noise=None, noise_seed=None, **kwargs):
        """units (B,Fr,n_unit), f0 (B,Fr,1) Hz, volume (B,Fr), spk_id (B,1)|(1,1) int64 1-based ->
        (signal (B,T), phase_frames (B,Fr,1), (harmonic (B,T), noise (B,T)))."""
        if units_frames.shape[0] == 0:
            return self._empty_result(f0_frames)
        if self._training_graph():
            pf, signal, harmonic, noise_out = _SynthTrainFn.apply(
                self, units_frames, f0_frames, volume_frames, spk_id, spk_mix_dict, initial_phase, infer, noise,
                noise_seed, *self.unit2ctrl.parameters())
            return signal, pf.unsqueeze(-1), (harmonic, noise_out)
        ctx, ps = self._front(f0_frames, initial_phase, infer, COMB_SINC)
        ctrl = self.unit2ctrl.forward_flat(units_frames, f0_frames, ps["phase_frames"], volume_frames, spk_id,
                                           spk_mix_dict)
        signal, harmonic, noise_out = self.synth_from_ctrl(ctx, ctrl, f0_frames, ps["comb"], noise, noise_seed)
        return signal, ps["phase_frames"].unsqueeze(-1), (harmonic, noise_out)


class Sins(_SynthBase):
    """Sinusoids additive synthesiser (reference `ddsp/vocoder.py:372-423`)."""

    def __init__(self, sampling_rate, block_size, n_harmonics, n_mag_allpass, n_mag_noise, n_unit=256, n_spk=1,
                 c=False):
        super().__init__(sampling_rate, block_size)
        print(" [DDSP Model] Sinusoids Additive Synthesiser")
        self.n_mags = (int(n_harmonics), int(n_mag_allpass), int(n_mag_noise))
        self.unit2ctrl = Unit2Control(n_unit, n_spk, {"amplitudes": n_harmonics, "group_delay": n_mag_allpass,
                                                      "noise_magnitude": n_mag_noise}, c)

    _comb_mode = COMB_NONE
    _front_wants = {"want_phase": True}

    def _train_forward(self, ctx, ctrl, ps, f0_frames, nargs):
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        rows, sr, hop = B * Fr, self._sr, self._hop
        nhm, na, nn_ = self.n_mags
        c2 = ctrl.reshape(rows, -1)
        sinusoids = ctx.sins_bank(c2, 0, nhm, f0_frames, ps["phase"], B, Fr, hop, sr)
        ir_ap = ctx.fir_from_ctrl(FIR_ALLPASS, c2, nhm, na, rows, sr)
        harmonic, _ = ctx.ltv_fir(sinusoids, ir_ap, B, Fr, hop, math=ctx.fir_math)
        ir_n = ctx.fir_from_ctrl(FIR_STATIC, c2, nhm + na, nn_, rows, sr)
        nz, exc, seed = nargs
        noise_out, signal = ctx.ltv_fir(nz, ir_n, B, Fr, hop, excitation=exc, noise_seed=seed, add_in=harmonic,
                                        math=ctx.fir_math)
        return (signal, harmonic, noise_out), (ps["phase"], sinusoids, ir_ap, ir_n)

    def _train_backward(self, ctx, ctrl, saved, f0_frames, nargs, d_outs):
        phase, sinusoids, ir_ap, ir_n = saved
        d_signal, d_harm, d_noise = d_outs
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        rows, sr, hop = B * Fr, self._sr, self._hop
        nhm, na, nn_ = self.n_mags
        nz, exc, seed = nargs
        d_h = _sum_grads(sinusoids, d_signal, d_harm)
        d_n = _sum_grads(sinusoids, d_signal, d_noise)
        c2 = ctrl.reshape(rows, -1)
        d_ctrl = torch.empty_like(c2)
        _, d_ir = ctx.ltv_fir_bwd(nz, ir_n, d_n, B, Fr, hop, excitation=exc, noise_seed=seed, want_d_audio=False)
        ctx.fir_from_ctrl_bwd(FIR_STATIC, c2, nhm + na, nn_, rows, sr, d_ir, d_ctrl)
        d_sin, d_ir = ctx.ltv_fir_bwd(sinusoids, ir_ap, d_h, B, Fr, hop)
        ctx.fir_from_ctrl_bwd(FIR_ALLPASS, c2, nhm, na, rows, sr, d_ir, d_ctrl)
        ctx.sins_bank_bwd(c2, 0, nhm, f0_frames, phase, d_sin, B, Fr, hop, sr, d_ctrl)
        return d_ctrl

    def synth_from_ctrl(self, ctx, ctrl, f0_frames, phase, noise=None, noise_seed=None):
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        rows, sr, hop = B * Fr, self._sr, self._hop
        nhm, na, nn_ = self.n_mags
        c2 = ctrl.reshape(rows, -1)
        sinusoids = ctx.sins_bank(c2, 0, nhm, f0_frames, phase, B, Fr, hop, sr)
        ir = ctx.fir_from_ctrl(FIR_ALLPASS, c2, nhm, na, rows, sr)
        harmonic, _ = ctx.ltv_fir(sinusoids, ir, B, Fr, hop, math=ctx.fir_math)
        ir = ctx.fir_from_ctrl(FIR_STATIC, c2, nhm + na, nn_, rows, sr)
        nz, exc, seed = self._noise_args(noise, noise_seed)
        noise_out, signal = ctx.ltv_fir(nz, ir, B, Fr, hop, excitation=exc, noise_seed=seed, add_in=harmonic,
                                        math=ctx.fir_math)
        return signal, harmonic, noise_out

    def forward(self, units_frames, f0_frames, volume_frames, spk_id, spk_mix_dict=None, initial_phase=None,
                infer=True, max_upsample_dim=32, noise=None, noise_seed=None):
        """Same contract as CombSub.forward except that the returned phase is sample-rate (B,T,1)
        (reference `ddsp/vocoder.py:423`).  `max_upsample_dim` is accepted and ignored: the bank kernel never
        materialises the (B,T,chunk) tensors the reference chunks to bound."""
        if units_frames.shape[0] == 0:
            return self._empty_result(f0_frames, sample_rate_phase=True)
        if self._training_graph():
            ph, signal, harmonic, noise_out = _SynthTrainFn.apply(
                self, units_frames, f0_frames, volume_frames, spk_id, spk_mix_dict, initial_phase, infer, noise,
                noise_seed, *self.unit2ctrl.parameters())
            return signal, ph.unsqueeze(-1), (harmonic, noise_out)
        ctx, ps = self._front(f0_frames, initial_phase, infer, COMB_NONE, want_phase=True)
        ctrl = self.unit2ctrl.forward_flat(units_frames, f0_frames, ps["phase_frames"], volume_frames, spk_id,
                                           spk_mix_dict)
        signal, harmonic, noise_out = self.synth_from_ctrl(ctx, ctrl, f0_frames, ps["phase"], noise, noise_seed)
        return signal, ps["phase"].unsqueeze(-1), (harmonic, noise_out)


class CombSubFast(_SynthBase):
    """Combtooth subtractive synthesiser, windowed spectral OLA variant (reference `ddsp/vocoder.py:426-492`)."""

    def __init__(self, sampling_rate, block_size, n_unit=256, n_spk=1, c=False):
        super().__init__(sampling_rate, block_size)
        print(" [DDSP Model] Combtooth Subtractive Synthesiser")
        self.register_buffer("window", torch.sqrt(torch.hann_window(2 * block_size)))
        nb = int(block_size) + 1
        self.unit2ctrl = Unit2Control(n_unit, n_spk, {"harmonic_magnitude": nb, "harmonic_phase": nb,
                                                      "noise_magnitude": nb}, c)

    _comb_mode = COMB_SINC_GATED
    _front_wants = {}

    def _train_forward(self, ctx, ctrl, ps, f0_frames, nargs):
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        nz, exc, seed = nargs
        signal = ctx.spectral_ola(ctrl.reshape(B * Fr, -1), ps["comb"], nz, exc, seed, B, Fr, self._hop)
        return (signal,), (ps["comb"],)

    def _train_backward(self, ctx, ctrl, saved, f0_frames, nargs, d_outs):
        (comb,) = saved
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        nz, exc, seed = nargs
        return ctx.spectral_ola_bwd(ctrl.reshape(B * Fr, -1), comb, nz, exc, seed, _sum_grads(comb, *d_outs), B, Fr,
                                    self._hop)

    def synth_from_ctrl(self, ctx, ctrl, comb, noise=None, noise_seed=None):
        B, Fr = ctrl.shape[0], ctrl.shape[1]
        nz, exc, seed = self._noise_args(noise, noise_seed)
        return ctx.spectral_ola(ctrl.reshape(B * Fr, -1), comb, nz, exc, seed, B, Fr, self._hop)

    def forward(self, units_frames, f0_frames, volume_frames, spk_id, spk_mix_dict=None, initial_phase=None,
                infer=True, noise=None, noise_seed=None, **kwargs):
        """Returns (signal, phase_frames (B,Fr,1), (signal, signal)) - the same tensor three times, like the
        reference (`ddsp/vocoder.py:492`)."""
        if units_frames.shape[0] == 0:
            return self._empty_result(f0_frames, shared=True)
        if self._training_graph():
            pf, signal = _SynthTrainFn.apply(self, units_frames, f0_frames, volume_frames, spk_id, spk_mix_dict,
                                             initial_phase, infer, noise, noise_seed, *self.unit2ctrl.parameters())
            return signal, pf.unsqueeze(-1), (signal, signal)
        ctx, ps = self._front(f0_frames, initial_phase, infer, COMB_SINC_GATED)
        ctrl = self.unit2ctrl.forward_flat(units_frames, f0_frames, ps["phase_frames"], volume_frames, spk_id,
                                           spk_mix_dict)
        signal = self.synth_from_ctrl(ctx, ctrl, ps["comb"], noise, noise_seed)
        return signal, ps["phase_frames"].unsqueeze(-1), (signal, signal)
