"""ctypes binding of libddsp_amd.so (the C ABI declared in include/ddsp_amd.h).

PyTorch is used only for device memory and streams: every call passes raw device pointers and the
current HIP stream.  There is NO CPU fallback: without the shared library or without a GPU the
calls raise.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libddsp_amd.so")

DDSP_OK, DDSP_ERR_ARG, DDSP_ERR_HIP, DDSP_ERR_OOM = 0, -1, -2, -3

COMB_NONE, COMB_SINC, COMB_SINC_GATED = 0, 1, 2
FIR_ALLPASS, FIR_DYNAMIC, FIR_STATIC = 0, 1, 2
EXC_AUDIO, EXC_UNIT_NOISE, EXC_GENERATE = 0, 1, 2
FIR_FP32, FIR_SPLIT_BF16 = 0, 3   # ddsp_ltv_fir `math` (include/ddsp_amd.h)
ABI_VERSION = 5                   # DDSP_ABI_VERSION of include/ddsp_amd.h (struct layouts: U2CWeights)
MATH_FP32, MATH_SPLIT_BF16 = 0, 3  # ddsp_ctx_set_math
ATTENTION_CAUSAL = 200            # ddsp_performer_attention: causal_linear_attention (pcmer.py:170-188)

_c = ctypes
_vp, _i64, _u64, _int, _f32, _f64 = _c.c_void_p, _c.c_int64, _c.c_uint64, _c.c_int, _c.c_float, _c.c_double


class U2CWeights(_c.Structure):
    """Mirror of `ddsp_u2c_weights` (include/ddsp_amd.h): device pointers into the state dict."""
    _fields_ = (
        [(n, _vp) for n in ("prenet_conv1_w", "prenet_conv1_b", "prenet_gn_w", "prenet_gn_b",
                            "prenet_conv2_w", "prenet_conv2_b",
                            "f0_w", "f0_b", "phase_w", "phase_b", "volume_w", "volume_b", "spk_table")]
        + [("n_spk", _int), ("n_unit", _int), ("n_out", _int), ("causal", _int)]
        + [(f"l{i}_{n}", _vp) for i in range(3) for n in (
            "norm_w", "norm_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "proj", "out_w", "out_b",
            "cm_ln_w", "cm_ln_b", "cm_pw1_w", "cm_pw1_b", "cm_dw_w", "cm_dw_b", "cm_pw2_w", "cm_pw2_b")]
        + [(n, _vp) for n in ("final_ln_w", "final_ln_b", "head_g", "head_v", "head_b")]
        + [("version", _c.c_uint64)]     # change counter of the weight values (0: prepare the weights on every call)
    )


class ProfEntry(_c.Structure):
    """Mirror of `ddsp_prof_entry`."""
    _fields_ = [("family", _int), ("name", _c.c_char * 36), ("launches", _i64), ("ms_total", _c.c_double),
                ("flops_total", _c.c_double), ("bytes_total", _c.c_double)]


CONV_X_SPLIT, CONV_ACT_SPLIT = 1, 2


def presplit(w):
    """fp32 matrix (rows, K), K % 8 == 0 -> the same bytes-per-row with every group of 8 values replaced by their 8 bf16 high
    parts and 8 bf16 remainders (round to nearest even, twice): the pre-split operand layout of the split-bf16 GEMM
    (gemm::Args::B_split / A_split), bit-identical to what the kernels produce when they split in the loop."""
    if w.dtype != torch.float32 or w.shape[-1] % 8:
        raise ValueError("presplit: fp32 with a multiple of 8 columns")
    g = w.contiguous().reshape(*w.shape[:-1], w.shape[-1] // 8, 8)
    hi = g.to(torch.bfloat16)
    lo = (g - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32).reshape(w.shape)


FAMILIES = ["phase_scan", "fir_act", "fir_dft_gemm", "ltv_fir", "u2c_prep", "u2c_gemm_conv3", "u2c_gemm_linear",
            "u2c_gemm_feat", "u2c_gemm_ctx", "u2c_gemm_attnout", "u2c_rowwise", "sins_bank", "spectral_ola", "rss_loss",
            "sola", "upsample", "other", "ltv_fir_bwd", "fir_synth_bwd", "u2c_bwd", "optim"]

# name -> (restype, argtypes); every symbol declared in include/ddsp_amd.h must be listed here
SIGNATURES = {
    "ddsp_abi_version": (_int, []),
    "ddsp_ctx_create": (_int, [_c.POINTER(_vp), _int]),
    "ddsp_ctx_destroy": (_int, [_vp]),
    "ddsp_last_error": (_c.c_char_p, [_vp]),
    "ddsp_ctx_reserve": (_int, [_vp, _u64]),
    "ddsp_ctx_poll_error": (_int, [_vp]),
    "ddsp_ctx_set_math": (_int, [_vp, _int]),
    "ddsp_ctx_get_math": (_int, [_vp]),
    "ddsp_upsample": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _int, _vp]),
    "ddsp_phase_scan": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "ddsp_fir_from_ctrl": (_int, [_vp, _vp, _int, _vp, _i64, _int, _vp, _i64, _int, _vp]),
    "ddsp_ltv_fir_bwd": (_int, [_vp, _vp, _vp, _int, _u64, _vp, _vp, _i64, _i64, _int, _int, _vp, _vp]),
    "ddsp_fir_from_ctrl_bwd": (_int, [_vp, _vp, _int, _vp, _i64, _int, _vp, _i64, _int, _vp, _vp, _i64]),
    "ddsp_sins_bank": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp, _i64, _i64, _int, _int, _vp]),
    "ddsp_spectral_ola": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _int, _u64, _i64, _i64, _int, _vp]),
    "ddsp_sins_bank_bwd": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _i64, _i64, _int, _int, _vp, _i64]),
    "ddsp_spectral_ola_bwd": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _int, _u64, _vp, _i64, _i64, _int, _vp, _i64]),
    "ddsp_rss_loss": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _c.POINTER(_int), _c.POINTER(_int), _int, _f32, _f32, _vp, _vp]),
    "ddsp_sola": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _vp]),
    "ddsp_volume_gate": (_int, [_vp, _vp, _vp, _vp, _f32, _i64, _i64, _int]),
    "ddsp_phase_vocoder": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "ddsp_volume_extract": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _vp]),
    "ddsp_align_units": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp]),
    "ddsp_gemm_res_ln": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _int]),
    "ddsp_conv1d_pair_supported": (_int, [_vp, _int, _int, _int]),
    "ddsp_conv1d_pair": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _f32, _vp, _vp]),
    "ddsp_retime_f0": (_int, [_vp, _vp, _vp, _i64, _f64, _f64, _f32, _f64, _i64, _vp]),
    "ddsp_resample_length": (_i64, [_i64, _int, _int]),
    "ddsp_resample": (_int, [_vp, _vp, _vp, _i64, _i64, _int, _int, _int, _vp]),
    "ddsp_conv1d": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _f32, _vp, _vp, _vp, _f32, _vp, _int]),
    "ddsp_nsf_source": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _f32, _vp]),
    "ddsp_nsf_noise_conv": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _int, _int, _int, _int, _i64, _vp]),
    "ddsp_nsf_post": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _f32, _vp]),
    "ddsp_nsf_mean": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _i64, _vp, _vp, _f32, _int]),
    "ddsp_log_mel": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _f32, _vp]),
    "ddsp_adamw_step": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i64]),
    "ddsp_adamw_step_multi": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _f32, _i64]),
    "ddsp_gemm_f32": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _i64, _int, _vp, _vp, _i64, _int, _int, _int, _int, _int]),
    "ddsp_performer_attention": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _int]),
    "ddsp_profile_begin": (_int, [_vp, _u64]),
    "ddsp_profile_mask": (_int, [_vp, _u64]),
    "ddsp_profile_end": (_int, [_vp, _c.POINTER(ProfEntry), _int, _c.POINTER(_int)]),
    "ddsp_unit2ctrl_fwd": (_int, [_vp, _vp, _c.POINTER(U2CWeights), _vp, _vp, _vp, _vp, _vp, _i64,
                                  _c.POINTER(_i64), _c.POINTER(_f32), _int, _i64, _i64, _vp]),
    "ddsp_unit2ctrl_bwd": (_int, [_vp, _vp, _c.POINTER(U2CWeights), _vp, _vp, _vp, _vp, _vp, _i64,
                                  _c.POINTER(_i64), _c.POINTER(_f32), _int, _i64, _i64, _vp, _c.POINTER(U2CWeights), _vp]),
    "ddsp_ltv_fir": (_int, [_vp, _vp, _vp, _int, _u64, _vp, _i64, _i64, _int, _int, _vp, _vp, _vp, _int]),
    "ddsp_unit2ctrl_keep_bytes": (_i64, [_c.POINTER(U2CWeights), _i64, _i64]),
    "ddsp_unit2ctrl_fwd_keep": (_int, [_vp, _vp, _c.POINTER(U2CWeights), _vp, _vp, _vp, _vp, _vp, _i64,
                                       _c.POINTER(_i64), _c.POINTER(_f32), _int, _i64, _i64, _vp, _i64, _vp]),
    "ddsp_unit2ctrl_bwd_kept": (_int, [_vp, _vp, _c.POINTER(U2CWeights), _vp, _vp, _vp, _vp, _vp, _i64,
                                       _c.POINTER(_i64), _c.POINTER(_f32), _int, _i64, _i64, _vp, _i64, _vp,
                                       _c.POINTER(U2CWeights)]),
}

_lib = None
_lib_lock = threading.Lock()


def load_library():
    """Loads libddsp_amd.so from the package tree; raises if it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: build it with `python ddsp-svc-official_amd/hipddsp/build.py` "
                    "(there is no CPU fallback for the synthesis path)")
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)   # AttributeError if the symbol is not exported
                fn.restype = res
                fn.argtypes = args
            if lib.ddsp_abi_version() != ABI_VERSION:
                raise RuntimeError(f"{LIB_PATH} has ABI version {lib.ddsp_abi_version()}, this binding needs {ABI_VERSION}: "
                                   "rebuild it (`python ddsp-svc-official_amd/hipddsp/build.py`)")
            _lib = lib
    return _lib


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libddsp_amd needs device tensors (no CPU fallback)")
    if not t.is_contiguous():
        raise ValueError("libddsp_amd needs contiguous tensors")
    return t.data_ptr()


class Context:
    """One `ddsp_ctx` (scratch arena + constant tables) bound to a device; use one per stream/thread."""

    def __init__(self, device):
        self.lib = load_library()
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("libddsp_amd runs on a HIP device only (no CPU fallback); got device=%r" % (device,))
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: libddsp_amd cannot run (no CPU fallback)")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        h = _vp()
        rc = self.lib.ddsp_ctx_create(ctypes.byref(h), self.device.index)
        if rc != DDSP_OK:
            raise RuntimeError(f"ddsp_ctx_create failed with {rc}")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ddsp_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------------------------
    def _stream(self):
        """The torch stream current on this thread.  A context's scratch arena and lazily built tables are ordered by ONE
        stream (include/ddsp_amd.h: one context per stream); when a caller switches streams under a cached context
        (`with torch.cuda.stream(s):`), the new stream first waits for everything the context queued on the old one, so a
        half-built table or a scratch region still being read is never touched.  Concurrency across streams
        needs a context per stream (`Context(device)` + `use_context`)."""
        s = torch.cuda.current_stream(self.device)
        last = getattr(self, "_last_stream", None)
        if last is not None and last != s and not torch.cuda.is_current_stream_capturing():
            s.wait_stream(last)
        self._last_stream = s
        return s.cuda_stream

    def _check(self, rc, what):
        if rc == DDSP_OK:
            return
        msg = self.lib.ddsp_last_error(self.handle)
        msg = msg.decode() if msg else ""
        if rc == DDSP_ERR_ARG:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")

    def call(self, name, *args):
        if getattr(self, "_frozen", False):
            raise RuntimeError("this ddsp context belongs to a captured HIP graph (its scratch arena and tables are "
                               "referenced by the graph's kernels); use another context for eager calls")
        rc = getattr(self.lib, name)(self.handle, self._stream(), *args)
        self._check(rc, name)

    def freeze(self):
        """After a HIP graph has captured calls made through this context: refuse every further eager call, so that
        nothing can regrow or overwrite the scratch memory the captured kernels point into."""
        self._frozen = True

    def poll_error(self):
        """Raises ValueError if a kernel of an earlier call met a device-side contract violation (a speaker id outside
        the table); synchronise the stream first to be sure the call has run."""
        self._check(self.lib.ddsp_ctx_poll_error(self.handle), "ddsp_ctx_poll_error")

    # -- product arithmetic of the inference contractions ---------------------------------------
    @property
    def math(self):
        return self.lib.ddsp_ctx_get_math(self.handle)

    @property
    def fir_math(self):
        """The `math` argument the model forwards pass to `ddsp_ltv_fir` for this context's mode."""
        return FIR_FP32 if self.math == MATH_FP32 else FIR_SPLIT_BF16

    def set_math(self, math):
        """MATH_SPLIT_BF16 (default) or MATH_FP32: `ddsp_ctx_set_math`; the model forwards pass the same value to
        `ddsp_ltv_fir`."""
        self._check(self.lib.ddsp_ctx_set_math(self.handle, int(math)), "ddsp_ctx_set_math")

    # -- measurement ---------------------------------------------------------------------------
    def profile_begin(self, families=None):
        """Arms HIP-event timing for the named kernel families (all when None)."""
        mask = 0
        for f in (families if families is not None else FAMILIES):
            mask |= 1 << FAMILIES.index(f)
        self._check(self.lib.ddsp_profile_begin(self.handle, mask), "ddsp_profile_begin")

    def profile_mask(self, families):
        """While armed: bracket only `families` from now on ([] pauses); the records taken so far are kept."""
        mask = 0
        for f in families:
            mask |= 1 << FAMILIES.index(f)
        self._check(self.lib.ddsp_profile_mask(self.handle, mask), "ddsp_profile_mask")

    def profile_end(self):
        """-> {family: {launches, ms_total, flops_total, bytes_total}} (waits for the recorded events)."""
        buf = (ProfEntry * 32)()
        n = _int(0)
        self._check(self.lib.ddsp_profile_end(self.handle, buf, 32, ctypes.byref(n)), "ddsp_profile_end")
        return {buf[i].name.decode(): {"launches": buf[i].launches, "ms_total": buf[i].ms_total,
                                       "flops_total": buf[i].flops_total, "bytes_total": buf[i].bytes_total}
                for i in range(n.value)}

    # -- a1 ------------------------------------------------------------------------------------
    def upsample(self, x, hop):
        B, Fr, C = x.shape
        x = x.contiguous().float()
        out = torch.empty(B, Fr * hop, C, device=x.device, dtype=torch.float32)
        self.call("ddsp_upsample", _ptr(x), B, Fr, C, int(hop), _ptr(out))
        return out

    # -- a1-a3 ---------------------------------------------------------------------------------
    def phase_scan(self, f0_frames, hop, sr, initial_phase=None, precise=True, comb_mode=COMB_NONE,
                   want_rot=False, want_phase=False, want_f0=False):
        f0 = f0_frames.reshape(f0_frames.shape[0], -1).contiguous().float()
        B, Fr = f0.shape
        T = Fr * hop
        dev = f0.device
        mk = lambda: torch.empty(B, T, device=dev, dtype=torch.float32)
        rot = mk() if want_rot else None
        phase = mk() if want_phase else None
        comb = mk() if comb_mode != COMB_NONE else None
        f0_up = mk() if want_f0 else None
        pf = torch.empty(B, Fr, device=dev, dtype=torch.float32)
        ip = None if initial_phase is None else initial_phase.reshape(-1).contiguous().float()
        self.call("ddsp_phase_scan", _ptr(f0), _ptr(ip), B, Fr, int(hop), int(sr), 1 if precise else 0,
                  int(comb_mode), _ptr(rot), _ptr(phase), _ptr(comb), _ptr(f0_up), _ptr(pf))
        return {"rot": rot, "phase": phase, "comb": comb, "f0_up": f0_up, "phase_frames": pf}

    # -- a4 ------------------------------------------------------------------------------------
    def unit2ctrl(self, weights, units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict, n_out):
        B, Fr, _ = units.shape
        dev = units.device
        units = units.contiguous().float()
        f0 = f0_frames.reshape(B, Fr).contiguous().float()
        ph = phase_frames.reshape(B, Fr).contiguous().float()
        vol = volume.reshape(B, Fr).contiguous().float()
        ctrl = torch.empty(B, Fr, n_out, device=dev, dtype=torch.float32)
        if spk_mix_dict is not None:
            n_mix = len(spk_mix_dict)
            ids = (_i64 * max(n_mix, 1))(*[int(k) for k in spk_mix_dict.keys()])
            ws = (_f32 * max(n_mix, 1))(*[float(v) for v in spk_mix_dict.values()])
            sid, n_sid = None, 0
        else:
            n_mix, ids, ws = 0, None, None
            sid = spk_id.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
            n_sid = sid.numel()
            if n_sid not in (1, B):
                raise ValueError(f"spk_id must hold 1 or B={B} ids, got {n_sid}")
        self.call("ddsp_unit2ctrl_fwd", ctypes.byref(weights), _ptr(units), _ptr(f0), _ptr(ph), _ptr(vol), _ptr(sid),
                  n_sid, ids, ws, n_mix, B, Fr, _ptr(ctrl))
        return ctrl

    def unit2ctrl_bwd(self, weights, grads, units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict, n_out,
                      d_ctrl, want_ctrl=False):
        """Back-propagates d_ctrl (B,Fr,n_out) into the tensors `grads` points at (same struct layout as `weights`)."""
        B, Fr, _ = units.shape
        dev = units.device
        units = units.contiguous().float()
        f0 = f0_frames.reshape(B, Fr).contiguous().float()
        ph = phase_frames.reshape(B, Fr).contiguous().float()
        vol = volume.reshape(B, Fr).contiguous().float()
        d_ctrl = d_ctrl.contiguous().float()
        ctrl = torch.empty(B, Fr, n_out, device=dev, dtype=torch.float32) if want_ctrl else None
        if spk_mix_dict is not None:
            n_mix = len(spk_mix_dict)
            ids = (_i64 * max(n_mix, 1))(*[int(k) for k in spk_mix_dict.keys()])
            ws = (_f32 * max(n_mix, 1))(*[float(v) for v in spk_mix_dict.values()])
            sid, n_sid = None, 0
        else:
            n_mix, ids, ws = 0, None, None
            sid = spk_id.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
            n_sid = sid.numel()
        self.call("ddsp_unit2ctrl_bwd", ctypes.byref(weights), _ptr(units), _ptr(f0), _ptr(ph), _ptr(vol), _ptr(sid),
                  n_sid, ids, ws, n_mix, B, Fr, _ptr(d_ctrl), ctypes.byref(grads), _ptr(ctrl))
        return ctrl

    def _u2c_inputs(self, units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict):
        B, Fr, _ = units.shape
        dev = units.device
        units = units.contiguous().float()
        f0 = f0_frames.reshape(B, Fr).contiguous().float()
        ph = phase_frames.reshape(B, Fr).contiguous().float()
        vol = volume.reshape(B, Fr).contiguous().float()
        if spk_mix_dict is not None:
            n_mix = len(spk_mix_dict)
            ids = (_i64 * max(n_mix, 1))(*[int(k) for k in spk_mix_dict.keys()])
            ws = (_f32 * max(n_mix, 1))(*[float(v) for v in spk_mix_dict.values()])
            sid, n_sid = None, 0
        else:
            n_mix, ids, ws = 0, None, None
            sid = spk_id.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
            n_sid = sid.numel()
            if n_sid not in (1, B):
                raise ValueError(f"spk_id must hold 1 or B={B} ids, got {n_sid}")
        return (units, f0, ph, vol, sid), (_ptr(units), _ptr(f0), _ptr(ph), _ptr(vol), _ptr(sid), n_sid, ids, ws, n_mix, B, Fr)

    def unit2ctrl_keep(self, weights, units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict, n_out):
        """Training forward (fp32 products) that keeps its activations: -> (ctrl (B,Fr,n_out), keep) where `keep` is the
        uint8 device tensor `unit2ctrl_bwd_kept` back-propagates from (the caller holds it until then)."""
        B, Fr, _ = units.shape
        nbytes = int(self.lib.ddsp_unit2ctrl_keep_bytes(ctypes.byref(weights), B, Fr))
        if nbytes < 0:
            raise ValueError("ddsp_unit2ctrl_keep_bytes: bad shape")
        keep = torch.empty(nbytes, device=units.device, dtype=torch.uint8)
        ctrl = torch.empty(B, Fr, n_out, device=units.device, dtype=torch.float32)
        hold, args = self._u2c_inputs(units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict)
        self.call("ddsp_unit2ctrl_fwd_keep", ctypes.byref(weights), *args, _ptr(keep), nbytes, _ptr(ctrl))
        return ctrl, keep

    def unit2ctrl_bwd_kept(self, weights, grads, units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict, keep, d_ctrl):
        """Back-propagates d_ctrl from the activations `unit2ctrl_keep` left in `keep` (no second forward)."""
        d_ctrl = d_ctrl.contiguous().float()
        hold, args = self._u2c_inputs(units, f0_frames, phase_frames, volume, spk_id, spk_mix_dict)
        self.call("ddsp_unit2ctrl_bwd_kept", ctypes.byref(weights), *args, _ptr(keep), keep.numel(), _ptr(d_ctrl),
                  ctypes.byref(grads))

    # -- a5-a6 ---------------------------------------------------------------------------------
    def fir_from_ctrl(self, mode, ctrl2d, col0, n_mag, rows, sr, f0_frames=None):
        """ctrl2d :: (rows, ld) contiguous; the filter's control values are columns [col0, col0+n_mag)."""
        ld = ctrl2d.shape[-1]
        n = 2 * (n_mag - 1)
        ir = torch.empty(rows, n, device=ctrl2d.device, dtype=torch.float32)
        base = _ptr(ctrl2d) + 4 * col0
        f0 = None if f0_frames is None else f0_frames.reshape(-1).contiguous().float()
        self.call("ddsp_fir_from_ctrl", int(mode), base, ld, int(n_mag), _ptr(f0), rows, int(sr), _ptr(ir))
        return ir

    # -- a7 ------------------------------------------------------------------------------------
    def ltv_fir(self, audio, ir, B, Fr, hop, excitation=EXC_AUDIO, noise_seed=0, add_in=None, want_out=True,
                math=FIR_FP32):
        """Returns (filtered | None, filtered + add_in | None).  math: FIR_FP32 or FIR_SPLIT_BF16 (inference)."""
        n = ir.shape[-1]
        mk = lambda: torch.empty(B, Fr * hop, device=ir.device, dtype=torch.float32)
        out = mk() if want_out else None
        out_sum = mk() if add_in is not None else None
        self.call("ddsp_ltv_fir", _ptr(audio), int(excitation), int(noise_seed), _ptr(ir), B, Fr, int(hop), int(n),
                  _ptr(add_in), _ptr(out), _ptr(out_sum), int(math))
        return out, out_sum


    # -- backward of a5-a8 ---------------------------------------------------------------------
    def ltv_fir_bwd(self, audio, ir, d_out, B, Fr, hop, excitation=EXC_AUDIO, noise_seed=0, want_d_audio=True,
                    want_d_ir=True):
        n = ir.shape[-1]
        d_out = d_out.contiguous().float()
        d_audio = torch.empty(B, Fr * hop, device=ir.device, dtype=torch.float32) if want_d_audio else None
        d_ir = torch.empty(B * Fr, n, device=ir.device, dtype=torch.float32) if want_d_ir else None
        self.call("ddsp_ltv_fir_bwd", _ptr(audio), int(excitation), int(noise_seed), _ptr(ir), _ptr(d_out), B, Fr,
                  int(hop), int(n), _ptr(d_audio), _ptr(d_ir))
        return d_audio, d_ir

    def fir_from_ctrl_bwd(self, mode, ctrl2d, col0, n_mag, rows, sr, d_ir, d_ctrl2d, f0_frames=None):
        """Writes d ctrl into columns [col0, col0+n_mag) of d_ctrl2d (rows, ld); d_ir is consumed (scaled in place)."""
        f0 = None if f0_frames is None else f0_frames.reshape(-1).contiguous().float()
        self.call("ddsp_fir_from_ctrl_bwd", int(mode), _ptr(ctrl2d) + 4 * col0, ctrl2d.shape[-1], int(n_mag), _ptr(f0),
                  rows, int(sr), _ptr(d_ir), _ptr(d_ctrl2d) + 4 * col0, d_ctrl2d.shape[-1])

    # -- front-end steps -----------------------------------------------------
    def volume_extract(self, audio, hop):
        """audio (B,T) fp32 -> (B, T//hop + 1) block RMS with numpy-'reflect' padding (ddsp/vocoder.py:116-137)."""
        audio = audio.contiguous().float()
        B, T = audio.shape
        out = torch.empty(B, T // int(hop) + 1, device=audio.device, dtype=torch.float32)
        if B == 0:
            return out
        self.call("ddsp_volume_extract", _ptr(audio), B, T, int(hop), _ptr(out))
        return out

    def align_units(self, units, n_frames, ratio):
        """units (B,Lu,C) -> (B,n_frames,C), row i = units[:, min(rint(fp32(ratio)*i), Lu-1)] (ddsp/vocoder.py:201-211)."""
        units = units.contiguous().float()
        B, Lu, C = units.shape
        out = torch.empty(B, int(n_frames), C, device=units.device, dtype=torch.float32)
        if B == 0 or int(n_frames) == 0:
            return out
        self.call("ddsp_align_units", _ptr(units), B, Lu, C, int(n_frames), float(ratio), _ptr(out))
        return out

    def retime_f0(self, f0, step_num, div, scale, step_dst, n_dst):
        """f0 (n,) device track -> (n_dst,): numpy.interp(i * step_dst; knots (step_num * j) / div, values fl32(f0 * scale)),
        end values held (enhancer.py:56-62), without leaving the device."""
        f0 = f0.reshape(-1).contiguous().float()
        out = torch.empty(int(n_dst), device=f0.device, dtype=torch.float32)
        self.call("ddsp_retime_f0", _ptr(f0), f0.numel(), float(step_num), float(div), float(scale), float(step_dst), int(n_dst),
                  _ptr(out))
        return out

    # -- sample-rate conversion ---------------------------------------------
    def resample(self, audio, orig_freq, new_freq, lowpass_filter_width=6):
        """audio (B,T) or (T,) -> (B, ceil(T*new/orig)): windowed-sinc polyphase (torchaudio.transforms.Resample's algorithm)."""
        flat = audio.dim() == 1
        x = (audio.reshape(1, -1) if flat else audio).contiguous().float()
        B, T = x.shape
        T_out = self.lib.ddsp_resample_length(T, int(orig_freq), int(new_freq))
        if T_out < 0:
            raise ValueError("resample: bad rates")
        out = torch.empty(B, T_out, device=x.device, dtype=torch.float32)
        if B and T:
            self.call("ddsp_resample", _ptr(x), B, T, int(orig_freq), int(new_freq), int(lowpass_filter_width), _ptr(out))
        return out[0] if flat else out

    # -- NSF-HiFiGAN post-net building blocks --------------------------------
    def conv1d(self, x, w_packed, bias, ktaps, dil, in_slope, residual=None, want_out=True, act_slope=None, w_split=None,
               x_split=False, act_split=False):
        """x (T,Cin), w_packed (Cout, ktaps*Cin) -> y (T,Cout) = conv_same(leaky_relu(x, in_slope)) + bias (+ residual).
        Returns y, or (y | None, leaky_relu(y, act_slope)) when act_slope is given (want_out=False skips y itself).
        Split operands (include/ddsp_amd.h, ddsp_conv1d): `w_split` = presplit(w_packed); `x_split`: x is in the split layout;
        `act_split`: write the activated copy in it."""
        T, Cin = x.shape
        Cout = w_packed.shape[0]
        if w_packed.shape[1] != ktaps * Cin:
            raise ValueError("conv1d: packed weight does not match (ktaps, Cin)")
        out = torch.empty(T, Cout, device=x.device, dtype=torch.float32) if want_out else None
        act = torch.empty(T, Cout, device=x.device, dtype=torch.float32) if act_slope is not None else None
        if out is None and act is None:
            raise ValueError("conv1d: nothing to return")
        self.call("ddsp_conv1d", _ptr(x), _ptr(w_packed), _ptr(bias), T, Cin, Cout, int(ktaps), int(dil), float(in_slope),
                  _ptr(residual), _ptr(out), _ptr(act), float(act_slope if act_slope is not None else 1.0), _ptr(w_split),
                  (CONV_X_SPLIT if x_split else 0) | (CONV_ACT_SPLIT if act_split else 0))
        return out if act_slope is None else (out, act)

    def gemm_res_ln(self, A_split, W_split, bias, res, gamma, beta, y_split=True, a_fp32=False):
        """X = res + A W^T + bias and Y = LayerNorm(X) * gamma + beta in one launch; W (256, K) pre-split, A (M, K) pre-split or
        (a_fp32) plain fp32 rows."""
        M, K = A_split.shape
        X = torch.empty(M, 256, device=A_split.device, dtype=torch.float32)
        Y = torch.empty(M, 256, device=A_split.device, dtype=torch.float32)
        self.call("ddsp_gemm_res_ln", _ptr(A_split), _ptr(W_split), _ptr(bias), _ptr(res), _ptr(gamma), _ptr(beta), int(M), int(K),
                  _ptr(X), _ptr(Y), (1 if y_split else 0) | (2 if a_fp32 else 0))
        return X, Y

    def conv1d_pair_supported(self, C, ktaps, dil):
        return bool(self.lib.ddsp_conv1d_pair_supported(self.handle, int(C), int(ktaps), int(dil)))

    def conv1d_pair(self, x, w1, b1, w2, b2, ktaps, dil, slope, want_out=True, want_act=False):
        """One ResBlock1 pair of a narrow stage: x (T,C) raw -> (x + c2(leaky_relu(c1_dil(leaky_relu(x)))), its activated copy);
        either may be skipped (None)."""
        T, C = x.shape
        out = torch.empty(T, C, device=x.device, dtype=torch.float32) if want_out else None
        act = torch.empty(T, C, device=x.device, dtype=torch.float32) if want_act else None
        self.call("ddsp_conv1d_pair", _ptr(x), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), T, C, int(ktaps), int(dil), float(slope),
                  _ptr(out), _ptr(act))
        return out, act

    def nsf_source(self, f0, rand_ini, lin_w, lin_b, upp, sr, sine_amp=0.1):
        L = f0.numel()
        out = torch.empty(L * int(upp), device=f0.device, dtype=torch.float32)
        self.call("ddsp_nsf_source", _ptr(f0), _ptr(rand_ini.contiguous().float()), _ptr(lin_w), _ptr(lin_b), L, int(upp),
                  int(sr), float(sine_amp), _ptr(out))
        return out

    def nsf_noise_conv(self, src, w, b, K, stride, pad, T_out):
        C = w.shape[0]
        out = torch.empty(int(T_out), C, device=src.device, dtype=torch.float32)
        self.call("ddsp_nsf_noise_conv", _ptr(src), src.numel(), _ptr(w), _ptr(b), C, int(K), int(stride), int(pad),
                  int(T_out), _ptr(out))
        return out

    def nsf_post(self, x, w, b, K, slope):
        T, C = x.shape
        out = torch.empty(T, device=x.device, dtype=torch.float32)
        self.call("ddsp_nsf_post", _ptr(x), _ptr(w), _ptr(b), T, C, int(K), float(slope), _ptr(out))
        return out

    def nsf_mean(self, terms, want_out=True, act_slope=None, act_split=False):
        """Mean of up to three tensors -> out, or (out | None, leaky_relu(out, act_slope)) when act_slope is given
        (`act_split`: that copy in the split operand layout, see conv1d)."""
        a = terms[0]
        out = torch.empty_like(a) if want_out else None
        act = torch.empty_like(a) if act_slope is not None else None
        b = terms[1] if len(terms) > 1 else None
        c = terms[2] if len(terms) > 2 else None
        self.call("ddsp_nsf_mean", _ptr(a), _ptr(b), _ptr(c), len(terms), a.numel(), _ptr(out), _ptr(act),
                  float(act_slope if act_slope is not None else 1.0), CONV_ACT_SPLIT if act_split else 0)
        return out if act_slope is None else (out, act)

    def log_mel(self, frames, dft_table, mel_basis, clip):
        n_frames, n_fft = frames.shape
        n_mels = mel_basis.shape[0]
        out = torch.empty(n_frames, n_mels, device=frames.device, dtype=torch.float32)
        self.call("ddsp_log_mel", _ptr(frames), _ptr(dft_table), _ptr(mel_basis), n_frames, n_fft, n_mels, float(clip), _ptr(out))
        return out

    # -- a15 optimiser -------------------------------------------------------------------------
    def adamw_step(self, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step):
        self.call("ddsp_adamw_step", _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(lr),
                  float(beta1), float(beta2), float(eps), float(weight_decay), int(step))

    def adamw_step_multi(self, params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, weight_decay, step):
        """One AdamW update of a whole list of parameter tensors (shared hyper-parameters and step): the pointer
        tables go to the library as host arrays, the library issues one launch per 24 tensors."""
        n = len(params)
        if not (len(grads) == len(exp_avgs) == len(exp_avg_sqs) == n):
            raise ValueError("adamw_step_multi: lists of different length")
        for t in (*params, *grads, *exp_avgs, *exp_avg_sqs):
            if t.dtype != torch.float32:
                raise ValueError("adamw_step_multi: fp32 tensors only")
        for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
            if not (g.numel() == m.numel() == v.numel() == p.numel()):
                raise ValueError("adamw_step_multi: a tensor and its state differ in size")
        arr = lambda ts: (_vp * n)(*[_ptr(t) for t in ts])
        numel = (_i64 * n)(*[p.numel() for p in params])
        self.call("ddsp_adamw_step_multi", n, arr(params), arr(grads), arr(exp_avgs), arr(exp_avg_sqs), numel, float(lr),
                  float(beta1), float(beta2), float(eps), float(weight_decay), int(step))

    # -- building block ------------------------------------------------------------------------
    def gemm(self, A, B, bias=None, a_k_contig=True, b_k_contig=True, tile=0, variant=0, out=None):
        """C = op(A) @ op(B) (+bias): A (M,K) or (K,M) when not a_k_contig; B (N,K) or (K,N) when not b_k_contig.
        `out`: an (M, N) matrix to write into (the residual variants of the wave-specialised tiles ADD to it)."""
        M, K = (A.shape if a_k_contig else A.shape[::-1])
        N = B.shape[0] if b_k_contig else B.shape[1]
        C = torch.empty(M, N, device=A.device, dtype=torch.float32) if out is None else out
        if C.shape != (M, N) or C.dtype != torch.float32 or not C.is_contiguous():
            raise ValueError("gemm: `out` must be a contiguous fp32 (M, N) matrix")
        for t in (A, B):
            if not t.is_cuda or t.stride(1) != 1 or t.dtype != torch.float32:
                raise ValueError("gemm operands must be fp32 device matrices with unit inner stride")
        self.call("ddsp_gemm_f32", A.data_ptr(), A.stride(0), int(a_k_contig), B.data_ptr(), B.stride(0), int(b_k_contig),
                  _ptr(bias), _ptr(C), N, M, N, K, int(tile), int(variant))
        return C

    def performer_attention(self, q, k, v, proj, B, Fr, math=MATH_SPLIT_BF16):
        """q, k, v (B*Fr, 512), proj (266, 64) -> merged heads (B*Fr, 512) before `to_out` (pcmer.py:69-77,123-159)."""
        out = torch.empty(B * Fr, 512, device=q.device, dtype=torch.float32)
        self.call("ddsp_performer_attention", _ptr(q), _ptr(k), _ptr(v), _ptr(proj), int(B), int(Fr), _ptr(out), int(math))
        return out

    # -- a10 -----------------------------------------------------------------------------------
    def sins_bank(self, ctrl2d, col0, n_harmonics, f0_frames, phase, B, Fr, hop, sr):
        out = torch.empty(B, Fr * hop, device=ctrl2d.device, dtype=torch.float32)
        f0 = f0_frames.reshape(-1).contiguous().float()
        self.call("ddsp_sins_bank", _ptr(ctrl2d) + 4 * col0, ctrl2d.shape[-1], int(n_harmonics), _ptr(f0), _ptr(phase),
                  B, Fr, int(hop), int(sr), _ptr(out))
        return out

    # -- a11 -----------------------------------------------------------------------------------
    def spectral_ola(self, ctrl2d, comb, noise, excitation, noise_seed, B, Fr, hop):
        out = torch.empty(B, Fr * hop, device=ctrl2d.device, dtype=torch.float32)
        self.call("ddsp_spectral_ola", _ptr(ctrl2d), ctrl2d.shape[-1], _ptr(comb), _ptr(noise), int(excitation),
                  int(noise_seed), B, Fr, int(hop), _ptr(out))
        return out

    def sins_bank_bwd(self, ctrl2d, col0, n_harmonics, f0_frames, phase, d_out, B, Fr, hop, sr, d_ctrl2d):
        f0 = f0_frames.reshape(-1).contiguous().float()
        self.call("ddsp_sins_bank_bwd", _ptr(ctrl2d) + 4 * col0, ctrl2d.shape[-1], int(n_harmonics), _ptr(f0),
                  _ptr(phase), _ptr(d_out.contiguous()), B, Fr, int(hop), int(sr), _ptr(d_ctrl2d) + 4 * col0,
                  d_ctrl2d.shape[-1])

    def spectral_ola_bwd(self, ctrl2d, comb, noise, excitation, noise_seed, d_out, B, Fr, hop):
        d_ctrl = torch.empty_like(ctrl2d)
        self.call("ddsp_spectral_ola_bwd", _ptr(ctrl2d), ctrl2d.shape[-1], _ptr(comb), _ptr(noise), int(excitation),
                  int(noise_seed), _ptr(d_out.contiguous()), B, Fr, int(hop), _ptr(d_ctrl), d_ctrl.shape[-1])
        return d_ctrl

    # -- a13 -----------------------------------------------------------------------------------
    def rss_loss(self, x_pred, x_true, n_ffts, alpha=1.0, eps=1e-7, want_grad=False, hops=None):
        """-> (loss (1,) device tensor, d loss/d x_pred (B,T) | None) for the given list of scales; `hops`: one hop per
        scale (default: hop = n_fft, the reference's overlap = 0)."""
        xp = x_pred.detach().contiguous().float()
        xt = x_true.detach().contiguous().float()
        B, T = xp.shape
        if xt.shape != xp.shape:
            raise ValueError(f"x_pred {tuple(xp.shape)} and x_true {tuple(xt.shape)} must have the same shape")
        if T % 4:
            raise ValueError("signal length must be a multiple of 4")
        arr = (_int * len(n_ffts))(*[int(n) for n in n_ffts])
        if hops is not None and len(hops) != len(n_ffts):
            raise ValueError("one hop per scale")
        harr = (_int * len(n_ffts))(*[int(h) for h in hops]) if hops is not None else None
        loss = torch.empty(1, device=xp.device, dtype=torch.float32)
        grad = torch.empty_like(xp) if want_grad else None
        self.call("ddsp_rss_loss", _ptr(xp), _ptr(xt), B, T, arr, harr, len(n_ffts), float(alpha), float(eps), _ptr(loss),
                  _ptr(grad))
        return loss, grad

    # -- a14 -----------------------------------------------------------------------------------
    def sola(self, audio, sola_buffer, block, xfade, search, delay):
        """audio (N,), sola_buffer (xfade,) updated in place -> (emitted (block,), shift int32 device tensor)."""
        audio = audio.contiguous().float()
        emitted = torch.empty(block, device=audio.device, dtype=torch.float32)
        shift = torch.empty(1, device=audio.device, dtype=torch.int32)
        self.call("ddsp_sola", _ptr(audio), audio.numel(), int(block), int(xfade), int(search), int(delay),
                  _ptr(sola_buffer), _ptr(emitted), _ptr(shift))
        return emitted, shift

    def phase_vocoder(self, a, b, fade_out, fade_in):
        """gui.py:14-31: cross-fade of the kept tail `a` into the new head `b` (both (n,)) with a phase-interpolated
        oscillator term; returns (n,)."""
        a, b = a.contiguous().float(), b.contiguous().float()
        fo, fi = fade_out.contiguous().float(), fade_in.contiguous().float()
        n = a.numel()
        if not (b.numel() == fo.numel() == fi.numel() == n):
            raise ValueError("phase_vocoder: a, b and the fade windows must have the same length")
        out = torch.empty(n, device=a.device, dtype=torch.float32)
        self.call("ddsp_phase_vocoder", _ptr(a), _ptr(b), _ptr(fo), _ptr(fi), n, _ptr(out))
        return out

    # -- a15 -----------------------------------------------------------------------------------
    def volume_gate_(self, signal, volume, threshold_db, hop):
        """In place: signal (B,T) *= upsample(dilate9(volume > 10^(dB/20)))."""
        B, T = signal.shape
        vol = volume.reshape(B, -1).contiguous().float()
        self.call("ddsp_volume_gate", _ptr(signal), _ptr(vol), float(10 ** (float(threshold_db) / 20)), B,
                  vol.shape[1], int(hop))
        return signal


_contexts = {}
_ctx_lock = threading.Lock()


_override = threading.local()


class use_context:
    """`with use_context(ctx):` makes `context_for` return `ctx` on this thread for that device - how a graph-capturing
    caller routes a model's library calls through a context of its own (graphed.GraphedSynth)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.prev = getattr(_override, "ctx", None)
        _override.ctx = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        _override.ctx = self.prev
        return False


def context_for(device):
    """Per (device, thread) context cache (one handle per stream/thread)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    forced = getattr(_override, "ctx", None)
    if forced is not None and dev.type == "cuda" and forced.device.index == idx:
        return forced
    key = (dev.type, idx, threading.get_ident())
    with _ctx_lock:
        ctx = _contexts.get(key)
        if ctx is None:
            ctx = Context(torch.device(dev.type, idx) if dev.type == "cuda" else dev)
            _contexts[key] = ctx
    return ctx
