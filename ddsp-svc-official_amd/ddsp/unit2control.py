"""`Unit2Control` with the reference's constructor, state-dict keys and forward contract
(reference `ddsp/unit2control.py:23-101`, `ddsp/pcmer.py`), executed by libddsp_amd.

The torch modules below are PARAMETER CONTAINERS only (their names reproduce the reference's
state-dict layout so upstream checkpoints load with `load_state_dict`); none of them has a forward of
its own.  `Unit2Control.forward` hands raw device pointers to `ddsp_unit2ctrl_fwd`.
Only the non-causal configuration (`c: false`, every shipped config) exists; `c=True` raises.
"""
import math

import os

import torch
import torch.nn as nn

import hipddsp

_WEIGHT_CACHE = os.environ.get("DDSP_U2C_WCACHE", "1") != "0"   # measurement aid: 0 = prepare the weights on every forward

NDIM = 256
N_LAYERS = 3
N_HEADS = 8
HEAD_DIM = 64
N_FEATURES = int(HEAD_DIM * math.log(HEAD_DIM))  # 266 random features (reference ddsp/pcmer.py:196)
DW_KERNEL = 31


class _Slot(nn.Module):
    """Placeholder keeping the index of a parameter-free layer inside an nn.Sequential."""


def _uniform_(t, bound):
    with torch.no_grad():
        return t.uniform_(-bound, bound)


class _Affine(nn.Module):
    """weight/bias pair initialised like torch's Linear / Conv1d (kaiming-uniform, a=sqrt(5))."""

    def __init__(self, w_shape, fan_in, ones=False):
        super().__init__()
        if ones:  # normalisation layers
            self.weight = nn.Parameter(torch.ones(w_shape))
            self.bias = nn.Parameter(torch.zeros(w_shape))
        else:
            bound = 1.0 / math.sqrt(fan_in)
            self.weight = nn.Parameter(_uniform_(torch.empty(w_shape), bound))
            self.bias = nn.Parameter(_uniform_(torch.empty(w_shape[0]), bound))


class _WeightNormHead(nn.Module):
    """Keys bias / weight_g / weight_v of the reference's `weight_norm(nn.Linear(...))` (:61)."""

    def __init__(self, n_in, n_out):
        super().__init__()
        bound = 1.0 / math.sqrt(n_in)
        v = _uniform_(torch.empty(n_out, n_in), bound)
        self.bias = nn.Parameter(_uniform_(torch.empty(n_out), bound))
        self.weight_g = nn.Parameter(v.norm(dim=1, keepdim=True).clone())
        self.weight_v = nn.Parameter(v)


def _orthogonal_gaussian_features(n_rows, n_cols):
    """Performer projection: stacked orthogonal blocks with chi-distributed row norms (the construction
    the reference uses for its fixed `projection_matrix` buffer, `ddsp/pcmer.py:80-120`)."""
    blocks = []
    remaining = n_rows
    while remaining > 0:
        q, _ = torch.linalg.qr(torch.randn(n_cols, n_cols), mode="reduced")
        take = min(remaining, n_cols)
        blocks.append(q.t()[:take])
        remaining -= take
    mat = torch.cat(blocks, dim=0)
    norms = torch.randn(n_rows, n_cols).norm(dim=1)
    return norms[:, None] * mat


class _FastAttention(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("projection_matrix", _orthogonal_gaussian_features(N_FEATURES, HEAD_DIM))


class _SelfAttention(nn.Module):
    def __init__(self, dim):
        super().__init__()
        inner = N_HEADS * HEAD_DIM
        self.fast_attention = _FastAttention()
        self.to_q = _Affine((inner, dim), dim)
        self.to_k = _Affine((inner, dim), dim)
        self.to_v = _Affine((inner, dim), dim)
        self.to_out = _Affine((dim, inner), inner)


class _ConvModule(nn.Module):
    def __init__(self, dim):
        super().__init__()
        inner = dim * 2
        self.net = nn.Sequential(
            _Affine((dim,), dim, ones=True),                 # 0 LayerNorm
            _Slot(),                                         # 1 transpose
            _Affine((inner * 2, dim, 1), dim),               # 2 pointwise conv
            _Slot(),                                         # 3 GLU
            _Affine((inner, 1, DW_KERNEL), DW_KERNEL),       # 4 depthwise conv
            _Slot(),                                         # 5 SiLU
            _Affine((dim, inner, 1), inner),                 # 6 pointwise conv
            _Slot(), _Slot(),                                # 7 transpose, 8 dropout(p=0)
        )


class _EncoderLayer(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = _Affine((dim,), dim, ones=True)
        self.attn = _SelfAttention(dim)
        self.local_mixer = _ConvModule(dim)


class _PCmer(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.Sequential(*[_EncoderLayer(dim) for _ in range(N_LAYERS)])


def split_to_dict(tensor, tensor_splits):
    """Views into the fused control matrix, in dict order (reference `ddsp/unit2control.py:10-20`)."""
    out = {}
    lo = 0
    for name, width in tensor_splits.items():
        out[name] = tensor[..., lo:lo + width]
        lo += width
    return out


class Unit2Control(nn.Module):
    def __init__(self, ndim_feat_i, n_spk, output_splits, c=False):
        super().__init__()
        # c = True: causal convolutions (taps at frames t-k+1 .. t) and causal linear attention, forward and backward.  The two
        # third-party primitives behind it (extorch.Conv1dEx(causal=True), fast_transformers.CausalDotProduct) are not in
        # the image: they are restated from their definitions, parity at that boundary is unpinned.
        self.causal = bool(c)
        self.n_unit = int(ndim_feat_i)
        self.n_spk = int(n_spk)
        self.output_splits = dict(output_splits)
        self.n_out = sum(self.output_splits.values())
        self.unit_prenet = nn.Sequential(
            _Slot(),
            _Affine((NDIM, self.n_unit, 3), self.n_unit * 3),
            _Affine((NDIM,), NDIM, ones=True),               # GroupNorm(4, 256)
            _Slot(),
            _Affine((NDIM, NDIM, 3), NDIM * 3),
            _Slot(),
        )
        self.f0_embed = _Affine((NDIM, 1), 1)
        self.phase_embed = _Affine((NDIM, 1), 1)
        self.volume_embed = _Affine((NDIM, 1), 1)
        self.spk_embed = nn.Module()
        self.spk_embed.weight = nn.Parameter(torch.randn(self.n_spk, NDIM))
        self.dec_post = nn.Sequential(_PCmer(NDIM), _Affine((NDIM,), NDIM, ones=True),
                                      _WeightNormHead(NDIM, self.n_out))
        self._packed_version = None

    # ---- raw pointer table ---------------------------------------------------------------------
    def _named_tensors(self):
        """(struct field, tensor) pairs in the order of `ddsp_u2c_weights`."""
        pre = self.unit_prenet
        out = [("prenet_conv1_w", pre[1].weight), ("prenet_conv1_b", pre[1].bias), ("prenet_gn_w", pre[2].weight),
               ("prenet_gn_b", pre[2].bias), ("prenet_conv2_w", pre[4].weight), ("prenet_conv2_b", pre[4].bias),
               ("f0_w", self.f0_embed.weight), ("f0_b", self.f0_embed.bias), ("phase_w", self.phase_embed.weight),
               ("phase_b", self.phase_embed.bias), ("volume_w", self.volume_embed.weight),
               ("volume_b", self.volume_embed.bias), ("spk_table", self.spk_embed.weight)]
        for i, layer in enumerate(self.dec_post[0].net):
            a, cm = layer.attn, layer.local_mixer.net
            vals = dict(norm_w=layer.norm.weight, norm_b=layer.norm.bias,
                        q_w=a.to_q.weight, q_b=a.to_q.bias, k_w=a.to_k.weight, k_b=a.to_k.bias,
                        v_w=a.to_v.weight, v_b=a.to_v.bias, proj=a.fast_attention.projection_matrix,
                        out_w=a.to_out.weight, out_b=a.to_out.bias,
                        cm_ln_w=cm[0].weight, cm_ln_b=cm[0].bias, cm_pw1_w=cm[2].weight, cm_pw1_b=cm[2].bias,
                        cm_dw_w=cm[4].weight, cm_dw_b=cm[4].bias, cm_pw2_w=cm[6].weight, cm_pw2_b=cm[6].bias)
            out += [(f"l{i}_{k}", v) for k, v in vals.items()]
        head = self.dec_post[2]
        out += [("final_ln_w", self.dec_post[1].weight), ("final_ln_b", self.dec_post[1].bias),
                ("head_g", head.weight_g), ("head_v", head.weight_v), ("head_b", head.bias)]
        return out

    def backward_flat(self, units, f0, phase, volume, spk_id, spk_mix_dict, d_ctrl, ctx=None, kept=None):
        """Gradients of every parameter for an upstream d_ctrl (B,Fr,n_out): {parameter tensor: gradient tensor}.
        `ctx`: the context of the forward call (autograd runs backward on its own thread; reusing the forward's
        context keeps one scratch arena and one profiler per model call).  `kept`: the activation region of a
        `forward_flat_keep` call on the same inputs and weights - without it the forward is re-run inside the call."""
        ctx = ctx or hipddsp.context_for(units.device)
        w, keep = self._weights_struct()
        g = hipddsp.U2CWeights()
        grads = {}
        # `training.GradBucket.zero()` arms `_grads_in_place`: the library then writes every gradient straight into the
        # parameter's `.grad` (a view of the bucket's freshly zeroed flat buffer) and autograd gets nothing to accumulate - no
        # per-parameter add launches (0.18 ms of a B = 32 step).  The library WRITES (`=`), so the token is consumed here: a
        # second backward pass before the next `zero()` returns its gradients to autograd, which accumulates them.
        in_place = getattr(self, "_grads_in_place", False)
        self._grads_in_place = False
        for name, t in self._named_tensors():
            if name.endswith("_proj"):
                continue
            if in_place and t.grad is not None and t.grad.is_contiguous() and t.grad.dtype == torch.float32:
                setattr(g, name, t.grad.data_ptr())
                continue
            gt = torch.empty_like(t, dtype=torch.float32, memory_format=torch.contiguous_format)
            grads[t] = gt
            setattr(g, name, gt.data_ptr())
        g.n_spk, g.n_unit, g.n_out = self.n_spk, self.n_unit, self.n_out
        if kept is not None:
            ctx.unit2ctrl_bwd_kept(w, g, units, f0, phase, volume, spk_id, spk_mix_dict, kept, d_ctrl)
        else:
            ctx.unit2ctrl_bwd(w, g, units, f0, phase, volume, spk_id, spk_mix_dict, self.n_out, d_ctrl)
        return grads

    def _tensor_slots(self):
        """(struct field, owning dict, key) for every tensor of `ddsp_u2c_weights`, built once: reading `d[k]` costs a dict
        lookup where `module.weight` goes through nn.Module.__getattr__ (the walk over the 75 tensors took longer on the host than
        a B = 1 forward takes on the device).  A Parameter replaced by setattr lands in the same dict and is seen; after replacing
        a whole SUBMODULE call `rebind()`."""
        slots = getattr(self, "_slots", None)
        if slots is None:
            by_id = {}
            for mod in self.modules():
                for d in (mod._parameters, mod._buffers):
                    for k, t in d.items():
                        if t is not None:
                            by_id[id(t)] = (d, k)
            slots = self._slots = [(name,) + by_id[id(t)] for name, t in self._named_tensors()]
        return slots

    def rebind(self):
        """Forget the cached tensor slots / weight struct (after a submodule of this network was replaced)."""
        self._slots = None
        self._ws = None

    def _weights_struct(self):
        slots = self._tensor_slots()
        tensors = [d[k] for _, d, k in slots]
        grad_mode = torch.is_grad_enabled()
        key = (grad_mode,) + tuple([(t.data_ptr(), t._version) for t in tensors])
        ws = getattr(self, "_ws", None)
        if ws is not None and ws[0] == key:
            return ws[1], ws[2]
        w = hipddsp.U2CWeights()
        keep = []
        for (name, _, _), t in zip(slots, tensors):
            if not t.is_cuda:
                raise RuntimeError("Unit2Control parameters must live on a HIP device (no CPU fallback)")
            t = t.detach()
            if not t.is_contiguous() or t.dtype != torch.float32:
                t = t.contiguous().float()
            keep.append(t)
            setattr(w, name, t.data_ptr())
        w.n_spk, w.n_unit, w.n_out = self.n_spk, self.n_unit, self.n_out
        w.causal = 1 if self.causal else 0
        # inference: the library may keep its prepared copies of these weights while their values stand (every in-place change of
        # a tensor - optimizer step, load_state_dict, copy_ - advances its `_version`); training steps prepare them every time
        # (the nonce tells two model objects apart whose tensors the allocator placed at the same addresses)
        if grad_mode or not _WEIGHT_CACHE:
            w.version = 0
        else:
            if not hasattr(self, "_weights_nonce"):
                self._weights_nonce = int.from_bytes(os.urandom(6), "little") << 16
            w.version = (self._weights_nonce + 1 + sum(int(t._version) for t in tensors)) & ((1 << 64) - 1)
        # (a copy made by `contiguous().float()` above has a new address every time: such a struct is not worth keeping)
        if all(a is b or a.data_ptr() == b.data_ptr() for a, b in zip(keep, tensors)):
            self._ws = (key, w, keep)
        return w, keep

    def forward_flat(self, units, f0, phase, volume, spk_id, spk_mix_dict=None):
        """(B, Fr, n_out) fused control matrix (the split views are taken by `forward`)."""
        ctx = hipddsp.context_for(units.device)
        w, keep = self._weights_struct()
        return ctx.unit2ctrl(w, units, f0, phase, volume, spk_id, spk_mix_dict, self.n_out)

    def forward_flat_keep(self, units, f0, phase, volume, spk_id, spk_mix_dict=None, ctx=None):
        """Training forward: (control matrix, kept activations) - PyTorch keeps a module's activations for `backward`
        (reference `solver.py:111-113`); here the library leaves them in one device region that `backward_flat(kept=...)`
        starts from, so the network runs once per step."""
        ctx = ctx or hipddsp.context_for(units.device)
        w, keep = self._weights_struct()
        return ctx.unit2ctrl_keep(w, units, f0, phase, volume, spk_id, spk_mix_dict, self.n_out)

    def forward(self, units, f0, phase, volume, spk_id, spk_mix_dict=None):
        """Same contract as the reference `Unit2Control.forward` (`ddsp/unit2control.py:68-101`):
        units (B,Fr,n_unit), f0 (B,Fr,1), phase (B,Fr), volume (B,Fr), spk_id (B,1)|(1,1) int64 1-based,
        spk_mix_dict {id: weight} or None -> dict of (B,Fr,width) views."""
        return split_to_dict(self.forward_flat(units, f0, phase, volume, spk_id, spk_mix_dict), self.output_splits)
