"""Oracle: the unit->control network (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional CPU restatement of reference `ddsp/unit2control.py:23-101` and
`ddsp/pcmer.py:11-63,69-77,123-159,191-251`, driven by a plain state dict with
the reference's key names (SURVEY.md section 5 "State-dict layout").

PARITY UNPINNED at the `extorch` boundary: `Conv1dEx(..., padding="same",
causal=False)` is read as a zero-padded `conv1d` and `Transpose(1, 2)` as
`x.transpose(1, 2)`; the third-party package is not installed and no reference
test covers it.  Causal mode (`c: true`, `causal=True` below) reads
`Conv1dEx(causal=True)` as a convolution whose k taps are the current and the k-1
previous frames (left zero padding) and `fast_transformers.CausalDotProduct` by its
definition (out_n = q_n . sum_{m<=n} k_m (x) v_m); the normaliser around it is the
reference's own `causal_linear_attention` (ddsp/pcmer.py:170-188).
"""
import math

import torch
import torch.nn.functional as F

N_LAYERS = 3
N_HEADS = 8
HEAD_DIM = 64


def split_sizes(x, splits):
    """ref: ddsp/unit2control.py:10-20.  `splits` is an ordered {name: width}."""
    names = list(splits.keys())
    parts = torch.split(x, [splits[k] for k in names], dim=-1)
    return {k: p for k, p in zip(names, parts)}


def _feature_map(x, proj, is_query):
    """Positive random features of the softmax kernel.  ref: ddsp/pcmer.py:123-159.

    x :: (B, h, N, d), proj :: (m, d).  Note the asymmetry the reference has: the
    query path subtracts the row max and adds eps OUTSIDE the exp, the key path
    adds eps INSIDE the exp and subtracts no max.
    """
    d = x.shape[-1]
    m = proj.shape[0]
    dn = d ** -0.25
    ratio = m ** -0.5
    dash = torch.einsum("bhnd,md->bhnm", dn * x, proj.to(x))
    diag = (x * x).sum(-1, keepdim=True) / 2.0 * dn * dn
    if is_query:
        return ratio * (torch.exp(dash - diag - dash.amax(dim=-1, keepdim=True)) + 1e-4)
    return ratio * torch.exp(dash - diag + 1e-4)


def _linear_attention(q, k, v):
    """Non-causal linear attention.  ref: ddsp/pcmer.py:69-77."""
    ksum = k.sum(dim=-2)
    dinv = 1.0 / (torch.einsum("bhnm,bhm->bhn", q, ksum) + 1e-8)
    ctx = torch.einsum("bhnm,bhne->bhme", k, v)
    return torch.einsum("bhme,bhnm,bhn->bhne", ctx, q, dinv)


def _causal_linear_attention(q, k, v, eps=1e-6):
    """ref: ddsp/pcmer.py:170-188 (CausalDotProduct restated from its definition)."""
    k_cumsum = k.cumsum(dim=-2) + eps
    d_inv = 1.0 / torch.einsum("...nd,...nd->...n", q, k_cumsum)
    ctx = torch.einsum("bhnm,bhne->bhnme", k, v).cumsum(dim=2)
    out = torch.einsum("bhnm,bhnme->bhne", q, ctx)
    return torch.einsum("...nd,...n->...nd", out, d_inv)


def _self_attention(x, sd, pre, causal=False):
    """ref: ddsp/pcmer.py:221-251."""
    B, N, _ = x.shape

    def heads(t):
        return t.reshape(B, N, N_HEADS, HEAD_DIM).permute(0, 2, 1, 3)

    q = heads(F.linear(x, sd[pre + "to_q.weight"], sd[pre + "to_q.bias"]))
    k = heads(F.linear(x, sd[pre + "to_k.weight"], sd[pre + "to_k.bias"]))
    v = heads(F.linear(x, sd[pre + "to_v.weight"], sd[pre + "to_v.bias"]))
    proj = sd[pre + "fast_attention.projection_matrix"]
    attn = _causal_linear_attention if causal else _linear_attention
    o = attn(_feature_map(q, proj, True), _feature_map(k, proj, False), v)
    o = o.permute(0, 2, 1, 3).reshape(B, N, N_HEADS * HEAD_DIM)
    return F.linear(o, sd[pre + "to_out.weight"], sd[pre + "to_out.bias"])


def _conv_same(x, w, b, causal, groups=1):
    """Conv1d with k taps: centred ("same") or, causal, on the current and the k-1 previous frames."""
    k = w.shape[-1]
    if causal:
        return F.conv1d(F.pad(x, (k - 1, 0)), w, b, groups=groups)
    return F.conv1d(x, w, b, padding=k // 2, groups=groups)


def _conv_module(x, sd, pre, causal=False):
    """LN - 1x1 conv - GLU - depthwise k=31 - SiLU - 1x1 conv.  ref: ddsp/pcmer.py:42-63."""
    d = x.shape[-1]
    y = F.layer_norm(x, (d,), sd[pre + "0.weight"], sd[pre + "0.bias"], 1e-5)
    y = y.transpose(1, 2)
    y = F.conv1d(y, sd[pre + "2.weight"], sd[pre + "2.bias"])
    y = F.glu(y, dim=1)
    wdw = sd[pre + "4.weight"]
    y = _conv_same(y, wdw, sd[pre + "4.bias"], causal, groups=wdw.shape[0])
    y = F.silu(y)
    y = F.conv1d(y, sd[pre + "6.weight"], sd[pre + "6.bias"])
    return y.transpose(1, 2)


def _encoder_layer(x, sd, pre, causal=False):
    """ref: ddsp/pcmer.py:20-38."""
    d = x.shape[-1]
    x = x + _self_attention(F.layer_norm(x, (d,), sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5),
                            sd, pre + "attn.", causal)
    x = x + _conv_module(x, sd, pre + "local_mixer.net.", causal)
    return x


def weight_normed(sd, pre):
    """Old-style weight norm: W = g * v / ||v||_2 (row-wise).  ref: ddsp/unit2control.py:61."""
    v = sd[pre + "weight_v"]
    g = sd[pre + "weight_g"]
    return v * (g / v.norm(dim=1, keepdim=True))


def unit2control(sd, units, f0, phase, volume, spk_id, spk_mix_dict, splits, return_flat=False, causal=False):
    """ref: ddsp/unit2control.py:68-101.

    sd      : state dict of the Unit2Control sub-module (keys without the "unit2ctrl." prefix)
    units   :: (B, Fr, n_unit)  f0 :: (B, Fr, 1)  phase, volume :: (B, Fr)  spk_id :: (B, 1) int64, 1-based
    """
    x = units.transpose(1, 2)
    x = _conv_same(x, sd["unit_prenet.1.weight"], sd["unit_prenet.1.bias"], causal)
    x = F.group_norm(x, 4, sd["unit_prenet.2.weight"], sd["unit_prenet.2.bias"], 1e-5)
    x = F.leaky_relu(x, 0.01)
    x = _conv_same(x, sd["unit_prenet.4.weight"], sd["unit_prenet.4.bias"], causal)
    x = x.transpose(1, 2)

    x = x + F.linear((1 + f0 / 700).log(), sd["f0_embed.weight"], sd["f0_embed.bias"]) \
          + F.linear(phase.unsqueeze(-1) / math.pi, sd["phase_embed.weight"], sd["phase_embed.bias"]) \
          + F.linear(volume.unsqueeze(-1), sd["volume_embed.weight"], sd["volume_embed.bias"])
    emb = sd["spk_embed.weight"]
    if spk_mix_dict is not None:
        for k, v in spk_mix_dict.items():
            x = x + v * emb[int(k) - 1][None, None, :]
    else:
        x = x + emb[spk_id - 1]                      # (B,1,256) broadcast over frames

    for i in range(N_LAYERS):
        x = _encoder_layer(x, sd, f"dec_post.0.net.{i}.", causal)
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd["dec_post.1.weight"], sd["dec_post.1.bias"], 1e-5)
    e = F.linear(x, weight_normed(sd, "dec_post.2."), sd["dec_post.2.bias"])
    if return_flat:
        return e
    return split_sizes(e, splits)
