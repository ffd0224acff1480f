"""Drop-in for the reference's `enhancer.py` (`Enhancer`, `NsfHifiGAN`) and the inference half of `nsf_hifigan/models.py`
(`load_model`, `Generator` with `SineGen` / `SourceModuleHnNSF`) and `nsf_hifigan/nvSTFT.py` (`STFT.get_mel`), executed by
libddsp_amd on the device (no CPU path).

    Enhancer(enhancer_type, enhancer_ckpt, device).enhance(audio (1,T), sample_rate, f0 (1,Fr,1), hop_size,
                                                           adaptive_key=0 | 'auto', silence_front=0) -> (audio (1,T'), sr)

Checkpoints are read with `torch.load(weights_only=True)` (`{'generator': state_dict}` next to a `config.json`, as
`nsf_hifigan/models.py:24-39` expects); weight-normed layers (`weight_g`, `weight_v`) are folded at load time like the
reference's `remove_weight_norm()`.  Two third-party boundaries are restated from their published algorithms and are
PARITY UNPINNED (neither package is in the image): librosa's Slaney mel filter bank (`nvSTFT.py:85`) and torchaudio's
resampler (`enhancer.py:50-53,69-73`, see resample.py).  Everything else is pinned by fixtures generated from the
reference's own `nsf_hifigan/models.py` (tests/golden/make_golden.py tier f).
"""
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

import hipddsp
from resample import Resample

PAIR_FUSION = os.environ.get("DDSP_CONV_PAIR", "1") != "0"   # measurement aid: 0 = two launches per residual pair

LRELU_SLOPE = 0.1


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


# ---- constant tables ------------------------------------------------------------------------------------------------------
def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults (Slaney scale, Slaney area normalisation),
    restated from librosa's published algorithm -> (n_mels, n_fft//2 + 1) float32."""
    def hz_to_mel(f):
        f = np.asanyarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-12) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asanyarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fmax = float(sr) / 2 if fmax is None else fmax
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


class STFT:
    """`nsf_hifigan/nvSTFT.py:52-119` for keyshift = 0, speed = 1: reflect padding, Hann(periodic) window, magnitude with the
    1e-9 floor, mel projection, log of the clamped value.  `get_mel(y (1,T)) -> (1, n_mels, frames)`."""

    def __init__(self, sr=22050, n_mels=80, n_fft=1024, win_size=1024, hop_length=256, fmin=20, fmax=11025, clip_val=1e-5):
        if win_size != n_fft:
            raise ValueError("only win_size == n_fft (every shipped NSF-HiFiGAN config) is built")
        self.target_sr, self.n_mels, self.n_fft, self.win_size, self.hop_length = sr, n_mels, n_fft, win_size, hop_length
        self.fmin, self.fmax, self.clip_val = fmin, fmax, clip_val
        self._tables = {}

    def _device_tables(self, device):
        key = str(device)
        if key not in self._tables:
            n, bins = self.n_fft, self.n_fft // 2 + 1
            ldm = (bins + 3) & ~3
            i = torch.arange(n, dtype=torch.float64)
            f = torch.arange(bins, dtype=torch.float64)
            ang = 2 * np.pi * torch.outer(f, i) / n
            win = torch.hann_window(n, dtype=torch.float64)
            tab = torch.zeros(2 * ldm, n, dtype=torch.float64)
            tab[0:2 * bins:2] = torch.cos(ang) * win
            tab[1:2 * bins:2] = -torch.sin(ang) * win
            mel = torch.zeros(self.n_mels, ldm)
            mel[:, :bins] = torch.from_numpy(mel_filterbank(self.target_sr, n, self.n_mels, self.fmin, self.fmax))
            self._tables[key] = (tab.float().to(device).contiguous(), mel.to(device).contiguous())
        return self._tables[key]

    def get_mel(self, y, keyshift=0, speed=1, center=False):
        if keyshift != 0 or speed != 1 or center:
            raise ValueError("only keyshift = 0, speed = 1, center = False (how the Enhancer calls it) is built")
        if not y.is_cuda:
            raise RuntimeError("STFT.get_mel runs on a HIP device only (no CPU fallback)")
        if y.shape[0] != 1:
            raise ValueError("one utterance per call, like the reference's Enhancer")
        n, hop = self.n_fft, self.hop_length
        pad_left = (n - hop) // 2
        pad_right = max((n - hop + 1) // 2, n - y.size(-1) - pad_left)
        mode = "reflect" if pad_right < y.size(-1) else "constant"
        yp = F.pad(y.unsqueeze(1), (pad_left, pad_right), mode=mode).squeeze(1)           # (memory movement only)
        frames = yp[0].unfold(0, n, hop).contiguous()                                      # (frames, n_fft)
        tab, mel = self._device_tables(y.device)
        out = hipddsp.context_for(y.device).log_mel(frames, tab, mel, self.clip_val)      # (frames, n_mels)
        return out.t().unsqueeze(0)


# ---- the generator -------------------------------------------------------------------------------------------------------------
def _fold_weight_norm(sd, prefix):
    """weight of a layer stored either plain or as (weight_g, weight_v) of torch.nn.utils.weight_norm (dim 0)."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"].float()
    g, v = sd[prefix + ".weight_g"].float(), sd[prefix + ".weight_v"].float()
    return v * (g / v.norm(dim=(1, 2), keepdim=True))


def _pack_conv(w):
    """Conv1d weight (Cout, Cin, k) -> (Cout, k*Cin), column = tap*Cin + ci (the GEMM's implicit-im2col order)."""
    return w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()


def _split_or_none(w_packed):
    """The packed weight in the split operand layout (hipddsp.presplit), or None where its rows are not whole groups of 8."""
    return hipddsp.presplit(w_packed) if w_packed.shape[1] % 8 == 0 else None


def _pack_conv_transpose(w, stride):
    """ConvTranspose1d weight (Cin, Cout, k), padding (k - stride)//2 -> the 3-tap convolution that produces all `stride`
    output phases at once: (stride*Cout, 3*Cin), row = r*Cout + co, column = tap*Cin + ci, value W[ci][co][r + p - (tap-1)*u]."""
    Cin, Cout, k = w.shape
    u, p = stride, (k - stride) // 2
    if (k - stride) % 2 or k < u:
        raise ValueError("unsupported ConvTranspose1d geometry")
    out = torch.zeros(u * Cout, 3 * Cin, dtype=w.dtype)
    covered = torch.zeros(u, k, dtype=torch.bool)
    for r in range(u):
        for tap in range(3):
            kk = r + p - (tap - 1) * u
            if 0 <= kk < k:
                out[r * Cout:(r + 1) * Cout, tap * Cin:(tap + 1) * Cin] = w[:, :, kk].t()
                covered[r, kk] = True
    # every kernel tap that reaches output phase r must have been placed
    for r in range(u):
        for kk in range(k):
            if (kk - r - p) % u == 0 and not covered[r, kk]:
                raise ValueError("ConvTranspose1d kernel too long for the 3-tap form")
    return out.contiguous()


class Generator(torch.nn.Module):
    """`nsf_hifigan/models.py:219-276` (inference).  Built from the reference's state dict; `forward(x (1, n_mels, L), f0 (1, L))
    -> (1, 1, L * prod(upsample_rates))`.  `rand_ini` (9,) injects the harmonics' random initial phases (the reference draws
    them with torch.rand; element 0 is forced to 0 like there)."""

    def __init__(self, h, state_dict=None):
        super().__init__()
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        self.upp = int(np.prod(h.upsample_rates))
        if str(h.resblock) != "1":
            raise ValueError("only ResBlock1 generators (every shipped NSF-HiFiGAN config) are built")
        if len(h.resblock_kernel_sizes) > 3:
            raise ValueError("at most three residual blocks per stage")
        self._packed = None
        if state_dict is not None:
            self.load_reference_state(state_dict)

    def load_reference_state(self, sd):
        h = self.h
        P = {}
        P["lin_w"] = sd["m_source.l_linear.weight"].float().reshape(-1).contiguous()
        P["lin_b"] = sd["m_source.l_linear.bias"].float().reshape(-1).contiguous()
        if P["lin_w"].numel() != 9:
            raise ValueError("the source module merges 9 harmonics")
        w = _fold_weight_norm(sd, "conv_pre")
        P["pre_w"], P["pre_b"], P["pre_k"] = _pack_conv(w), sd["conv_pre.bias"].float().contiguous(), w.shape[2]
        P["pre_ws"] = _split_or_none(P["pre_w"])
        ch = h.upsample_initial_channel
        P["ups"], P["noise"], P["res"] = [], [], []
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            w = _fold_weight_norm(sd, f"ups.{i}")
            if w.shape[0] != ch // (2 ** i) or w.shape[1] != ch // (2 ** (i + 1)):
                raise ValueError("upsampling layer shape does not match the config")
            b = sd[f"ups.{i}.bias"].float()
            w_up = _pack_conv_transpose(w, u)
            P["ups"].append((w_up, _split_or_none(w_up), b.repeat(u).contiguous(), u, w.shape[1]))
            nw = sd[f"noise_convs.{i}.weight"].float()
            if i + 1 < len(h.upsample_rates):
                s = int(np.prod(h.upsample_rates[i + 1:]))
                geo = (2 * s, s, s // 2)
            else:
                geo = (1, 1, 0)
            if nw.shape[2] != geo[0]:
                raise ValueError("noise convolution shape does not match the config")
            P["noise"].append((nw.reshape(nw.shape[0], -1).contiguous(), sd[f"noise_convs.{i}.bias"].float().contiguous(), geo))
            blocks = []
            for j, (k, dils) in enumerate(zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes)):
                n = i * self.num_kernels + j
                convs = []
                for t, d in enumerate(dils):
                    w1 = _fold_weight_norm(sd, f"resblocks.{n}.convs1.{t}")
                    w2 = _fold_weight_norm(sd, f"resblocks.{n}.convs2.{t}")
                    p1, p2 = _pack_conv(w1), _pack_conv(w2)
                    convs.append((p1, _split_or_none(p1), sd[f"resblocks.{n}.convs1.{t}.bias"].float().contiguous(), int(d),
                                  p2, _split_or_none(p2), sd[f"resblocks.{n}.convs2.{t}.bias"].float().contiguous(), int(k)))
                blocks.append(convs)
            P["res"].append(blocks)
        w = _fold_weight_norm(sd, "conv_post")
        P["post_w"] = w[0].t().contiguous().reshape(-1)          # (k, C) tap-major
        P["post_b"], P["post_k"] = sd["conv_post.bias"].float().contiguous(), w.shape[2]
        self._packed = P
        self._dev = None

    def to(self, device=None, *a, **k):
        self._target = torch.device(device) if device is not None else None
        return self

    def _on(self, device):
        if self._dev is None or self._dev[0] != device:
            def mv(x):
                if torch.is_tensor(x):
                    return x.to(device)
                if isinstance(x, (list, tuple)):
                    return type(x)(mv(y) for y in x)
                return x
            self._dev = (device, {k: mv(v) for k, v in self._packed.items()})
        return self._dev[1]

    @torch.no_grad()
    def forward(self, x, f0, rand_ini=None):
        if not x.is_cuda:
            raise RuntimeError("the NSF-HiFiGAN generator runs on a HIP device only (no CPU fallback)")
        if x.shape[0] != 1:
            raise ValueError("one utterance per call, like the reference's Enhancer")
        P = self._on(x.device)
        c = hipddsp.context_for(x.device)
        L = x.shape[-1]
        f0 = f0.reshape(-1)[:L].contiguous().float()
        if rand_ini is None:
            rand_ini = torch.rand(9)
        rand_ini = rand_ini.clone().float().reshape(9)
        rand_ini[0] = 0
        src = c.nsf_source(f0, rand_ini.to(x.device), P["lin_w"], P["lin_b"], self.upp, int(self.h.sampling_rate), 0.1)
        # Every convolution below reads leaky_relu(., 0.1) of its producer's result (models.py:60-62, 251): the producers
        # write that activated copy themselves (`act_slope`), next to the raw result where a residual path or the stage mean
        # needs it, so that the consumers take their input as it is (in_slope = 1) and run on the LDS-DMA GEMM.  With
        # split-bf16 products (the context's default) the activated copies of the 64-channel-multiple stages are written in the
        # split operand layout and the weights were converted at load: those convolutions split nothing in their loops.
        use_split = c.math == hipddsp.MATH_SPLIT_BF16

        def can_split(cin, cout):
            """A convolution can write its activated output split when it runs on the DMA kernel itself (Cin % 32 == 0) and
            its output rows are whole 64-column tiles."""
            return use_split and cin % 32 == 0 and cout % 64 == 0

        n_mels = x.shape[1]
        ch0 = int(self.h.upsample_initial_channel)
        act_s = can_split(n_mels, ch0)               # is the current activated tensor in the split layout?
        _, cur_act = c.conv1d(x[0].t().contiguous(), P["pre_w"], P["pre_b"], P["pre_k"], 1, 1.0, want_out=False,
                              act_slope=LRELU_SLOPE, w_split=P["pre_ws"] if act_s else None, act_split=act_s)   # (L, C0)
        T, cin = L, ch0
        cur = None
        for i in range(self.num_upsamples):
            w_up, w_up_s, b_up, u, cout = P["ups"][i]
            nw, nb, (nk, ns, npad) = P["noise"][i]
            T_out = T * u
            x_source = c.nsf_noise_conv(src, nw, nb, nk, ns, npad, T_out)                          # (T_out, cout)
            s_out = can_split(cin, cout)
            # a narrow stage whose residual pairs run fused (x in, x out, activations on load: csrc/nsf.hip, conv_pair*) needs no
            # activated copies at all
            fused = PAIR_FUSION and all(c.conv1d_pair_supported(cout, k, d) for convs in P["res"][i] for (_, _, _, d, _, _, _, k) in convs)
            up, up_act = c.conv1d(cur_act, w_up, b_up, 3, 1, 1.0, residual=x_source.reshape(T, u * cout),
                                  act_slope=None if fused else LRELU_SLOPE,
                                  w_split=w_up_s if (act_s or s_out) else None, x_split=act_s, act_split=s_out), None
            if not fused:
                up, up_act = up
            cur, cur_act = up.reshape(T_out, cout), None if fused else up_act.reshape(T_out, cout)
            T = T_out
            outs = []
            s = s_out                                 # inside a stage every convolution is cout -> cout
            ws_ok = use_split and cout % 32 == 0      # pre-split weights alone still save the weight half of the in-loop split
            for convs in P["res"][i]:
                xr, xr_act = cur, cur_act
                for t, (w1, w1s, b1, d, w2, w2s, b2, k) in enumerate(convs):
                    if fused:
                        xr, _ = c.conv1d_pair(xr, w1, b1, w2, b2, k, d, LRELU_SLOPE)
                        continue
                    _, xt_act = c.conv1d(xr_act, w1, b1, k, d, 1.0, want_out=False, act_slope=LRELU_SLOPE,
                                         w_split=w1s if ws_ok else None, x_split=s, act_split=s)
                    if t + 1 < len(convs):
                        xr, xr_act = c.conv1d(xt_act, w2, b2, k, 1, 1.0, residual=xr, act_slope=LRELU_SLOPE,
                                              w_split=w2s if ws_ok else None, x_split=s, act_split=s)
                    else:
                        xr = c.conv1d(xt_act, w2, b2, k, 1, 1.0, residual=xr, w_split=w2s if ws_ok else None, x_split=s)
                outs.append(xr)
            if i + 1 < self.num_upsamples:
                act_s = use_split and cout % 64 == 0   # the mean kernel writes either layout
                _, cur_act = c.nsf_mean(outs, want_out=False, act_slope=LRELU_SLOPE, act_split=act_s)
            else:
                cur = c.nsf_mean(outs)
            cin = cout
        audio = c.nsf_post(cur, P["post_w"], P["post_b"], P["post_k"], 0.01)
        return audio.reshape(1, 1, -1)

    __call__ = forward


def load_model(model_path, device="cuda"):
    """`nsf_hifigan/models.py:24-39`: config.json beside the checkpoint, `cp_dict['generator']`."""
    with open(os.path.join(os.path.split(model_path)[0], "config.json")) as fh:
        h = AttrDict(json.load(fh))
    cp = torch.load(model_path, map_location="cpu", weights_only=True)
    gen = Generator(h, cp["generator"])
    gen.to(device)
    return gen, h


class NsfHifiGAN(torch.nn.Module):
    """`enhancer.py:81-101`."""

    def __init__(self, model_path, device=None):
        super().__init__()
        self.device = "cuda" if device is None else device
        print("| Load HifiGAN: ", model_path)
        self.model, self.h = load_model(model_path, device=self.device)
        self._stft = None

    def sample_rate(self):
        return self.h.sampling_rate

    def hop_size(self):
        return self.h.hop_size

    def forward(self, audio, f0, rand_ini=None):
        h = self.h
        if self._stft is None:
            self._stft = STFT(h.sampling_rate, h.num_mels, h.n_fft, h.win_size, h.hop_size, h.fmin, h.fmax)
        with torch.no_grad():
            mel = self._stft.get_mel(audio)
            enhanced = self.model(mel, f0[:, :mel.size(-1)], rand_ini=rand_ini)
            return enhanced.reshape(1, -1), h.sampling_rate


class Enhancer:
    """`enhancer.py:9-78`, same constructor and `enhance` signature."""

    def __init__(self, enhancer_type, enhancer_ckpt, device=None):
        self.device = "cuda" if device is None else device
        if enhancer_type == "nsf-hifigan":
            self.enhancer = NsfHifiGAN(enhancer_ckpt, device=self.device)
        else:
            raise ValueError(f" [x] Unknown enhancer: {enhancer_type}")
        self.resample_kernel = {}
        self.enhancer_sample_rate = self.enhancer.sample_rate()
        self.enhancer_hop_size = self.enhancer.hop_size()

    # -- the three decisions of `enhance` that do not touch audio ------------------------------------------------------
    @staticmethod
    def _front_cut(silence_front, sample_rate, hop_size):
        """(frames, samples, seconds) of the silent front that is skipped and padded back (enhancer.py:27-31,76-77)."""
        frames = int(silence_front * sample_rate / hop_size)
        seconds = frames * hop_size / sample_rate
        return frames, int(np.round(seconds * sample_rate)), seconds

    def _working_rate(self, adaptive_key, f0):
        """The rate the generator is run at for a key shift of `adaptive_key` semitones (enhancer.py:33-43): the audio is
        treated as if sampled at 100 * round(sr_e * 2^(key/12) / 100) Hz, which moves every pitch down by the key.
        'auto' picks the smallest non-negative key that brings the highest f0 under 760 Hz - that one number decides
        tensor LENGTHS, so it is the single value read back from the device (a max over a few hundred frames)."""
        if isinstance(adaptive_key, str):
            if adaptive_key != "auto":
                raise ValueError(f"adaptive_key must be a number or 'auto', got {adaptive_key!r}")
            adaptive_key = max(0, np.ceil(12 * np.log2(float(torch.max(f0) / 760))))
            print("auto_adaptive_key: " + str(int(adaptive_key)))
        shrink = 2 ** (-float(adaptive_key) / 12)
        rate = 100 * int(np.round(self.enhancer_sample_rate / shrink / 100))
        return rate, self.enhancer_sample_rate / rate, shrink

    def _resampled(self, x, rate_in, rate_out):
        if rate_in == rate_out:
            return x
        pair = (int(rate_in), int(rate_out))
        if pair not in self.resample_kernel:
            self.resample_kernel[pair] = Resample(rate_in, rate_out, lowpass_filter_width=128)
        return self.resample_kernel[pair](x)

    def enhance(self, audio, sample_rate, f0, hop_size, adaptive_key=0, silence_front=0, rand_ini=None):
        """audio (1,T), f0 (1,n_frames,1) -> (enhanced (1,T'), enhancer sample rate); reference `enhancer.py:24-78`.
        Device only: the f0 track is re-timed to the generator's frames by `ddsp_retime_f0`, nothing is copied to the host
        (with adaptive_key='auto' one scalar - the highest f0 - is read back, see `_working_rate`)."""
        cut_frames, cut_samples, cut_seconds = self._front_cut(silence_front, sample_rate, hop_size)
        audio, f0 = audio[:, cut_samples:], f0[:, cut_frames:, :]
        work_rate, pitch_scale, shrink = self._working_rate(adaptive_key, f0)
        audio_res = self._resampled(audio, sample_rate, work_rate)
        n_frames = int(audio_res.size(-1) // self.enhancer_hop_size + 1)
        # f0 of frame i of the generator = the track (scaled by pitch_scale, its time axis divided by it) at i * hop_e / sr_e
        f0_res = hipddsp.context_for(audio.device).retime_f0(f0, hop_size / sample_rate, pitch_scale, pitch_scale,
                                                               self.enhancer_hop_size / self.enhancer_sample_rate, n_frames)[None]
        enhanced, sr_e = self.enhancer(audio_res, f0_res, rand_ini=rand_ini)
        if shrink != 0:
            enhanced = self._resampled(enhanced, work_rate, sr_e)
        if cut_frames > 0:
            enhanced = F.pad(enhanced, (int(np.round(sr_e * cut_seconds)), 0))
        return enhanced, sr_e
