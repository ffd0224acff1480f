"""ORACLE (test infrastructure only): CPU restatement of the NSF-HiFiGAN post-net the reference's `Enhancer` runs
(`nsf_hifigan/models.py:106-276` SineGen / SourceModuleHnNSF / Generator with ResBlock1; `nsf_hifigan/nvSTFT.py:65-119`
`STFT.get_mel`; `enhancer.py:24-78` `Enhancer.enhance`).  Pinned by tests/golden/ref_enhancer.npz, which
tests/golden/make_golden.py (tier f) produced by running the reference's own `nsf_hifigan/models.py` (imported as is: it
needs only torch and numpy) on seeded weights.  PARITY UNPINNED at two third-party boundaries that are not in the image:
librosa's mel filter bank (restated in the product's `enhancer.mel_filterbank`, passed in here) and torchaudio's
resampler (oracle/resample.py)."""
import numpy as np
import torch
import torch.nn.functional as F

LRELU = 0.1


def fold(sd, prefix):
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    g, v = sd[prefix + ".weight_g"], sd[prefix + ".weight_v"]
    return torch._weight_norm(v, g, 0)


def sine_source(sd, f0, upp, sr, rand_ini, sine_amp=0.1):
    """ref: models.py:141-177,212-216.  f0 (1, L) -> (1, L*upp, 1) merged source.  rand_ini (1, 9) replaces torch.rand."""
    f0 = f0.unsqueeze(-1)
    fn = torch.multiply(f0, torch.arange(1, 10).reshape(1, 1, -1))
    rad = (fn / sr) % 1
    ri = rand_ini.clone()
    ri[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + ri
    tmp = torch.cumsum(rad.double(), 1).float()
    tmp *= upp
    tmp = F.interpolate(tmp.transpose(2, 1), scale_factor=upp, mode="linear", align_corners=True).transpose(2, 1)
    rad_up = F.interpolate(rad.transpose(2, 1), scale_factor=upp, mode="nearest").transpose(2, 1)
    tmp %= 1
    idx = (tmp[:, 1:, :] - tmp[:, :-1, :]) < 0
    shift = torch.zeros_like(rad_up)
    shift[:, 1:, :] = idx * -1.0
    sines = torch.sin(torch.cumsum(rad_up.double() + shift.double(), dim=1) * 2 * np.pi).float() * sine_amp
    return torch.tanh(F.linear(sines, sd["m_source.l_linear.weight"], sd["m_source.l_linear.bias"]))


def generator(sd, h, mel, f0, rand_ini):
    """ref: models.py:251-272.  mel (1, n_mels, L), f0 (1, L) -> (1, 1, L*upp)."""
    rates, ksz = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
    upp = int(np.prod(rates))
    nk = len(h["resblock_kernel_sizes"])
    src = sine_source(sd, f0, upp, h["sampling_rate"], rand_ini).transpose(1, 2)
    x = F.conv1d(mel, fold(sd, "conv_pre"), sd["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, LRELU)
        x = F.conv_transpose1d(x, fold(sd, f"ups.{i}"), sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if i + 1 < len(rates):
            s = int(np.prod(rates[i + 1:]))
            xs = F.conv1d(src, sd[f"noise_convs.{i}.weight"], sd[f"noise_convs.{i}.bias"], stride=s, padding=s // 2)
        else:
            xs = F.conv1d(src, sd[f"noise_convs.{i}.weight"], sd[f"noise_convs.{i}.bias"])
        x = x + xs
        acc = None
        for j, (kk, dils) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            n = i * nk + j
            y = x
            for t, d in enumerate(dils):
                yt = F.leaky_relu(y, LRELU)
                yt = F.conv1d(yt, fold(sd, f"resblocks.{n}.convs1.{t}"), sd[f"resblocks.{n}.convs1.{t}.bias"], dilation=d,
                              padding=(kk * d - d) // 2)
                yt = F.leaky_relu(yt, LRELU)
                yt = F.conv1d(yt, fold(sd, f"resblocks.{n}.convs2.{t}"), sd[f"resblocks.{n}.convs2.{t}.bias"],
                              padding=(kk - 1) // 2)
                y = yt + y
            acc = y if acc is None else acc + y
        x = acc / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, fold(sd, "conv_post"), sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def log_mel(y, h, mel_basis, clip_val=1e-5):
    """ref: nvSTFT.py:65-117 for keyshift 0.  y (1, T) -> (1, n_mels, frames); mel_basis (n_mels, n_fft//2+1) from the caller."""
    n, hop = h["n_fft"], h["hop_size"]
    pad_left = (n - hop) // 2
    pad_right = max((n - hop + 1) // 2, n - y.size(-1) - pad_left)
    mode = "reflect" if pad_right < y.size(-1) else "constant"
    yp = F.pad(y.unsqueeze(1), (pad_left, pad_right), mode=mode).squeeze(1)
    spec = torch.stft(yp, n, hop_length=hop, win_length=n, window=torch.hann_window(n), center=False, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    mag = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(mel_basis, mag), min=clip_val))
