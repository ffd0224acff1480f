"""Oracle: random-scale spectral loss (TEST INFRASTRUCTURE, see oracle/__init__.py).

CPU restatement of reference `ddsp/loss.py:7-43`.  PARITY UNPINNED at the
torchaudio boundary: `torchaudio.transforms.Spectrogram(n_fft=N, hop_length=N,
power=1, normalized=True, center=False)` (`ddsp/loss.py:14`) is not installed;
it is restated from its documented semantics as a Hann(periodic)-windowed,
non-overlapping, one-sided STFT magnitude divided by sqrt(sum(window**2)).
"""
import torch


def hop_length(n_fft, overlap=0):
    """ref: ddsp/loss.py:13."""
    return int(n_fft * (1 - overlap))


def spectrogram(x, n_fft, overlap=0):
    """(B, T) -> (B, n_fft//2+1, 1 + (T - n_fft)//hop) magnitudes, window-normalised."""
    win = torch.hann_window(n_fft, dtype=x.dtype, device=x.device)
    st = torch.stft(x, n_fft, hop_length=hop_length(n_fft, overlap), win_length=n_fft, window=win, center=False,
                    onesided=True, return_complex=True)
    return st.abs() / win.pow(2).sum().sqrt()


def sss_loss(x_true, x_pred, n_fft, alpha=1.0, eps=1e-7, overlap=0):
    """One scale: spectral convergence + log-magnitude L1.  ref: ddsp/loss.py:16-25."""
    s_t = spectrogram(x_true, n_fft, overlap) + eps
    s_p = spectrogram(x_pred, n_fft, overlap) + eps
    conv = torch.mean(torch.linalg.norm(s_t - s_p, dim=(1, 2)) / torch.linalg.norm(s_t + s_p, dim=(1, 2)))
    logt = torch.nn.functional.l1_loss(s_t.log(), s_p.log())
    return conv + alpha * logt


def rss_loss(x_pred, x_true, n_ffts, alpha=1.0, eps=1e-7, overlap=0):
    """Mean of sss_loss over the given scales.  ref: ddsp/loss.py:37-43 (the reference
    draws `n_ffts = torch.randint(fft_min, fft_max, (n_scale,))`; callers pass the draw)."""
    total = 0.0
    for n in n_ffts:
        total = total + sss_loss(x_true, x_pred, int(n), alpha, eps, overlap)
    return total / len(n_ffts)


def draw_scales(fft_min, fft_max, n_scale, generator=None):
    """The reference's draw (`ddsp/loss.py:39`): integers in [fft_min, fft_max)."""
    return [int(v) for v in torch.randint(fft_min, fft_max, (n_scale,), generator=generator)]
