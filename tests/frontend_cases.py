"""Shapes and seeds of tests/golden/glue_frontend.npz (the same tables as tests/golden/make_golden.py tier d)."""
import numpy as np
import torch

FRONTEND_VOLUME = [(88064, 512), (44100, 512), (1000, 64), (513, 512), (700, 441), (5000, 7)]      # (T, hop)
FRONTEND_ALIGN = [(88064, 44100, 512, 101, 256), (32000, 16000, 320, 101, 64), (16000, 16000, 480, 40, 8),
                  (88064, 44100, 512, 60, 12), (48000, 48000, 557, 52, 768)]                        # (n, sr, hop, Lu, C)


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def volume_audio(i):
    T, hop = FRONTEND_VOLUME[i]
    return _rng(700 + i).uniform(-1, 1, size=T).astype(np.float32), hop


def align_units_input(i):
    n, sr, hop, Lu, C = FRONTEND_ALIGN[i]
    return torch.from_numpy(_rng(800 + i).standard_normal((1, Lu, C)).astype(np.float32)), n, sr, hop


# ---- phase-vocoder cross-fade (tests/golden/glue_phase_vocoder.npz; the same inputs as make_golden.py pv_inputs) ----
PV_SIZES = [1764, 441, 64]
SR = 44100


def pv_inputs(i):
    n = PV_SIZES[i]
    r = _rng(900 + i)
    t = np.arange(n) / SR
    a = 0.3 * np.sin(2 * np.pi * 220.0 * t + 0.4) + 0.1 * np.sin(2 * np.pi * 1330.0 * t + 1.1) + 0.01 * r.standard_normal(n)
    b = 0.3 * np.sin(2 * np.pi * 220.0 * t + 1.3) + 0.1 * np.sin(2 * np.pi * 1330.0 * t - 0.6) + 0.01 * r.standard_normal(n)
    fi = torch.sin(np.pi * torch.arange(0, 1, 1 / n) / 2)[:n] ** 2
    f32 = lambda x: torch.from_numpy(np.asarray(x, dtype=np.float32))
    return f32(a), f32(b), 1 - fi, fi
