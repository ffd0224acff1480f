"""Seeded inputs of the tier-E fixtures (tests/golden/ref_loss.npz, ref_train_step.npz, ref_gui_stream.npz,
ref_offline_glue.npz).  tests/golden/make_golden.py imports these when it runs the reference; the tests import them to
rebuild the same inputs, so only the reference's OUTPUTS are stored."""
import numpy as np
import torch

SR, HOP = 44100, 512


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def f32(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


# ---- G8: spectral loss (ddsp/loss.py) ---------------------------------------------------------------------------------
LOSS_SCALES = [256, 257, 1000, 1531, 2047]          # smallest, a prime, an even mid size, an odd composite, the largest
LOSS_BT = (3, 88064)
RSS_SEED = 11                                       # torch.manual_seed before RSSLoss.forward (its randint draw)
# (n_fft, overlap) of tests/golden/ref_loss_overlap.npz: hops 250, 128, 1432 (a frame count of 1 + (T - N)//hop each),
# and one whose hop int(N * (1 - overlap)) is decided by the floating-point product (1531 * 0.3 = 459.3 -> 459)
LOSS_OVERLAP_CASES = [(1000, 0.75), (257, 0.5), (2047, 0.3), (1531, 0.7)]


def loss_signals(seed=1, B=LOSS_BT[0], T=LOSS_BT[1]):
    """(x_pred, x_true): noise plus a partial each, a semitone apart."""
    r = _rng(seed)
    t = np.arange(T) / SR
    xt = 0.1 * r.standard_normal((B, T)) + 0.2 * np.sin(2 * np.pi * 220 * t)[None]
    xp = 0.1 * r.standard_normal((B, T)) + 0.15 * np.sin(2 * np.pi * 233 * t + 0.3)[None]
    return f32(xp), f32(xt)


# ---- G9: training steps (solver.py:110-114, train.py:41-45) -----------------------------------------------------------------
TRAIN_B, TRAIN_FR, TRAIN_STEPS = 3, 40, 3
TRAIN_LR, TRAIN_WD = 0.0005, 0.0                    # configs/combsub.yaml: train.lr, train.weight_decay
TRAIN_WEIGHT_SEED, TRAIN_INPUT_SEED, TRAIN_DRAW_SEED = 20240117 + 21, 20240117 + 22, 5


def train_target(B=TRAIN_B, Fr=TRAIN_FR):
    """Target audio 0.1*N(0,1)."""
    return f32(0.1 * _rng(TRAIN_INPUT_SEED + 1).standard_normal((B, Fr * HOP)))


# ---- G10: GUI stream (gui.py:367-430) -----------------------------------------------------------------------------------
GUI_SR, GUI_BLOCK_TIME, GUI_XFADE_TIME, GUI_BUFFER_NUM = 44100, 0.2, 0.04, 4
GUI_BLOCKS = 8
GUI_MODEL_LEN = 44544                               # 87 frames of 512: what the model returns for the 44 100-sample window


def gui_model_output(k):
    """What `svc_model.infer` returns at block k: a drifting 196 Hz partial with a random time offset plus noise."""
    r = _rng(3000 + k)
    t = (np.arange(GUI_MODEL_LEN) + k * int(GUI_BLOCK_TIME * GUI_SR) + r.integers(-200, 200)) / GUI_SR
    return f32(0.3 * np.sin(2 * np.pi * 196.0 * t) + 0.05 * np.sin(2 * np.pi * 588.0 * t + 0.7)
               + 0.01 * r.standard_normal(GUI_MODEL_LEN))


def gui_indata(k):
    """The (block, 2) input block PortAudio hands to the callback (only its mono mix enters the sliding window)."""
    n = int(GUI_BLOCK_TIME * GUI_SR)
    return _rng(3100 + k).uniform(-0.5, 0.5, size=(n, 2)).astype(np.float32)


# ---- volume gate as SvcDDSP.infer applies it (gui.py:103-112,125-127) ------------------------------------------------------
GATE_T, GATE_THRESHOLD = 44100, -45                 # 44 100 samples -> 87 volume frames -> 44 544 output samples


def gate_audio():
    """Input audio of the sliding window: a partial whose level dips below the threshold in two stretches."""
    r = _rng(3200)
    t = np.arange(GATE_T) / SR
    env = np.ones(GATE_T)
    env[9000:14000] = 1e-4
    env[30000:30700] = 1e-4                         # shorter than the 9-frame dilation: stays open
    env[38000:] = 3e-3
    return (0.2 * env * np.sin(2 * np.pi * 330 * t) + 1e-5 * r.standard_normal(GATE_T)).astype(np.float32)


def gate_model_output():
    return f32(_rng(3201).uniform(-1, 1, size=(1, GUI_MODEL_LEN)))


# ---- slice cross-fade (main.py:50-57) ------------------------------------------------------------------------------------------
def crossfade_cases():
    r = _rng(3300)
    return [(r.standard_normal(5000), r.standard_normal(4000), 4200),
            (r.standard_normal(300), r.standard_normal(1000), 0),          # fade over all of a
            (r.standard_normal(777), r.standard_normal(50), 740)]          # b shorter than a; fade 37 samples


# ---- tier F: NSF-HiFiGAN post-net (nsf_hifigan/models.py) -------------------------------------------------------------------------
# a small generator (fixtures stay small) with every structural feature of the shipped one: two transposed-convolution
# geometries (k = 2u with u = 4 and u = 2), strided and 1-tap noise convolutions, three residual blocks per stage with
# kernel sizes 3 / 7 / 11 and dilations 1 / 3 / 5
NSF_CONFIG = {"resblock": "1", "upsample_rates": [4, 4, 2], "upsample_kernel_sizes": [8, 8, 4], "upsample_initial_channel": 64,
              "resblock_kernel_sizes": [3, 7, 11], "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
              "num_mels": 16, "sampling_rate": 44100, "hop_size": 32, "n_fft": 128, "win_size": 128, "fmin": 40, "fmax": 16000}
NSF_L = 23                      # frames
NSF_WEIGHT_SEED, NSF_INPUT_SEED = 4242, 4243


def nsf_state_dict(cfg=NSF_CONFIG, seed=NSF_WEIGHT_SEED):
    """A generator checkpoint in the reference's key layout (weight_g / weight_v where it applies weight_norm), seeded.
    Weight scales are chosen so that activations stay O(1) through the stack (the reference's 0.01-std init would make the
    output vanish and test nothing)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    def wn(prefix, shape, fan_in, gain=1.0):
        v = rn(*shape)
        sd[prefix + ".weight_v"] = v
        sd[prefix + ".weight_g"] = (v.norm(dim=(1, 2), keepdim=True) * (0.9 + 0.2 * torch.rand(shape[0], 1, 1, generator=g))
                                    * (gain / np.sqrt(fan_in)))
        sd[prefix + ".bias"] = rn(shape[1] if "ups" in prefix else shape[0], scale=0.05)

    ch = cfg["upsample_initial_channel"]
    sd["m_source.l_linear.weight"] = rn(1, 9, scale=0.6)
    sd["m_source.l_linear.bias"] = rn(1, scale=0.1)
    wn("conv_pre", (ch, cfg["num_mels"], 7), cfg["num_mels"] * 7)
    rates = cfg["upsample_rates"]
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(rates, cfg["upsample_kernel_sizes"])):
        cin, cout = ch // 2 ** i, ch // 2 ** (i + 1)
        wn(f"ups.{i}", (cin, cout, k), cin * k / u)
        kk = 2 * int(np.prod(rates[i + 1:])) if i + 1 < len(rates) else 1
        sd[f"noise_convs.{i}.weight"] = rn(cout, 1, kk, scale=0.5 / np.sqrt(kk))
        sd[f"noise_convs.{i}.bias"] = rn(cout, scale=0.05)
        for j, ks in enumerate(cfg["resblock_kernel_sizes"]):
            for t in range(3):
                wn(f"resblocks.{i * nk + j}.convs1.{t}", (cout, cout, ks), cout * ks)
                wn(f"resblocks.{i * nk + j}.convs2.{t}", (cout, cout, ks), cout * ks)
    wn("conv_post", (1, ch // 2 ** len(rates), 7), ch // 2 ** len(rates) * 7, gain=0.25)
    return sd


def nsf_inputs(cfg=NSF_CONFIG, L=NSF_L, seed=NSF_INPUT_SEED):
    r = _rng(seed)
    mel = f32(r.standard_normal((1, cfg["num_mels"], L)))
    f0 = 220.0 * 2.0 ** (0.3 * np.sin(np.arange(L) / 5.0)) + 5.0 * r.standard_normal(L)
    f0[3:6] = 0.0                                   # an unvoiced stretch
    rand_ini = f32(r.random((1, 9)))
    return mel, f32(f0[None]), rand_ini


def nsf_audio(T=2000, seed=NSF_INPUT_SEED + 1):
    r = _rng(seed)
    t = np.arange(T) / 44100
    return f32((0.4 * np.sin(2 * np.pi * 330 * t) + 0.05 * r.standard_normal(T))[None])
