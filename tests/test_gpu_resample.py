"""Device-side resampler against an fp64 evaluation of torchaudio's published algorithm
(oracle/resample.py).  PARITY UNPINNED at the torchaudio boundary (package absent, no reference fixture)."""
import numpy as np
import pytest
import torch

from oracle import resample as OR

pytestmark = pytest.mark.gpu


def _signal(seed, B, T, sr):
    r = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(T) / sr
    x = 0.3 * np.sin(2 * np.pi * 220 * t) + 0.1 * np.sin(2 * np.pi * 5000 * t + 0.3) + 0.05 * r.standard_normal((B, T))
    return torch.from_numpy(x.astype(np.float32))


@pytest.mark.parametrize("orig,new,lpw,T", [(44100, 48000, 128, 44544), (48000, 44100, 128, 20000), (44100, 16000, 128, 30001),
                                            (44100, 46700, 128, 9000), (16000, 44100, 6, 777), (44100, 22050, 16, 4),
                                            (32000, 48000, 64, 1),
                                            # reduced rates 4409 / 4800: the window of 8 frames does not fit the LDS -> the
                                            # one-output-per-thread kernel; (7, 3): a table narrower than a workgroup
                                            (44090, 48000, 16, 9000), (7, 3, 4, 100), (44100, 52400, 128, 30000)])
def test_resample_against_fp64(ctx, dev, orig, new, lpw, T):
    x = _signal(T, 2, T, orig)
    want = OR.resample(x, orig, new, lpw, dtype=torch.float64)
    got = ctx.resample(x.to(dev), orig, new, lpw)
    assert got.shape == want.shape == (2, int(np.ceil(new * T / orig)))
    err = float((got.cpu().double() - want).abs().max())
    assert err < 3e-6 * max(1.0, float(want.abs().max())), err      # (400-term fp32 sums)
    # the fp32 CPU restatement is no closer to fp64 than the kernel is
    cpu32 = float((OR.resample(x, orig, new, lpw).double() - want).abs().max())
    assert err < max(3 * cpu32, 5e-7)


def test_resample_module_mirror(dev, lib_path):
    """The torchaudio-shaped mirror: constructor, .to(), call on (T,), (B,T) and (B,1,T); identity when the rates agree;
    no CPU path.  A tone keeps its frequency and level through 44.1k -> 48k -> 44.1k."""
    from resample import Resample
    rs = Resample(44100, 48000, lowpass_filter_width=128).to(dev)
    x = _signal(3, 3, 22050, 44100).to(dev)
    y = rs(x)
    assert y.shape == (3, 24000)
    assert rs(x[0]).shape == (24000,) and torch.equal(rs(x[0]), y[0])
    assert rs(x[:, None]).shape == (3, 1, 24000)
    back = Resample(48000, 44100, lowpass_filter_width=128).to(dev)(y)
    core = slice(2000, 20000)
    assert float((back[:, core] - x[:, core]).abs().max()) < 0.35            # band-limited copy (white noise above the
    t = np.arange(22050) / 44100                                              # passband edge is what differs)
    tone = torch.from_numpy((0.3 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32)).to(dev)
    rt = Resample(48000, 44100, lowpass_filter_width=128)(Resample(44100, 48000, lowpass_filter_width=128)(tone))
    assert float((rt[core] - tone[core]).abs().max()) < 2e-3
    assert Resample(44100, 44100)(x) is x
    with pytest.raises(RuntimeError):
        rs(x.cpu())
    with pytest.raises(ValueError):
        Resample(44100, 48000, resampling_method="sinc_interp_kaiser")


def test_resample_table_cache_evicts(ctx, dev):
    """tap tables are cached per context, least recently used evicted beyond eight (a server running
    `Enhancer.enhance(adaptive_key='auto')` keeps asking for new rate pairs).  Twenty distinct pairs, the first one used again
    at the end (re-built after its eviction), a pair whose reduced `orig` is >= 32768 (the old key overflowed an int)."""
    x = _signal(5, 1, 3000, 44100)
    first = None
    for i in range(20):
        new = 44100 + 100 * (i + 1)
        got = ctx.resample(x.to(dev), 44100, new, 16)
        if i == 0:
            first = got.clone()
    again = ctx.resample(x.to(dev), 44100, 44200, 16)
    assert torch.equal(again, first)
    want = OR.resample(x, 44100, 44200, 16, dtype=torch.float64)
    assert float((again.cpu().double() - want).abs().max()) < 3e-6
    y = _signal(6, 1, 40000, 40001)[:, :40000]
    got = ctx.resample(y.to(dev), 40001, 3, 4)               # reduced rates 40001 / 3
    want = OR.resample(y, 40001, 3, 4, dtype=torch.float64)
    assert got.shape == want.shape and float((got.cpu().double() - want).abs().max()) < 3e-6 * max(1.0, float(want.abs().max()))
