"""`Resample(orig_freq, new_freq, lowpass_filter_width=6)` with the constructor, `.to(device)` and call signature of
`torchaudio.transforms.Resample` as the reference uses it (`gui.py:399-404`: model rate -> audio-device rate;
`enhancer.py:50-53,69-73`: around the adaptive-key trick), executed by libddsp_amd on the device (no CPU path).
torchaudio itself is not installed: the algorithm is its published windowed-sinc polyphase resampler (Hann window,
rolloff 0.99) and parity at that boundary is unpinned; tests/test_gpu_resample.py holds the kernels to an fp64
evaluation of the same formulas."""
import torch

import hipddsp


class Resample(torch.nn.Module):
    def __init__(self, orig_freq=16000, new_freq=16000, resampling_method="sinc_interp_hann", lowpass_filter_width=6,
                 rolloff=0.99, beta=None, *, dtype=None):
        super().__init__()
        if resampling_method not in ("sinc_interp_hann", "sinc_interpolation"):
            raise ValueError("only the Hann-windowed sinc method (torchaudio's default, the one the reference uses) is built")
        if rolloff != 0.99:
            raise ValueError("only torchaudio's default rolloff 0.99 is built")
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self.lowpass_filter_width = int(lowpass_filter_width)

    def forward(self, waveform):
        if self.orig_freq == self.new_freq:
            return waveform
        if not waveform.is_cuda:
            raise RuntimeError("Resample runs on a HIP device only (no CPU fallback)")
        shape = waveform.shape
        flat = waveform.reshape(-1, shape[-1])
        out = hipddsp.context_for(waveform.device).resample(flat, self.orig_freq, self.new_freq, self.lowpass_filter_width)
        return out.reshape(*shape[:-1], out.shape[-1])
