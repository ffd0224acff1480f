"""The splice half of the reference's real-time callback (`gui.py:373-430`) around the device path: sliding input
window bookkeeping is the caller's (PortAudio), the SOLA search + sin^2 cross-fade + tail hand-over run as two
kernels on the stream's own GPU with no host synchronisation.  One `Splicer` per stream (eight
independent streams = eight replicas, no collective)."""
import torch

import hipddsp


def phase_vocoder(a, b, fade_out, fade_in):
    """`gui.phase_vocoder` (`gui.py:14-31`) on the device: all four arguments are (n,) device tensors."""
    return hipddsp.context_for(a.device).phase_vocoder(a, b, fade_out, fade_in)


class Splicer:
    def __init__(self, samplerate, block_time, crossfade_time, device, search_time=0.01, delay_time=0.02,
                 use_phase_vocoder=False):
        """Sizes as the reference derives them (`gui.py:319-322`)."""
        self.block = int(block_time * samplerate)
        self.xfade = int(crossfade_time * samplerate)
        self.search = int(search_time * samplerate)
        self.delay = int(delay_time * samplerate)
        self.device = torch.device(device)
        self.buffer = torch.zeros(self.xfade, device=self.device)      # `sola_buffer`, gui.py:347
        self.last_shift = None
        # gui.py:349-351 windows, used by the optional phase-vocoder splice (gui.py:417-423)
        self.use_phase_vocoder = bool(use_phase_vocoder)
        self.fade_in = torch.sin(torch.pi * torch.arange(0, 1, 1 / self.xfade, device=self.device)[:self.xfade] / 2) ** 2
        self.fade_out = 1 - self.fade_in

    @property
    def ctx(self):
        """The calling thread's context (the audio callback runs on PortAudio's thread, not on the one that built the
        splicer: each thread gets its own scratch arena)."""
        return hipddsp.context_for(self.device)

    def input_frames(self, buffer_num):
        """Length of the sliding input window (`gui.py:323-325`)."""
        return max(self.block + self.xfade + self.search + 2 * self.delay, (1 + buffer_num) * self.block)

    def push(self, audio):
        """audio (N,) model output for the current window -> (block,) samples to play (mono; the reference
        duplicates them to two channels on the host, `gui.py:430`)."""
        kept = self.buffer.clone() if self.use_phase_vocoder else None
        emitted, shift = self.ctx.sola(audio, self.buffer, self.block, self.xfade, self.search, self.delay)
        self.last_shift = shift
        if self.use_phase_vocoder:
            # head of the new block at the SOLA shift (the shift stays on the device: indexed gather, no host sync)
            start = audio.numel() - self.block - self.xfade - self.search - self.delay
            idx = start + shift.to(torch.int64) + torch.arange(self.xfade, device=self.device)
            head = audio.reshape(-1).index_select(0, idx)
            emitted[:self.xfade] = self.ctx.phase_vocoder(kept, head, self.fade_out, self.fade_in)
        return emitted
