"""The splice half of the reference's real-time callback (`gui.py:373-430`) around the device path: sliding input
window bookkeeping is the caller's (PortAudio), the SOLA search + sin^2 cross-fade + tail hand-over run as two
kernels on the stream's own GPU with no host synchronisation.  One `Splicer` per stream (eight
independent streams = eight replicas, no collective)."""
import torch

import hipddsp


class Splicer:
    def __init__(self, samplerate, block_time, crossfade_time, device, search_time=0.01, delay_time=0.02):
        """Sizes as the reference derives them (`gui.py:319-322`)."""
        self.block = int(block_time * samplerate)
        self.xfade = int(crossfade_time * samplerate)
        self.search = int(search_time * samplerate)
        self.delay = int(delay_time * samplerate)
        self.device = torch.device(device)
        self.ctx = hipddsp.context_for(self.device)
        self.buffer = torch.zeros(self.xfade, device=self.device)      # `sola_buffer`, gui.py:347
        self.last_shift = None

    def input_frames(self, buffer_num):
        """Length of the sliding input window (`gui.py:323-325`)."""
        return max(self.block + self.xfade + self.search + 2 * self.delay, (1 + buffer_num) * self.block)

    def push(self, audio):
        """audio (N,) model output for the current window -> (block,) samples to play (mono; the reference
        duplicates them to two channels on the host, `gui.py:430`)."""
        emitted, shift = self.ctx.sola(audio, self.buffer, self.block, self.xfade, self.search, self.delay)
        self.last_shift = shift
        return emitted
