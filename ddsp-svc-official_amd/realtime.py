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


class StreamRenderer:
    """One real-time stream on one GPU: the device half of the reference's callback chain
    (`gui.GUI.audio_callback`, gui.py:367-433, calling `gui.SvcDDSP.infer`, gui.py:69-140).

    Per block of `block` input samples:
      1. the sliding input window takes the block (`input_wav[:] = append(input_wav[block:], indata)`, gui.py:373-374);
      2. frame volume of the window (`Volume_Extractor.extract`, gui.py:105-106) -> `ddsp_volume_extract`;
      3. units / f0 of the window from the caller's analysis front end (`features(window) -> (units (1,Fr,C), f0 (1,Fr,1))`:
         the f0 extractor and the units encoder are third-party models outside this path);
      4. the synthesiser forward (gui.py:125-126), eager or replayed from a HIP graph (`graphed.GraphedSynth`);
      5. `output *= mask` with the 9-frame dilated volume gate (gui.py:107-112,127) -> `ddsp_volume_gate`, in place;
      6. SOLA search + cross-fade + tail hand-over (gui.py:405-430) -> `Splicer.push`.
    Nothing in the chain synchronises with the host; the returned (block,) tensor is what the callback copies out.
    Streams are independent: eight streams are eight renderers on eight GPUs (replicas only)."""

    def __init__(self, model, samplerate, block_time, crossfade_time, device, buffer_num=4, threshold_db=-45.0, spk_id=1,
                 features=None, use_graph=True, use_phase_vocoder=False):
        self.model = model.eval()
        self.device = torch.device(device)
        self.hop = int(model.block_size)
        self.threshold_db = float(threshold_db)
        self.splicer = Splicer(samplerate, block_time, crossfade_time, self.device, use_phase_vocoder=use_phase_vocoder)
        self.block = self.splicer.block
        self.n_in = self.splicer.input_frames(buffer_num)
        self.frames = self.n_in // self.hop + 1                      # frames of the window (f0 / units / volume alike)
        self.window = torch.zeros(self.n_in, device=self.device)   # `self.input_wav`, gui.py:346
        self.spk_id = torch.full((1, 1), int(spk_id), dtype=torch.int64, device=self.device)
        self.features = features
        self.graph = None
        if use_graph:
            import graphed
            self.graph = graphed.GraphedSynth(self.model, 1, self.frames)

    @torch.no_grad()
    def push_block(self, block_in, units=None, f0=None, noise=None):
        """block_in (block,) device samples of the stream -> (block,) samples to play.  `units` (1, Fr, C) and `f0`
        (1, Fr, 1) of the CURRENT window may be passed instead of a `features` callable; `noise` (1, Fr*hop) in [0, 1)
        replaces the fresh draw (parity tests)."""
        if block_in.numel() != self.block:
            raise ValueError(f"StreamRenderer: a block is {self.block} samples, got {block_in.numel()}")
        self.window = torch.cat([self.window[self.block:], block_in.reshape(-1).to(self.device, torch.float32)])
        ctx = hipddsp.context_for(self.device)
        volume = ctx.volume_extract(self.window[None], self.hop)       # (1, Fr)
        if units is None or f0 is None:
            if self.features is None:
                raise ValueError("StreamRenderer: pass units and f0, or construct it with a `features` callable")
            units, f0 = self.features(self.window)
        if units.shape[1] != self.frames or f0.shape[1] != self.frames or volume.shape[1] != self.frames:
            raise ValueError(f"StreamRenderer: the window has {self.frames} frames")
        if self.graph is not None:
            sig = self.graph(units, f0, volume, self.spk_id, noise=noise)[0]
        elif noise is not None:
            sig = self.model(units, f0, volume, self.spk_id, noise=noise)[0]
        else:
            sig = self.model(units, f0, volume, self.spk_id)[0]
        ctx.volume_gate_(sig, volume, self.threshold_db, self.hop)
        return self.splicer.push(sig[0])
