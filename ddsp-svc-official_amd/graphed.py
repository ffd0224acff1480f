"""HIP-graph replay of a synthesiser's inference forward for one fixed (batch, frames) shape.

The real-time caller (`gui.py:360-430`: one 0.2 s block at a time, B = 1, 87 frames) is launch-bound: ~50 kernels of
a few microseconds each, paced by Python + ctypes + hipLaunch on the host.  `GraphedSynth` captures the whole
`model(...)` call once (`torch.cuda.graph`: stream capture also records the kernels libddsp_amd launches on that
stream) and replays it with one host call per block.

What makes the capture safe:
  * the model's library calls go through a `hipddsp.Context` of their own (`hipddsp.use_context`): the captured kernels
    point into THAT context's scratch arena, tables and zero page, which are sized by eager warm-up runs before the
    capture and frozen afterwards (`Context.freeze`) - no later eager call can regrow or reuse them;
  * the noise excitation comes from a static device tensor that is refilled before every replay (the eager path draws
    a host seed per call, which a graph would freeze);
  * inputs are copied into static tensors, outputs are static tensors valid until the next replay.
Weights are read by the captured kernels at replay time (the weight-preparation kernel is part of the graph), so
in-place weight updates are honoured; `spk_mix_dict` and `initial_phase` are host-side arguments and not supported
here.  Forward only.
"""
import torch

import hipddsp


class GraphedSynth:
    def __init__(self, model, B, Fr, warmup=3):
        p = next(model.parameters())
        if not p.is_cuda:
            raise RuntimeError("GraphedSynth needs the model on a HIP device (no CPU fallback)")
        self.model = model.eval()
        self.device = p.device
        self.B, self.Fr = int(B), int(Fr)
        hop = int(model.block_size)
        n_unit = model.unit2ctrl.n_unit
        dev = self.device
        self.units = torch.zeros(B, Fr, n_unit, device=dev)
        self.f0 = torch.full((B, Fr, 1), 220.0, device=dev)
        self.volume = torch.zeros(B, Fr, device=dev)
        self.spk_id = torch.ones(B, 1, dtype=torch.int64, device=dev)
        self.noise = torch.rand(B, Fr * hop, device=dev)
        self.ctx = hipddsp.Context(dev)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side), hipddsp.use_context(self.ctx), torch.no_grad():
            for _ in range(max(1, warmup)):          # first-use allocations (arena, tables) happen here, eagerly
                self._run()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with hipddsp.use_context(self.ctx), torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self._run()
        self.ctx.freeze()

    def _run(self):
        return self.model(self.units, self.f0, self.volume, self.spk_id, noise=self.noise)

    @torch.no_grad()
    def __call__(self, units, f0, volume, spk_id, noise=None):
        """Same positional inputs as `model.forward`; returns the model's result tuple (static tensors, overwritten by
        the next call).  `noise` (B, T) in [0, 1) replaces the fresh uniform draw (parity tests)."""
        if tuple(units.shape[:2]) != (self.B, self.Fr):
            raise ValueError(f"GraphedSynth captured for (B, Fr) = {(self.B, self.Fr)}, got {tuple(units.shape[:2])}")
        self.units.copy_(units)
        self.f0.copy_(f0.reshape(self.f0.shape))
        self.volume.copy_(volume.reshape(self.volume.shape))
        self.spk_id.copy_(spk_id.expand_as(self.spk_id) if spk_id.shape[0] == 1 else spk_id.reshape(self.spk_id.shape))
        if noise is None:
            self.noise.uniform_()
        else:
            self.noise.copy_(noise)
        self.graph.replay()
        return self.out
