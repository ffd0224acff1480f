"""ORACLE (test infrastructure only - never imported by the product path): CPU restatement of the two front-end steps
of SURVEY 8(f) rank 2.  Pinned by tests/golden/glue_frontend.npz, which tests/golden/make_golden.py (tier d) produced by
running the reference's own `Volume_Extractor.extract` and `Units_Encoder.encode` (the latter on an instance whose
encoder model is a stand-in returning prepared units - the alignment code that runs is the reference's)."""
import numpy as np
import torch


def volume_extract(audio, hop):
    """ref: ddsp/vocoder.py:125-137.  audio (T,) float32 numpy -> (T//hop + 1,) RMS per block of the reflect-padded signal."""
    n_frames = len(audio) // hop + 1
    padded = np.pad(audio, (hop // 2, (hop + 1) // 2), mode="reflect")
    sq = padded ** 2
    return np.sqrt(np.array([np.mean(sq[n * hop:(n + 1) * hop]) for n in range(n_frames)]))


def align_units(units, n_samples, sample_rate, hop_size, encoder_sample_rate=16000, encoder_hop_size=320):
    """ref: ddsp/vocoder.py:201-211.  units (1, Lu, C) -> (1, n_samples // hop_size + 1, C), nearest frame, half-to-even."""
    n_frames = n_samples // hop_size + 1
    ratio = (hop_size / sample_rate) / (encoder_hop_size / encoder_sample_rate)
    idx = torch.clamp(torch.round(ratio * torch.arange(n_frames)).long(), max=units.size(1) - 1)
    return torch.gather(units, 1, idx.unsqueeze(0).unsqueeze(-1).repeat([1, 1, units.size(-1)]))
