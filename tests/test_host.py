"""CPU-side checks: the C ABI library builds, loads and exports every declared symbol; the drop-in classes keep
the reference's constructor / state-dict contract; host logic (sharding, slicing glue); no CPU fallback."""
import os
import re

import numpy as np
import pytest
import torch

import synthetic
from conftest import GOLDEN, PKG, ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ddsp_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ddsp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(lib_path):
    import ctypes
    import hipddsp
    syms = _declared_symbols()
    assert len(syms) >= 15
    lib = ctypes.CDLL(lib_path)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ddsp_amd.h but not exported"
        assert s in hipddsp.SIGNATURES, f"{s} has no ctypes signature in hipddsp"
    assert set(hipddsp.SIGNATURES) == set(syms)
    assert hipddsp.load_library().ddsp_abi_version() >= 1


def test_struct_layouts_match_header():
    """ctypes mirrors of the C structs: sizes that the natural-alignment rules of the header imply."""
    import ctypes
    import hipddsp
    n_ptr = 13 + 3 * 19 + 5
    assert ctypes.sizeof(hipddsp.U2CWeights) == 8 * n_ptr + 16 + 8   # 4 ints, the 64-bit change counter (ABI 5)
    assert hipddsp.U2CWeights.version.offset == 8 * n_ptr + 16 and hipddsp.load_library().ddsp_abi_version() == hipddsp.ABI_VERSION
    assert ctypes.sizeof(hipddsp.ProfEntry) == 4 + 36 + 8 + 3 * 8


def test_no_cpu_fallback(lib_path):
    import hipddsp
    with pytest.raises(RuntimeError):
        hipddsp.Context("cpu")
    model, _ = synthetic.build_model("CombSubFast", seed=1)
    inp = synthetic.make_inputs(2, 1, 4)
    with pytest.raises(RuntimeError):
        model(inp["units"], inp["f0"], inp["volume"], inp["spk_id"])
    from ddsp.loss import RSSLoss
    with pytest.raises(RuntimeError):
        RSSLoss(256, 2048, 4)(torch.zeros(1, 4096), torch.zeros(1, 4096))
    from ddsp.core import upsample
    with pytest.raises(RuntimeError):
        upsample(torch.zeros(1, 2, 1), 512)


@pytest.mark.parametrize("name,n_params", [("CombSub", 3500032), ("Sins", None), ("CombSubFast", None)])
def test_state_dict_contract(name, n_params):
    """Keys / shapes of SURVEY section 5 'State-dict layout' (checked against the reference by load_state_dict(strict)
    when the goldens were generated)."""
    m, cfg = synthetic.build_model(name, seed=3)
    sd = m.state_dict()
    for k in ["sampling_rate", "block_size", "unit2ctrl.unit_prenet.1.weight", "unit2ctrl.unit_prenet.2.bias",
              "unit2ctrl.unit_prenet.4.bias", "unit2ctrl.f0_embed.weight", "unit2ctrl.phase_embed.bias",
              "unit2ctrl.volume_embed.weight", "unit2ctrl.spk_embed.weight",
              "unit2ctrl.dec_post.0.net.0.norm.weight", "unit2ctrl.dec_post.0.net.2.attn.to_q.weight",
              "unit2ctrl.dec_post.0.net.1.attn.fast_attention.projection_matrix",
              "unit2ctrl.dec_post.0.net.0.local_mixer.net.0.weight", "unit2ctrl.dec_post.0.net.0.local_mixer.net.2.weight",
              "unit2ctrl.dec_post.0.net.0.local_mixer.net.4.weight", "unit2ctrl.dec_post.0.net.0.local_mixer.net.6.bias",
              "unit2ctrl.dec_post.1.weight", "unit2ctrl.dec_post.2.bias", "unit2ctrl.dec_post.2.weight_g",
              "unit2ctrl.dec_post.2.weight_v"]:
        assert k in sd, k
    assert sd["sampling_rate"].dtype == torch.int64 and sd["sampling_rate"].dim() == 0
    assert sd["unit2ctrl.dec_post.0.net.0.attn.fast_attention.projection_matrix"].shape == (266, 64)
    assert sd["unit2ctrl.dec_post.0.net.0.local_mixer.net.4.weight"].shape == (512, 1, 31)
    assert sd["unit2ctrl.dec_post.2.weight_g"].shape[1] == 1
    if name == "CombSubFast":
        assert sd["window"].shape == (1024,) and sd["unit2ctrl.dec_post.2.weight_v"].shape == (1539, 256)
    if n_params:
        assert sum(p.numel() for p in m.parameters()) == n_params
    m.train()
    m.eval()
    assert len(list(m.parameters())) > 60


def test_load_model_roundtrip(tmp_path):
    from ddsp.vocoder import load_model
    m, cfg = synthetic.build_model("CombSub", seed=4)
    conf = {"data": {"sampling_rate": 44100, "block_size": 512, "encoder_out_channels": 256},
            "model": {"type": "CombSub", "n_mag_allpass": 256, "n_mag_harmonic": 512, "n_mag_noise": 256, "n_spk": 100,
                      "c": False}}
    import yaml
    with open(tmp_path / "config.yaml", "w") as fh:
        yaml.safe_dump(conf, fh)
    torch.save({"global_step": 7, "model": m.state_dict()}, tmp_path / "model_7.pt")
    m2, args = load_model(str(tmp_path / "model_7.pt"), device="cpu")
    assert args.model.type == "CombSub" and args.data.block_size == 512 and not m2.training
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
    conf["model"]["type"] = "Nope"
    with open(tmp_path / "config.yaml", "w") as fh:
        yaml.safe_dump(conf, fh)
    with pytest.raises(ValueError):
        load_model(str(tmp_path / "model_7.pt"))
    from ddsp.vocoder import CombSub
    mc = CombSub(44100, 512, 256, 512, 256, c=True)          # causal mode: same parameters, inference only
    assert mc.unit2ctrl.causal and set(mc.state_dict()) == set(m.state_dict())


def test_sharding_rows():
    import sharding
    for n, w in [(64, 8), (10, 4), (3, 8), (0, 2)]:
        spans = [sharding.shard_rows(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    batch = {"units": torch.zeros(10, 5, 4), "spk_id": torch.ones(1, 1, dtype=torch.long)}
    sh = sharding.shard_batch(batch, 4, 3)
    assert sh["units"].shape[0] == 2 and sh["spk_id"].shape[0] == 1


def test_offline_cross_fade_matches_fixture():
    import infer_offline
    z = np.load(os.path.join(GOLDEN, "glue_offline.npz"))
    assert np.array_equal(infer_offline.cross_fade(z["a"], z["b"], int(z["idx"])), z["crossfaded"])


def test_enhancer_surface():
    from enhancer import Enhancer
    with pytest.raises(ValueError):
        Enhancer("other", "x")
    with pytest.raises(FileNotFoundError):          # like the reference: the config.json beside the checkpoint is read first
        Enhancer("nsf-hifigan", "/nonexistent/model")
