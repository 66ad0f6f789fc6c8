"""CPU: the C-ABI library loads and exports every symbol include/cfgpp.h declares; UNet config
tables reproduce the published parameter totals; FLOP model matches SURVEY.md 8(d)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from cfgpp_amd import _lib
    from cfgpp_amd.build import build
    build(verbose=False)
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "cfgpp.h")).read()
    declared = set(re.findall(r"\b(cfgpp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cfgpp_unet_config"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/cfgpp.h but not exported"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
    assert set(_lib.PROTOTYPES) == declared
    assert _lib.last_error() == "" or isinstance(_lib.last_error(), str)


def test_param_totals_match_published_sizes():
    from cfgpp_amd.unet_config import SD15, SDXL, param_count
    assert param_count(SD15) == 859_520_964       # 859.5 M
    assert param_count(SDXL) == 2_567_463_684     # 2567.5 M


def test_flop_model():
    from cfgpp_amd.unet_config import SD15, SDXL, unet_flops_per_row
    assert abs(unet_flops_per_row(SD15, 64, 64) / 1e12 - 0.800) < 0.01
    assert abs(unet_flops_per_row(SDXL, 128, 128) / 1e12 - 6.71) < 0.05


def test_synthetic_weights_are_deterministic_and_fp16_exact():
    import torch
    from cfgpp_amd.unet_config import TINY_SD, param_shapes
    from cfgpp_amd.weights import synth_state_dict
    a, b = synth_state_dict(TINY_SD, 0), synth_state_dict(TINY_SD, 0)
    assert list(a) == list(param_shapes(TINY_SD))
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], a[k].half().float())
    assert not torch.equal(a["conv_in.weight"], synth_state_dict(TINY_SD, 1)["conv_in.weight"])


def test_oracle_unet_runs_and_is_finite():
    import torch
    from cfgpp_amd.unet_config import TINY_XL as cfg
    from cfgpp_amd.weights import synth_state_dict
    from oracle.unet_ref import UNetRef
    net = UNetRef(cfg, synth_state_dict(cfg))
    out = net(torch.randn(2, 4, 16, 16), 500.0, torch.randn(2, 77, cfg.cross_attention_dim) * 0.5,
              {"text_embeds": torch.randn(1, cfg.addition_pooled_dim), "time_ids": torch.tensor([[128.0, 128, 0, 0, 128, 128]])})["sample"]
    assert out.shape == (2, 4, 16, 16) and torch.isfinite(out).all()


def test_safetensors_loader_streams_diffusers_keys(tmp_path):
    """weights.load_safetensors_iter (the real-checkpoint path of HipEngine / vae_weights=) yields every key
    of a diffusers-layout file unchanged."""
    import torch
    from safetensors.torch import save_file
    from cfgpp_amd.unet_config import TINY_SD, param_shapes
    from cfgpp_amd.weights import load_safetensors_iter, synth_tensor
    shapes = dict(list(param_shapes(TINY_SD).items())[:12])
    sd = {k: synth_tensor(k, s, 0).half() for k, s in shapes.items()}
    path = str(tmp_path / "unet.safetensors")
    save_file(sd, path)
    got = dict(load_safetensors_iter(path))
    assert set(got) == set(sd)
    for k in sd:
        assert got[k].dtype == torch.float16 and torch.equal(got[k], sd[k])


def test_clip_text_towers_shapes_and_determinism():
    """opt-in CLIP text towers (transformers architecture, random init, hash tokenizer): SD1.5 / SDXL contracts"""
    import torch
    from cfgpp_amd.conditioning import ClipTextTower, HashTokenizer
    ids = HashTokenizer()(["a photo of a cat", ""])
    assert ids.shape == (2, 77) and ids[0, 0] == 49406 and ids[1, 1] == 49407 and ids[0, 6] == 49407
    assert (HashTokenizer(pad_id=0)(["x"])[0, 3:] == 0).all()
    enc = ClipTextTower.clip_l(layers=1)
    hs, pooled = enc(["a photo of a cat", "a dog"])
    assert hs.shape == (2, 77, 768) and hs.dtype == torch.float16 and pooled is None and torch.isfinite(hs.float()).all()
    assert torch.equal(hs, ClipTextTower.clip_l(layers=1)(["a photo of a cat", "a dog"])[0])       # seeded init
    l_pen = ClipTextTower.clip_l(layers=2, penultimate=True)
    g = ClipTextTower.open_clip_bigg(layers=2)
    h1, _ = l_pen(["a cat"]); h2, p2 = g(["a cat"])
    assert torch.cat([h1, h2], -1).shape == (1, 77, 2048) and p2.shape == (1, 1280)
