"""The example CLIs (twins of the reference's examples/text_to_img.py / inversion.py) end to end on the CPU with
an injected mock engine and stub VAE: flag parsing, solver plumbing, PNG output."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))

from mock_engine import MockEngine, StubVAE          # noqa: E402


def _unet(z, t, ehs, te, ti):
    return (0.1 * z + 0.01 * ehs.float().mean(dim=(1, 2)).view(-1, 1, 1, 1)).half()


def test_text_to_img_cli_writes_pngs(tmp_path):
    from PIL import Image
    import text_to_img
    kw = dict(engine=MockEngine(_unet, (8, 8)), vae=StubVAE(0.18215), latent_hw=(8, 8))
    text_to_img.main(["--method", "ddim_cfg++", "--cfg_guidance", "0.6", "--NFE", "3", "--prompt", "a cat", "--device", "cpu",
                      "--batch", "2", "--draw", "--workdir", str(tmp_path)], solver_kwargs=kw)
    for i in range(2):
        im = Image.open(tmp_path / "result" / f"generated_{i}.png")
        assert im.size == (64, 64) and im.mode == "RGB"
    assert len(list((tmp_path / "record").rglob("*.png"))) > 0          # draw_* callbacks fired


def test_mscoco_cli_one_image_per_caption(tmp_path, monkeypatch):
    """twin of the reference's examples/text_to_mscoco.py: blank lines skipped, files named by caption index, batches of
    --batch captions (ragged last batch), --limit; under a 2-rank environment each rank writes its contiguous shard"""
    from PIL import Image
    import text_to_mscoco
    caps = tmp_path / "caps.txt"
    caps.write_text("a cat\n\n  a dog on a sofa  \na red bus\n\na plate of food\ntwo birds\n")
    assert text_to_mscoco.read_captions(caps) == ["a cat", "a dog on a sofa", "a red bus", "a plate of food", "two birds"]
    out = tmp_path / "coco"
    kw = dict(engine=MockEngine(_unet, (8, 8)), vae=StubVAE(0.18215), latent_hw=(8, 8))
    base = ["--prompt_dir", str(caps), "--method", "ddim_cfg++", "--cfg_guidance", "0.6", "--NFE", "2", "--device", "cpu", "--no_draw"]
    n = text_to_mscoco.main(base + ["--batch", "2", "--workdir", str(out)], solver_kwargs=kw)
    assert n == 5 and sorted(p.name for p in out.glob("*.png")) == [f"{i:05d}.png" for i in range(5)]
    assert Image.open(out / "00004.png").size == (64, 64)
    # batch 1 == the reference's loop; --limit
    out1 = tmp_path / "coco1"
    kw = dict(engine=MockEngine(_unet, (8, 8)), vae=StubVAE(0.18215), latent_hw=(8, 8))
    assert text_to_mscoco.main(base + ["--limit", "3", "--workdir", str(out1)], solver_kwargs=kw) == 3
    assert sorted(p.name for p in out1.glob("*.png")) == ["00000.png", "00001.png", "00002.png"]
    # two ranks: contiguous shards 0..2 and 3..4, same file names as the single-process run
    monkeypatch.setenv("WORLD_SIZE", "2")
    for rank, expect in ((0, [0, 1, 2]), (1, [3, 4])):
        monkeypatch.setenv("RANK", str(rank))
        outr = tmp_path / f"coco_r{rank}"
        kw = dict(engine=MockEngine(_unet, (8, 8)), vae=StubVAE(0.18215), latent_hw=(8, 8))
        assert text_to_mscoco.main(base + ["--batch", "2", "--workdir", str(outr)], solver_kwargs=kw) == len(expect)
        assert sorted(p.name for p in outr.glob("*.png")) == [f"{i:05d}.png" for i in expect]
        for i in expect:      # explicit per-caption seeds: a caption's image does not depend on how the captions were sharded
            assert np.array_equal(np.asarray(Image.open(outr / f"{i:05d}.png")), np.asarray(Image.open(out / f"{i:05d}.png")))


def test_save_image_normalize_matches_torchvision_rule(tmp_path):
    from PIL import Image
    from cfgpp_amd.callback_util import save_image
    x = torch.rand(1, 3, 4, 5) * 0.5 + 0.2
    save_image(x, tmp_path / "n.png", normalize=True)
    lo, hi = float(x.min()), float(x.max())
    want = ((x - lo) / (hi - lo)).mul(255).add(0.5).clamp(0, 255).to(torch.uint8)[0].permute(1, 2, 0).numpy()
    assert np.array_equal(np.asarray(Image.open(tmp_path / "n.png")), want)
    save_image(x, tmp_path / "p.png")
    assert np.array_equal(np.asarray(Image.open(tmp_path / "p.png")), x.mul(255).add(0.5).to(torch.uint8)[0].permute(1, 2, 0).numpy())


def test_inversion_cli_reconstructs(tmp_path):
    from PIL import Image
    import inversion
    src = tmp_path / "src.png"
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 255, (64, 64, 3), dtype=np.uint8)).save(src)
    kw = dict(engine=MockEngine(_unet, (8, 8)), vae=StubVAE(0.18215))
    inversion.main(["--img_path", str(src), "--img_size", "64", "--method", "ddim_inversion_cfg++", "--cfg_guidance", "0.6",
                    "--NFE", "4", "--prompt", "noise", "--device", "cpu", "--workdir", str(tmp_path)], solver_kwargs=kw)
    im = Image.open(tmp_path / "result" / "reconstruct.png")
    assert im.size == (64, 64)
    assert torch.isfinite(torch.from_numpy(np.asarray(im, dtype=np.float32))).all()


def test_solver_with_clip_text_tower():
    """the opt-in CLIP tower plugs into the SD1.5 solver through the same ``text_encoder=`` seam as the synthetic one"""
    import types
    from cfgpp_amd.conditioning import ClipTextTower
    from cfgpp_amd.latent_diffusion import get_solver
    eng = MockEngine(_unet, (8, 8))
    s = get_solver("ddim_cfg++", solver_config=types.SimpleNamespace(num_sampling=2), device="cpu", engine=eng,
                   text_encoder=ClipTextTower.clip_l(layers=1), latent_hw=(8, 8), vae=StubVAE(0.18215))
    img = s.sample(prompt=["", "a cat"], cfg_guidance=0.6)
    assert img.shape == (1, 3, 64, 64) and torch.isfinite(img).all()
    assert eng.ehs.shape == (2, 77, 768) and eng.ehs.dtype == torch.float16


def _tiny_checkpoint_dir(root, sdxl=False):
    """diffusers-layout directory with tiny CLIP text encoder(s) + toy BPE vocabulary (no unet / vae)"""
    import json
    from safetensors.torch import save_file
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from test_capi_and_config import _toy_clip_vocab
    vocab, merges = _toy_clip_vocab()
    towers = [("text_encoder", "tokenizer", False, 48)] + ([("text_encoder_2", "tokenizer_2", True, 80)] if sdxl else [])
    for enc, tok, proj, hidden in towers:
        (root / enc).mkdir(parents=True); (root / tok).mkdir(parents=True)
        (root / tok / "vocab.json").write_text(json.dumps(vocab, ensure_ascii=False), encoding="utf-8")
        (root / tok / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
        cfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=2,
                             num_attention_heads=4, max_position_embeddings=77, projection_dim=32,
                             bos_token_id=vocab["<|startoftext|>"], eos_token_id=vocab["<|endoftext|>"], pad_token_id=1)
        torch.manual_seed(11)
        model = (CLIPTextModelWithProjection(cfg) if proj else CLIPTextModel(cfg)).eval()
        cfg.to_json_file(str(root / enc / "config.json"))
        save_file({k: v.contiguous() for k, v in model.state_dict().items()}, str(root / enc / "model.safetensors"))
    return root


def test_model_dir_flag_plugs_real_text_path(tmp_path, capsys):
    """--model_dir: CLIP tower + BPE tokenizer from a diffusers-layout directory replace the synthetic text encoder;
    absent unet / vae folders are reported and the synthetic stand-ins stay"""
    import text_to_img
    from cfgpp_amd.checkpoint import solver_kwargs_from_dir
    from cfgpp_amd.conditioning import ClipTextTower
    ckpt = _tiny_checkpoint_dir(tmp_path / "sd15_ckpt")
    found, missing = solver_kwargs_from_dir(ckpt, sdxl=False, device="cpu")
    assert missing == ["unet", "vae"] and isinstance(found["text_encoder"], ClipTextTower)
    eng = MockEngine(_unet, (8, 8))
    text_to_img.main(["--method", "ddim_cfg++", "--cfg_guidance", "0.6", "--NFE", "2", "--prompt", "a photo of a cat", "--device", "cpu",
                      "--model_dir", str(ckpt), "--workdir", str(tmp_path / "out")], solver_kwargs=dict(engine=eng, vae=StubVAE(0.18215), latent_hw=(8, 8)))
    assert "no unet, vae there" in capsys.readouterr().out
    assert eng.ehs.shape == (2, 77, 48) and eng.ehs.dtype == torch.float16        # the tiny tower's width, not the synthetic 768
    want = found["text_encoder"](["a photo of a cat"])[0]
    assert torch.equal(eng.ehs[1:2].cpu(), want)
    # SDXL: two towers, concatenated hidden states, pooled from the projected second tower
    xl = _tiny_checkpoint_dir(tmp_path / "xl_ckpt", sdxl=True)
    found, missing = solver_kwargs_from_dir(xl, sdxl=True, device="cpu")
    t1, t2 = found["text_encoder"]
    assert t1.penultimate and t2.penultimate and t2.proj and not t1.proj and t2.tok.pad_id == t2.tok.vocab["!"]
    h1, _ = t1(["a cat"]); h2, p2 = t2(["a cat"])
    assert torch.cat([h1, h2], -1).shape == (1, 77, 128) and p2.shape == (1, 32)
    # SDXL never takes the pipeline's own <dir>/vae (latent_sdxl.py:44 swaps in sdxl-vae-fp16-fix: the stock VAE overflows in fp16)
    from safetensors.torch import save_file
    (xl / "vae").mkdir(); (xl / "unet").mkdir()
    save_file({"w": torch.zeros(1)}, str(xl / "vae" / "diffusion_pytorch_model.safetensors"))
    save_file({"w": torch.zeros(1)}, str(xl / "unet" / "diffusion_pytorch_model.fp16.safetensors"))
    found, missing = solver_kwargs_from_dir(xl, sdxl=True, device="cpu")
    assert "vae_weights" not in found and any(m.startswith("vae") for m in missing) and found["unet_weights"].endswith(".fp16.safetensors")
    (xl / "vae_fp16_fix").mkdir()
    save_file({"w": torch.zeros(1)}, str(xl / "vae_fp16_fix" / "diffusion_pytorch_model.safetensors"))
    found, missing = solver_kwargs_from_dir(xl, sdxl=True, device="cpu")
    assert found["vae_weights"].endswith(os.path.join("vae_fp16_fix", "diffusion_pytorch_model.safetensors")) and not any(m.startswith("vae") for m in missing)
    assert solver_kwargs_from_dir(xl, sdxl=True, device="cpu", vae_dir=xl / "vae")[0]["vae_weights"].endswith(os.path.join("vae", "diffusion_pytorch_model.safetensors"))
    # fp16-variant text-encoder weights (model.fp16.safetensors, the common SDXL download) are found too
    os.rename(xl / "text_encoder" / "model.safetensors", xl / "text_encoder" / "model.fp16.safetensors")
    found, missing = solver_kwargs_from_dir(xl, sdxl=True, device="cpu")
    assert "text_encoder" in found and "text_encoder" not in missing
