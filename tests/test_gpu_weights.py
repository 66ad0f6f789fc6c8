"""Real-checkpoint path on the GPU: diffusers-layout ``.safetensors`` files streamed into the HIP engines
must give exactly the results of the same tensors handed over in memory - UNet (``unet_weights=``), VAE (``vae_weights=``,
including checkpoints that still use the pre-0.15 attention key names) and the SDXL-Lightning single-file UNet
(``light_model_ckpt=``, reference latent_sdxl.py:378-390)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


def _save(sd, path, half=True):
    from safetensors.torch import save_file
    save_file({k: (v.half() if half and v.is_floating_point() else v).contiguous() for k, v in sd.items()}, str(path))
    return str(path)


def test_engine_from_safetensors_file_is_bit_identical_to_in_memory_weights(tmp_path):
    need_gpu()
    from cfgpp_amd.hip_engine import HipEngine
    from cfgpp_amd.unet_config import TINY_SD as cfg
    from cfgpp_amd.weights import synth_state_dict
    sd = {k: v.half() for k, v in synth_state_dict(cfg, 3).items()}
    path = _save(sd, tmp_path / "unet.safetensors")
    g = torch.Generator().manual_seed(0)
    uc, c = torch.randn(1, 77, cfg.cross_attention_dim, generator=g).half().cuda(), torch.randn(2, 77, cfg.cross_attention_dim, generator=g).half().cuda()
    z = torch.randn(2, 4, 16, 16, generator=g).cuda()
    outs = []
    for weights in (sd, path):
        eng = HipEngine(cfg, max_batch=2, latent_hw=(16, 16), weights=weights)
        eng.set_context(uc, c)
        e_uc, e_c = eng.predict(z, 481.0)
        outs.append(torch.cat([e_uc, e_c]).clone())
        del eng
    assert torch.isfinite(outs[0].float()).all() and float(outs[0].float().abs().max()) > 0
    assert torch.equal(outs[0], outs[1])


def test_solver_from_unet_and_vae_files_with_legacy_vae_keys(tmp_path):
    """get_solver(unet_weights=path, vae_weights=path): whole text-to-image job; the VAE file uses the deprecated
    query / key / value / proj_attn names with 1x1-conv-shaped weights (how the original SD VAE checkpoints ship)."""
    need_gpu()
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD as cfg
    from cfgpp_amd.vae import synth_vae_state_dict
    from cfgpp_amd.weights import synth_state_dict
    usd = {k: v.half() for k, v in synth_state_dict(cfg, 1).items()}
    vsd = {k: v.half() for k, v in synth_vae_state_dict(2).items()}
    legacy = {}
    renames = {".to_q.": ".query.", ".to_k.": ".key.", ".to_v.": ".value.", ".to_out.0.": ".proj_attn."}
    n_renamed = 0
    for k, v in vsd.items():
        k2 = k
        if ".attentions." in k:
            for new, old in renames.items():
                if new in k:
                    k2 = k.replace(new, old); n_renamed += 1
                    if v.dim() == 2:
                        v = v[:, :, None, None]          # conv-shaped, as in the original checkpoints
        legacy[k2] = v
    assert n_renamed >= 16                                # encoder + decoder mid-block attention: 4 layers x (w, b) x 2
    upath, vpath = _save(usd, tmp_path / "unet.safetensors"), _save(legacy, tmp_path / "vae.safetensors")
    sc = types.SimpleNamespace(num_sampling=3)
    common = dict(solver_config=sc, device="cuda", unet_config=cfg, max_batch=2, latent_hw=(16, 16))
    imgs = []
    for uw, vw in ((usd, vsd), (upath, vpath)):
        s = get_solver("ddim_cfg++", unet_weights=uw, vae_weights=vw, **common)
        imgs.append(s.sample(prompt=["", ["a cat", "a dog"]], cfg_guidance=0.6, seeds=[4, 5]).clone())
        del s
    assert imgs[0].shape == (2, 3, 128, 128) and torch.isfinite(imgs[0]).all() and float(imgs[0].std()) > 0
    assert torch.equal(imgs[0], imgs[1])


def test_lightning_single_file_checkpoint_becomes_the_unet(tmp_path):
    """SDXLLightning(light_model_ckpt=<file>): the file's tensors ARE the engine's UNet (the reference loads them into the
    base pipeline's UNet, latent_sdxl.py:378-390); a missing file keeps the synthetic weights (no checkpoint offline)."""
    need_gpu()
    from cfgpp_amd.latent_sdxl import get_solver
    from cfgpp_amd.unet_config import TINY_XL as cfg
    from cfgpp_amd.weights import synth_state_dict
    sd = {k: v.half() for k, v in synth_state_dict(cfg, 7).items()}
    ckpt = _save(sd, tmp_path / "sdxl_lightning_4step_unet.safetensors")
    sc = types.SimpleNamespace(num_sampling=4)
    common = dict(solver_config=sc, device="cuda", unet_config=cfg, max_batch=1, latent_hw=(16, 16))
    kw = dict(cfg_guidance=1.0, target_size=(128, 128), original_size=(128, 128), seeds=[9], return_latents=True)
    from_file = get_solver("ddim_cfg++_lightning", light_model_ckpt=ckpt, **common)
    pe = from_file.get_text_embed("", ["a cat"], "", ["a cat"])
    a = from_file.sample(prompt_embeds=pe, **kw).clone()
    in_memory = get_solver("ddim_cfg++_lightning", unet_weights=sd, text_encoder=from_file.text_encoder, **common)
    b = in_memory.sample(prompt_embeds=pe, **kw).clone()
    missing = get_solver("ddim_cfg++_lightning", light_model_ckpt=str(tmp_path / "absent.safetensors"),
                         text_encoder=from_file.text_encoder, **common)      # seed-0 synthetic weights
    c = missing.sample(prompt_embeds=pe, **kw).clone()
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    assert not torch.equal(a, c)
