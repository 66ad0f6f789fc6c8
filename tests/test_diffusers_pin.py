"""Opportunistic pin of the UNet / VAE oracles: the reference's UNet and VAE arithmetic lives in
``diffusers==0.27.1`` (environment.yaml:87), which this image does not carry, so ``oracle/unet_ref.py`` and
``oracle/vae_ref.py`` are marked "parity unpinned".  WHEREVER diffusers is importable these tests build the real
``UNet2DConditionModel`` / ``AutoencoderKL`` from the config below, load the SAME synthetic state dict and compare: if they
ever run, row (c) is pinned.  Present twice - once in the CPU suite, once under ``-m gpu`` - so that either box can do it."""
import pytest
import torch


def _unet_kwargs(cfg, hw):
    L = cfg.num_levels
    down = tuple("CrossAttnDownBlock2D" if cfg.level_has_attn[i] else "DownBlock2D" for i in range(L))
    up = tuple("CrossAttnUpBlock2D" if cfg.level_has_attn[L - 1 - i] else "UpBlock2D" for i in range(L))
    kw = dict(sample_size=hw, in_channels=cfg.in_channels, out_channels=cfg.out_channels, down_block_types=down, up_block_types=up,
              mid_block_type="UNetMidBlock2DCrossAttn", block_out_channels=tuple(cfg.block_out_channels),
              layers_per_block=cfg.layers_per_block, cross_attention_dim=cfg.cross_attention_dim,
              attention_head_dim=tuple(cfg.num_heads),          # (diffusers' historical misnomer: this IS the head count)
              norm_num_groups=cfg.norm_groups, transformer_layers_per_block=tuple(cfg.transformer_depth))
    if cfg.addition_embed:
        kw.update(use_linear_projection=True, addition_embed_type="text_time", addition_time_embed_dim=cfg.addition_time_embed_dim,
                  projection_class_embeddings_input_dim=6 * cfg.addition_time_embed_dim + cfg.addition_pooled_dim)
    return kw


def _pin_unet(cfg_name):
    diffusers = pytest.importorskip("diffusers")
    from cfgpp_amd.unet_config import CONFIGS
    from cfgpp_amd.weights import synth_state_dict
    from oracle.unet_ref import UNetRef
    cfg = CONFIGS[cfg_name]
    hw = 16
    sd = synth_state_dict(cfg, 0)
    model = diffusers.UNet2DConditionModel(**_unet_kwargs(cfg, hw)).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    g = torch.Generator().manual_seed(1)
    z = torch.randn(2, 4, hw, hw, generator=g)
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=g) * 0.5
    ack = None
    if cfg.addition_embed:
        ack = {"text_embeds": torch.randn(2, cfg.addition_pooled_dim, generator=g) * 0.5,
               "time_ids": torch.tensor([[128.0, 128, 0, 0, 128, 128]] * 2)}
    for t in (981.0, 1.0):
        with torch.no_grad():
            want = model(z, torch.tensor([t, t]), encoder_hidden_states=ehs, added_cond_kwargs=ack).sample
        got = UNetRef(cfg, sd, round_io=False)(z, t, ehs, ack)["sample"]
        rel = float((got - want).norm() / want.norm())
        assert rel < 1e-5, (cfg_name, t, rel)          # same ATen ops in fp32: rounding-order noise only


def _pin_vae():
    diffusers = pytest.importorskip("diffusers")
    from cfgpp_amd.vae import synth_vae_state_dict
    from oracle.vae_ref import VAERef
    sd = synth_vae_state_dict(0)
    model = diffusers.AutoencoderKL(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                                    up_block_types=("UpDecoderBlock2D",) * 4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                                    latent_channels=4, norm_num_groups=32, sample_size=128, scaling_factor=0.18215).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    ref = VAERef(0.18215, state_dict=sd)
    g = torch.Generator().manual_seed(2)
    z = torch.randn(1, 4, 16, 16, generator=g)
    img = torch.rand(1, 3, 128, 128, generator=g) * 2 - 1
    with torch.no_grad():
        want_dec = model.decode(z / 0.18215).sample
        want_mom = model.encode(img).latent_dist
    rel = float((ref.decode(z) - want_dec).norm() / want_dec.norm())
    assert rel < 1e-5, rel
    mean, logvar = ref.encode_moments(img)
    assert float((mean - want_mom.mean).norm() / want_mom.mean.norm()) < 1e-5
    assert float((logvar - want_mom.logvar).norm() / want_mom.logvar.norm()) < 1e-5


@pytest.mark.parametrize("cfg_name", ["tiny_sd", "tiny_xl"])
def test_unet_oracle_equals_diffusers(cfg_name):
    _pin_unet(cfg_name)


def test_vae_oracle_equals_diffusers():
    _pin_vae()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name", ["tiny_sd", "tiny_xl"])
def test_unet_oracle_equals_diffusers_on_the_gpu_box(cfg_name):
    _pin_unet(cfg_name)


@pytest.mark.gpu
def test_vae_oracle_equals_diffusers_on_the_gpu_box():
    _pin_vae()
