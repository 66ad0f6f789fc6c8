"""UNet architecture configs (SD1.5, SDXL and small test nets) and the diffusers
state-dict key/shape table they imply.

The key names are diffusers-0.27.1 ``UNet2DConditionModel`` names (the model the
reference loads at latent_diffusion.py:63 / latent_sdxl.py:40); a real
``unet/diffusion_pytorch_model.safetensors`` can be fed to
``HipUNet.load_state_dict`` unchanged.  Structural self-check: the SD1.5 and
SDXL tables sum to the published 859.5 M / 2567.5 M parameters
(tests/test_unet_config.py).
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class UNetConfig:
    name: str
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    level_has_attn: Tuple[int, ...] = (1, 1, 1, 0)
    transformer_depth: Tuple[int, ...] = (1, 1, 1, 1)      # per level; the last one is also the mid block's
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    addition_embed: int = 0
    addition_time_embed_dim: int = 256
    addition_pooled_dim: int = 1280
    norm_groups: int = 32
    sample_size: int = 64          # default latent H = W
    vae_scale: float = 0.18215

    @property
    def num_levels(self) -> int:
        return len(self.block_out_channels)

    @property
    def temb_dim(self) -> int:
        return 4 * self.block_out_channels[0]


SD15 = UNetConfig(name="sd15")
SDXL = UNetConfig(name="sdxl", block_out_channels=(320, 640, 1280), level_has_attn=(0, 1, 1),
                  transformer_depth=(1, 2, 10), num_heads=(5, 10, 20), cross_attention_dim=2048,
                  addition_embed=1, sample_size=128, vae_scale=0.13025)
# small nets with the same topology for fast parity tests (channels multiple of 64, 32 groups)
TINY_SD = UNetConfig(name="tiny_sd", block_out_channels=(64, 128, 128), level_has_attn=(1, 1, 0),
                     transformer_depth=(1, 1, 1), num_heads=(2, 4, 4), cross_attention_dim=64, sample_size=16)
TINY_XL = UNetConfig(name="tiny_xl", block_out_channels=(64, 128), level_has_attn=(0, 1),
                     transformer_depth=(1, 2), num_heads=(1, 2), cross_attention_dim=128,
                     addition_embed=1, addition_time_embed_dim=32, addition_pooled_dim=64, sample_size=16,
                     vae_scale=0.13025)

CONFIGS = {c.name: c for c in (SD15, SDXL, TINY_SD, TINY_XL)}


def param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    """diffusers state-dict key -> shape, in module order."""
    P: "OrderedDict[str, tuple]" = OrderedDict()
    L = cfg.num_levels
    c0 = cfg.block_out_channels[0]
    temb = cfg.temb_dim

    def lin(p, o, i, bias=True):
        P[p + ".weight"] = (o, i)
        if bias:
            P[p + ".bias"] = (o,)

    def conv(p, o, i, k):
        P[p + ".weight"] = (o, i, k, k)
        P[p + ".bias"] = (o,)

    def norm(p, c):
        P[p + ".weight"] = (c,)
        P[p + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cout, cin, 3)
        lin(p + ".time_emb_proj", cout, temb)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cout, cin, 1)

    def transformer(p, c, depth):
        norm(p + ".norm", c)
        if cfg.addition_embed:           # SDXL: use_linear_projection=True
            lin(p + ".proj_in", c, c)
        else:                            # SD1.5: 1x1 convs
            conv(p + ".proj_in", c, c, 1)
        for k in range(depth):
            b = f"{p}.transformer_blocks.{k}"
            norm(b + ".norm1", c)
            lin(b + ".attn1.to_q", c, c, False)
            lin(b + ".attn1.to_k", c, c, False)
            lin(b + ".attn1.to_v", c, c, False)
            lin(b + ".attn1.to_out.0", c, c)
            norm(b + ".norm2", c)
            lin(b + ".attn2.to_q", c, c, False)
            lin(b + ".attn2.to_k", c, cfg.cross_attention_dim, False)
            lin(b + ".attn2.to_v", c, cfg.cross_attention_dim, False)
            lin(b + ".attn2.to_out.0", c, c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", 8 * c, c)
            lin(b + ".ff.net.2", c, 4 * c)
        if cfg.addition_embed:
            lin(p + ".proj_out", c, c)
        else:
            conv(p + ".proj_out", c, c, 1)

    conv("conv_in", c0, cfg.in_channels, 3)
    lin("time_embedding.linear_1", temb, c0)
    lin("time_embedding.linear_2", temb, temb)
    if cfg.addition_embed:
        lin("add_embedding.linear_1", temb, 6 * cfg.addition_time_embed_dim + cfg.addition_pooled_dim)
        lin("add_embedding.linear_2", temb, temb)
    ch = c0
    for i in range(L):
        co = cfg.block_out_channels[i]
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", ch, co)
            if cfg.level_has_attn[i]:
                transformer(f"down_blocks.{i}.attentions.{j}", co, cfg.transformer_depth[i])
            ch = co
        if i != L - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    cm = cfg.block_out_channels[-1]
    resnet("mid_block.resnets.0", cm, cm)
    transformer("mid_block.attentions.0", cm, cfg.transformer_depth[-1])
    resnet("mid_block.resnets.1", cm, cm)
    prev = cm
    for i in range(L):
        lvl = L - 1 - i
        co = cfg.block_out_channels[lvl]
        cin_lvl = cfg.block_out_channels[max(lvl - 1, 0)]
        for j in range(cfg.layers_per_block + 1):
            skip = cin_lvl if j == cfg.layers_per_block else co
            rin = (prev if j == 0 else co) + skip
            resnet(f"up_blocks.{i}.resnets.{j}", rin, co)
            if cfg.level_has_attn[lvl]:
                transformer(f"up_blocks.{i}.attentions.{j}", co, cfg.transformer_depth[lvl])
        if i != L - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        prev = co
    norm("conv_norm_out", c0)
    conv("conv_out", cfg.out_channels, c0, 3)
    return P


def param_count(cfg: UNetConfig) -> int:
    n = 0
    for s in param_shapes(cfg).values():
        k = 1
        for d in s:
            k *= d
        n += k
    return n


def unet_flops_per_row(cfg: UNetConfig, H: int, W: int, tokens: int = 77) -> float:
    """Algorithmic FLOPs (2*MAC) of one UNet forward for ONE batch row: every conv /
    linear / attention matmul; norms and elementwise excluded.
    Cross-attention K/V projections of the text are excluded (step-invariant)."""
    L = cfg.num_levels
    temb = cfg.temb_dim
    mac = 0.0

    def res(cin, cout, hw):
        m = hw * cout * 9 * cin + hw * cout * 9 * cout + cout * temb
        if cin != cout:
            m += hw * cout * cin
        return m

    def tf(c, depth, hw, heads):
        d = c // heads
        m = 2 * hw * c * c                                   # proj_in, proj_out
        per = (3 * hw * c * c + hw * c * c                   # qkv, out
               + hw * c * c + hw * c * c                     # cross q, out
               + hw * c * 8 * c + hw * 4 * c * c             # ff
               + 2 * heads * hw * hw * d + 2 * heads * hw * tokens * d)
        return m + depth * per

    h, w = H, W
    c0 = cfg.block_out_channels[0]
    mac += h * w * c0 * 9 * cfg.in_channels
    mac += c0 * temb + temb * temb
    ch = c0
    for i in range(L):
        co = cfg.block_out_channels[i]
        for j in range(cfg.layers_per_block):
            mac += res(ch, co, h * w)
            if cfg.level_has_attn[i]:
                mac += tf(co, cfg.transformer_depth[i], h * w, cfg.num_heads[i])
            ch = co
        if i != L - 1:
            h //= 2
            w //= 2
            mac += h * w * co * 9 * co
    cm = cfg.block_out_channels[-1]
    mac += 2 * res(cm, cm, h * w) + tf(cm, cfg.transformer_depth[-1], h * w, cfg.num_heads[-1])
    prev = cm
    for i in range(L):
        lvl = L - 1 - i
        co = cfg.block_out_channels[lvl]
        cin_lvl = cfg.block_out_channels[max(lvl - 1, 0)]
        for j in range(cfg.layers_per_block + 1):
            skip = cin_lvl if j == cfg.layers_per_block else co
            rin = (prev if j == 0 else co) + skip
            mac += res(rin, co, h * w)
            if cfg.level_has_attn[lvl]:
                mac += tf(co, cfg.transformer_depth[lvl], h * w, cfg.num_heads[lvl])
        if i != L - 1:
            h *= 2
            w *= 2
            mac += h * w * co * 9 * co
        prev = co
    mac += h * w * cfg.out_channels * 9 * c0
    return 2.0 * mac
