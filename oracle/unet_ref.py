"""CPU oracle: plain-PyTorch fp32 restatement of the SD1.5 / SDXL UNet forward.
TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

PARITY UNPINNED at this boundary: the UNet arithmetic of the reference lives in
the third-party ``diffusers==0.27.1`` (environment.yaml:87), which is neither
vendored in /root/reference nor installable here, and the reference has no
tests.  This file restates the published diffusers-0.27.1
``UNet2DConditionModel`` semantics for the two configs the reference loads
(call sites latent_diffusion.py:63,146-156; latent_sdxl.py:40,170-183), is
structurally checked by the exact parameter totals (859.5 M / 2567.5 M) and by
consuming diffusers state-dict keys unchanged, and is what every HIP kernel is
compared against.

Same ATen ops diffusers issues: conv2d, group_norm, layer_norm, linear,
scaled_dot_product_attention, exact (erf) GELU, nearest 2x interpolate.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float().reshape(-1, 1) * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class UNetRef:
    def __init__(self, cfg, sd: Dict[str, torch.Tensor], round_io: bool = True):
        """``round_io``: round the latent input and the sinusoid through fp16 as the
        fp16 engine / the autocast reference do (everything else stays fp32)."""
        self.cfg = cfg
        self.sd = {k: v.float() for k, v in sd.items()}
        self.round_io = round_io

    # -- primitives ------------------------------------------------------------
    def _lin(self, p, x):
        return F.linear(x, self.sd[p + ".weight"], self.sd.get(p + ".bias"))

    def _conv(self, p, x, stride=1, pad=1):
        w = self.sd[p + ".weight"]
        if w.dim() == 2:
            w = w[:, :, None, None]
        return F.conv2d(x, w, self.sd[p + ".bias"], stride=stride, padding=pad)

    def _gn(self, p, x, eps=1e-5):
        return F.group_norm(x, self.cfg.norm_groups, self.sd[p + ".weight"], self.sd[p + ".bias"], eps)

    def _ln(self, p, x):
        return F.layer_norm(x, (x.shape[-1],), self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-5)

    def _resnet(self, p, x, emb):
        h = self._conv(p + ".conv1", F.silu(self._gn(p + ".norm1", x)))
        h = h + self._lin(p + ".time_emb_proj", F.silu(emb))[:, :, None, None]
        h = self._conv(p + ".conv2", F.silu(self._gn(p + ".norm2", h)))
        if (p + ".conv_shortcut.weight") in self.sd:
            x = self._conv(p + ".conv_shortcut", x, pad=0)
        return x + h

    def _attn(self, p, x, ctx, heads):
        q = self._lin(p + ".to_q", x)
        k = self._lin(p + ".to_k", ctx)
        v = self._lin(p + ".to_v", ctx)
        B, N, Cc = q.shape
        d = Cc // heads
        q = q.view(B, N, heads, d).transpose(1, 2)
        k = k.view(B, -1, heads, d).transpose(1, 2)
        v = v.view(B, -1, heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).reshape(B, N, Cc)
        return self._lin(p + ".to_out.0", o)

    def _tblock(self, p, x, ctx, heads):
        h = self._ln(p + ".norm1", x)
        x = x + self._attn(p + ".attn1", h, h, heads)
        x = x + self._attn(p + ".attn2", self._ln(p + ".norm2", x), ctx, heads)
        h = self._lin(p + ".ff.net.0.proj", self._ln(p + ".norm3", x))
        a, gate = h.chunk(2, dim=-1)
        x = x + self._lin(p + ".ff.net.2", a * F.gelu(gate))
        return x

    def _transformer(self, p, x, ctx, depth, heads):
        B, Cc, H, W = x.shape
        res = x
        h = self._gn(p + ".norm", x, eps=1e-6)
        wpi = self.sd[p + ".proj_in.weight"]
        if wpi.dim() == 4:       # conv1x1 then flatten (SD1.5)
            h = self._conv(p + ".proj_in", h, pad=0)
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, Cc)
        else:                    # flatten then linear (SDXL)
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, Cc)
            h = self._lin(p + ".proj_in", h)
        for k in range(depth):
            h = self._tblock(f"{p}.transformer_blocks.{k}", h, ctx, heads)
        if wpi.dim() == 4:
            h = h.reshape(B, H, W, Cc).permute(0, 3, 1, 2)
            h = self._conv(p + ".proj_out", h, pad=0)
        else:
            h = self._lin(p + ".proj_out", h)
            h = h.reshape(B, H, W, Cc).permute(0, 3, 1, 2)
        return h + res

    # -- forward ---------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, sample, timestep, encoder_hidden_states, added_cond_kwargs: Optional[dict] = None):
        cfg = self.cfg
        L = cfg.num_levels
        x = sample.float()
        if self.round_io:
            x = x.half().float()
        R = x.shape[0]
        ctx = encoder_hidden_states.float()
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(R)
        temb = timestep_embedding(t, cfg.block_out_channels[0])
        if self.round_io:
            temb = temb.half().float()
        emb = self._lin("time_embedding.linear_2", F.silu(self._lin("time_embedding.linear_1", temb)))
        if cfg.addition_embed:
            te = added_cond_kwargs["text_embeds"].float()
            ti = added_cond_kwargs["time_ids"].float()
            tproj = timestep_embedding(ti.flatten(), cfg.addition_time_embed_dim)
            if self.round_io:
                tproj = tproj.half().float()
            tproj = tproj.reshape(te.shape[0], -1)
            add = torch.cat([te, tproj], dim=-1)
            aug = self._lin("add_embedding.linear_2", F.silu(self._lin("add_embedding.linear_1", add)))
            emb = emb + aug            # broadcasts when the cond batch is 1 (quirk Q7)
        x = self._conv("conv_in", x)
        skips = [x]
        for i in range(L):
            for j in range(cfg.layers_per_block):
                x = self._resnet(f"down_blocks.{i}.resnets.{j}", x, emb)
                if cfg.level_has_attn[i]:
                    x = self._transformer(f"down_blocks.{i}.attentions.{j}", x, ctx, cfg.transformer_depth[i],
                                          cfg.num_heads[i])
                skips.append(x)
            if i != L - 1:
                x = self._conv(f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
                skips.append(x)
        x = self._resnet("mid_block.resnets.0", x, emb)
        x = self._transformer("mid_block.attentions.0", x, ctx, cfg.transformer_depth[-1], cfg.num_heads[-1])
        x = self._resnet("mid_block.resnets.1", x, emb)
        for i in range(L):
            lvl = L - 1 - i
            for j in range(cfg.layers_per_block + 1):
                x = torch.cat([x, skips.pop()], dim=1)
                x = self._resnet(f"up_blocks.{i}.resnets.{j}", x, emb)
                if cfg.level_has_attn[lvl]:
                    x = self._transformer(f"up_blocks.{i}.attentions.{j}", x, ctx, cfg.transformer_depth[lvl],
                                          cfg.num_heads[lvl])
            if i != L - 1:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = self._conv(f"up_blocks.{i}.upsamplers.0.conv", x)
        x = F.silu(self._gn("conv_norm_out", x))
        x = self._conv("conv_out", x)
        return {"sample": x}
