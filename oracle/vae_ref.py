"""fp32 torch restatement of diffusers-0.27.1 AutoencoderKL (SD1.5 ``vae`` / ``madebyollin/sdxl-vae-fp16-fix``:
block_out_channels (128,256,512,512), 2 layers per block, 4 latent channels).

TEST INFRASTRUCTURE ONLY: parity reference of the HIP VAE engine (``tests/test_gpu_vae.py``), the VAE leg of
``bench.py``'s ``cpu_baseline`` and the decode used by CPU mock-engine tests.  Nothing under ``cfgpp_amd/``
imports it.  PARITY UNPINNED at the third-party boundary: the arithmetic lives in diffusers (not in
/root/reference, not installable here); this file restates its published semantics and consumes diffusers
state-dict keys unchanged.

Reference call sites: latent_diffusion.py:117-129 (scale 0.18215), latent_sdxl.py:44,150-164
(``vae.config.scaling_factor`` = 0.13025).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from cfgpp_amd.vae import synth_vae_state_dict


class VAERef:
    """Functional AutoencoderKL on torch ops (CPU, fp32)."""

    def __init__(self, scaling_factor: float, device="cpu", dtype=torch.float32, state_dict=None, seed: int = 0):
        sd = state_dict if state_dict is not None else synth_vae_state_dict(seed)
        self.sd = {k: v.to(device=device, dtype=dtype) for k, v in sd.items()}
        self.scaling_factor = scaling_factor
        self.device, self.dtype = device, dtype

    def _conv(self, p, x, stride=1, pad=1):
        return F.conv2d(x, self.sd[p + ".weight"], self.sd[p + ".bias"], stride=stride, padding=pad)

    def _gn(self, p, x):
        return F.group_norm(x, 32, self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-6)

    def _res(self, p, x):
        h = self._conv(p + ".conv1", F.silu(self._gn(p + ".norm1", x)))
        h = self._conv(p + ".conv2", F.silu(self._gn(p + ".norm2", h)))
        if (p + ".conv_shortcut.weight") in self.sd:
            x = self._conv(p + ".conv_shortcut", x, pad=0)
        return x + h

    def _attn(self, p, x):
        B, C, H, W = x.shape
        h = self._gn(p + ".group_norm", x).view(B, C, H * W).transpose(1, 2)
        q = F.linear(h, self.sd[p + ".to_q.weight"], self.sd[p + ".to_q.bias"])
        k = F.linear(h, self.sd[p + ".to_k.weight"], self.sd[p + ".to_k.bias"])
        v = F.linear(h, self.sd[p + ".to_v.weight"], self.sd[p + ".to_v.bias"])
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        o = F.linear(o, self.sd[p + ".to_out.0.weight"], self.sd[p + ".to_out.0.bias"])
        return x + o.transpose(1, 2).reshape(B, C, H, W)

    @torch.no_grad()
    def decode_raw(self, z):
        x = self._conv("post_quant_conv", z.to(self.dtype), pad=0)
        x = self._conv("decoder.conv_in", x)
        x = self._res("decoder.mid_block.resnets.0", x)
        x = self._attn("decoder.mid_block.attentions.0", x)
        x = self._res("decoder.mid_block.resnets.1", x)
        for i in range(4):
            for j in range(3):
                x = self._res(f"decoder.up_blocks.{i}.resnets.{j}", x)
            if i != 3:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = self._conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", x)
        x = F.silu(self._gn("decoder.conv_norm_out", x))
        return self._conv("decoder.conv_out", x)

    @torch.no_grad()
    def encode_moments(self, img):
        x = self._conv("encoder.conv_in", img.to(self.dtype))
        for i in range(4):
            for j in range(2):
                x = self._res(f"encoder.down_blocks.{i}.resnets.{j}", x)
            if i != 3:
                x = F.pad(x, (0, 1, 0, 1))
                x = self._conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", x, stride=2, pad=0)
        x = self._res("encoder.mid_block.resnets.0", x)
        x = self._attn("encoder.mid_block.attentions.0", x)
        x = self._res("encoder.mid_block.resnets.1", x)
        x = F.silu(self._gn("encoder.conv_norm_out", x))
        x = self._conv("encoder.conv_out", x)
        x = self._conv("quant_conv", x, pad=0)
        mean, logvar = x.chunk(2, dim=1)
        return mean, logvar.clamp(-30.0, 20.0)

    # reference-facing API -------------------------------------------------------
    def decode(self, zt):
        """zt -> image in [-1, 1]-ish, fp32 (latent_diffusion.py:123-129 / latent_sdxl.py:155-164)."""
        return self.decode_raw(zt / self.scaling_factor).float()

    def encode(self, x, sample: bool = True, generator=None):
        """x -> latent: posterior sample * scaling_factor (latent_diffusion.py:117-121)."""
        mean, logvar = self.encode_moments(x)
        if sample:
            noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
            mean = mean + torch.exp(0.5 * logvar) * noise
        return mean * self.scaling_factor
