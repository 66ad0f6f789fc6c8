"""AutoencoderKL decode / encode.

``HipVAE``   - decoder AND encoder on the hand-written HIP kernels (``csrc/vae.hip`` behind the C ABI:
               implicit-GEMM convs incl. fused upsample / asymmetric stride-2 downsample, GroupNorm+SiLU,
               GEMM-softmax-GEMM mid-block attention, quant_conv + posterior kernel).  This is what
               ``solver.decode()`` / ``solver.encode()`` run on the GPU and what the benchmark times.
The fp32 torch restatement used as its parity reference lives in ``oracle/vae_ref.py`` (test
infrastructure; never imported from here).
Weights: seeded synthetic in exact diffusers shapes / key names (no checkpoints offline).

Reference call sites: latent_diffusion.py:117-129 (scale 0.18215),
latent_sdxl.py:44,150-164 (``vae.config.scaling_factor`` = 0.13025).
"""
from __future__ import annotations

import hashlib
from typing import Dict

import torch

CH = (128, 256, 512, 512)


def vae_param_shapes() -> Dict[str, tuple]:
    P: Dict[str, tuple] = {}

    def conv(p, o, i, k):
        P[p + ".weight"] = (o, i, k, k)
        P[p + ".bias"] = (o,)

    def norm(p, c):
        P[p + ".weight"] = (c,)
        P[p + ".bias"] = (c,)

    def lin(p, o, i):
        P[p + ".weight"] = (o, i)
        P[p + ".bias"] = (o,)

    def res(p, i, o):
        norm(p + ".norm1", i)
        conv(p + ".conv1", o, i, 3)
        norm(p + ".norm2", o)
        conv(p + ".conv2", o, o, 3)
        if i != o:
            conv(p + ".conv_shortcut", o, i, 1)

    def attn(p, c):
        norm(p + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(f"{p}.{n}", c, c)

    # encoder
    conv("encoder.conv_in", CH[0], 3, 3)
    ch = CH[0]
    for i, co in enumerate(CH):
        for j in range(2):
            res(f"encoder.down_blocks.{i}.resnets.{j}", ch, co)
            ch = co
        if i != len(CH) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    res("encoder.mid_block.resnets.0", ch, ch)
    attn("encoder.mid_block.attentions.0", ch)
    res("encoder.mid_block.resnets.1", ch, ch)
    norm("encoder.conv_norm_out", ch)
    conv("encoder.conv_out", 8, ch, 3)
    conv("quant_conv", 8, 8, 1)
    # decoder
    conv("post_quant_conv", 4, 4, 1)
    conv("decoder.conv_in", CH[-1], 4, 3)
    ch = CH[-1]
    res("decoder.mid_block.resnets.0", ch, ch)
    attn("decoder.mid_block.attentions.0", ch)
    res("decoder.mid_block.resnets.1", ch, ch)
    for i, co in enumerate(reversed(CH)):
        for j in range(3):
            res(f"decoder.up_blocks.{i}.resnets.{j}", ch, co)
            ch = co
        if i != len(CH) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("decoder.conv_norm_out", ch)
    conv("decoder.conv_out", 3, ch, 3)
    return P


def synth_vae_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    sd = {}
    for k, shape in vae_param_shapes().items():
        g = torch.Generator().manual_seed(int.from_bytes(hashlib.sha256(f"vae{seed}:{k}".encode()).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF)
        leaf = k.rsplit(".", 1)[-1]
        if "norm" in k:
            t = (1.0 + 0.1 * torch.randn(shape, generator=g)) if leaf == "weight" else 0.05 * torch.randn(shape, generator=g)
        elif leaf == "bias":
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan = 1
            for d in shape[1:]:
                fan *= d
            gain = 0.5 if (".conv2." in k or ".to_out." in k) else 1.0
            t = torch.randn(shape, generator=g) * gain / fan ** 0.5
        sd[k] = t
    return sd


# older AutoencoderKL checkpoints (e.g. the published SD1.5 VAE) name the mid-block attention projections
# query / key / value / proj_attn - sometimes as 1x1-conv-shaped [C, C, 1, 1] weights, which the engine accepts
# (same element count); diffusers >= 0.15 renames them on load exactly like this
_DEPRECATED_ATTN = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}


def _canonical_vae_key(k: str) -> str:
    if ".attentions." in k:
        for old, new in _DEPRECATED_ATTN.items():
            if old in k:
                return k.replace(old, new)
    return k


class HipVAE:
    """VAE whose ``decode`` and ``encode`` run on libcfgpp_hip.so (no MIOpen, no torch conv).
    ``with_encoder=False`` skips loading the encoder weights (text-to-image solvers never encode)."""

    def __init__(self, scaling_factor: float, latent_hw, max_batch: int = 1, device=None, state_dict=None, seed: int = 0,
                 with_encoder: bool = True):
        import ctypes as C

        from . import _lib
        from ._lib import CfgppError, check
        if not torch.cuda.is_available():
            raise CfgppError("HipVAE needs a ROCm GPU; the HIP path has no CPU fallback")
        self.lib = _lib.load()
        dev = torch.device(device if device is not None else "cuda")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.scaling_factor = float(scaling_factor)
        self.h, self.w = int(latent_hw[0]), int(latent_hw[1])
        self.max_batch = int(max_batch)
        self._sd = state_dict if state_dict is not None else synth_vae_state_dict(seed)
        torch.cuda.set_device(self.device)      # one device per process: the engine's device is the current device (see HipEngine)
        self._h = self.lib.cfgpp_vae_create(self.h, self.w, self.max_batch, self.scaling_factor, self.device.index)
        if not self._h:
            raise CfgppError("cfgpp_vae_create failed: " + _lib.last_error())
        for k, v in self._sd.items():
            k = _canonical_vae_key(k)
            if not with_encoder and (k.startswith("encoder.") or k.startswith("quant_conv.")):
                continue
            t = v.detach().cpu().contiguous()
            dt = 1 if t.dtype == torch.float16 else 0
            if dt == 0:
                t = t.to(torch.float32)
            shape = (C.c_long * t.dim())(*t.shape)
            check(self.lib.cfgpp_vae_load_tensor(self._h, k.encode(), t.data_ptr(), dt, shape, t.dim()), f"cfgpp_vae_load_tensor({k})")
        check(self.lib.cfgpp_vae_finalize(self._h), "cfgpp_vae_finalize")
        self.with_encoder = bool(with_encoder)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
            self.lib.cfgpp_vae_destroy(h)
            self._h = None

    def profile(self, zt: torch.Tensor) -> list:
        """one decode with HIP events between the launches: [(family, description, us, GFLOP)] in plan order"""
        import ctypes as C
        from ._lib import check
        zt = zt.to(self.device, torch.float32).contiguous()
        img = torch.empty((zt.shape[0], 3, 8 * self.h, 8 * self.w), dtype=torch.float32, device=self.device)
        buf = C.create_string_buffer(1 << 20)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.cfgpp_vae_profile(self._h, zt.data_ptr(), img.data_ptr(), int(zt.shape[0]), stream, buf, 1 << 20), "cfgpp_vae_profile")
        rows = []
        for line in buf.value.decode().strip().split("\n"):
            _, kind, desc, us, gf = line.split("\t")
            rows.append((int(kind), desc, float(us), float(gf)))
        return rows

    def decode_image(self, zt: torch.Tensor) -> torch.Tensor:
        """zt -> ``(decode(zt) / 2 + 0.5).clamp(0, 1)`` with the post-processing of the solvers' ``sample()`` folded into
        the decoder's last kernel (latent_diffusion.py:676-677)."""
        return self.decode(zt, _entry="cfgpp_vae_decode_image")

    def decode(self, zt: torch.Tensor, _entry: str = "cfgpp_vae_decode") -> torch.Tensor:
        """zt [B,4,h,w] -> image [B,3,8h,8w] fp32 on the GPU (reference: latent_diffusion.py:123-129)."""
        from ._lib import CfgppError, check
        z = zt.to(device=self.device, dtype=torch.float32).contiguous()
        B = int(z.shape[0])
        if tuple(z.shape[1:]) != (4, self.h, self.w) or B > self.max_batch:
            raise CfgppError(f"HipVAE.decode: latent {tuple(z.shape)} does not fit engine [<= {self.max_batch}, 4, {self.h}, {self.w}]")
        img = torch.empty((B, 3, 8 * self.h, 8 * self.w), dtype=torch.float32, device=self.device)
        check(getattr(self.lib, _entry)(self._h, z.data_ptr(), img.data_ptr(), B, torch.cuda.current_stream(self.device).cuda_stream), _entry)
        return img

    def encode(self, x, sample: bool = True, generator=None, noise=None, return_moments: bool = False):
        """x [B,3,8h,8w] in [-1,1] -> latent [B,4,h,w] fp32 = posterior sample * scaling_factor
        (reference: latent_diffusion.py:117-121).  The posterior noise is drawn on the HOST (``generator`` or the
        global CPU RNG, like the reference's initial latent) so runs are reproducible across devices; pass
        ``noise`` to pin it, ``sample=False`` for the posterior mean."""
        from ._lib import CfgppError, check
        if not self.with_encoder:
            raise CfgppError("HipVAE.encode: engine was built with with_encoder=False")
        img = x.to(device=self.device, dtype=torch.float32).contiguous()
        B = int(img.shape[0])
        if tuple(img.shape[1:]) != (3, 8 * self.h, 8 * self.w) or B > self.max_batch:
            raise CfgppError(f"HipVAE.encode: image {tuple(img.shape)} does not fit engine [<= {self.max_batch}, 3, {8 * self.h}, {8 * self.w}]")
        if sample and noise is None:
            noise = torch.randn((B, 4, self.h, self.w), generator=generator)
        nz = None if (not sample or noise is None) else noise.to(device=self.device, dtype=torch.float32).contiguous()
        z = torch.empty((B, 4, self.h, self.w), dtype=torch.float32, device=self.device)
        mom = torch.empty((B, 8, self.h, self.w), dtype=torch.float32, device=self.device) if return_moments else None
        check(self.lib.cfgpp_vae_encode(self._h, img.data_ptr(), None if nz is None else nz.data_ptr(), z.data_ptr(),
                                        None if mom is None else mom.data_ptr(), B, torch.cuda.current_stream(self.device).cuda_stream),
              "cfgpp_vae_encode")
        return (z, mom) if return_moments else z

    def encode_flops(self, B: int) -> float:
        return float(self.lib.cfgpp_vae_encode_flops(self._h, int(B)))

    def flops(self, B: int) -> float:
        return float(self.lib.cfgpp_vae_flops(self._h, int(B)))
