"""Stub environment that lets the *reference's own* sampler modules import and
run on CPU in the build container (there is no ``diffusers`` here).

This file is OUR code: it fakes the third-party objects the reference pulls in
(`diffusers` schedulers / pipelines, `munch`, `torchvision.utils.save_image`)
with the smallest objects that satisfy the attribute accesses the reference
makes.  It is used only by ``make_golden.py`` (to record golden vectors from
the reference) and never travels as reference source.

The scripted UNet is *pointwise* (eps at a pixel depends only on z at that
pixel, on t and on per-row scalars derived from the conditioning), so spatial
crops of a trajectory are self-consistent and fixtures stay small.
"""
from __future__ import annotations

import hashlib
import sys
import types

import numpy as np
import torch

# ----------------------------------------------------------------------------
# scripted components
# ----------------------------------------------------------------------------


def prompt_to_seed(prompt) -> int:
    if isinstance(prompt, (list, tuple)):
        prompt = "|".join(prompt)
    return int.from_bytes(hashlib.sha256(prompt.encode("utf-8")).digest()[:4], "little")


def fake_embed(prompt, shape, scale=0.5) -> torch.Tensor:
    g = torch.Generator().manual_seed(prompt_to_seed(prompt))
    return torch.randn(shape, generator=g) * scale


def pointwise_eps(z, t, ehs, text_embeds=None, time_ids=None, out_dtype=torch.float16):
    """The scripted UNet.  z [R,4,H,W], t [R] or [1], ehs [R,77,D]."""
    zf = z.float()
    R = zf.shape[0]
    tt = (t.float().reshape(-1) / 1000.0)
    if tt.numel() == 1:
        tt = tt.expand(R)
    tt = tt.view(R, 1, 1, 1)
    ctx = ehs.float().mean(dim=(1, 2)).view(-1, 1, 1, 1) * 40.0
    if ctx.shape[0] == 1:
        ctx = ctx.expand(R, 1, 1, 1)
    add = torch.zeros(R, 1, 1, 1)
    if text_embeds is not None:
        te = text_embeds.float().mean(dim=-1).view(-1, 1, 1, 1) * 10.0
        ti = time_ids.float().sum(dim=-1).view(-1, 1, 1, 1) * 1e-4
        add = (te + ti).expand(R, 1, 1, 1) if te.shape[0] == 1 else (te + ti)
    ch = torch.arange(zf.shape[1], dtype=torch.float32).view(1, -1, 1, 1)
    eps = zf * (0.35 + 0.5 * tt) + 0.25 * torch.sin(1.7 * zf + ctx + 0.3 * ch) + 0.1 * torch.tanh(ctx + add)
    return eps.to(out_dtype)


class _Cfg(types.SimpleNamespace):
    pass


class FakeUNet:
    def __init__(self, sdxl=False, out_dtype=torch.float16):
        self.out_dtype = out_dtype
        self.calls = []
        self.config = _Cfg(sample_size=128 if sdxl else 64, addition_time_embed_dim=256)
        self.add_embedding = _Cfg(linear_1=_Cfg(in_features=2816))

    def __call__(self, sample, timestep, encoder_hidden_states=None, added_cond_kwargs=None):
        te = ti = None
        if added_cond_kwargs is not None:
            te, ti = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
        self.calls.append(dict(rows=int(sample.shape[0]), t=timestep.detach().clone(),
                               z_dtype=str(sample.dtype),
                               te_rows=None if te is None else int(te.shape[0])))
        eps = pointwise_eps(sample, timestep, encoder_hidden_states, te, ti, self.out_dtype)
        return {"sample": eps}


class _TokOut(types.SimpleNamespace):
    pass


class FakeTokenizer:
    model_max_length = 77

    def __call__(self, prompt, **kw):
        # carry the prompt string through `.input_ids.to(device)`
        class _Ids:
            def __init__(self, p):
                self.p = p

            def to(self, device):
                return self
        return _TokOut(input_ids=_Ids(prompt))


class FakeTextEncoder:
    def __init__(self, dim, pooled_dim=None, tag=""):
        self.dim, self.pooled_dim, self.tag = dim, pooled_dim, tag

    def __call__(self, ids, output_hidden_states=False):
        p = ids.p
        p = p if isinstance(p, str) else "|".join(p)
        hs = fake_embed(self.tag + p, (1, 77, self.dim))
        if not output_hidden_states:
            return (hs,)

        class _Out:
            def __init__(o):
                o.hidden_states = [hs * 0.1, hs, hs * 3.0]   # [-2] is `hs`
                o._pooled = fake_embed(self.tag + "pool" + p, (1, self.pooled_dim or self.dim))

            def __getitem__(o, i):
                assert i == 0
                return o._pooled
        return _Out()


class _LatentDist:
    def __init__(self, z):
        self.z = z

    def sample(self):
        return self.z


class FakeVAE:
    """encode = 8x8 average pool of a fixed channel mix; decode = its adjoint-ish."""

    latent_dtype = None      # set to torch.float16 to emulate the reference's fp16 VAE (its latents are fp16)

    def __init__(self, scaling_factor):
        self.config = _Cfg(scaling_factor=scaling_factor, block_out_channels=[128, 256, 512, 512],
                           force_upcast=False)
        self.dtype = torch.float32

    def to(self, *a, **k):
        return self

    def encode(self, x):
        xf = x.float()
        z = torch.nn.functional.avg_pool2d(xf, 8)
        mix = torch.tensor([[1.0, 0.2, -0.3], [0.1, 0.9, 0.4], [-0.5, 0.3, 0.8], [0.3, -0.6, 0.5]])
        z = torch.einsum("oc,bchw->bohw", mix, z) * 4.0
        if FakeVAE.latent_dtype is not None:
            z = z.to(FakeVAE.latent_dtype)
        return _Cfg(latent_dist=_LatentDist(z))

    def decode(self, z):
        zf = z.float()
        mix = torch.tensor([[0.5, 0.1, -0.2, 0.1], [0.1, 0.4, 0.2, -0.3], [-0.2, 0.2, 0.4, 0.2]])
        x = torch.einsum("oc,bchw->bohw", mix, zf)
        x = torch.nn.functional.interpolate(x, scale_factor=8, mode="nearest")
        return _Cfg(sample=torch.tanh(x * 0.25))


class FakeDDIMScheduler:
    def __init__(self):
        from cfgpp_amd import schedule as S
        self.alphas_cumprod = S.alphas_cumprod()
        self.final_alpha_cumprod = self.alphas_cumprod[0].clone()
        self.timesteps = torch.arange(999, -1, -1)
        self.config = _Cfg()

    @classmethod
    def from_pretrained(cls, *a, **k):
        return cls()

    def set_timesteps(self, n, device=None):
        ratio = 1000 // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + 1
        self.timesteps = torch.from_numpy(ts)


class FakeEulerScheduler:
    def __init__(self):
        from cfgpp_amd import schedule as S
        self.alphas_cumprod = S.alphas_cumprod()
        self.timesteps = torch.arange(999, -1, -1).float()

    @classmethod
    def from_config(cls, *a, **k):
        assert k.get("timestep_spacing") == "trailing"
        return cls()

    def set_timesteps(self, n, device=None):
        ts = np.round(np.arange(1000, 0, -1000 / n)) - 1
        self.timesteps = torch.from_numpy(ts.astype(np.float32))


class FakeSDPipe:
    last = None

    def __init__(self, unet_dtype):
        self.vae = FakeVAE(0.18215)
        self.tokenizer = FakeTokenizer()
        self.text_encoder = FakeTextEncoder(768, tag="L")
        self.unet = FakeUNet(False, unet_dtype)
        FakeSDPipe.last = self

    @classmethod
    def from_pretrained(cls, key, torch_dtype=torch.float16):
        return cls(torch_dtype)   # eps dtype follows pipe_dtype (fp16 = the real autocast behaviour)

    def to(self, device):
        return self


class FakeSDXLPipe:
    last = None

    def __init__(self):
        self.tokenizer = FakeTokenizer()
        self.tokenizer_2 = FakeTokenizer()
        self.text_encoder = FakeTextEncoder(768, 768, tag="L")
        self.text_encoder_2 = FakeTextEncoder(1280, 1280, tag="G")
        self.unet = FakeUNet(True, torch.float16)
        self.scheduler = _Cfg(config=_Cfg())
        FakeSDXLPipe.last = self

    @classmethod
    def from_pretrained(cls, key, torch_dtype=torch.float16):
        return cls()

    @classmethod
    def from_single_file(cls, path, torch_dtype=torch.float16):
        return cls()

    def to(self, device):
        return self


class FakeAutoencoderKL:
    @classmethod
    def from_pretrained(cls, key, torch_dtype=torch.float16):
        return FakeVAE(0.13025)


def install(reference_root="/root/reference"):
    """Register the stub modules and put the reference on sys.path."""
    d = types.ModuleType("diffusers")
    d.DDIMScheduler = FakeDDIMScheduler
    d.StableDiffusionPipeline = FakeSDPipe
    d.AutoencoderKL = FakeAutoencoderKL
    d.StableDiffusionXLPipeline = FakeSDXLPipe
    d.UNet2DConditionModel = object
    d.EulerDiscreteScheduler = FakeEulerScheduler
    dm = types.ModuleType("diffusers.models")
    dap = types.ModuleType("diffusers.models.attention_processor")
    for n in ("AttnProcessor2_0", "LoRAAttnProcessor2_0", "LoRAXFormersAttnProcessor", "XFormersAttnProcessor"):
        setattr(dap, n, type(n, (), {}))
    sys.modules["diffusers"] = d
    sys.modules["diffusers.models"] = dm
    sys.modules["diffusers.models.attention_processor"] = dap
    m = types.ModuleType("munch")
    m.munchify = lambda dct: types.SimpleNamespace(**dct)
    sys.modules["munch"] = m
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.save_image = lambda *a, **k: None
    tv.utils = tvu
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.utils"] = tvu
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)
