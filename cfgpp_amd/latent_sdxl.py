"""SDXL / SDXL-Lightning solvers on the MI355X-native engine.  Same registry
names, ``sample()`` / ``reverse_process()`` / ``inversion()`` signatures,
conditioning plumbing and index quirks as the reference's ``latent_sdxl.py``
(registry :15-28, SDXL :32-363, SDXLLightning :366-418, solvers :425-1025).

Batch extension: ``prompt1=[null, [text_1..text_B]]``.
UNet rows are ``[uc_1..uc_B, c_1..c_B]``; the added conditioning follows the
reference: ``[neg_1..neg_B, pos_1..pos_B]`` when lambda not in {0, 1}, else the
positive rows only, which the UNet then applies to BOTH halves (quirk Q7).
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import torch

from .registry import Registry
from . import coeffs as K
from .conditioning import SyntheticTextEncoder, as_list
from .latent_diffusion import StableDiffusion, _progress
from .schedule import get_sigmas_karras
from .unet_config import SDXL as SDXL_CFG

# ---- solver registry (names: Appendix A of SURVEY.md) ----
__SOLVER__ = Registry("Solver")
register_solver = __SOLVER__.register        # @register_solver(name)
get_solver = __SOLVER__.create               # get_solver(name, solver_config=..., device=..., **kw)


class SDXL(StableDiffusion):
    """reference: latent_sdxl.py:32-363."""

    unet_config = SDXL_CFG
    quantize = True

    def __init__(self, solver_config, model_key: str = "stabilityai/stable-diffusion-xl-base-1.0",
                 dtype=torch.float16, device="cuda", **kwargs):
        kwargs.setdefault("pipe_dtype", dtype)
        cfg = kwargs.get("unet_config", self.unet_config)
        if kwargs.get("text_encoder") is None:
            # two encoders: CLIP-L (768) || OpenCLIP-bigG (D-768), pooled from the second one
            d2 = cfg.cross_attention_dim - min(768, cfg.cross_attention_dim // 2)
            kwargs["text_encoder"] = (SyntheticTextEncoder(cfg.cross_attention_dim - d2, cfg.addition_pooled_dim, tag="clip_l"),
                                      SyntheticTextEncoder(d2, cfg.addition_pooled_dim, tag="clip_g"))
        super().__init__(solver_config, model_key=model_key, device=device, **kwargs)
        self.vae_scale_factor = 8
        self.default_sample_size = self.cfg.sample_size

    # ------------------------------------------------------------------ text
    @torch.no_grad()
    def _text_embed(self, prompt, encoder, clip_skip=None):
        """reference: latent_sdxl.py:76-93.  ``clip_skip`` selects ``hidden_states[-(clip_skip + 2)]`` there; an
        encoder that exposes its hidden states takes it as a keyword, one that cannot honour it must not
        silently ignore it."""
        if clip_skip is None:
            hs, pooled = encoder(as_list(prompt))
        else:
            try:
                hs, pooled = encoder(as_list(prompt), clip_skip=int(clip_skip))
            except TypeError as e:
                raise NotImplementedError(f"clip_skip={clip_skip}: text encoder {type(encoder).__name__} returns one "
                                          "hidden state only (no clip_skip keyword)") from e
        return hs, pooled

    @torch.no_grad()
    def get_text_embed(self, null_prompt_1, prompt_1, null_prompt_2=None, prompt_2=None, clip_skip=None):
        """-> null_prompt_embeds, prompt_embeds, pool_null_embed, pool_prompt_embed
        (reference: latent_sdxl.py:95-128; pooled output always from the last encoder used)."""
        enc1, enc2 = self.text_encoder
        pe1, pool = self._text_embed(prompt_1, enc1, clip_skip)
        pe = [pe1]
        if prompt_2 is not None:
            pe2, pool = self._text_embed(prompt_2, enc2, clip_skip)
            pe.append(pe2)
        ne1, pool_null = self._text_embed(null_prompt_1, enc1, clip_skip)
        ne = [ne1]
        if null_prompt_2 is not None:
            ne2, pool_null = self._text_embed(null_prompt_2, enc2, clip_skip)
            ne.append(ne2)
        dev = self.work_device
        return (torch.cat(ne, dim=-1).to(dev), torch.cat(pe, dim=-1).to(dev), pool_null.to(dev), pool.to(dev))

    def _get_add_time_ids(self, original_size, crops_coords_top_left, target_size, dtype, text_encoder_projection_dim):
        add_time_ids = list(original_size + crops_coords_top_left + target_size)
        passed = self.cfg.addition_time_embed_dim * len(add_time_ids) + text_encoder_projection_dim
        expected = 6 * self.cfg.addition_time_embed_dim + self.cfg.addition_pooled_dim
        assert expected == passed, (
            f"Model expects an added time embedding vector of length {expected}, but a vector of {passed} was created.")
        return torch.tensor([add_time_ids], dtype=dtype)

    # ------------------------------------------------------------------ UNet
    def predict_noise(self, zt, t, uc, c, added_cond_kwargs):
        """reference: latent_sdxl.py:167-185."""
        self._ensure_context(uc, c, added_cond_kwargs)
        noise_uc, noise_c = self.engine.predict(zt, float(t))
        if uc is None:
            return noise_c, noise_c
        if c is None:
            return noise_uc, noise_uc
        return noise_uc, noise_c

    def _ensure_context(self, uc, c, added_cond_kwargs):
        if uc is None and c is None:
            raise ValueError("predict_noise needs at least one of uc / c")
        a = c if uc is None else uc
        b = uc if c is None else c
        te, ti = added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]
        key = (a.data_ptr(), b.data_ptr(), te.data_ptr(), ti.data_ptr(), tuple(te.shape), tuple(a.shape), tuple(b.shape),
               a._version, b._version, te._version, ti._version)
        if getattr(self, "_ctx_key", None) != key:
            B = max(int(a.shape[0]), int(b.shape[0]))
            rows = 2 * B
            # cond rows: 2B ([neg.., pos..]) or B / 1 (positive only -> applied to both halves, quirk Q7)
            if te.shape[0] not in (rows, B, 1):
                raise ValueError(f"text_embeds has {te.shape[0]} rows for a UNet batch of {rows}")
            te_full = te if te.shape[0] == rows else te.repeat(rows // te.shape[0], 1)
            ti_full = ti if ti.shape[0] == rows else ti.repeat(rows // ti.shape[0], 1)
            self.engine.set_context(a, b, te_full, ti_full)
            self._ctx_key = key
            self._ctx_keep = (a, b, te, ti)

    # ------------------------------------------------------------------ sample
    def _sizes(self, original_size, target_size):
        h = self.default_sample_size * self.vae_scale_factor
        return (original_size or (h, h)), (target_size or (h, h))

    def _cond_kwargs(self, pool_pos, pool_null, cfg_guidance, original_size, crops, target_size,
                     neg_original_size, neg_crops, neg_target_size, dtype):
        B = int(pool_pos.shape[0])
        dim = int(pool_pos.shape[-1])
        ids = self._get_add_time_ids(original_size, crops, target_size, dtype, dim).repeat(B, 1)
        if neg_original_size is not None and neg_target_size is not None:
            neg_ids = self._get_add_time_ids(neg_original_size, neg_crops, neg_target_size, dtype, dim).repeat(B, 1)
        else:
            neg_ids = ids
        te = pool_pos
        if cfg_guidance != 0.0 and cfg_guidance != 1.0:      # do cfg (latent_sdxl.py:249-252)
            te = torch.cat([pool_null.expand(B, -1), pool_pos], dim=0)
            ids = torch.cat([neg_ids, ids], dim=0)
        return {"text_embeds": te.to(self.work_device), "time_ids": ids.to(self.work_device)}

    @torch.no_grad()
    def sample(self, prompt1=["", ""], prompt2=["", ""], cfg_guidance: float = 5.0,
               original_size: Optional[Tuple[int, int]] = None, crops_coords_top_left: Tuple[int, int] = (0, 0),
               target_size: Optional[Tuple[int, int]] = None, negative_original_size: Optional[Tuple[int, int]] = None,
               negative_crops_coords_top_left: Tuple[int, int] = (0, 0),
               negative_target_size: Optional[Tuple[int, int]] = None, clip_skip: Optional[int] = None, **kwargs):
        original_size, target_size = self._sizes(original_size, target_size)
        pe = kwargs.pop("prompt_embeds", None)
        if pe is not None:          # (null_embeds, embeds, pool_null, pool)
            null_e, emb, pool_null, pool = (x.to(self.work_device, torch.float16) for x in pe)
        else:
            null_e, emb, pool_null, pool = self.get_text_embed(prompt1[0], prompt1[1], prompt2[0], prompt2[1], clip_skip)
        add_cond_kwargs = self._cond_kwargs(pool, pool_null, cfg_guidance, original_size, crops_coords_top_left, target_size,
                                            negative_original_size, negative_crops_coords_top_left, negative_target_size,
                                            emb.dtype)
        self._batch = int(emb.shape[0])
        ret_lat = kwargs.pop("return_latents", False)
        zt = self.reverse_process(null_e, emb, cfg_guidance, add_cond_kwargs, target_size, **kwargs)
        if ret_lat:
            return zt
        return self._finish(zt)

    # ------------------------------------------------------------------ latents
    def initialize_latent(self, method: str = "random", src_img: Optional[torch.Tensor] = None,
                          add_cond_kwargs: Optional[dict] = None, **kwargs):
        if method == "ddim":
            z0 = kwargs.get("src_latent")
            if z0 is None:
                assert src_img is not None, "src_img must be provided for inversion"
                z0 = self.encode(src_img.to(self.dtype))
            z = self.inversion(z0, kwargs.get("uc"), kwargs.get("c"), kwargs.get("cfg_guidance", 0.0), add_cond_kwargs)
        elif method == "npi":
            assert src_img is not None, "src_img must be provided for inversion"
            z = self.inversion(self.encode(src_img.to(self.dtype)), kwargs.get("c"), kwargs.get("c"), 1.0, add_cond_kwargs)
        elif method == "random":
            size = tuple(kwargs.get("size", (1, 4, 128, 128)))
            z = self._randn(size, kwargs.get("seeds")).to(self.work_device)
        elif method == "random_kdiffusion":
            size = tuple(kwargs.get("latent_dim", (1, 4, 128, 128)))
            sigmas = kwargs.get("sigmas", [14.6146])
            z = self._randn(size, kwargs.get("seeds"))
            z = (z * (sigmas[0] ** 2 + 1) ** 0.5).to(self.work_device)
        else:
            raise NotImplementedError
        return z

    def _latent_size(self, shape):
        B = getattr(self, "_batch", 1)
        return (B, 4, shape[1] // self.vae_scale_factor, shape[0] // self.vae_scale_factor)

    def _split_cond_for_inversion(self, cfg_guidance, add_cond_kwargs):
        # lambda in {0,1}: add_cond_kwargs is reduced IN PLACE to its last row (latent_sdxl.py:302-305)
        # batch extension: the reference is batch-1, where "[-1]" is THE positive row; with B chains the
        # positive rows are the last B ([neg_1..neg_B, pos_1..pos_B] or already [pos_1..pos_B])
        if cfg_guidance == 0.0 or cfg_guidance == 1.0:
            B = getattr(self, "_batch", 1)
            if B <= 1:
                add_cond_kwargs["text_embeds"] = add_cond_kwargs["text_embeds"][-1].unsqueeze(0)
                add_cond_kwargs["time_ids"] = add_cond_kwargs["time_ids"][-1].unsqueeze(0)
            else:
                add_cond_kwargs["text_embeds"] = add_cond_kwargs["text_embeds"][-B:]
                add_cond_kwargs["time_ids"] = add_cond_kwargs["time_ids"][-B:]

    def _inversion_xl(self, z0, uc, c, cfg_guidance, add_cond_kwargs, cfgpp):
        self._split_cond_for_inversion(cfg_guidance, add_cond_kwargs)
        zt = self._own_latent(z0)
        self._ensure_context(uc, c, add_cond_kwargs)
        done = self._graph_loop(zt, list(reversed(self.scheduler.timesteps)), uc, c, cfg_guidance, cfgpp, False, lambda t: (
            self.tables.ddim_sqrt_coeffs(t, inversion=True), "tw" if int(t) - self.tables.skip < 0 else None))
        if done is not None:
            return done[1]
        z0t = torch.empty_like(zt)
        for t in _progress(reversed(self.scheduler.timesteps), "DDIM inversion"):
            sqrt4 = self.tables.ddim_sqrt_coeffs(t, inversion=True)
            noise_uc, noise_c = self.predict_noise(zt, t, uc, c, add_cond_kwargs)
            dev_a = "tw" if int(t) - self.tables.skip < 0 else None      # self.alpha(t - skip) = the DEVICE-resident final alpha
            self._ddim_update(zt, z0t, noise_uc, noise_c, cfg_guidance, sqrt4, cfgpp, False, dev_a)
        return zt

    def inversion(self, z0, uc, c, cfg_guidance, add_cond_kwargs):
        """CFG inversion (reference: latent_sdxl.py:301-320)."""
        return self._inversion_xl(z0, uc, c, cfg_guidance, add_cond_kwargs, cfgpp=False)

    def reverse_process(self, *args, **kwargs):
        raise NotImplementedError

    # ------------------------------------------------------------------ k-diffusion helpers
    def sigma_to_t(self, sigma, quantize=None):
        return self.tables.sigma_to_t(sigma, self.quantize if quantize is None else quantize)

    def _ddim_xl(self, null_e, emb, cfg_guidance, add_cond_kwargs, shape, cfgpp, callback_fn, wrap, zt=None, desc="SDXL",
                 seeds=None):
        if zt is None:
            zt = self.initialize_latent(size=self._latent_size(shape), seeds=seeds)
        zt = self._own_latent(zt)
        ts = self.scheduler.timesteps.int() if wrap else self.scheduler.timesteps
        if callback_fn is None:
            self._ensure_context(null_e, emb, add_cond_kwargs)
            done = self._graph_loop(zt, ts, null_e, emb, cfg_guidance, False, cfgpp, lambda t: (
                self.tables.ddim_sqrt_coeffs(t, wrap=wrap), "rn" if (not wrap and int(t) - self.tables.skip < 0) else None))
            if done is not None:
                return done[0]
        z0t = torch.empty_like(zt)
        for step, t in enumerate(_progress(ts, desc)):
            sqrt4 = self.tables.ddim_sqrt_coeffs(t, wrap=wrap)
            noise_uc, noise_c = self.predict_noise(zt, t, null_e, emb, add_cond_kwargs)
            dev_a = "rn" if (not wrap and int(t) - self.tables.skip < 0) else None
            self._ddim_update(zt, z0t, noise_uc, noise_c, cfg_guidance, sqrt4, False, cfgpp, dev_a)
            if callback_fn is not None:
                self._run_callback(callback_fn, step, t, z0t, zt)
        return z0t          # for the last step, do not add noise

    def _kdiff_xl(self, null_e, emb, cfg_guidance, add_cond_kwargs, shape, callback_fn, *, sigmas, n_steps, x0_scale_mode,
                  input_mode, xl_form, variant, solver, t_of_sigma, ret, seeds=None):
        size = self._latent_size(shape)
        if x0_scale_mode == "kdiff":       # randn * sqrt(sigma0^2 + 1) in fp32, then cast (latent_sdxl.py:290-294)
            x = self.initialize_latent(method="random_kdiffusion", latent_dim=size, sigmas=sigmas, seeds=seeds).to(torch.float16)
        else:                              # randn cast to fp16, then * sigma0 (latent_sdxl.py:882-884)
            x = self.initialize_latent(method="random", size=size, seeds=seeds).to(torch.float16)
            x = x * sigmas[0]
        x = x.contiguous()
        xc, den, old = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        have_old = False
        for i in _progress(range(n_steps), "SDXL"):
            sigma = sigmas[i]
            new_t = t_of_sigma(sigma)
            if input_mode == 0:
                self.engine.kdiff_input(x, xc, K.kdiff_input_scale_sd(sigma, self.scalar_semantics), 0)
            else:
                self.engine.kdiff_input(x, xc, float(self._alphas_2m[i].clone().sqrt()), 1)
            noise_uc, noise_c = self.predict_noise(xc, new_t, null_e, emb, add_cond_kwargs)
            first = (solver == "euler") or (not have_old)
            coef, euler = K.kdiff_coeffs(cfg_guidance, sigmas, i, first, xl_form=xl_form, semantics=self.scalar_semantics)
            self.engine.step_kdiff(x, den, old, noise_uc, noise_c, coef, variant, xl_form, euler, solver != "euler")
            have_old = True
            if callback_fn is not None:
                self._run_callback(callback_fn, i, new_t, den, x)
        return den if ret == "den" else x


class SDXLLightning(SDXL):
    """reference: latent_sdxl.py:366-418 (Euler "trailing" timesteps, no final_alpha_cumprod)."""
    scheduler_kind = "lightning"

    def __init__(self, solver_config, base_model_key: str = "stabilityai/stable-diffusion-xl-base-1.0",
                 light_model_ckpt: str = "ckpt/sdxl_lightning_4step_unet.safetensors", dtype=torch.float16,
                 device="cuda", **kwargs):
        # the reference swaps the Lightning UNet state dict into the base pipeline (latent_sdxl.py:378-390): here the
        # checkpoint file - when it exists - IS the engine's UNet weights (no checkpoints exist offline: synthetic then)
        self.light_model_ckpt = light_model_ckpt
        import os
        if kwargs.get("unet_weights") is None and kwargs.get("engine") is None and light_model_ckpt and os.path.exists(light_model_ckpt):
            kwargs["unet_weights"] = light_model_ckpt
        SDXL.__init__(self, solver_config, model_key=base_model_key, dtype=dtype, device=device, **kwargs)


# ======== plain-CFG solvers (eps_hat used for both the x0 estimate and the renoising) ========
@register_solver("ddim")
class BaseDDIM(SDXL):
    """reference: latent_sdxl.py:425-467."""
    cfgpp = False

    def reverse_process(self, null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape=(1024, 1024),
                        callback_fn=None, **kwargs):
        return self._ddim_xl(null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape, self.cfgpp,
                             callback_fn, wrap=True, zt=kwargs.get("latents"), seeds=kwargs.get("seeds"))


@register_solver("euler")
class Euler(SDXL):
    """Karras Euler, VE casted (reference: latent_sdxl.py:469-517)."""
    quantize = True
    variant = 0

    def reverse_process(self, null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape=(1024, 1024),
                        callback_fn=None, **kwargs):
        sigmas = self.tables.karras_sigmas()
        return self._kdiff_xl(null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape, callback_fn,
                              sigmas=sigmas, n_steps=len(self.scheduler.timesteps), x0_scale_mode="kdiff", input_mode=0,
                              xl_form=False, variant=self.variant, solver="euler", t_of_sigma=self.timestep, ret="den",
                              seeds=kwargs.get("seeds"))


class _LightningMixin:
    def reverse_process(self, null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape=(1024, 1024),
                        callback_fn=None, **kwargs):
        assert cfg_guidance == 1.0, "CFG should be turned off in the lightning version"
        return super().reverse_process(null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape,
                                       callback_fn, **kwargs)


@register_solver("ddim_lightning")
class BaseDDIMLight(_LightningMixin, BaseDDIM, SDXLLightning):
    """reference: latent_sdxl.py:519-539."""

    def __init__(self, **kwargs):
        SDXLLightning.__init__(self, **kwargs)


@register_solver("euler_lightning")
class EulerLight(_LightningMixin, Euler, SDXLLightning):
    """reference: latent_sdxl.py:541-567."""

    def __init__(self, **kwargs):
        SDXLLightning.__init__(self, **kwargs)


@register_solver("ddim_edit")
class EditWardSwapDDIM(BaseDDIM):
    """Invert with the source prompt, regenerate with the target (reference: latent_sdxl.py:569-706)."""

    @torch.no_grad()
    def sample(self, prompt1=["", "", ""], prompt2=["", "", ""], cfg_guidance: float = 5.0,
               original_size: Optional[Tuple[int, int]] = None, crops_coords_top_left: Tuple[int, int] = (0, 0),
               target_size: Optional[Tuple[int, int]] = None, negative_original_size: Optional[Tuple[int, int]] = None,
               negative_crops_coords_top_left: Tuple[int, int] = (0, 0),
               negative_target_size: Optional[Tuple[int, int]] = None, clip_skip: Optional[int] = None, **kwargs):
        original_size, target_size = self._sizes(original_size, target_size)
        pe = kwargs.pop("prompt_embeds", None)
        if pe is not None:      # (null, src, tgt, pool_null, pool_src, pool_tgt)
            null_e, src_e, tgt_e, pool_null, pool_src, pool_tgt = (x.to(self.work_device, torch.float16) for x in pe)
        else:
            null_e, src_e, pool_null, pool_src = self.get_text_embed(prompt1[0], prompt1[1], prompt2[0], prompt2[1], clip_skip)
            _, tgt_e, _, pool_tgt = self.get_text_embed(prompt1[0], prompt1[2], prompt2[0], prompt2[2], clip_skip)
        mk = lambda pool: self._cond_kwargs(pool, pool_null, cfg_guidance, original_size, crops_coords_top_left,  # noqa: E731
                                            target_size, negative_original_size, negative_crops_coords_top_left,
                                            negative_target_size, src_e.dtype)
        add_src, add_tgt = mk(pool_src), mk(pool_tgt)
        self._batch = int(src_e.shape[0])
        ret_lat = kwargs.pop("return_latents", False)
        zt = self.reverse_process(null_e, src_e, tgt_e, cfg_guidance, add_src, add_tgt, **kwargs)
        if ret_lat:
            return zt
        return self._finish(zt)

    def reverse_process(self, null_prompt_embeds, src_prompt_embeds, tgt_prompt_embed, cfg_guidance,
                        add_src_cond_kwargs, add_tgt_cond_kwargs, callback_fn=None, **kwargs):
        zt = self.initialize_latent(method="ddim", src_img=kwargs.get("src_img", None), src_latent=kwargs.get("src_latent"),
                                    uc=null_prompt_embeds, c=src_prompt_embeds, cfg_guidance=cfg_guidance,
                                    add_cond_kwargs=add_src_cond_kwargs)
        # forward loop with the TARGET prompt; guarded alpha() here (latent_sdxl.py:679-703), unlike `ddim`
        return self._ddim_xl(null_prompt_embeds, tgt_prompt_embed, cfg_guidance, add_tgt_cond_kwargs, None, self.cfgpp,
                             callback_fn, wrap=False, zt=zt)


# ======== CFG++ solvers (renoise with eps_uc; small-lambda regime) ========
@register_solver("ddim_cfg++")
class BaseDDIMCFGpp(BaseDDIM):
    """reference: latent_sdxl.py:713-755."""
    cfgpp = True


@register_solver("euler_cfg++")
class EulerCFGpp(SDXL):
    """Euler CFG++ on the DDIM-timestep sigmas (reference: latent_sdxl.py:757-808)."""
    quantize = True

    def reverse_process(self, null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape=(1024, 1024),
                        callback_fn=None, **kwargs):
        total_sigmas = (1 - self.total_alphas).sqrt() / self.total_alphas.sqrt()
        sigmas = total_sigmas[torch.round(self.scheduler.timesteps.float()).int()]
        sigmas = torch.cat([sigmas, torch.tensor([0.0])])
        return self._kdiff_xl(null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape, callback_fn,
                              sigmas=sigmas, n_steps=len(self.scheduler.timesteps), x0_scale_mode="kdiff", input_mode=0,
                              xl_form=False, variant=1, solver="euler", t_of_sigma=self.timestep, ret="den",
                              seeds=kwargs.get("seeds"))


@register_solver("euler_cfg++_lightning")
class EulerCFGppLight(_LightningMixin, EulerCFGpp, SDXLLightning):
    """reference: latent_sdxl.py:810-836."""

    def __init__(self, **kwargs):
        SDXLLightning.__init__(self, **kwargs)


@register_solver("ddim_cfg++_lightning")
class BaseDDIMCFGppLight(_LightningMixin, BaseDDIMCFGpp, SDXLLightning):
    """reference: latent_sdxl.py:838-858."""

    def __init__(self, **kwargs):
        SDXLLightning.__init__(self, **kwargs)


@register_solver("dpm++_2m_cfgpp")
class DPMpp2mCFGppSolver(SDXL):
    """DPM-Solver++(2M) CFG++, SDXL flavour: sigmas from the DDIM timesteps, NFE-1 UNet calls,
    c_in = sqrt(abar), c_out = -sigma, (uncond_denoised - old) in the 2nd-order term, returns x
    (reference: latent_sdxl.py:860-930)."""
    quantize = True

    def reverse_process(self, null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape=(1024, 1024),
                        callback_fn=None, **kwargs):
        alphas = self.scheduler.alphas_cumprod[self.scheduler.timesteps.int()]
        sigmas = (1 - alphas).sqrt() / alphas.sqrt()
        self._alphas_2m = alphas
        return self._kdiff_xl(null_prompt_embeds, prompt_embeds, cfg_guidance, add_cond_kwargs, shape, callback_fn,
                              sigmas=sigmas, n_steps=len(self.scheduler.timesteps) - 1, x0_scale_mode="mul", input_mode=1,
                              xl_form=True, variant=2, solver="dpm2m", t_of_sigma=self.sigma_to_t, ret="x",
                              seeds=kwargs.get("seeds"))


@register_solver("dpm++_2m_cfgpp_lightning")
class DPMpp2mCFGppLightningSolver(_LightningMixin, DPMpp2mCFGppSolver, SDXLLightning):
    """reference: latent_sdxl.py:932-952."""

    def __init__(self, **kwargs):
        SDXLLightning.__init__(self, **kwargs)


@register_solver("ddim_edit_cfg++")
class EditWardSwapDDIMCFGpp(EditWardSwapDDIM):
    """CFG++ inversion + CFG++ regeneration (reference: latent_sdxl.py:954-1025)."""
    cfgpp = True

    @torch.no_grad()
    def inversion(self, z0, uc, c, cfg_guidance, add_cond_kwargs):
        return self._inversion_xl(z0, uc, c, cfg_guidance, add_cond_kwargs, cfgpp=True)


# API symmetry with SD1.5 (BASELINE.json config 5 names "SDXL ddim_inversion_cfg++"): the reference
# registers no such SDXL solver; invert + reconstruct is `ddim_edit_cfg++` with target == source.
@register_solver("ddim_inversion_cfg++")
class InversionDDIMCFGpp(EditWardSwapDDIMCFGpp):
    @torch.no_grad()
    def sample(self, prompt1=["", ""], prompt2=["", ""], **kwargs):
        p1 = list(prompt1) + [prompt1[1]] if len(prompt1) == 2 else prompt1
        p2 = list(prompt2) + [prompt2[1]] if len(prompt2) == 2 else prompt2
        return super().sample(prompt1=p1, prompt2=p2, **kwargs)


if __name__ == "__main__":
    print(f"Possble solvers: {[x for x in __SOLVER__.keys()]}")
