"""SD1.5 solvers on the MI355X-native engine - same registry names, ``Solver``
class API, scheduler-index conventions and callback protocol as the reference's
``latent_diffusion.py`` (registry :13-26, wrapper :54-241, solvers :247-1010),
re-built so that the loop body is two asynchronous launches: the HIP UNet
(``engine.predict``) and one fused step kernel.  The host only walks
pre-computed fp32 coefficient tables; it never synchronises with the device
inside the loop unless a callback asks for tensors.

Extension beyond the reference: batches.  ``prompt=[null,
[text_1..text_B]]`` (or ``prompt_embeds=(uc, c)``) with ``seeds=[s_1..s_B]`` runs
B independent chains; chain b equals the reference run with ``set_seed(s_b)``.
"""
from __future__ import annotations

import os
import sys
from typing import Any, Callable, Dict, List, Optional

import torch

from . import coeffs as K
from ._lib import CfgppError
from .conditioning import SyntheticTextEncoder, as_list
from .registry import Registry
from .schedule import SchedulerTables, get_ancestral_step, get_sigmas_karras  # noqa: F401
from .unet_config import SD15, UNetConfig

# ---- solver registry (names: Appendix A of SURVEY.md) ----
__SOLVER__ = Registry("Solver")
register_solver = __SOLVER__.register        # @register_solver(name)
get_solver = __SOLVER__.create               # get_solver(name, solver_config=..., device=..., **kw)


class _SchedulerView:
    """The two attributes of ``self.scheduler`` the reference's solvers touch."""

    def __init__(self, tables: SchedulerTables):
        self.timesteps = tables.timesteps
        self.alphas_cumprod = tables.alphas_cumprod
        self.final_alpha_cumprod = tables.final_alpha_cumprod


def _progress(it, desc):
    try:
        from tqdm import tqdm
        return tqdm(it, desc=desc, leave=False, disable=None)
    except Exception:  # noqa: BLE001
        return it


class StableDiffusion:
    """Model wrapper (reference: latent_diffusion.py:54-241)."""

    unet_config: UNetConfig = SD15
    scheduler_kind = "ddim"
    latent_scale = 8

    def __init__(self, solver_config, model_key: str = "runwayml/stable-diffusion-v1-5",
                 device: Optional[torch.device] = None, **kwargs):
        self.device = device
        self.model_key = model_key
        self.dtype = kwargs.get("pipe_dtype", torch.float16)
        # how a 0-dim fp32 scalar written first in `s * fp16_tensor` enters the product (coeffs.py): the
        # reference RUNS on the GPU, where torch keeps it fp32 ("cuda"); the golden vectors were recorded on
        # torch-CPU, which rounds it to fp16 first ("cpu": the default only when a test injects an engine).
        self.scalar_semantics = kwargs.get("scalar_semantics", "cpu" if kwargs.get("engine") is not None else "cuda")
        cfg = kwargs.get("unet_config", self.unet_config)
        self.cfg = cfg
        self.latent_hw = tuple(kwargs.get("latent_hw", (cfg.sample_size, cfg.sample_size)))
        self.max_batch = int(kwargs.get("max_batch", 1))

        # scheduler tables (host)
        self.tables = SchedulerTables(solver_config.num_sampling, self.scheduler_kind)
        self.scheduler = _SchedulerView(self.tables)
        self.total_alphas = self.tables.total_alphas
        self.sigmas = self.tables.sigmas
        self.log_sigmas = self.tables.log_sigmas
        self.skip = self.tables.skip
        self.final_alpha_cumprod = self.tables.final_alpha_cumprod

        # engine: the HIP UNet + fused step kernels.  No fallback.
        engine = kwargs.get("engine")
        if engine is None:
            from .hip_engine import HipEngine
            engine = HipEngine(cfg, max_batch=self.max_batch, latent_hw=self.latent_hw, device=device,
                               weights=kwargs.get("unet_weights", "synthetic"), weight_seed=kwargs.get("weight_seed", 0))
        self.engine = engine
        self.unet = engine
        self.work_device = getattr(engine, "device", torch.device("cpu"))

        # boundary components (off the per-step path)
        self.text_encoder = kwargs.get("text_encoder") or SyntheticTextEncoder(cfg.cross_attention_dim, None)
        self.vae = kwargs.get("vae")
        self._vae_kwargs = dict(seed=kwargs.get("vae_seed", 0))
        vw = kwargs.get("vae_weights")                  # diffusers AutoencoderKL state dict, or a safetensors path
        if vw is not None:
            if isinstance(vw, str):
                from .weights import load_safetensors_iter
                vw = dict(load_safetensors_iter(vw))
            self._vae_kwargs["state_dict"] = vw

    # ------------------------------------------------------------------ reference API
    def __call__(self, *args: Any, **kwargs: Any) -> Any:
        self.sample(*args, **kwargs)        # return value dropped, as in the reference (quirk Q4)

    def sample(self, *args: Any, **kwargs: Any) -> Any:
        raise NotImplementedError("Solver must implement sample() method.")

    def alpha(self, t):
        return self.tables.alpha(t)

    @torch.no_grad()
    def get_text_embed(self, null_prompt, prompt):
        """-> (null_text_embed [1 or B,77,D], text_embed [B,77,D]), fp16."""
        uc, _ = self.text_encoder(as_list(null_prompt))
        c, _ = self.text_encoder(as_list(prompt))
        return uc.to(self.work_device), c.to(self.work_device)

    def _get_vae(self):
        if self.vae is None:
            if self.work_device.type != "cuda":      # CPU is only reachable with an injected (test) engine: inject the VAE too
                raise CfgppError("decode/encode need the HIP VAE (ROCm GPU); on CPU pass vae=<object with decode/encode>")
            from .vae import HipVAE                  # decoder + encoder on the HIP kernels; no fallback
            self.vae = HipVAE(self.cfg.vae_scale, self.latent_hw, max_batch=self.max_batch, device=self.work_device,
                              **self._vae_kwargs)
        return self.vae

    def encode(self, x):
        """xt -> zt (posterior sample * scale; latent_diffusion.py:117-121).  The reference's VAE runs in
        ``pipe_dtype`` (fp16), so the latent it returns - and with it the whole inversion / edit chain - is
        fp16; the HIP encoder's fp32 posterior sample is rounded to that dtype here."""
        return self._get_vae().encode(x.to(self.work_device)).to(self.dtype)

    def decode(self, zt):
        """zt -> xt (latent_diffusion.py:123-129)."""
        return self._get_vae().decode(zt.to(self.work_device))

    def predict_noise(self, zt: torch.Tensor, t, uc: Optional[torch.Tensor], c: Optional[torch.Tensor]):
        """epsilon_theta for null and condition (latent_diffusion.py:131-158).  One
        UNet launch over rows [uc_1..uc_B, c_1..c_B]; ``zt`` is read twice by index."""
        self._ensure_context(uc, c)
        noise_uc, noise_c = self.engine.predict(zt, float(t))
        if uc is None:
            return noise_c, noise_c
        if c is None:
            return noise_uc, noise_uc
        return noise_uc, noise_c

    def _ensure_context(self, uc, c):
        """(re)build the engine's conditioning when the embeddings changed (once per sampling loop)"""
        if uc is None and c is None:
            raise ValueError("predict_noise needs at least one of uc / c")
        a = c if uc is None else uc
        b = uc if c is None else c
        key = (a.data_ptr(), b.data_ptr(), tuple(a.shape), tuple(b.shape), a._version, b._version)
        if getattr(self, "_ctx_key", None) != key:
            self._set_context(a, b)
            self._ctx_key = key
            self._ctx_keep = (a, b)

    # ------------------------------------------------------------------ whole-loop graph replay
    def _graph_loop(self, zt, ts, uc, c, lam, tweedie_uc, renoise_uc, coeff_of):
        """The DDIM loops with callback_fn None as hipGraph replays of ONE captured step (UNet + fused update; include/cfgpp.h:
        cfgpp_sample_graph_ddim) when the engine offers it ($CFGPP_GRAPH=1 on the HIP engine): the per-step scalars - exactly
        what the eager loop hands to ``predict`` / ``step_ddim`` - are computed up front.  ``coeff_of(t) -> (sqrt4, device_alpha)``.
        Returns (z0t, zt) or None when the eager loop has to run (mock engine, switch off)."""
        if not getattr(self.engine, "graph_enabled", False):
            return None
        z_half = zt.dtype == torch.float16
        steps = []
        for t in ts:
            sqrt4, dev_a = coeff_of(t)
            co = K.ddim_coeffs_pinned(sqrt4, eps_half=True, semantics=self.scalar_semantics, z_half=z_half, device_alpha=dev_a)
            steps.append((float(t), *[float(v) for v in co]))
        single = "c" if uc is None else ("uc" if c is None else "")
        return self.engine.ddim_loop_graph(zt, steps, lam, tweedie_uc, renoise_uc, single)

    def _set_context(self, uc, c):
        self.engine.set_context(uc, c)

    # ------------------------------------------------------------------ latents
    def initialize_latent(self, method: str = "random", src_img: Optional[torch.Tensor] = None, **kwargs):
        if method == "ddim":
            z = self.inversion(self.encode(src_img.to(self.dtype)), kwargs.get("uc"), kwargs.get("c"),
                               cfg_guidance=kwargs.get("cfg_guidance", 0.0))
        elif method == "npi":
            z = self.inversion(self.encode(src_img.to(self.dtype)), kwargs.get("c"), kwargs.get("c"), cfg_guidance=1.0)
        elif method in ("random", "random_kdiffusion"):
            size = tuple(kwargs.get("latent_dim", (1, self.cfg.in_channels) + self.latent_hw))
            z = self._randn(size, kwargs.get("seeds"))
            if method == "random_kdiffusion":
                sigmas = kwargs.get("sigmas", [14.6146])
                z = z * (sigmas[0] ** 2 + 1) ** 0.5
            z = z.to(self.work_device)
        else:
            raise NotImplementedError
        return z

    @staticmethod
    def _randn(size, seeds=None) -> torch.Tensor:
        """CPU-generator noise, as the reference draws it (latent_diffusion.py:200).
        With ``seeds`` chain b is ``torch.manual_seed(s_b); torch.randn(1, ...)``."""
        if seeds is None:
            return torch.randn(size)
        if len(seeds) != size[0]:
            raise ValueError(f"{len(seeds)} seeds for a batch of {size[0]}")
        out = []
        for s in seeds:
            g = torch.Generator().manual_seed(int(s))
            out.append(torch.randn((1,) + tuple(size[1:]), generator=g))
        return torch.cat(out, dim=0)

    # ------------------------------------------------------------------ k-diffusion helpers
    def timestep(self, sigma):
        return self.tables.timestep(sigma)

    def to_d(self, x, sigma, denoised):
        return (x - denoised) / sigma.item()

    get_ancestral_step = staticmethod(get_ancestral_step)

    def calculate_input(self, x, sigma):
        return x / (sigma ** 2 + 1) ** 0.5

    def calculate_denoised(self, x, model_pred, sigma):
        return x - model_pred * sigma

    # ------------------------------------------------------------------ fused loops
    def _ddim_update(self, zt, z0t, noise_uc, noise_c, lam, sqrt4, tweedie_uc, renoise_uc, device_alpha=None):
        co = K.ddim_coeffs_pinned(sqrt4, eps_half=(noise_uc.dtype == torch.float16), semantics=self.scalar_semantics,
                                  z_half=(zt.dtype == torch.float16), device_alpha=device_alpha)
        self.engine.step_ddim(zt, z0t, noise_uc, noise_c, lam, co, tweedie_uc, renoise_uc)

    def _own_latent(self, z):
        """private, contiguous copy on the engine device (the loops update it in place); fp16 stays fp16
        (the reference's inversion / edit latents), everything else runs as fp32 like ``torch.randn``."""
        dt = torch.float16 if z.dtype == torch.float16 else torch.float32
        return z.detach().to(device=self.work_device, dtype=dt, copy=True).contiguous()

    def _run_callback(self, callback_fn, step, t, z0t, zt):
        kw = callback_fn(step, t, {"z0t": z0t.detach(), "zt": zt.detach(), "decode": self.decode})
        if kw["z0t"] is not z0t:
            z0t.copy_(kw["z0t"])
        if kw["zt"] is not zt:
            zt.copy_(kw["zt"])

    def _ddim_forward(self, zt, uc, c, cfg_guidance, cfgpp: bool, callback_fn=None, desc="SD", wrap_index=False):
        """DDIM / DDIM-CFG++ reverse loop (latent_diffusion.py:272-294 / 652-674).
        ``zt`` [B,4,H,W] on the engine device: fp32 (text-to-image: ``torch.randn``) or fp16 (after an
        inversion that started from the fp16 VAE latent); a private copy is updated in place; returns (z0t, zt)."""
        zt = self._own_latent(zt)
        ts = self.scheduler.timesteps
        ts = ts.int() if wrap_index else ts
        if callback_fn is None:
            self._ensure_context(uc, c)
            done = self._graph_loop(zt, ts, uc, c, cfg_guidance, False, cfgpp, lambda t: (
                self.tables.ddim_sqrt_coeffs(t, wrap=wrap_index), "rn" if (not wrap_index and int(t) - self.tables.skip < 0) else None))
            if done is not None:
                return done
        z0t = torch.empty_like(zt)
        for step, t in enumerate(_progress(ts, desc)):
            # at = alpha(t), at_prev = alpha(t - skip); SDXL loops index the shifted table
            # unguarded (quirk Q3, wrap_index).  sqrt(at) etc. come from the pinned tables.
            sqrt4 = self.tables.ddim_sqrt_coeffs(t, wrap=wrap_index)
            noise_uc, noise_c = self.predict_noise(zt, t, uc, c)
            # alpha(t - skip) of the last step is `final_alpha_cumprod.to(device)`: a DEVICE scalar (coeffs.py)
            dev_a = "rn" if (not wrap_index and int(t) - self.tables.skip < 0) else None
            self._ddim_update(zt, z0t, noise_uc, noise_c, cfg_guidance, sqrt4, False, cfgpp, dev_a)
            if callback_fn is not None:
                self._run_callback(callback_fn, step, t, z0t, zt)
        return z0t, zt

    def _ddim_inversion(self, z0, uc, c, cfg_guidance, cfgpp: bool):
        """DDIM inversion (latent_diffusion.py:160-182 CFG, 888-910 CFG++)."""
        zt = self._own_latent(z0)
        self._ensure_context(uc, c)
        done = self._graph_loop(zt, list(reversed(self.scheduler.timesteps)), uc, c, cfg_guidance, cfgpp, False, lambda t: (
            self.tables.ddim_sqrt_coeffs(t, inversion=True), "tw" if int(t) - self.tables.skip < 0 else None))
        if done is not None:
            return done[1]
        z0t = torch.empty_like(zt)
        for t in _progress(reversed(self.scheduler.timesteps), "DDIM Inversion"):
            sqrt4 = self.tables.ddim_sqrt_coeffs(t, inversion=True)     # a_tw = alpha(t-skip), a_rn = alpha(t)
            noise_uc, noise_c = self.predict_noise(zt, t, uc, c)
            dev_a = "tw" if int(t) - self.tables.skip < 0 else None
            self._ddim_update(zt, z0t, noise_uc, noise_c, cfg_guidance, sqrt4, cfgpp, False, dev_a)
        return zt

    @torch.no_grad()
    def inversion(self, z0, uc, c, cfg_guidance: float = 1.0):
        return self._ddim_inversion(z0, uc, c, cfg_guidance, cfgpp=False)

    def _finish(self, z):
        if os.environ.get("CFGPP_TRACE"):
            torch.cuda.synchronize() if z.is_cuda else None
            print("[cfgpp] sampling loop done, decoding", file=sys.stderr, flush=True)
        vae = self._get_vae()
        if hasattr(vae, "decode_image"):        # HIP VAE: `/ 2 + 0.5` and the clamp are folded into its last kernel
            img = vae.decode_image(z.to(self.work_device))
        else:
            img = (self.decode(z) / 2 + 0.5).clamp(0, 1)
        return img.detach().cpu()

    def _embeds(self, prompt, kwargs, n_cond=1):
        """(uc, c_1, ..) either from ``prompt_embeds=`` or from the text encoder."""
        pe = kwargs.get("prompt_embeds")
        if pe is not None:
            return tuple(e.to(self.work_device, torch.float16) for e in pe)
        out = []
        uc = None
        for k in range(n_cond):
            uc, c = self.get_text_embed(null_prompt=prompt[0], prompt=prompt[1 + k])
            out.append(c)
        return (uc, *out)

    def _batch_of(self, c, kwargs):
        return int(c.shape[0])

    # ------------------------------------------------------------------ k-diffusion loop
    def _kdiff_loop(self, uc, c, cfg_guidance, variant: int, solver: str, callback_fn=None, seeds=None):
        """Euler / DPM++2M (CFG: variant 0, CFG++: variant 1) on Karras sigmas, fp16 latent
        (latent_diffusion.py:302-346, 454-503, 682-723, 830-879)."""
        B = self._batch_of(c, None)
        sigmas = self.tables.karras_sigmas()
        x = self.initialize_latent(method="random_kdiffusion", latent_dim=(B, self.cfg.in_channels) + self.latent_hw,
                                   sigmas=sigmas, seeds=seeds).to(torch.float16).contiguous()
        xc = torch.empty_like(x)
        den = torch.empty_like(x)
        old = torch.empty_like(x)
        have_old = False
        n = len(self.scheduler.timesteps)
        for i in _progress(range(n), "SD"):
            sigma = sigmas[i]
            new_t = self.timestep(sigma)
            self.engine.kdiff_input(x, xc, K.kdiff_input_scale_sd(sigma, self.scalar_semantics), 0)
            noise_uc, noise_c = self.predict_noise(xc, new_t, uc, c)
            first = (solver == "euler") or (not have_old)
            coef, euler = K.kdiff_coeffs(cfg_guidance, sigmas, i, first, xl_form=False, semantics=self.scalar_semantics)
            self.engine.step_kdiff(x, den, old, noise_uc, noise_c, coef, variant, False, euler, solver != "euler")
            have_old = True
            if callback_fn is not None:
                self._run_callback(callback_fn, i, new_t, den, x)
        return den, x


    # ------------------------------------------------------------------ ancestral k-diffusion loops
    def _ancestral_loop(self, uc, c, cfg_guidance, cfgpp: bool, two_stage: bool, callback_fn=None, seeds=None):
        """Euler-ancestral (latent_diffusion.py:349-390 / 726-766) and DPM-Solver++(2S)-ancestral
        (:393-451 / 769-827; two UNet calls per step) on Karras sigmas, fp16 latent.  The injected noise
        is drawn with ``engine.randn_like`` (device RNG on the GPU, like the reference's torch.randn_like)."""
        B = self._batch_of(c, None)
        lam = cfg_guidance
        sigmas = self.tables.karras_sigmas()
        x = self.initialize_latent(method="random_kdiffusion", latent_dim=(B, self.cfg.in_channels) + self.latent_hw,
                                   sigmas=sigmas, seeds=seeds).to(torch.float16).contiguous()
        xc, den, uden = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        x2, den2, uden2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        variant = 1 if cfgpp else 0
        sem = self.scalar_semantics
        t_fn = lambda sg: sg.log().neg()       # noqa: E731
        sigma_fn = lambda tt: tt.neg().exp()   # noqa: E731
        first = lambda sc: K._first(K._s(sc), True, sem)   # noqa: E731  scalar written first in `s * fp16`
        n = len(self.scheduler.timesteps)
        for i in _progress(range(n), "SD"):
            sigma = sigmas[i]
            new_t = self.timestep(sigma)
            sigma_down, sigma_up = get_ancestral_step(sigmas[i], sigmas[i + 1])
            self.engine.kdiff_input(x, xc, K.kdiff_input_scale_sd(sigma, self.scalar_semantics), 0)
            noise_uc, noise_c = self.predict_noise(xc, new_t, uc, c)
            if (not two_stage) or float(sigma_down) == 0.0:
                # Euler step down to sigma_down: x = den + ((x - d_from)/sigma) * sigma_down
                # (`/ sigma.item()` in to_d, latent_diffusion.py:216-218, is a python-number divisor: K.divisor applies the backend rule)
                coef = [float(lam), float(sigma), 0.0, K.divisor(sigma.item(), sem), float(sigma_down), 0.0, 0.0, 1.0, 0.0]
                self.engine.step_kdiff(x, den, None, noise_uc, noise_c, coef, variant, False, True, False)
            else:
                self.engine.kdiff_denoise(x, noise_uc, noise_c, lam, float(sigma), den, uden)
                t, t_next = t_fn(sigmas[i]), t_fn(K._s(sigma_down))
                r = 1 / 2
                h = t_next - t
                s_mid = t + r * h
                # x_2 = (sigma(s)/sigma(t)) * x - expm1(-h r) * (den | uden)
                self.engine.lincomb(x2, x, uden if cfgpp else den, None, first(sigma_fn(s_mid) / sigma_fn(t)),
                                    first((-h * r).expm1()), 0)
                sigma_s = sigma_fn(s_mid)
                t_2 = self.timestep(sigma_s)
                self.engine.kdiff_input(x2, xc, K.kdiff_input_scale_sd(sigma_s, self.scalar_semantics), 0)
                nuc2, nc2 = self.predict_noise(xc, t_2, uc, c)
                self.engine.kdiff_denoise(x2, nuc2, nc2, lam, float(sigma_s), den2, uden2)
                ratio_n = first(sigma_fn(t_next) / sigma_fn(t))
                if cfgpp:   # x = den_2 - exp(-h) * uden_2 + ratio * x
                    self.engine.lincomb(x, x, den2, uden2, ratio_n, first(torch.exp(-h)), 1)
                else:       # x = ratio * x - expm1(-h) * den_2
                    self.engine.lincomb(x, x, den2, None, ratio_n, first((-h).expm1()), 0)
            if sigmas[i + 1] > 0:
                self.engine.lincomb(x, x, self.engine.randn_like(x), None, float(sigma_up), 0.0, 2)
            if callback_fn is not None:
                self._run_callback(callback_fn, i, new_t, den, x)
        return den, x


# ======== plain-CFG solvers (eps_hat used for both the x0 estimate and the renoising) ========
def _sd_prompts(prompt):
    return prompt


@register_solver("ddim")
class BaseDDIM(StableDiffusion):
    """Basic DDIM solver for SD (reference: latent_diffusion.py:247-299)."""
    cfgpp = False

    @torch.no_grad()
    def sample(self, cfg_guidance=7.5, prompt=["", ""], callback_fn=None, **kwargs):
        uc, c = self._embeds(prompt, kwargs)
        B = int(c.shape[0])
        zt = kwargs.get("latents")
        if zt is None:
            zt = self.initialize_latent(latent_dim=(B, self.cfg.in_channels) + self.latent_hw, seeds=kwargs.get("seeds"))
        z0t, zt = self._ddim_forward(zt.to(self.work_device), uc, c, cfg_guidance, self.cfgpp, callback_fn)
        if kwargs.get("return_latents"):
            return z0t, zt
        return self._finish(z0t)


@register_solver("euler")
class EulerCFGSolver(StableDiffusion):
    """Karras Euler, VE casted (reference: latent_diffusion.py:302-346)."""
    variant, solver = 0, "euler"

    @torch.no_grad()
    def sample(self, cfg_guidance, prompt=["", ""], callback_fn=None, **kwargs):
        uc, c = self._embeds(prompt, kwargs)
        den, x = self._kdiff_loop(uc, c, cfg_guidance, self.variant, self.solver, callback_fn, kwargs.get("seeds"))
        out = den if self.solver == "euler" else x      # Euler decodes `denoised`, 2M decodes `x`
        if kwargs.get("return_latents"):
            return den, x
        return self._finish(out)


@register_solver("euler_a")
class EulerAncestralCFGSolver(StableDiffusion):
    """Karras Euler + ancestral sampling (reference: latent_diffusion.py:349-390)."""
    cfgpp, two_stage = False, False

    @torch.no_grad()
    def sample(self, cfg_guidance, prompt=["", ""], callback_fn=None, **kwargs):
        uc, c = self._embeds(prompt, kwargs)
        den, x = self._ancestral_loop(uc, c, cfg_guidance, self.cfgpp, self.two_stage, callback_fn, kwargs.get("seeds"))
        if kwargs.get("return_latents"):
            return den, x
        return self._finish(x if self.two_stage else den)     # Euler-a decodes `denoised`, 2S-a decodes `x`


@register_solver("dpm++_2s_a")
class DPMpp2sAncestralCFGSolver(EulerAncestralCFGSolver):
    """DPM-Solver++(2S) ancestral, two UNet calls per step (reference: latent_diffusion.py:393-451)."""
    cfgpp, two_stage = False, True


@register_solver("dpm++_2m")
class DPMpp2mCFGSolver(EulerCFGSolver):
    """DPM-Solver++(2M) with CFG (reference: latent_diffusion.py:454-503)."""
    variant, solver = 0, "dpm2m"


@register_solver("ddim_inversion")
class InversionDDIM(BaseDDIM):
    """Invert with CFG then reconstruct (reference: latent_diffusion.py:506-558)."""

    def _invert(self, src_img, uc, c, cfg_guidance, kwargs):
        z0 = kwargs.get("src_latent")
        if z0 is None:
            z0 = self.encode(src_img.to(self.dtype))
        return self.inversion(z0, uc, c, cfg_guidance=cfg_guidance)

    @torch.no_grad()
    def sample(self, src_img=None, cfg_guidance=7.5, prompt=["", "", ""], callback_fn=None, **kwargs):
        uc, c = self._embeds(prompt, kwargs)
        zt = self._invert(src_img, uc, c, cfg_guidance, kwargs)
        z0t, zt = self._ddim_forward(zt, uc, c, cfg_guidance, self.cfgpp, callback_fn)
        if kwargs.get("return_latents"):
            return z0t, zt
        return self._finish(z0t)


@register_solver("ddim_edit")
class EditWordSwapDDIM(InversionDDIM):
    """Invert with the source prompt, regenerate with the target prompt
    (reference: latent_diffusion.py:561-612)."""

    @torch.no_grad()
    def sample(self, src_img=None, cfg_guidance=7.5, prompt=["", "", ""], callback_fn=None, **kwargs):
        uc, src_c, tgt_c = self._embeds(prompt, kwargs, n_cond=2)
        zt = self._invert(src_img, uc, src_c, cfg_guidance, kwargs)
        z0t, zt = self._ddim_forward(zt, uc, tgt_c, cfg_guidance, self.cfgpp, callback_fn, desc="DDIM-edit")
        if kwargs.get("return_latents"):
            return z0t, zt
        return self._finish(z0t)


# ======== CFG++ solvers (renoise with eps_uc; small-lambda regime) ========
@register_solver("ddim_cfg++")
class BaseDDIMCFGpp(BaseDDIM):
    """DDIM with CFG++: renoise with eps_uc (reference: latent_diffusion.py:621-679)."""
    cfgpp = True


@register_solver("euler_cfg++")
class EulerCFGppSolver(EulerCFGSolver):
    """reference: latent_diffusion.py:682-723 (d from uncond_denoised)."""
    variant, solver = 1, "euler"


@register_solver("euler_a_cfg++")
class EulerAncestralCFGppSolver(EulerAncestralCFGSolver):
    """reference: latent_diffusion.py:726-766 (d from uncond_denoised)."""
    cfgpp, two_stage = True, False


@register_solver("dpm++_2s_a_cfg++")
class DPMpp2sAncestralCFGppSolver(EulerAncestralCFGSolver):
    """reference: latent_diffusion.py:769-827 (x_2 from uncond_denoised; x = D_2 - e^{-h} D_uc,2 + ratio x)."""
    cfgpp, two_stage = True, True


@register_solver("dpm++_2m_cfg++")
class DPMpp2mCFGppSolver(EulerCFGSolver):
    """DPM-Solver++(2M) with CFG++ (reference: latent_diffusion.py:830-879)."""
    variant, solver = 1, "dpm2m"


@register_solver("ddim_inversion_cfg++")
class InversionDDIMCFGpp(InversionDDIM):
    """CFG++ inversion (x0 from eps_uc, renoise eps_hat) + CFG++ reconstruction
    (reference: latent_diffusion.py:882-957)."""
    cfgpp = True

    @torch.no_grad()
    def inversion(self, z0, uc, c, cfg_guidance: float = 1.0):
        return self._ddim_inversion(z0, uc, c, cfg_guidance, cfgpp=True)


@register_solver("ddim_edit_cfg++")
class EditWordSwapDDIMCFGpp(EditWordSwapDDIM):
    """reference: latent_diffusion.py:959-1010."""
    cfgpp = True

    @torch.no_grad()
    def inversion(self, z0, uc, c, cfg_guidance: float = 1.0):
        return self._ddim_inversion(z0, uc, c, cfg_guidance, cfgpp=True)


if __name__ == "__main__":
    print(f"Possble solvers: {[x for x in __SOLVER__.keys()]}")
