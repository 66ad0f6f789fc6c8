#!/usr/bin/env python
"""Record golden vectors G1-G7 from the REFERENCE's own
sampler code, run here on CPU behind the stub environment in ``_stub_env.py``.

    python tests/golden/make_golden.py          # writes tests/golden/sampler_golden.npz (+ .json)

Only runs in the build container (needs /root/reference).  The outputs are
plain data: inputs (latents, eps the scripted UNet returned, coefficients) and
the reference's outputs (z0t / zt per step, final results).  Nothing of the
reference's source is stored.
"""
from __future__ import annotations

import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _stub_env as stub  # noqa: E402

stub.install()
warnings.filterwarnings("ignore")

import latent_diffusion as ref_sd  # noqa: E402  (the reference)
import latent_sdxl as ref_xl  # noqa: E402
from utils import callback_util as ref_cb  # noqa: E402

CROP = 8
NULL = "low quality,jpeg artifacts,blurry,poorly drawn,ugly,worst quality,"
PROMPT = "a photo of an astronaut riding a horse on mars"
PROMPT2 = "a photo of an astronaut riding a zebra on mars"

out: dict[str, np.ndarray] = {}
meta: dict = {}


def crop(x):
    return x.detach()[..., :CROP, :CROP].contiguous()


def npy(x):
    x = x.detach().cpu() if isinstance(x, torch.Tensor) else torch.as_tensor(x)
    if x.dtype == torch.bfloat16:
        x = x.float()
    return x.numpy()


class Recorder:
    """callback_fn that records (step, t, z0t, zt) crops."""

    def __init__(self):
        self.steps, self.ts, self.z0t, self.zt, self.dt = [], [], [], [], []

    def __call__(self, step, t, kw):
        self.steps.append(int(step))
        self.ts.append(float(t))
        self.z0t.append(npy(crop(kw["z0t"])))
        self.zt.append(npy(crop(kw["zt"])))
        self.dt.append((str(kw["z0t"].dtype), str(kw["zt"].dtype)))
        return kw

    def dump(self, prefix):
        out[prefix + "/cb_step"] = np.array(self.steps)
        out[prefix + "/cb_t"] = np.array(self.ts, dtype=np.float64)
        out[prefix + "/z0t"] = np.stack(self.z0t)
        out[prefix + "/zt"] = np.stack(self.zt)
        meta[prefix + "/dtypes"] = self.dt[0]


def hook_unet(unet):
    """Wrap the fake UNet so inputs / outputs are recorded (cropped)."""
    log = dict(z=[], t=[], eps=[], rows=[], te_rows=[])
    inner = unet.__call__

    def call(sample, timestep, encoder_hidden_states=None, added_cond_kwargs=None):
        r = inner(sample, timestep, encoder_hidden_states=encoder_hidden_states,
                  added_cond_kwargs=added_cond_kwargs)
        log["z"].append(npy(crop(sample)))
        log["t"].append(npy(timestep.float()))
        log["eps"].append(npy(crop(r["sample"])))
        log["rows"].append(int(sample.shape[0]))
        te = None if added_cond_kwargs is None else int(added_cond_kwargs["text_embeds"].shape[0])
        log["te_rows"].append(-1 if te is None else te)
        return r
    unet_cls = type(unet)
    # instance-level override of __call__ needs a subclass
    sub = type("HookedUNet", (unet_cls,), {"__call__": lambda self, *a, **k: call(*a, **k)})
    unet.__class__ = sub
    return log


def dump_unet_log(prefix, log):
    out[prefix + "/unet_z"] = np.stack(log["z"])          # [calls, rows, 4, 8, 8]
    out[prefix + "/unet_t"] = np.stack(log["t"])
    out[prefix + "/unet_eps"] = np.stack(log["eps"])
    out[prefix + "/unet_rows"] = np.array(log["rows"])
    out[prefix + "/unet_te_rows"] = np.array(log["te_rows"])


def cfg(n):
    return types.SimpleNamespace(num_sampling=n)


# ----------------------------------------------------------------------------
# G7 registry
# ----------------------------------------------------------------------------
meta["G7/sd_names"] = list(ref_sd.__SOLVER__.keys())
meta["G7/sdxl_names"] = list(ref_xl.__SOLVER__.keys())
try:
    ref_sd.get_solver("nope")
except ValueError as e:
    meta["G7/unknown_msg"] = str(e)
try:
    ref_sd.register_solver("ddim")(object)
except ValueError as e:
    meta["G7/dup_msg"] = str(e)

# ----------------------------------------------------------------------------
# G1 scheduler tables
# ----------------------------------------------------------------------------
for nfe in (50, 10):
    s = ref_sd.get_solver("ddim_cfg++", solver_config=cfg(nfe), device="cpu", pipe_dtype=torch.float16)
    ts = s.scheduler.timesteps
    out[f"G1/sd{nfe}/timesteps"] = npy(ts)
    out[f"G1/sd{nfe}/at"] = np.array([float(s.alpha(t)) for t in ts], dtype=np.float32)
    out[f"G1/sd{nfe}/at_prev"] = np.array([float(s.alpha(t - s.skip)) for t in ts], dtype=np.float32)
    meta[f"G1/sd{nfe}/skip"] = int(s.skip)
    if nfe == 50:
        out["G1/total_alphas"] = npy(s.total_alphas)
        out["G1/sigmas"] = npy(s.sigmas)
        out["G1/karras50"] = npy(ref_sd.get_sigmas_karras(50, s.sigmas.min(), s.sigmas.max(), rho=7.0))
        kar = ref_sd.get_sigmas_karras(50, s.sigmas.min(), s.sigmas.max(), rho=7.0)
        out["G1/karras50_timestep"] = np.array([int(s.timestep(kar[i])) for i in range(50)])

s = ref_xl.get_solver("ddim_cfg++", solver_config=cfg(50), device="cpu")
ts = s.scheduler.timesteps.int()
out["G1/sdxl50/timesteps"] = npy(ts)
out["G1/sdxl50/at"] = np.array([float(s.scheduler.alphas_cumprod[t]) for t in ts], dtype=np.float32)
out["G1/sdxl50/at_next"] = np.array([float(s.scheduler.alphas_cumprod[t - s.skip]) for t in ts], dtype=np.float32)
s = ref_xl.get_solver("ddim_cfg++_lightning", solver_config=cfg(4), device="cpu")
ts = s.scheduler.timesteps.int()
out["G1/light4/timesteps_f"] = npy(s.scheduler.timesteps)
out["G1/light4/timesteps"] = npy(ts)
out["G1/light4/at"] = np.array([float(s.scheduler.alphas_cumprod[t]) for t in ts], dtype=np.float32)
out["G1/light4/at_next"] = np.array([float(s.scheduler.alphas_cumprod[t - s.skip]) for t in ts], dtype=np.float32)
meta["G1/light4/skip"] = int(s.skip)

# ----------------------------------------------------------------------------
# SD1.5 trajectories (G2, G3, G4)
# ----------------------------------------------------------------------------


def run_sd(name, tag, nfe, lam, pipe_dtype=torch.float16, seed=42, src=False, prompts=None, **kw):
    torch.manual_seed(seed)
    np.random.seed(seed)
    s = ref_sd.get_solver(name, solver_config=cfg(nfe), device="cpu", pipe_dtype=pipe_dtype)
    log = hook_unet(s.unet)
    rec = Recorder()
    prompts = prompts or [NULL, PROMPT]
    args = dict(cfg_guidance=lam, prompt=prompts, callback_fn=rec)
    if src:
        g = torch.Generator().manual_seed(7)
        src_img = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
        args["src_img"] = src_img
    img = s.sample(**args)
    rec.dump(tag)
    dump_unet_log(tag, log)
    out[tag + "/img_crop"] = npy(img[..., :64, :64])
    meta[tag] = dict(name=name, nfe=nfe, lam=lam, seed=seed, eps_dtype=str(pipe_dtype), prompts=prompts,
                     src=src)
    return s


run_sd("ddim_cfg++", "G2/sd_ddim_cfgpp_h", 50, 0.6)
run_sd("ddim_cfg++", "G2/sd_ddim_cfgpp_f", 50, 0.6, pipe_dtype=torch.float32)
run_sd("ddim", "G2/sd_ddim_cfg_h", 10, 7.5)
run_sd("ddim_inversion_cfg++", "G3/sd_inv_cfgpp", 10, 0.6, src=True)
run_sd("ddim_inversion", "G3/sd_inv_cfg", 10, 2.0, src=True)
run_sd("ddim_edit_cfg++", "G3/sd_edit_cfgpp", 10, 0.6, src=True, prompts=[NULL, PROMPT, PROMPT2])
run_sd("dpm++_2m_cfg++", "G4/sd_dpm2m_cfgpp", 20, 0.6)
run_sd("dpm++_2m", "G4/sd_dpm2m_cfg", 10, 7.5)
run_sd("euler_cfg++", "G4/sd_euler_cfgpp", 10, 0.6)
run_sd("euler", "G4/sd_euler_cfg", 10, 7.5)
# ancestral samplers: the noise comes from the global CPU generator (torch.randn_like on CPU), drawn after z_T
run_sd("euler_a", "G4/sd_euler_a_cfg", 8, 7.5)
run_sd("euler_a_cfg++", "G4/sd_euler_a_cfgpp", 8, 0.6)
run_sd("dpm++_2s_a", "G4/sd_dpm2s_a_cfg", 8, 7.5)
run_sd("dpm++_2s_a_cfg++", "G4/sd_dpm2s_a_cfgpp", 8, 0.6)

# ----------------------------------------------------------------------------
# SDXL trajectories (G2-G5)
# ----------------------------------------------------------------------------


def run_xl(name, tag, nfe, lam, seed=42, src=False, prompts=None):
    torch.manual_seed(seed)
    np.random.seed(seed)
    s = ref_xl.get_solver(name, solver_config=cfg(nfe), device="cpu")
    log = hook_unet(s.unet)
    rec = Recorder()
    prompts = prompts or [NULL, PROMPT]
    args = dict(prompt1=prompts, prompt2=prompts, cfg_guidance=lam, target_size=(64, 64),
                original_size=(64, 64), callback_fn=rec)
    if src:
        g = torch.Generator().manual_seed(7)
        args["src_img"] = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
    img = s.sample(**args)
    rec.dump(tag)
    dump_unet_log(tag, log)
    out[tag + "/img"] = npy(img)
    meta[tag] = dict(name=name, nfe=nfe, lam=lam, seed=seed, prompts=prompts, src=src)
    return s


run_xl("ddim_cfg++", "G2/xl_ddim_cfgpp", 50, 0.6)
run_xl("ddim", "G2/xl_ddim_cfg", 10, 5.0)
run_xl("ddim_cfg++_lightning", "G2/xl_light_ddim_cfgpp", 4, 1.0)
run_xl("ddim_edit_cfg++", "G3/xl_edit_cfgpp", 10, 0.6, src=True, prompts=[NULL, PROMPT, PROMPT2])
run_xl("ddim_edit_cfg++", "G3/xl_edit_cfgpp_recon", 10, 0.6, src=True, prompts=[NULL, PROMPT, PROMPT])
run_xl("ddim_edit", "G3/xl_edit_cfg", 10, 3.0, src=True, prompts=[NULL, PROMPT, PROMPT2])
run_xl("dpm++_2m_cfgpp", "G4/xl_dpm2m_cfgpp", 20, 0.6)
run_xl("dpm++_2m_cfgpp_lightning", "G4/xl_light_dpm2m_cfgpp", 4, 1.0)
run_xl("euler_cfg++", "G4/xl_euler_cfgpp", 10, 0.6)

# G5: conditioning as seen by the UNet
for lam, tag in ((0.6, "G5/xl_cond_l06"), (1.0, "G5/xl_cond_l10")):
    seen = {}
    s = ref_xl.get_solver("ddim_cfg++", solver_config=cfg(2), device="cpu")
    orig = s.reverse_process

    def spy(null_embeds, embeds, cfg_guidance, add_cond_kwargs, shape, **kw):
        seen["text_embeds"] = add_cond_kwargs["text_embeds"].clone()
        seen["time_ids"] = add_cond_kwargs["time_ids"].clone()
        seen["null"] = null_embeds.clone()
        seen["emb"] = embeds.clone()
        return orig(null_embeds, embeds, cfg_guidance, add_cond_kwargs, shape, **kw)
    s.reverse_process = spy
    torch.manual_seed(0)
    s.sample(prompt1=[NULL, PROMPT], prompt2=[NULL, PROMPT], cfg_guidance=lam, target_size=(1024, 1024)
             if False else (64, 64))
    out[tag + "/text_embeds"] = npy(seen["text_embeds"])
    out[tag + "/time_ids"] = npy(seen["time_ids"])
    meta[tag + "/shapes"] = dict(text_embeds=list(seen["text_embeds"].shape), time_ids=list(seen["time_ids"].shape),
                                 null=list(seen["null"].shape), emb=list(seen["emb"].shape))
s = ref_xl.get_solver("ddim_cfg++", solver_config=cfg(2), device="cpu")
out["G5/time_ids_1024"] = npy(s._get_add_time_ids((1024, 1024), (0, 0), (1024, 1024), torch.float16, 1280))
# in-place mutation of add_cond_kwargs by inversion when lambda in {0,1}
s = ref_xl.get_solver("ddim_edit_cfg++", solver_config=cfg(2), device="cpu")
ack = {"text_embeds": torch.arange(2 * 1280).float().view(2, 1280) * 1e-4, "time_ids": torch.ones(2, 6)}
z0 = torch.zeros(1, 4, 8, 8)
uc = stub.fake_embed("u", (1, 77, 2048))
c = stub.fake_embed("c", (1, 77, 2048))
s.inversion(z0, uc, c, 1.0, ack)
meta["G5/inversion_mutates"] = dict(text_embeds=list(ack["text_embeds"].shape), time_ids=list(ack["time_ids"].shape),
                                    first=float(ack["text_embeds"][0, 1]))

# ----------------------------------------------------------------------------
# G6 callback protocol
# ----------------------------------------------------------------------------
from pathlib import Path  # noqa: E402
import tempfile  # noqa: E402

for freq in (1, 5):
    fired = []

    class Probe(ref_cb.DiffusionCallback):
        def callback(self, step, t, kw):
            fired.append(int(step))
            return kw
    p = Probe(frequency=freq, workdir=Path(tempfile.gettempdir()))
    for st in range(20):
        p(st, torch.tensor(981 - 20 * st), {})
    meta[f"G6/fired_freq{freq}"] = fired
# state replacement: a callback that overwrites zt must change the trajectory
torch.manual_seed(42)
s = ref_sd.get_solver("ddim_cfg++", solver_config=cfg(5), device="cpu", pipe_dtype=torch.float16)
log = hook_unet(s.unet)


def overwrite(step, t, kw):
    kw["zt"] = kw["zt"] * 0.5
    return kw


s.sample(cfg_guidance=0.6, prompt=[NULL, PROMPT], callback_fn=overwrite)
dump_unet_log("G6/replace", log)
meta["G6/callbacks"] = list(ref_cb.__CALLBACK__.keys())

np.savez_compressed(os.path.join(HERE, "sampler_golden.npz"), **out)
with open(os.path.join(HERE, "sampler_golden.json"), "w") as f:
    json.dump(meta, f, indent=1, sort_keys=True)
print("wrote", len(out), "arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")

# ----------------------------------------------------------------------------
# G3h: the inversion / edit paths with an fp16 VAE, as the reference runs them (pipe_dtype = fp16: the
# latent `vae.encode(...)` returns is fp16 and STAYS fp16 through inversion() and the regeneration loop,
# every op rounding to fp16).  Separate file so that sampler_golden.npz keeps regenerating bit-identically.
# ----------------------------------------------------------------------------
out.clear()
meta.clear()
stub.FakeVAE.latent_dtype = torch.float16
run_sd("ddim_inversion_cfg++", "G3h/sd_inv_cfgpp", 10, 0.6, src=True)
run_sd("ddim_inversion", "G3h/sd_inv_cfg", 10, 2.0, src=True)
run_sd("ddim_edit_cfg++", "G3h/sd_edit_cfgpp", 10, 0.6, src=True, prompts=[NULL, PROMPT, PROMPT2])
run_xl("ddim_edit_cfg++", "G3h/xl_edit_cfgpp", 10, 0.6, src=True, prompts=[NULL, PROMPT, PROMPT2])
run_xl("ddim_edit_cfg++", "G3h/xl_edit_cfgpp_recon", 10, 0.6, src=True, prompts=[NULL, PROMPT, PROMPT])
run_xl("ddim_edit", "G3h/xl_edit_cfg", 10, 3.0, src=True, prompts=[NULL, PROMPT, PROMPT2])
stub.FakeVAE.latent_dtype = None
np.savez_compressed(os.path.join(HERE, "sampler_golden_h16.npz"), **out)
with open(os.path.join(HERE, "sampler_golden_h16.json"), "w") as f:
    json.dump(meta, f, indent=1, sort_keys=True)
print("wrote", len(out), "fp16-latent arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")

# ----------------------------------------------------------------------------
# G8 (round 3): the registry names that had no trajectory of their own - SDXL `euler`, `ddim_lightning`,
# `euler_lightning`, `euler_cfg++_lightning` (latent_sdxl.py:469, 519, 541, 810) - and the 'npi' initialisation
# (null-prompt inversion with the prompt embedding on both rows at cfg_guidance = 1; latent_diffusion.py:193-197,
# latent_sdxl.py:280-286).  A third file, so the first two keep regenerating bit-identically.
# ----------------------------------------------------------------------------
out.clear()
meta.clear()
run_xl("euler", "G8/xl_euler_cfg", 10, 5.0)
run_xl("ddim_lightning", "G8/xl_light_ddim_cfg", 4, 1.0)
run_xl("euler_lightning", "G8/xl_light_euler_cfg", 4, 1.0)
run_xl("euler_cfg++_lightning", "G8/xl_light_euler_cfgpp", 4, 1.0)


def run_npi(kind, tag, latent_dtype):
    stub.FakeVAE.latent_dtype = latent_dtype
    torch.manual_seed(42)
    g = torch.Generator().manual_seed(7)
    if kind == "sd":
        s = ref_sd.get_solver("ddim_cfg++", solver_config=cfg(10), device="cpu", pipe_dtype=torch.float16)
        log = hook_unet(s.unet)
        src_img = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
        uc, c = s.get_text_embed(null_prompt=NULL, prompt=PROMPT)
        z = s.initialize_latent(method="npi", src_img=src_img, uc=uc, c=c)
    else:
        s = ref_xl.get_solver("ddim_cfg++", solver_config=cfg(10), device="cpu")
        log = hook_unet(s.unet)
        src_img = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
        null_e, e, pool_null, pool = s.get_text_embed(NULL, PROMPT, NULL, PROMPT)
        ack = {"text_embeds": torch.cat([pool_null, pool], dim=0), "time_ids": torch.ones(2, 6)}
        z = s.initialize_latent(method="npi", src_img=src_img, add_cond_kwargs=ack, uc=null_e, c=e)
        meta[tag + "/ack_rows_after"] = [int(ack["text_embeds"].shape[0]), int(ack["time_ids"].shape[0])]
    dump_unet_log(tag, log)
    out[tag + "/z"] = npy(crop(z))
    meta[tag] = dict(kind=kind, nfe=10, z_dtype=str(z.dtype))
    stub.FakeVAE.latent_dtype = None


run_npi("sd", "G8/sd_npi", None)
run_npi("sd", "G8/sd_npi_h", torch.float16)
run_npi("xl", "G8/xl_npi", None)
run_npi("xl", "G8/xl_npi_h", torch.float16)
np.savez_compressed(os.path.join(HERE, "sampler_golden_r3.npz"), **out)
with open(os.path.join(HERE, "sampler_golden_r3.json"), "w") as f:
    json.dump(meta, f, indent=1, sort_keys=True)
print("wrote", len(out), "round-3 arrays;", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")
