"""Host logic of the Solver classes (registry, index rules, coefficient tables,
conditioning assembly, callbacks, batching) against the reference's golden vectors,
with the CPU mock engine injected in place of the HIP engine."""
import types

import numpy as np
import pytest
import torch

from _stub_env import fake_embed, pointwise_eps
from mock_engine import MockEngine, StubVAE

import cfgpp_amd.latent_diffusion as sd
import cfgpp_amd.latent_sdxl as xl


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def cfgn(n):
    return types.SimpleNamespace(num_sampling=n)


def scripted_unet(z, t, ehs, te, ti):
    return pointwise_eps(z, torch.as_tensor(float(t)).reshape(1), ehs, te, ti)


class StubSDText:
    """the embeddings the golden run's fake CLIP produced"""
    def __call__(self, prompts):
        return torch.cat([fake_embed("L" + p, (1, 77, 768)) for p in prompts]), None


class StubXLText:
    def __init__(self, tag, dim):
        self.tag, self.dim = tag, dim

    def __call__(self, prompts):
        hs = torch.cat([fake_embed(self.tag + p, (1, 77, self.dim)) for p in prompts])
        pooled = torch.cat([fake_embed(self.tag + "pool" + p, (1, self.dim)) for p in prompts])
        return hs, pooled


class Rec:
    def __init__(self):
        self.steps, self.ts, self.z0t, self.zt = [], [], [], []

    def __call__(self, step, t, kw):
        self.steps.append(int(step)); self.ts.append(float(t))
        self.z0t.append(kw["z0t"].clone()); self.zt.append(kw["zt"].clone())
        return kw


def make_sd(name, nfe):
    eng = MockEngine(scripted_unet)
    s = sd.get_solver(name, solver_config=cfgn(nfe), device="cpu", engine=eng, text_encoder=StubSDText(), latent_hw=(8, 8),
                      vae=StubVAE(0.18215))
    return s, eng


def make_xl(name, nfe):
    eng = MockEngine(scripted_unet)
    s = xl.get_solver(name, solver_config=cfgn(nfe), device="cpu", engine=eng,
                      text_encoder=(StubXLText("L", 768), StubXLText("G", 1280)), latent_hw=(8, 8), vae=StubVAE(0.13025))
    return s, eng


def zT_sd(seed=42):
    torch.manual_seed(seed)
    return torch.randn(1, 4, 64, 64)[..., :8, :8].contiguous()


# ------------------------------------------------------------------ G7 registry
def test_registry_names_and_errors(golden):
    _, meta = golden
    assert list(sd.__SOLVER__.keys()) == meta["G7/sd_names"]
    ours = list(xl.__SOLVER__.keys())
    assert ours[: len(meta["G7/sdxl_names"])] == meta["G7/sdxl_names"]
    assert ours[len(meta["G7/sdxl_names"]):] == ["ddim_inversion_cfg++"]      # documented addition
    with pytest.raises(ValueError) as e:
        sd.get_solver("nope")
    assert str(e.value) == meta["G7/unknown_msg"]
    with pytest.raises(ValueError) as e:
        sd.register_solver("ddim")(object)
    assert str(e.value) == meta["G7/dup_msg"]


# ------------------------------------------------------------------ G1 tables
def test_scheduler_tables(golden):
    g, meta = golden
    for nfe in (50, 10):
        s, _ = make_sd("ddim_cfg++", nfe)
        ts = s.scheduler.timesteps
        assert np.array_equal(ts.numpy(), g[f"G1/sd{nfe}/timesteps"])
        assert s.skip == meta[f"G1/sd{nfe}/skip"]
        assert np.array_equal(np.array([float(s.alpha(t)) for t in ts], dtype=np.float32), g[f"G1/sd{nfe}/at"])
        assert np.array_equal(np.array([float(s.alpha(t - s.skip)) for t in ts], dtype=np.float32), g[f"G1/sd{nfe}/at_prev"])
    s, _ = make_sd("ddim_cfg++", 50)
    assert np.array_equal(s.total_alphas.numpy(), g["G1/total_alphas"])
    assert np.allclose(s.sigmas.numpy(), g["G1/sigmas"], rtol=1e-6)
    kar = s.tables.karras_sigmas()
    assert np.allclose(kar.numpy(), g["G1/karras50"], rtol=1e-5)
    assert [int(s.timestep(kar[i])) for i in range(50)] == list(g["G1/karras50_timestep"])
    x, _ = make_xl("ddim_cfg++", 50)
    ts = x.scheduler.timesteps.int()
    assert np.array_equal(ts.numpy(), g["G1/sdxl50/timesteps"])
    assert np.array_equal(np.array([float(x.tables.alpha_wrap(t)) for t in ts], dtype=np.float32), g["G1/sdxl50/at"])
    assert np.array_equal(np.array([float(x.tables.alpha_wrap(int(t) - x.skip)) for t in ts], dtype=np.float32), g["G1/sdxl50/at_next"])  # Q3 wrap
    l, _ = make_xl("ddim_cfg++_lightning", 4)
    assert np.array_equal(l.scheduler.timesteps.numpy(), g["G1/light4/timesteps_f"])
    assert l.skip == meta["G1/light4/skip"] and l.final_alpha_cumprod is None
    ts = l.scheduler.timesteps.int()
    assert np.array_equal(np.array([float(l.tables.alpha_wrap(int(t) - l.skip)) for t in ts], dtype=np.float32), g["G1/light4/at_next"])


# ------------------------------------------------------------------ G2 / G3 / G4 trajectories
@pytest.mark.parametrize("name,tag,nfe,lam", [("ddim_cfg++", "G2/sd_ddim_cfgpp_h", 50, 0.6), ("ddim", "G2/sd_ddim_cfg_h", 10, 7.5)])
def test_sd_ddim_trajectory(golden, name, tag, nfe, lam):
    g, meta = golden
    s, eng = make_sd(name, nfe)
    rec = Rec()
    z0t, zt = s.sample(cfg_guidance=lam, prompt=meta[tag]["prompts"], callback_fn=rec, latents=zT_sd(), return_latents=True)
    assert rec.steps == list(g[tag + "/cb_step"]) and rec.ts == list(g[tag + "/cb_t"])
    assert torch.equal(torch.stack(rec.z0t), T(g[tag + "/z0t"])) and torch.equal(torch.stack(rec.zt), T(g[tag + "/zt"]))
    assert all(c["z_dtype"] == torch.float32 for c in eng.calls)


@pytest.mark.parametrize("name,tag,lam", [("ddim_inversion_cfg++", "G3/sd_inv_cfgpp", 0.6), ("ddim_inversion", "G3/sd_inv_cfg", 2.0),
                                           ("ddim_edit_cfg++", "G3/sd_edit_cfgpp", 0.6)])
def test_sd_inversion_trajectory(golden, name, tag, lam):
    g, meta = golden
    s, eng = make_sd(name, 10)
    rec = Rec()
    z0 = T(g[tag + "/unet_z"])[0][0:1]            # the VAE-encoded source latent the reference inverted
    s.sample(src_img=None, src_latent=z0, cfg_guidance=lam, prompt=meta[tag]["prompts"], callback_fn=rec, return_latents=True)
    uz = T(g[tag + "/unet_z"])
    assert len(eng.calls) == uz.shape[0] == 20
    for i, c in enumerate(eng.calls):            # every UNet input of inversion + regeneration
        assert torch.equal(c["z"], uz[i][0:1]), f"unet call {i}"
    assert torch.equal(torch.stack(rec.z0t), T(g[tag + "/z0t"]))


@pytest.mark.parametrize("name,tag,lam", [("ddim_inversion_cfg++", "G3h/sd_inv_cfgpp", 0.6), ("ddim_inversion", "G3h/sd_inv_cfg", 2.0),
                                           ("ddim_edit_cfg++", "G3h/sd_edit_cfgpp", 0.6)])
def test_sd_inversion_trajectory_fp16_latent(golden_h16, name, tag, lam):
    """the reference's real dtype flow: fp16 VAE latent -> fp16 inversion + regeneration (ddim_step_h kernel path)"""
    g, meta = golden_h16
    s, eng = make_sd(name, 10)
    rec = Rec()
    z0 = T(g[tag + "/unet_z"])[0][0:1]
    assert z0.dtype == torch.float16
    z0_before = z0.clone()
    s.sample(src_img=None, src_latent=z0, cfg_guidance=lam, prompt=meta[tag]["prompts"], callback_fn=rec, return_latents=True)
    assert torch.equal(z0, z0_before)              # the caller's latent is not updated in place
    uz = T(g[tag + "/unet_z"])
    assert len(eng.calls) == uz.shape[0] == 20
    for i, c in enumerate(eng.calls):
        assert c["z_dtype"] == torch.float16 and torch.equal(c["z"], uz[i][0:1]), f"unet call {i}"
    assert torch.equal(torch.stack(rec.z0t), T(g[tag + "/z0t"])) and torch.equal(torch.stack(rec.zt), T(g[tag + "/zt"]))


@pytest.mark.parametrize("name,tag,lam", [("ddim_edit_cfg++", "G3h/xl_edit_cfgpp", 0.6), ("ddim_edit", "G3h/xl_edit_cfg", 3.0),
                                           ("ddim_inversion_cfg++", "G3h/xl_edit_cfgpp_recon", 0.6)])
def test_sdxl_edit_trajectory_fp16_latent(golden_h16, name, tag, lam):
    g, meta = golden_h16
    s, eng = make_xl(name, 10)
    rec = Rec()
    p = meta[tag]["prompts"]
    if name == "ddim_inversion_cfg++":
        p = p[:2]
    z0 = T(g[tag + "/unet_z"])[0][0:1]
    s.sample(prompt1=p, prompt2=p, cfg_guidance=lam, target_size=(64, 64), original_size=(64, 64), callback_fn=rec,
             src_latent=z0, return_latents=True)
    uz = T(g[tag + "/unet_z"])
    assert len(eng.calls) == uz.shape[0]
    for i, c in enumerate(eng.calls):
        assert c["z_dtype"] == torch.float16 and torch.equal(c["z"], uz[i][0:1]), f"unet call {i}"
    assert torch.equal(torch.stack(rec.z0t), T(g[tag + "/z0t"]))


def test_solver_encode_returns_pipe_dtype_and_latents_are_not_updated_in_place():
    s, _ = make_sd("ddim_inversion_cfg++", 4)
    x = torch.rand(1, 3, 64, 64) * 2 - 1
    assert s.encode(x).dtype == torch.float16          # the reference's fp16 VAE -> fp16 latent
    z = zT_sd()
    keep = z.clone()
    make_sd("ddim_cfg++", 4)[0].sample(cfg_guidance=0.6, prompt=["", "x"], latents=z, return_latents=True)
    assert torch.equal(z, keep)


def test_context_cache_sees_in_place_edits():
    """the cross-attention context is cached per embedding tensor; an in-place edit must invalidate it"""
    s, eng = make_sd("ddim_cfg++", 2)
    uc, c = s.get_text_embed("bad", ["a cat"])
    z = zT_sd()
    s.predict_noise(z, 981, uc, c)
    s.predict_noise(z, 961, uc, c)
    assert len(eng.contexts) == 1
    c.mul_(0.5)
    s.predict_noise(z, 941, uc, c)
    assert len(eng.contexts) == 2


def test_scalar_semantics_defaults():
    """product (HIP engine) = "cuda" semantics; an injected (test) engine defaults to the golden "cpu" semantics"""
    s, _ = make_sd("ddim_cfg++", 2)
    assert s.scalar_semantics == "cpu"
    eng = MockEngine(scripted_unet)
    s2 = sd.get_solver("ddim_cfg++", solver_config=cfgn(2), device="cpu", engine=eng, text_encoder=StubSDText(), latent_hw=(8, 8),
                       vae=StubVAE(0.18215), scalar_semantics="cuda")
    a = s.sample(cfg_guidance=0.6, prompt=["", "x"], latents=zT_sd(), return_latents=True)[0]
    b = s2.sample(cfg_guidance=0.6, prompt=["", "x"], latents=zT_sd(), return_latents=True)[0]
    assert not torch.equal(a, b) and float((a - b).abs().max()) < 5e-3
    import inspect
    src = inspect.getsource(sd.StableDiffusion.__init__)
    assert '"cpu" if kwargs.get("engine") is not None else "cuda"' in src


def test_clip_skip_is_not_silently_ignored():
    x, _ = make_xl("ddim_cfg++", 2)
    with pytest.raises(NotImplementedError):
        x.get_text_embed("bad", "a cat", "bad", "a cat", clip_skip=1)


@pytest.mark.parametrize("name,tag,nfe,lam", [("dpm++_2m_cfg++", "G4/sd_dpm2m_cfgpp", 20, 0.6), ("dpm++_2m", "G4/sd_dpm2m_cfg", 10, 7.5),
                                              ("euler_cfg++", "G4/sd_euler_cfgpp", 10, 0.6), ("euler", "G4/sd_euler_cfg", 10, 7.5)])
def test_sd_kdiff_trajectory(golden, name, tag, nfe, lam, monkeypatch):
    g, meta = golden
    s, eng = make_sd(name, nfe)
    # the reference draws a 64x64 latent; our 8x8 test latent is its crop
    monkeypatch.setattr(s, "_randn", lambda size, seeds=None: (torch.manual_seed(42), torch.randn(1, 4, 64, 64))[1][..., :8, :8].contiguous())
    rec = Rec()
    s.sample(cfg_guidance=lam, prompt=meta[tag]["prompts"], callback_fn=rec)
    assert rec.ts == list(g[tag + "/cb_t"])
    assert torch.equal(torch.stack(rec.z0t), T(g[tag + "/z0t"])) and torch.equal(torch.stack(rec.zt), T(g[tag + "/zt"]))
    assert all(c["z_dtype"] == torch.float16 for c in eng.calls)


@pytest.mark.parametrize("name,tag,nfe,lam", [("ddim_cfg++", "G2/xl_ddim_cfgpp", 50, 0.6), ("ddim", "G2/xl_ddim_cfg", 10, 5.0),
                                              ("ddim_cfg++_lightning", "G2/xl_light_ddim_cfgpp", 4, 1.0),
                                              ("dpm++_2m_cfgpp", "G4/xl_dpm2m_cfgpp", 20, 0.6),
                                              ("dpm++_2m_cfgpp_lightning", "G4/xl_light_dpm2m_cfgpp", 4, 1.0),
                                              ("euler_cfg++", "G4/xl_euler_cfgpp", 10, 0.6)])
def test_sdxl_trajectory(golden, name, tag, nfe, lam):
    g, meta = golden
    s, eng = make_xl(name, nfe)
    rec = Rec()
    p = meta[tag]["prompts"]
    s.sample(prompt1=p, prompt2=p, cfg_guidance=lam, target_size=(64, 64), original_size=(64, 64), callback_fn=rec, seeds=[42],
             return_latents=True)
    assert torch.equal(torch.stack(rec.z0t), T(g[tag + "/z0t"])) and torch.equal(torch.stack(rec.zt), T(g[tag + "/zt"]))
    # conditioning rows as the reference's UNet saw them (Q7: lambda == 1 -> positive rows only)
    assert eng.contexts[0]["rows"] == 2
    want_rows = int(g[tag + "/unet_te_rows"][0])
    assert s._ctx_keep[2].shape[0] == want_rows


@pytest.mark.parametrize("name,tag,lam", [("ddim_edit_cfg++", "G3/xl_edit_cfgpp", 0.6), ("ddim_edit", "G3/xl_edit_cfg", 3.0),
                                           ("ddim_inversion_cfg++", "G3/xl_edit_cfgpp_recon", 0.6)])
def test_sdxl_edit_trajectory(golden, name, tag, lam):
    g, meta = golden
    s, eng = make_xl(name, 10)
    rec = Rec()
    p = meta[tag]["prompts"]
    if name == "ddim_inversion_cfg++":
        p = p[:2]
    z0 = T(g[tag + "/unet_z"])[0][0:1]
    s.sample(prompt1=p, prompt2=p, cfg_guidance=lam, target_size=(64, 64), original_size=(64, 64), callback_fn=rec,
             src_latent=z0, return_latents=True)
    uz = T(g[tag + "/unet_z"])
    assert len(eng.calls) == uz.shape[0]
    for i, c in enumerate(eng.calls):
        assert torch.equal(c["z"], uz[i][0:1]), f"unet call {i}"
    assert torch.equal(torch.stack(rec.z0t), T(g[tag + "/z0t"]))


@pytest.mark.parametrize("name,tag,nfe,lam", [("euler", "G8/xl_euler_cfg", 10, 5.0), ("ddim_lightning", "G8/xl_light_ddim_cfg", 4, 1.0),
                                              ("euler_lightning", "G8/xl_light_euler_cfg", 4, 1.0),
                                              ("euler_cfg++_lightning", "G8/xl_light_euler_cfgpp", 4, 1.0)])
def test_sdxl_trajectory_remaining_names(golden_r3, name, tag, nfe, lam):
    """the four latent_sdxl registry names that shared a loop with a tested name but had no golden trajectory of their
    own (latent_sdxl.py:469, 519, 541, 810): every UNet input and every (z0t, zt) of the reference's run"""
    g, meta = golden_r3
    s, eng = make_xl(name, nfe)
    rec = Rec()
    p = meta[tag]["prompts"]
    s.sample(prompt1=p, prompt2=p, cfg_guidance=lam, target_size=(64, 64), original_size=(64, 64), callback_fn=rec, seeds=[42],
             return_latents=True)
    uz, ut = T(g[tag + "/unet_z"]), T(g[tag + "/unet_t"])
    assert len(eng.calls) == uz.shape[0] == nfe
    for i, c in enumerate(eng.calls):
        assert c["z"].dtype == uz.dtype and torch.equal(c["z"], uz[i][0:1]), f"unet call {i}"
        assert float(c["t"]) == float(ut[i][0]), (i, float(c["t"]), float(ut[i][0]))
    assert torch.equal(torch.stack(rec.z0t), T(g[tag + "/z0t"])) and torch.equal(torch.stack(rec.zt), T(g[tag + "/zt"]))
    assert [float(t) for t in rec.ts] == [float(t) for t in g[tag + "/cb_t"]]
    assert s._ctx_keep[2].shape[0] == int(g[tag + "/unet_te_rows"][0])          # Q7: lambda == 1 -> positive rows only
    if "lightning" in name:
        with pytest.raises(AssertionError):                                      # "CFG should be turned off in the lightning version"
            s.sample(prompt1=p, prompt2=p, cfg_guidance=0.6, target_size=(64, 64), original_size=(64, 64), seeds=[42], return_latents=True)


@pytest.mark.parametrize("kind,tag", [("sd", "G8/sd_npi"), ("sd", "G8/sd_npi_h"), ("xl", "G8/xl_npi"), ("xl", "G8/xl_npi_h")])
def test_npi_initialisation(golden_r3, kind, tag):
    """``initialize_latent(method='npi')`` (latent_diffusion.py:193-197, latent_sdxl.py:280-286): inversion of the encoded
    source with the PROMPT embedding on both rows at cfg_guidance = 1 - fp32 and fp16 (the reference's real dtype) latents"""
    g, meta = golden_r3
    null = "low quality,jpeg artifacts,blurry,poorly drawn,ugly,worst quality,"
    prompt = "a photo of an astronaut riding a horse on mars"
    uz = T(g[tag + "/unet_z"])
    z0 = uz[0][0:1].clone()                        # the latent the reference's VAE stub produced = first UNet input
    if kind == "sd":
        s, eng = make_sd("ddim_cfg++", 10)
        s.encode = lambda x: z0
        uc, c = s.get_text_embed(null_prompt=null, prompt=prompt)
        z = s.initialize_latent(method="npi", src_img=torch.zeros(1, 3, 64, 64), uc=uc, c=c)
    else:
        s, eng = make_xl("ddim_cfg++", 10)
        s.encode = lambda x: z0
        null_e, e, pool_null, pool = s.get_text_embed(null, prompt, null, prompt)
        ack = {"text_embeds": torch.cat([pool_null, pool], dim=0), "time_ids": torch.ones(2, 6)}
        z = s.initialize_latent(method="npi", src_img=torch.zeros(1, 3, 64, 64), add_cond_kwargs=ack, uc=null_e, c=e)
        assert [int(ack["text_embeds"].shape[0]), int(ack["time_ids"].shape[0])] == meta[tag + "/ack_rows_after"]      # reduced in place
    assert len(eng.calls) == uz.shape[0] == 10
    for i, c_ in enumerate(eng.calls):
        assert c_["z"].dtype == uz.dtype and torch.equal(c_["z"], uz[i][0:1]), f"unet call {i}"
    want = T(g[tag + "/z"])
    assert str(z.dtype) == meta[tag]["z_dtype"] and torch.equal(z.cpu(), want)
    # both rows carried the prompt embedding: the two eps halves of every call are the same tensor
    ue = T(g[tag + "/unet_eps"])
    assert torch.equal(ue[:, 0], ue[:, 1])


# ------------------------------------------------------------------ G5 conditioning
def test_sdxl_conditioning(golden):
    g, meta = golden
    for lam, tag in ((0.6, "G5/xl_cond_l06"), (1.0, "G5/xl_cond_l10")):
        s, eng = make_xl("ddim_cfg++", 2)
        p = ["low quality,jpeg artifacts,blurry,poorly drawn,ugly,worst quality,", "a photo of an astronaut riding a horse on mars"]
        seen = {}
        orig = s.reverse_process

        def spy(null_e, emb, lam_, ack, shape, **kw):
            seen.update(te=ack["text_embeds"].clone(), ti=ack["time_ids"].clone(), null=null_e, emb=emb)
            return orig(null_e, emb, lam_, ack, shape, **kw)
        s.reverse_process = spy
        s.sample(prompt1=p, prompt2=p, cfg_guidance=lam, target_size=(64, 64), seeds=[0], return_latents=True)
        assert list(seen["te"].shape) == meta[tag + "/shapes"]["text_embeds"]
        assert list(seen["ti"].shape) == meta[tag + "/shapes"]["time_ids"]
        assert list(seen["emb"].shape) == meta[tag + "/shapes"]["emb"]
        assert np.array_equal(seen["te"].float().numpy(), g[tag + "/text_embeds"])
        assert np.array_equal(seen["ti"].float().numpy(), g[tag + "/time_ids"].astype(np.float32))
    s, _ = make_xl("ddim_cfg++", 2)
    assert np.array_equal(s._get_add_time_ids((1024, 1024), (0, 0), (1024, 1024), torch.float16, 1280).float().numpy(),
                          g["G5/time_ids_1024"].astype(np.float32))
    # in-place reduction of add_cond_kwargs by inversion when lambda in {0,1}
    s, _ = make_xl("ddim_edit_cfg++", 2)
    ack = {"text_embeds": torch.arange(2 * 1280).float().view(2, 1280) * 1e-4, "time_ids": torch.ones(2, 6)}
    s.inversion(torch.zeros(1, 4, 8, 8), fake_embed("u", (1, 77, 2048)), fake_embed("c", (1, 77, 2048)), 1.0, ack)
    m = meta["G5/inversion_mutates"]
    assert list(ack["text_embeds"].shape) == m["text_embeds"] and list(ack["time_ids"].shape) == m["time_ids"]
    assert abs(float(ack["text_embeds"][0, 1]) - m["first"]) < 1e-9
    with pytest.raises(AssertionError):
        make_xl("ddim_cfg++_lightning", 4)[0].sample(prompt1=["", "x"], prompt2=["", "x"], cfg_guidance=0.6, target_size=(64, 64))


# ------------------------------------------------------------------ G6 callbacks
def test_callback_protocol(golden, tmp_path):
    g, meta = golden
    from cfgpp_amd import callback_util as cb
    assert list(cb.__CALLBACK__.keys()) == meta["G6/callbacks"]
    for freq in (1, 5):
        fired = []

        class Probe(cb.DiffusionCallback):
            def callback(self, step, t, kw):
                fired.append(int(step))
                return kw
        p = Probe(frequency=freq, workdir=tmp_path)
        for st in range(20):
            p(st, torch.tensor(981 - 20 * st), {})
        assert fired == meta[f"G6/fired_freq{freq}"]
    # the returned zt REPLACES the loop state
    s, eng = make_sd("ddim_cfg++", 5)

    def overwrite(step, t, kw):
        kw["zt"] = kw["zt"] * 0.5
        return kw
    s.sample(cfg_guidance=0.6, prompt=meta["G2/sd_ddim_cfgpp_h"]["prompts"], callback_fn=overwrite, latents=zT_sd(), return_latents=True)
    uz = T(g["G6/replace/unet_z"])
    for i, c in enumerate(eng.calls):
        assert torch.equal(c["z"], uz[i][0:1])
    # draw_* callbacks call decode() and write one file named by int(t)
    comp = cb.ComposeCallback(workdir=tmp_path, callbacks=["draw_noisy", "draw_tweedie"], frequency=1)
    comp(0, torch.tensor(981), {"z0t": torch.zeros(1, 4, 2, 2), "zt": torch.zeros(1, 4, 2, 2), "decode": lambda z: torch.zeros(1, 3, 16, 16)})
    assert any(f.name.startswith("x0_981") for f in (tmp_path / "record/tweedie").iterdir())
    assert any(f.name.startswith("xt_981") for f in (tmp_path / "record/noisy").iterdir())


# ------------------------------------------------------------------ batching extension
def test_batched_chains_equal_independent_runs():
    prompts = ["a cat", "a dog", "a bird"]
    s, _ = make_sd("ddim_cfg++", 6)
    zb = torch.cat([StableZ(sd_) for sd_ in (1, 2, 3)])
    z0b, _ = s.sample(cfg_guidance=0.6, prompt=["bad", prompts], latents=zb.clone(), return_latents=True)
    for b in range(3):
        s1, _ = make_sd("ddim_cfg++", 6)
        z01, _ = s1.sample(cfg_guidance=0.6, prompt=["bad", prompts[b]], latents=zb[b:b + 1].clone(), return_latents=True)
        assert torch.equal(z0b[b:b + 1], z01)
    # seeds=[...] reproduces per-chain CPU-generator noise
    a = sd.StableDiffusion._randn((2, 4, 8, 8), seeds=[7, 9])
    torch.manual_seed(9)
    assert torch.equal(a[1:2], torch.randn(1, 4, 8, 8))


def StableZ(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, 4, 8, 8, generator=g)


@pytest.mark.parametrize("name,tag,lam", [("euler_a", "G4/sd_euler_a_cfg", 7.5), ("euler_a_cfg++", "G4/sd_euler_a_cfgpp", 0.6),
                                           ("dpm++_2s_a", "G4/sd_dpm2s_a_cfg", 7.5), ("dpm++_2s_a_cfg++", "G4/sd_dpm2s_a_cfgpp", 0.6)])
def test_sd_ancestral_trajectory(golden, name, tag, lam, monkeypatch):
    """ancestral samplers: the reference draws its noise from the global CPU generator after z_T, at 64x64;
    replay the same draws (cropped) through the mock engine."""
    g, meta = golden
    s, eng = make_sd(name, 8)
    monkeypatch.setattr(s, "_randn", lambda size, seeds=None: (torch.manual_seed(42), torch.randn(1, 4, 64, 64))[1][..., :8, :8].contiguous())
    eng.randn_like = lambda x: torch.randn_like(torch.empty(1, 4, 64, 64, dtype=x.dtype))[..., :8, :8].contiguous()
    rec = Rec()
    s.sample(cfg_guidance=lam, prompt=meta[tag]["prompts"], callback_fn=rec)
    uz = T(g[tag + "/unet_z"])
    assert len(eng.calls) == uz.shape[0]
    for i, c in enumerate(eng.calls):
        assert torch.equal(c["z"], uz[i][0:1]), f"unet call {i}"
    assert torch.equal(torch.stack(rec.z0t), T(g[tag + "/z0t"])) and torch.equal(torch.stack(rec.zt), T(g[tag + "/zt"]))


def test_product_path_has_no_cpu_fallback():
    from cfgpp_amd._lib import CfgppError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(CfgppError):
        sd.get_solver("ddim_cfg++", solver_config=cfgn(5), device="cuda")
