"""-m gpu: parity of the BASELINE configs' own flows and sizes (C2-C5) on the MI355X.

* every ``latent_sdxl`` solver on the bench path (``ddim_cfg++`` = C3, ``ddim_cfg++_lightning`` = C4,
  ``ddim_edit_cfg++`` / ``ddim_inversion_cfg++`` = C5, plus ``dpm++_2m_cfgpp``) runs through the HIP engine
  (UNet + fused step kernels + HIP VAE encode) against the SAME solver class on the CPU mock engine driving
  ``oracle/unet_ref.py`` - mirrors ``test_gpu_unet.py::test_sd_chain_vs_oracle`` for ``latent_sdxl.py:715-755,
  838-858,860-930,954-1025``;
* (the real SD1.5 / SDXL nets at the benchmark's own sizes: tests/test_gpu_realsize.py);
* attention and the VAE at the benchmark's own geometries;
* the fp16-latent step kernel (inversion / edit dtype flow of the reference) bit-exact vs the golden vectors.

Tolerances are 2x what was measured on the MI355X (recorded in gpurun_out/parity_<run>.jsonl by these tests).
Both sides of a chain test use the same scalar semantics ("cuda" = the product default, see cfgpp_amd/coeffs.py).
"""
import json
import os
import time
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


_RUN_TAG = f"{time.strftime('%Y%m%d_%H%M%S')}_{os.getpid()}"


def record(test, **kw):
    """measured errors of THIS run -> gpurun_out/parity_<start time>_<pid>.jsonl (one file per pytest process: numbers of
    different builds never mix; tolerances are set from these)"""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"parity_{_RUN_TAG}.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=test, **kw)) + "\n")
    except OSError:
        pass


def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ------------------------------------------------------------------ fp16-latent step kernel vs the reference's golden
@pytest.mark.parametrize("tag,lam,cfgpp", [("G3h/sd_inv_cfgpp", 0.6, True), ("G3h/sd_inv_cfg", 2.0, False),
                                           ("G3h/xl_edit_cfgpp", 0.6, True), ("G3h/xl_edit_cfg", 3.0, False)])
def test_ddim_fp16_latent_kernel_vs_golden(golden_h16, tag, lam, cfgpp):
    need_gpu()
    from cfgpp_amd import engine as E
    from cfgpp_amd.coeffs import ddim_coeffs_pinned
    from cfgpp_amd.schedule import SchedulerTables
    g, _ = golden_h16
    tb = SchedulerTables(10)
    z, e, z0, zt = T(g[tag + "/unet_z"]), T(g[tag + "/unet_eps"]), T(g[tag + "/z0t"]), T(g[tag + "/zt"])
    off = z.shape[0] - z0.shape[0]

    def run(i, s4, tw, rn):
        zi = z[i][0:1].contiguous().cuda()
        z0o = torch.empty_like(zi)
        co = ddim_coeffs_pinned(s4, eps_half=True, semantics="cpu", z_half=True)       # the golden is a torch-CPU recording
        E.step_ddim(zi, z0o, e[i][0:1].contiguous().cuda(), e[i][1:2].contiguous().cuda(), lam, co, tw, rn)
        return z0o.cpu(), zi.cpu()
    for i, t in enumerate(reversed(tb.timesteps)):
        _, b = run(i, tb.ddim_sqrt_coeffs(t, inversion=True), cfgpp, False)
        assert b.dtype == torch.float16 and torch.equal(b, z[i + 1][0:1]), f"{tag} inversion step {i}"
    for i, t in enumerate(tb.timesteps):
        a, b = run(off + i, tb.ddim_sqrt_coeffs(t), False, cfgpp)
        assert torch.equal(a, z0[i]) and torch.equal(b, zt[i]), f"{tag} step {i}"


def test_ddim_fp16_latent_kernel_full_size_vs_oracle():
    """[8,4,128,128] fp16 latents, both scalar semantics: kernel == oracle, bit for bit"""
    need_gpu()
    from cfgpp_amd import engine as E
    from cfgpp_amd.coeffs import ddim_coeffs_pinned
    from cfgpp_amd.schedule import SchedulerTables
    from oracle import sampler as O
    tb = SchedulerTables(50)
    g = torch.Generator().manual_seed(11)
    shape = (8, 4, 128, 128)
    z = (torch.randn(shape, generator=g) * 2).half()
    eu, ec = torch.randn(shape, generator=g).half(), torch.randn(shape, generator=g).half()
    for sem in ("cuda", "cpu"):
        for (tw, rn, inv) in ((False, True, False), (True, False, True)):
            s4 = tb.ddim_sqrt_coeffs(tb.timesteps[11], inversion=inv)
            a, b = O.ddim_step(z, eu, ec, 0.6, None, None, tw, rn, sqrt4=s4, semantics=sem)
            zd, z0d = z.clone().cuda(), torch.empty_like(z).cuda()
            E.step_ddim(zd, z0d, eu.cuda(), ec.cuda(), 0.6, ddim_coeffs_pinned(s4, True, sem, z_half=True), tw, rn)
            assert torch.equal(z0d.cpu(), a) and torch.equal(zd.cpu(), b), (sem, tw, rn)


# ------------------------------------------------------------------ latent_sdxl solvers on the HIP engine (C3 / C4 / C5 flows)
class _PinnedNoiseVAE:
    """``solver.encode`` with the posterior noise pinned (the reference draws it from the device RNG); wraps either the
    HIP VAE or the CPU restatement so both sides sample the SAME posterior point."""

    def __init__(self, kind, scale, hw, B, noise, sd):
        self.noise = noise
        if kind == "hip":
            from cfgpp_amd.vae import HipVAE
            self.v = HipVAE(scale, hw, max_batch=B, state_dict=sd)
            self.enc = lambda x: self.v.encode(x, noise=self.noise)
        else:
            from oracle.vae_ref import VAERef
            self.v = VAERef(scale, device="cpu", dtype=torch.float32, state_dict=sd)

            def enc(x):
                mean, logvar = self.v.encode_moments(x.float().cpu())
                return (mean + torch.exp(0.5 * logvar) * self.noise) * scale
            self.enc = enc

    def encode(self, x):
        return self.enc(x)

    def decode(self, z):
        return self.v.decode(z)


def _xl_pair(name, nfe, B, hw=16):
    """(hip solver, cpu mock solver) of one latent_sdxl registry name on TINY_XL, sharing weights and text encoders"""
    from cfgpp_amd.latent_sdxl import get_solver
    from cfgpp_amd.unet_config import TINY_XL as cfg
    from cfgpp_amd.weights import synth_state_dict
    from mock_engine import MockEngine
    from oracle.unet_ref import UNetRef
    sc = types.SimpleNamespace(num_sampling=nfe)
    hip = get_solver(name, solver_config=sc, device="cuda", unet_config=cfg, max_batch=B, latent_hw=(hw, hw),
                     scalar_semantics="cuda")
    net = UNetRef(cfg, synth_state_dict(cfg, 0))

    def unet(z, t, ehs, te, ti):
        ack = {"text_embeds": te.float(), "time_ids": ti.float()}
        return net(z.float(), t, ehs.float(), ack)["sample"].half()
    ref = get_solver(name, solver_config=sc, device="cpu", unet_config=cfg, max_batch=B, latent_hw=(hw, hw),
                     text_encoder=hip.text_encoder, engine=MockEngine(unet, (hw, hw)), scalar_semantics="cuda")
    return hip, ref, cfg


# tolerance = ~2x the chain rel-L2 measured on the MI355X (profiles/r02/parity_r02.jsonl): 6.0e-4, 5.1e-4, 2.2e-3, 1.6e-3, 8.7e-4
XL_CASES = [("ddim_cfg++", 20, 0.6, 1.5e-3), ("ddim_cfg++_lightning", 4, 1.0, 1.5e-3), ("dpm++_2m_cfgpp", 10, 0.6, 5e-3),
            ("ddim", 8, 5.0, 4e-3), ("euler_cfg++", 8, 0.6, 2e-3)]


@pytest.mark.parametrize("name,nfe,lam,tol", XL_CASES)
def test_sdxl_solver_chain_vs_oracle(name, nfe, lam, tol):
    """C3 (ddim_cfg++) / C4 (ddim_cfg++_lightning, lambda = 1: Q7 broadcast of the positive conditioning) whole loops,
    B = 2 chains: HIP UNet + fused step kernels vs UNetRef + the CPU emulation of the reference's arithmetic."""
    need_gpu()
    B = 2
    hip, ref, cfg = _xl_pair(name, nfe, B)
    prompts = ["a cat", "a dog"]
    pe = hip.get_text_embed("bad", prompts, "bad", prompts)
    kw = dict(cfg_guidance=lam, target_size=(128, 128), original_size=(128, 128), seeds=[21, 22], return_latents=True)
    a = hip.sample(prompt_embeds=pe, **kw)
    b = ref.sample(prompt_embeds=tuple(x.cpu() for x in pe), **kw)
    rel = rel_l2(a, b)
    record("sdxl_chain", name=name, nfe=nfe, rel_l2=rel)
    assert a.shape == (B, 4, 16, 16) and torch.isfinite(a.float()).all() and rel < tol, f"{name}: chain rel-L2 {rel:.3e}"
    if lam == 1.0:      # Q7: the UNet saw the positive conditioning rows only
        assert hip._ctx_keep[2].shape[0] == B


# inversion amplifies the per-forward fp16 noise (eps rel-L2 1e-3) and the latent itself is fp16 here: measured chain
# rel-L2 1.2e-2 (CFG++, lambda 0.6), 2.8e-2 (plain CFG, omega 3) after 8 + 8 steps; tolerance = 2x that
# The last case is C5 at its REAL length, 50 + 50 NFE (latent_sdxl.py:956-1025).  Expected growth: every forward adds an
# independent eps error of rel-L2 ~1e-3 (fp16 GEMM inputs) and every fp16 latent update a rounding of 2^-11; an inversion
# step amplifies what it inherits by at most sqrt(a_t / a_{t-skip}) and the regeneration contracts it again, so the chain
# error grows like sqrt(#forwards) rather than linearly: 8 + 8 steps measure 1.2e-2, 50 + 50 are bounded at 2.5x that, i.e.
# sqrt(100 / 16), with the same 2x margin as the short cases -> 6e-2.
@pytest.mark.parametrize("name,lam,tol,nfe", [("ddim_edit_cfg++", 0.6, 2.5e-2, 8), ("ddim_inversion_cfg++", 0.6, 2.5e-2, 8), ("ddim_edit", 3.0, 6e-2, 8),
                                              ("ddim_edit_cfg++", 0.6, 6e-2, 50)])
def test_sdxl_invert_edit_vs_oracle(name, lam, tol, nfe):
    """C5: VAE encode (HIP kernels, pinned posterior noise) -> fp16 latent -> CFG++ inversion -> regeneration, B = 2,
    against the CPU restatement of the same flow (latent_sdxl.py:954-1025)."""
    need_gpu()
    from cfgpp_amd.vae import synth_vae_state_dict
    B, hw = 2, 16
    hip, ref, cfg = _xl_pair(name, nfe, B)
    g = torch.Generator().manual_seed(5)
    img = torch.rand((B, 3, 8 * hw, 8 * hw), generator=g) * 2 - 1
    noise = torch.randn((B, 4, hw, hw), generator=g)
    vsd = synth_vae_state_dict(0)
    hip.vae = _PinnedNoiseVAE("hip", cfg.vae_scale, (hw, hw), B, noise, vsd)
    ref.vae = _PinnedNoiseVAE("cpu", cfg.vae_scale, (hw, hw), B, noise, vsd)
    src, tgt = ["a cat", "a dog"], ["a tiger", "a wolf"]
    if name == "ddim_inversion_cfg++":
        n_e, s_e, p_n, p_s = hip.get_text_embed("bad", src, "bad", src)
        pe = (n_e, s_e, s_e, p_n, p_s, p_s)
    else:
        n_e, s_e, p_n, p_s = hip.get_text_embed("bad", src, "bad", src)
        _, t_e, _, p_t = hip.get_text_embed("bad", tgt, "bad", tgt)
        pe = (n_e, s_e, t_e, p_n, p_s, p_t)
    kw = dict(cfg_guidance=lam, target_size=(128, 128), original_size=(128, 128), return_latents=True)
    z_hip = hip.encode(img)
    z_ref = ref.encode(img)
    assert z_hip.dtype == torch.float16                 # the reference's fp16 VAE latent
    rel_z = rel_l2(z_hip, z_ref)
    a = hip.sample(prompt_embeds=pe, src_img=img, **kw)
    b = ref.sample(prompt_embeds=tuple(x.cpu() for x in pe), src_img=img, **kw)
    rel = rel_l2(a, b)
    record("sdxl_invert_edit", name=name, nfe=nfe, rel_l2=rel, rel_encode=rel_z)
    assert a.dtype == torch.float16 and torch.isfinite(a.float()).all()
    assert rel_z < 2e-3 and rel < tol, f"{name}: encode rel-L2 {rel_z:.3e}, chain rel-L2 {rel:.3e}"      # encode measured 8.1e-4


def test_sd_invert_with_hip_vae_vs_oracle():
    """SD1.5 ddim_inversion_cfg++ from an IMAGE: HIP VAE encode -> fp16 latent chain vs the CPU restatement"""
    need_gpu()
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD as cfg
    from cfgpp_amd.vae import synth_vae_state_dict
    from cfgpp_amd.weights import synth_state_dict
    from mock_engine import MockEngine
    from oracle.unet_ref import UNetRef
    B, hw = 2, 16
    sc = types.SimpleNamespace(num_sampling=6)
    hip = get_solver("ddim_inversion_cfg++", solver_config=sc, device="cuda", unet_config=cfg, max_batch=B, scalar_semantics="cuda")
    net = UNetRef(cfg, synth_state_dict(cfg, 0))
    ref = get_solver("ddim_inversion_cfg++", solver_config=sc, device="cpu", unet_config=cfg, max_batch=B, text_encoder=hip.text_encoder,
                     engine=MockEngine(lambda z, t, ehs, te, ti: net(z.float(), t, ehs.float())["sample"].half(), (hw, hw)),
                     scalar_semantics="cuda")
    g = torch.Generator().manual_seed(6)
    img = torch.rand((B, 3, 8 * hw, 8 * hw), generator=g) * 2 - 1
    noise = torch.randn((B, 4, hw, hw), generator=g)
    vsd = synth_vae_state_dict(0)
    hip.vae = _PinnedNoiseVAE("hip", cfg.vae_scale, (hw, hw), B, noise, vsd)
    ref.vae = _PinnedNoiseVAE("cpu", cfg.vae_scale, (hw, hw), B, noise, vsd)
    uc, c = hip.get_text_embed("bad", ["a cat", "a dog"])
    a = hip.sample(src_img=img, cfg_guidance=0.6, prompt_embeds=(uc, c), return_latents=True)[0]
    b = ref.sample(src_img=img, cfg_guidance=0.6, prompt_embeds=(uc.cpu(), c.cpu()), return_latents=True)[0]
    rel = rel_l2(a, b)
    record("sd_invert_hip_vae", rel_l2=rel)
    assert a.dtype == torch.float16 and rel < 2e-2, rel          # measured 9.5e-3 (6 + 6 steps, fp16 latent)


@pytest.mark.parametrize("name,lam,tol", [("euler_a_cfg++", 0.6, 2e-3), ("dpm++_2s_a_cfg++", 0.6, 2e-3), ("euler_a", 7.5, 1e-2)])
def test_ancestral_solver_chain_vs_oracle_pinned_noise(name, lam, tol):
    """ancestral samplers with the injected noise pinned on both sides (the reference draws it from the device RNG)"""
    need_gpu()
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD as cfg
    from cfgpp_amd.weights import synth_state_dict
    from mock_engine import MockEngine
    from oracle.unet_ref import UNetRef
    B, nfe = 2, 6
    sc = types.SimpleNamespace(num_sampling=nfe)
    hip = get_solver(name, solver_config=sc, device="cuda", unet_config=cfg, max_batch=B, scalar_semantics="cuda")
    net = UNetRef(cfg, synth_state_dict(cfg, 0))
    eng = MockEngine(lambda z, t, ehs, te, ti: net(z.float(), t, ehs.float())["sample"].half(), (16, 16))
    ref = get_solver(name, solver_config=sc, device="cpu", unet_config=cfg, max_batch=B, text_encoder=hip.text_encoder, engine=eng,
                     scalar_semantics="cuda")
    g = torch.Generator().manual_seed(8)
    noises = [torch.randn((B, 4, 16, 16), generator=g).half() for _ in range(nfe)]
    it_h, it_r = iter(noises), iter(noises)
    hip.engine.randn_like = lambda x: next(it_h).to(x.device)
    eng.randn_like = lambda x: next(it_r)
    uc, c = hip.get_text_embed("bad", ["a cat", "a dog"])
    a = hip.sample(cfg_guidance=lam, prompt_embeds=(uc, c), seeds=[3, 4], return_latents=True)[1]
    b = ref.sample(cfg_guidance=lam, prompt_embeds=(uc.cpu(), c.cpu()), seeds=[3, 4], return_latents=True)[1]
    rel = rel_l2(a, b)
    record("ancestral_chain", name=name, rel_l2=rel)
    assert torch.isfinite(a.float()).all() and rel < tol, f"{name}: rel-L2 {rel:.3e}"      # measured 7.1e-4, 8.8e-4, 4.2e-3


# ------------------------------------------------------------------ the benchmark's own sizes
# (the real SD1.5 / SDXL nets - forwards at every bench plan's row count and the real-net chains - live in
#  tests/test_gpu_realsize.py: one subprocess per case, against fixtures recorded from the oracle in the build container)
ATTN_CASES = [(1, 2, 4096, 4096, 40), (1, 2, 4096, 4096, 64), (2, 3, 1024, 1024, 64), (1, 2, 4096, 77, 40), (1, 2, 4096, 77, 64),
              (1, 2, 1024, 77, 64), (1, 2, 1024, 1024, 80), (1, 1, 256, 256, 160)]


@pytest.mark.parametrize("B,h,Nq,Nk,d", ATTN_CASES)
def test_attention_at_unet_sizes(B, h, Nq, Nk, d):
    """the UNet's own attention geometries (SD1.5 64x64: N = 4096, d = 40; SDXL: N = 4096 / 1024, d = 64; cross: 77 keys)"""
    need_gpu()
    import hip_ops as H
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(40 + d)
    q, k, v = (torch.randn((B, h, n, d), generator=g).half().float() for n in (Nq, Nk, Nk))
    q = q * 1.5                     # some rows with a peaked softmax
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, Nq, h * d)
    hq, hk, hvt, qp, kp = H.make_heads(q, k, v)
    got = H.attention(hq, hk, hvt, B, h, d, Nq, Nk, qp, kp)
    st = H.err_stats(got, ref)
    record("attention", B=B, h=h, Nq=Nq, Nk=Nk, d=d, **st)
    assert st["finite"] and st["rel_l2"] < 1.2e-3, st           # measured 4.6e-4 .. 5.7e-4


@pytest.mark.parametrize("mode", [1])
def test_attention_online_softmax_rescale_branch(mode):
    """a key whose score dwarfs the running max at a LATE tile forces the re-reference branch (the branch is rare on
    random data); full-tensor fp64 reference"""
    need_gpu()
    import hip_ops as H
    g = torch.Generator().manual_seed(77)
    B, h, N, d = 1, 1, 1024, 64
    q, k, v = (torch.randn((B, h, N, d), generator=g).half().float() for _ in range(3))
    k[0, 0, 900] = q[0, 0, 5] * 4.0          # query 5 (and its neighbours in the wave) meets a huge score at tile 14
    k[0, 0, 130] = q[0, 0, 700] * 3.0
    s = (q.double() @ k.double().transpose(-1, -2)) / d ** 0.5
    ref = (torch.softmax(s, -1) @ v.double()).transpose(1, 2).reshape(B, N, h * d).float()
    hq, hk, hvt, qp, kp = H.make_heads(q, k, v)
    H.lib().cfgpp_attention_set_dma(mode)
    try:
        got = H.attention(hq, hk, hvt, B, h, d, N, N, qp, kp)
    finally:
        H.lib().cfgpp_attention_set_dma(1)
    st = H.err_stats(got, ref)
    record("attention_rescale", mode=mode, **st)
    assert st["finite"] and st["rel_l2"] < 1.5e-3 and st["max_abs"] < 2e-2, st


def test_vae_decode_at_512():
    """C2's decode: 64x64 latent -> 512x512 image (mid-block attention over 4096 tokens, one 512-wide head)"""
    need_gpu()
    from cfgpp_amd.vae import HipVAE, synth_vae_state_dict
    from oracle.vae_ref import VAERef
    sd = synth_vae_state_dict(0)
    g = torch.Generator().manual_seed(1)
    z = torch.randn((1, 4, 64, 64), generator=g) * 0.18215 * 1.5
    hip = HipVAE(0.18215, (64, 64), max_batch=1, state_dict=sd, with_encoder=False)
    img = hip.decode(z.cuda()).cpu()
    t0 = time.time()
    ref = VAERef(0.18215, device="cpu", dtype=torch.float32, state_dict=sd).decode(z)
    rel = rel_l2(img, ref)
    record("vae_decode_512", rel_l2=rel, cpu_ref_s=round(time.time() - t0, 1))
    assert img.shape == (1, 3, 512, 512) and torch.isfinite(img).all() and rel < 3e-3, rel          # measured 1.2e-3


def test_vae_decode_and_encode_at_1024():
    """SDXL's VAE passes (latent_sdxl.py:150-164): 128x128 latent <-> 1024x1024 image - the mid-block attention runs
    over 16384 tokens with one 512-wide head - decode AND encode (pinned posterior noise) vs the fp32 restatement."""
    need_gpu()
    from cfgpp_amd.vae import HipVAE, synth_vae_state_dict
    from oracle.vae_ref import VAERef
    sd = synth_vae_state_dict(0)
    scale = 0.13025
    g = torch.Generator().manual_seed(3)
    z = torch.randn((1, 4, 128, 128), generator=g) * scale * 1.5
    x = torch.rand((1, 3, 1024, 1024), generator=g) * 2 - 1
    noise = torch.randn((1, 4, 128, 128), generator=g)
    hip = HipVAE(scale, (128, 128), max_batch=1, state_dict=sd)
    img = hip.decode(z.cuda()).cpu()
    zz, mom = hip.encode(x, noise=noise, return_moments=True)
    zz, mom = zz.cpu(), mom.cpu()
    t0 = time.time()
    ref = VAERef(scale, device="cpu", dtype=torch.float32, state_dict=sd)
    ref_img = ref.decode(z)
    mean, logvar = ref.encode_moments(x)
    ref_z = (mean + torch.exp(0.5 * logvar) * noise) * scale
    rel_d, rel_m, rel_z = rel_l2(img, ref_img), rel_l2(mom, torch.cat([mean, logvar], dim=1)), rel_l2(zz, ref_z)
    record("vae_1024", rel_decode=rel_d, rel_moments=rel_m, rel_latent=rel_z, cpu_ref_s=round(time.time() - t0, 1))
    assert img.shape == (1, 3, 1024, 1024) and torch.isfinite(img).all() and torch.isfinite(zz).all()
    assert rel_d < 3e-3 and rel_m < 1e-2 and rel_z < 1e-2, (rel_d, rel_m, rel_z)      # 512^2: decode 1.2e-3, encode moments 8e-4


def test_groupnorm_large_mean_small_variance():
    """|mean| = 100 sigma inside a group (what the fp16-fix VAE / deep SDXL activations look like): E[x^2] - mean^2 in
    fp32 loses the variance to cancellation; the kernel must use a shifted / two-pass form like torch does."""
    need_gpu()
    import hip_ops as H
    import torch.nn.functional as F
    out = {}
    for (N, C, Hh, Ww, mean, sig) in ((2, 320, 64, 64, 100.0, 1.0), (2, 1280, 16, 16, -50.0, 0.5), (1, 640, 32, 32, 200.0, 2.0),
                                      (2, 1280, 8, 8, 30.0, 0.25)):
        g = torch.Generator().manual_seed(C)
        x = (torch.randn((N, C, Hh, Ww), generator=g) * sig + mean).half().float()
        gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        ref = F.silu(F.group_norm(x.double(), 32, gam.double(), bet.double(), 1e-5)).float()
        got = H.from_pn(H.groupnorm(H.to_pn(x), None, gam.to(H.DEV), bet.to(H.DEV), 32, 1e-5, 1))
        st = H.err_stats(got, ref)
        out[f"{C}x{Hh}"] = st
        record("groupnorm_large_mean", C=C, hw=Hh, mean=mean, sigma=sig, **st)
        assert st["finite"] and st["rel_l2"] < 6e-4, (C, Hh, st)       # measured 2.1e-4 (fp16 output rounding)
