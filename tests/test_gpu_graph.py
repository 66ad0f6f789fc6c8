"""-m gpu: whole-loop hipGraph replay (include/cfgpp.h: cfgpp_sample_graph_ddim) against the eager loop of
the SAME solver on the SAME engine: latents must be bit-identical - the captured step takes its scalars {t, c1..c4} from a device
table instead of kernel arguments, nothing else differs.  Covers the reference loops latent_diffusion.py:653-674 (ddim_cfg++),
272-294 (ddim), 160-182 / 888-910 (inversion, fp16 latents) and latent_sdxl.py:730-752, 838-858 (SDXL, Lightning lambda == 1)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


def _both(monkeypatch, run):
    monkeypatch.setenv("CFGPP_GRAPH", "0")
    eager = run()
    monkeypatch.setenv("CFGPP_GRAPH", "1")
    graph = run()
    again = run()                     # second call: the cached graph, a fresh step table and counter
    return eager, graph, again


def _same(a, b):
    return all(x.dtype == y.dtype and torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("name,lam", [("ddim_cfg++", 0.6), ("ddim", 7.5)])
def test_sd_ddim_graph_replay_is_bit_identical(monkeypatch, name, lam):
    need_gpu()
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD as cfg
    s = get_solver(name, solver_config=types.SimpleNamespace(num_sampling=7), device="cuda", unet_config=cfg, max_batch=2)
    uc, c = s.get_text_embed("bad", ["a cat", "a dog"])

    def run():
        return [t.clone() for t in s.sample(cfg_guidance=lam, prompt_embeds=(uc, c), seeds=[5, 6], return_latents=True)]
    eager, graph, again = _both(monkeypatch, run)
    assert eager[0].dtype == torch.float32 and torch.isfinite(eager[0]).all()
    assert _same(eager, graph) and _same(eager, again)


def test_graph_replay_follows_new_prompts_seeds_and_step_counts(monkeypatch):
    """one engine, one captured graph: other embeddings (set_context again), other seeds and another NFE (longer table) all go
    through the same buffers and must match the eager loop each time"""
    need_gpu()
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD as cfg
    for nfe, prompts, seeds in ((4, ["a", "b"], [1, 2]), (9, ["c", "d"], [3, 4]), (4, ["e", "f"], [7, 8])):
        s = get_solver("ddim_cfg++", solver_config=types.SimpleNamespace(num_sampling=nfe), device="cuda", unet_config=cfg, max_batch=2)
        uc, c = s.get_text_embed("bad", prompts)

        def run():
            return [t.clone() for t in s.sample(cfg_guidance=0.6, prompt_embeds=(uc, c), seeds=seeds, return_latents=True)]
        eager, graph, again = _both(monkeypatch, run)
        assert _same(eager, graph) and _same(eager, again), (nfe, prompts)


def test_sd_inversion_graph_replay_fp16_latents(monkeypatch):
    """ddim_inversion_cfg++ from a latent: fp16 latent chain, inversion (tweedie with eps_uc) then regeneration"""
    need_gpu()
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD as cfg
    s = get_solver("ddim_inversion_cfg++", solver_config=types.SimpleNamespace(num_sampling=6), device="cuda", unet_config=cfg, max_batch=2)
    uc, c = s.get_text_embed("bad", ["a cat", "a dog"])
    g = torch.Generator().manual_seed(4)
    z0 = (torch.randn((2, 4, 16, 16), generator=g) * 0.8).half()

    def run():
        return [t.clone() for t in s.sample(src_latent=z0.cuda(), cfg_guidance=0.6, prompt_embeds=(uc, c), return_latents=True)]
    eager, graph, again = _both(monkeypatch, run)
    assert eager[0].dtype == torch.float16 and torch.isfinite(eager[0].float()).all()
    assert _same(eager, graph) and _same(eager, again)


@pytest.mark.parametrize("name,lam,nfe", [("ddim_cfg++", 0.6, 5), ("ddim_cfg++_lightning", 1.0, 4), ("ddim", 5.0, 5)])
def test_sdxl_ddim_graph_replay_is_bit_identical(monkeypatch, name, lam, nfe):
    need_gpu()
    from cfgpp_amd.latent_sdxl import get_solver
    from cfgpp_amd.unet_config import TINY_XL as cfg
    s = get_solver(name, solver_config=types.SimpleNamespace(num_sampling=nfe), device="cuda", unet_config=cfg, max_batch=2)
    p = ["a cat", "a dog"]
    pe = s.get_text_embed("bad", p, "bad", p)

    def run():
        out = s.sample(prompt_embeds=pe, cfg_guidance=lam, target_size=(128, 128), original_size=(128, 128), seeds=[1, 2],
                       return_latents=True)
        return [out.clone()]
    eager, graph, again = _both(monkeypatch, run)
    assert torch.isfinite(eager[0].float()).all()
    assert _same(eager, graph) and _same(eager, again)


def test_graph_replay_with_a_callback_stays_eager(monkeypatch):
    """callbacks need the host between steps: with one, the loop must run eagerly even when the switch is on"""
    need_gpu()
    from cfgpp_amd.latent_diffusion import get_solver
    from cfgpp_amd.unet_config import TINY_SD as cfg
    monkeypatch.setenv("CFGPP_GRAPH", "1")
    s = get_solver("ddim_cfg++", solver_config=types.SimpleNamespace(num_sampling=3), device="cuda", unet_config=cfg, max_batch=1)
    uc, c = s.get_text_embed("bad", ["a cat"])
    seen = []

    def cb(step, t, kw):
        seen.append((step, int(t)))
        return kw
    s.sample(cfg_guidance=0.6, prompt_embeds=(uc, c), seeds=[5], callback_fn=cb, return_latents=True)
    assert [k for k, _ in seen] == [0, 1, 2]
