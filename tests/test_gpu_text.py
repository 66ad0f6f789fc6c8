"""First-contact test of the CLIP text tower on the HIP kernels (csrc/text.hip, cfgpp_amd/text.py) against
`transformers` - the library the reference's text encoders come from - on the same weights and token ids.

First run on hardware in round 3 (4 / 4 parametrisations green); the tower stays opt-in for the solvers
(``text_encoder=HipClipTextTower.from_dir(...)``)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("hidden,heads,layers,inter,act,proj", [(128, 2, 3, 256, "quick_gelu", None), (128, 2, 3, 256, "gelu", 64),
                                                                (768, 12, 12, 3072, "quick_gelu", None), (1280, 20, 4, 5120, "gelu", 1280)])
def test_hip_text_tower_vs_transformers(hidden, heads, layers, inter, act, proj):
    _need()
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from cfgpp_amd.conditioning import HashTokenizer
    from cfgpp_amd.text import HipClipTextTower
    tok = HashTokenizer(0 if proj else None)
    cfg = CLIPTextConfig(vocab_size=tok.VOCAB, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                         num_attention_heads=heads, max_position_embeddings=77, hidden_act=act, projection_dim=proj or hidden,
                         bos_token_id=tok.BOS, eos_token_id=tok.EOS, pad_token_id=1)
    torch.manual_seed(3)
    ref = (CLIPTextModelWithProjection(cfg) if proj else CLIPTextModel(cfg)).eval()
    sd = {k: v.half() for k, v in ref.state_dict().items()}           # the engine stores fp16 matrices: compare like with like
    ref.load_state_dict({k: v.float() for k, v in sd.items()})
    prompts = ["a photo of a cat", "", "two dogs playing in the park, photorealistic 4k " * 3, "low quality,jpeg artifacts,blurry"]
    ids = tok(prompts)
    with torch.no_grad():
        out = ref(input_ids=ids, output_hidden_states=True)
    for penultimate in (False, True):
        tower = HipClipTextTower(tok.VOCAB, hidden, layers, heads, inter, act, proj, sd, tok, penultimate=penultimate, max_batch=3)
        hs, pooled = tower(prompts)
        want = out.hidden_states[-2] if penultimate else out.last_hidden_state
        assert hs.shape == (len(prompts), 77, hidden) and hs.dtype == torch.float16 and torch.isfinite(hs.float()).all()
        assert _rel(hs, want) < 4e-3, (penultimate, _rel(hs, want))
        if proj:
            assert pooled.shape == (len(prompts), proj) and _rel(pooled, out.text_embeds) < 4e-3, _rel(pooled, out.text_embeds)
        else:
            assert pooled is None
        if penultimate:
            assert _rel(tower(prompts, clip_skip=1)[0], out.hidden_states[-3]) < 4e-3
            assert _rel(tower(prompts, clip_skip=layers - 1)[0], out.hidden_states[0]) < 2e-3      # the embeddings themselves
        del tower


def test_checkpoint_directory_takes_the_hip_text_towers(tmp_path):
    """``checkpoint.solver_kwargs_from_dir`` on a GPU device (what ``--model_dir`` of the example CLIs calls) builds
    ``HipClipTextTower``s; through the SDXL solver's ``get_text_embed`` (latent_sdxl.py:76-128: hidden_states[-2] of both towers
    concatenated, pooled output of the second, "!" padding of the second tokenizer) they reproduce `transformers` on the files
    of a small diffusers-layout checkpoint directory."""
    _need()
    import json
    import types
    from safetensors.torch import save_file
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection, CLIPTokenizer
    from test_capi_and_config import _toy_clip_vocab
    from cfgpp_amd.checkpoint import solver_kwargs_from_dir
    from cfgpp_amd.latent_sdxl import get_solver
    from cfgpp_amd.text import HipClipTextTower
    from cfgpp_amd.unet_config import TINY_XL
    vocab, merges = _toy_clip_vocab()
    refs = {}
    for enc, tokd, hidden, heads, proj in (("text_encoder", "tokenizer", 128, 2, None), ("text_encoder_2", "tokenizer_2", 192, 3, 64)):
        (tmp_path / tokd).mkdir()
        (tmp_path / tokd / "vocab.json").write_text(json.dumps(vocab, ensure_ascii=False), encoding="utf-8")
        (tmp_path / tokd / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
        cfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=3, num_attention_heads=heads,
                             max_position_embeddings=77, hidden_act="gelu" if proj else "quick_gelu", projection_dim=proj or hidden,
                             bos_token_id=vocab["<|startoftext|>"], eos_token_id=vocab["<|endoftext|>"], pad_token_id=1)
        torch.manual_seed(11 + hidden)
        ref = (CLIPTextModelWithProjection(cfg) if proj else CLIPTextModel(cfg)).eval()
        sd = {k: v.half() for k, v in ref.state_dict().items()}
        ref.load_state_dict({k: v.float() for k, v in sd.items()})
        (tmp_path / enc).mkdir()
        cfg.to_json_file(str(tmp_path / enc / "config.json"))
        save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / enc / "model.safetensors"))
        refs[enc] = ref
    kw, missing = solver_kwargs_from_dir(tmp_path, sdxl=True, device="cuda")
    assert all(isinstance(t, HipClipTextTower) for t in kw["text_encoder"]), kw
    assert "text_encoder" not in missing and "text_encoder_2" not in missing
    cfg_xl = TINY_XL
    solver = get_solver("ddim_cfg++", solver_config=types.SimpleNamespace(num_sampling=2), device="cuda", unet_config=cfg_xl, max_batch=2,
                        latent_hw=(16, 16), text_encoder=kw["text_encoder"])
    prompts = ["a photo of a cat", "two dogs playing in the park!"]
    null_e, pos_e, null_p, pos_p = solver.get_text_embed("low quality", prompts, "low quality", prompts)
    tok1 = CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges])
    tok2 = CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges], pad_token="!")
    with torch.no_grad():
        outs = []
        for plist in (["low quality"], prompts):
            o1 = refs["text_encoder"](input_ids=tok1(plist, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids,
                                      output_hidden_states=True)
            o2 = refs["text_encoder_2"](input_ids=tok2(plist, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids,
                                        output_hidden_states=True)
            outs.append((torch.cat([o1.hidden_states[-2], o2.hidden_states[-2]], dim=-1), o2.text_embeds))
    assert pos_e.shape == (2, 77, 128 + 192) and pos_p.shape == (2, 64)
    assert _rel(null_e, outs[0][0]) < 4e-3 and _rel(pos_e, outs[1][0]) < 4e-3, (_rel(null_e, outs[0][0]), _rel(pos_e, outs[1][0]))
    assert _rel(null_p, outs[0][1]) < 4e-3 and _rel(pos_p, outs[1][1]) < 4e-3
