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
