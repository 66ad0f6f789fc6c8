"""Boundary inputs of the hot path: text embeddings.

The reference runs CLIP text encoders once per prompt
(latent_diffusion.py:92-115, latent_sdxl.py:76-128); that is OFF the per-step
path, and neither tokenizer files nor encoder weights exist in this
environment.  The solver therefore takes any callable
``encode(list[str]) -> (hidden [n,77,D], pooled [n,P] | None)``; the default is
a deterministic synthetic encoder (seeded by the prompt text) so that the whole
loop can run and be benchmarked on synthetic prompts.  Pre-computed embeddings
can be passed to ``sample(prompt_embeds=...)`` to bypass it entirely.
"""
from __future__ import annotations

import hashlib
from typing import List, Optional, Tuple

import torch


def _seed(text: str, tag: str) -> int:
    return int.from_bytes(hashlib.sha256((tag + "\x00" + text).encode("utf-8")).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF


class SyntheticTextEncoder:
    """Deterministic stand-in for CLIP: N(0, 0.5^2) hidden states per prompt string."""

    def __init__(self, hidden_dim: int, pooled_dim: Optional[int] = None, tokens: int = 77, tag: str = "clip"):
        self.hidden_dim, self.pooled_dim, self.tokens, self.tag = hidden_dim, pooled_dim, tokens, tag

    def __call__(self, prompts: List[str]) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        hs, pooled = [], []
        for p in prompts:
            g = torch.Generator().manual_seed(_seed(p, self.tag))
            hs.append(torch.randn((self.tokens, self.hidden_dim), generator=g) * 0.5)
            if self.pooled_dim:
                pooled.append(torch.randn((self.pooled_dim,), generator=g) * 0.5)
        h = torch.stack(hs).to(torch.float16)
        return h, (torch.stack(pooled).to(torch.float16) if self.pooled_dim else None)


def as_list(p) -> List[str]:
    return [p] if isinstance(p, str) else list(p)


class HashTokenizer:
    """Vocabulary-free stand-in for the CLIP BPE tokenizer (no vocab.json / merges.txt exist offline): lower-cased
    word pieces hashed into the 49 406 ordinary ids, BOS 49406 / EOS 49407, padded to 77 with ``pad_id``
    (EOS for CLIP-L, 0 for OpenCLIP-bigG, as the SDXL tokenizers do).  Same token layout as the real one -
    [BOS, words..., EOS, pad...] - so the towers below see realistic sequences and the EOS pooling works."""

    BOS, EOS, VOCAB = 49406, 49407, 49408

    def __init__(self, pad_id: Optional[int] = None, length: int = 77):
        self.pad_id = self.EOS if pad_id is None else pad_id
        self.length = length

    def __call__(self, prompts: List[str]) -> torch.Tensor:
        import re
        out = torch.full((len(prompts), self.length), self.pad_id, dtype=torch.long)
        for i, p in enumerate(prompts):
            words = re.findall(r"[a-z0-9]+|[^\sa-z0-9]", p.lower())[: self.length - 2]
            ids = [self.BOS] + [_seed(w, "tok") % 49406 for w in words] + [self.EOS]
            out[i, : len(ids)] = torch.tensor(ids)
        return out


class ClipTextTower:
    """A CLIP text transformer on torch ops (``transformers`` architecture): the reference's
    ``text_encoder(tokens)[0]`` (SD1.5, latent_diffusion.py:105-113) or ``hidden_states[-2]`` + projected pooled
    output (SDXL, latent_sdxl.py:76-93).  Runs once per prompt, off the per-step path, so it stays on
    torch-ROCm ops.  Weights: a ``transformers`` state dict / safetensors path, or seeded random init (no
    checkpoint exists offline).  Opt-in: ``get_solver(..., text_encoder=ClipTextTower.clip_l())``.

    ``encode(list[str]) -> (hidden [n,77,D] fp16, pooled [n,P] fp16 | None)`` like every text encoder here."""

    def __init__(self, hidden: int, layers: int, heads: int, intermediate: int, act: str, proj_dim: Optional[int],
                 penultimate: bool, pad_id: Optional[int], device="cpu", dtype=torch.float32, weights=None, seed: int = 0):
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
        cfg = CLIPTextConfig(vocab_size=HashTokenizer.VOCAB, hidden_size=hidden, intermediate_size=intermediate,
                             num_hidden_layers=layers, num_attention_heads=heads, max_position_embeddings=77,
                             hidden_act=act, projection_dim=proj_dim or hidden, bos_token_id=HashTokenizer.BOS,
                             eos_token_id=HashTokenizer.EOS, pad_token_id=1)
        torch.manual_seed(seed)
        self.model = (CLIPTextModelWithProjection(cfg) if proj_dim else CLIPTextModel(cfg)).eval()
        if weights is not None:
            if isinstance(weights, str):
                from .weights import load_safetensors_iter
                weights = dict(load_safetensors_iter(weights))
            self.model.load_state_dict(weights)
        self.model.to(device=device, dtype=dtype)
        self.tok = HashTokenizer(pad_id)
        self.penultimate, self.proj = penultimate, bool(proj_dim)
        self.device = device

    @classmethod
    def clip_l(cls, **kw):        # SD1.5 text encoder / SDXL text_encoder (12 layers, 768 wide)
        kw.setdefault("penultimate", False)
        return cls(768, kw.pop("layers", 12), 12, 3072, "quick_gelu", None, kw.pop("penultimate"), None, **kw)

    @classmethod
    def open_clip_bigg(cls, **kw):  # SDXL text_encoder_2 (32 layers, 1280 wide, projected pooled output)
        return cls(1280, kw.pop("layers", 32), 20, 5120, "gelu", 1280, True, 0, **kw)

    @torch.no_grad()
    def __call__(self, prompts: List[str], clip_skip: Optional[int] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """``clip_skip`` (SDXL only): ``hidden_states[-(clip_skip + 2)]`` instead of ``[-2]`` (latent_sdxl.py:88-92)."""
        ids = self.tok(prompts).to(self.device)
        out = self.model(input_ids=ids, output_hidden_states=True)
        if clip_skip is not None:
            if not self.penultimate:
                raise NotImplementedError("clip_skip is defined for the SDXL (penultimate-layer) towers only")
            hs = out.hidden_states[-(int(clip_skip) + 2)]
        else:
            hs = out.hidden_states[-2] if self.penultimate else out.last_hidden_state
        pooled = out.text_embeds if self.proj else None
        return hs.to(torch.float16), (None if pooled is None else pooled.to(torch.float16))
