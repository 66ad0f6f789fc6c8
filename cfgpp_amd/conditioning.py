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
