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


def _bytes_to_unicode() -> dict:
    """the byte <-> printable-character table of byte-level BPE (GPT-2 / CLIP): printable latin-1 bytes map to
    themselves, the other 68 bytes to code points 256 .."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, extra = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + extra)
            extra += 1
    return table


class ClipBpeTokenizer:
    """The CLIP byte-level BPE tokenizer from its two files (``vocab.json``: token -> id, ``merges.txt``: one merge per
    line after the ``#version`` header) - what ``CLIPTokenizer.from_pretrained(model_key, subfolder="tokenizer")`` loads in
    the reference (latent_diffusion.py:92-104, latent_sdxl.py:56-75).  No vocabulary ships offline, so the solvers default
    to ``HashTokenizer``; with the files of a real checkpoint this class gives the real ids.

    Algorithm (pinned against ``transformers.CLIPTokenizer`` on the same files in tests/test_capi_and_config.py):
    NFC -> runs of whitespace to one space -> lower case; split with CLIP's pattern (special tokens, English
    contractions, letter runs, single digits, runs of other symbols); UTF-8 bytes -> printable characters; greedy
    lowest-rank pair merging with ``</w>`` on the last character; unknown pieces -> ``<|endoftext|>``; ``[BOS] + ids[:75] +
    [EOS]`` padded to 77 with the pad token (EOS for CLIP-L, "!" = id 0 for SDXL's second tokenizer)."""

    PATTERN = r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"

    def __init__(self, vocab, merges, pad_token: Optional[str] = None, length: int = 77):
        """``pad_token``: None = pad with ``<|endoftext|>`` (CLIP-L); ``"!"`` for SDXL's second tokenizer.  A pad token that
        is an ordinary vocabulary entry is also a SPECIAL token of the HF tokenizer: every occurrence in a prompt is cut out
        before the BPE and mapped to its own id ("!!" -> [0, 0], not ["!", "!</w>"]) - a quirk of that tokenizer kept here."""
        import json
        import regex
        if isinstance(vocab, (str, bytes)) or hasattr(vocab, "__fspath__"):
            with open(vocab, "r", encoding="utf-8") as f:
                vocab = json.load(f)
        if isinstance(merges, (str, bytes)) or hasattr(merges, "__fspath__"):
            with open(merges, "r", encoding="utf-8") as f:
                lines = f.read().strip().split("\n")
            merges = [ln for ln in lines if ln and not ln.startswith("#version")]
        self.vocab = dict(vocab)
        self.ranks = {tuple(m.split(" ")) if isinstance(m, str) else tuple(m): i for i, m in enumerate(merges)}
        self.BOS, self.EOS = self.vocab["<|startoftext|>"], self.vocab["<|endoftext|>"]
        self.VOCAB = max(self.vocab.values()) + 1
        self.pad_id = self.EOS if pad_token is None else self.vocab[pad_token]
        self.length = length
        self._b2u = _bytes_to_unicode()
        self._pat = regex.compile(self.PATTERN)
        specials = ["<|startoftext|>", "<|endoftext|>"] + ([pad_token] if pad_token not in (None, "<|startoftext|>", "<|endoftext|>") else [])
        self._special = regex.compile("(" + "|".join(regex.escape(t) for t in sorted(specials, key=len, reverse=True)) + ")")
        self._special_set = set(specials)
        self._cache = {}

    def _bpe(self, piece: str) -> List[str]:
        if piece in self._cache:
            return self._cache[piece]
        word = list(piece[:-1]) + [piece[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for a, b in zip(word[:-1], word[1:]):
                r = self.ranks.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i + 1 < len(word) and (word[i], word[i + 1]) == best:
                    merged.append(word[i] + word[i + 1]); i += 2
                else:
                    merged.append(word[i]); i += 1
            word = merged
        self._cache[piece] = word
        return word

    def encode(self, text: str) -> List[int]:
        """ids of one prompt without BOS / EOS / padding"""
        import re
        import unicodedata
        ids = []
        for seg in self._special.split(text):               # special tokens are cut out of the RAW text
            if seg in self._special_set:
                ids.append(self.vocab[seg])
                continue
            seg = re.sub(r"\s+", " ", unicodedata.normalize("NFC", seg)).lower()
            for piece in self._pat.findall(seg):
                mapped = "".join(self._b2u[b] for b in piece.encode("utf-8"))
                ids.extend(self.vocab.get(tok, self.EOS) for tok in self._bpe(mapped))
        return ids

    def __call__(self, prompts: List[str]) -> torch.Tensor:
        out = torch.full((len(prompts), self.length), self.pad_id, dtype=torch.long)
        for i, p in enumerate(prompts):
            ids = [self.BOS] + self.encode(p)[: self.length - 2] + [self.EOS]
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
                 penultimate: bool, pad_id: Optional[int], device="cpu", dtype=torch.float32, weights=None, seed: int = 0,
                 tokenizer=None):
        """``tokenizer``: ``prompts -> LongTensor [n, 77]`` with ``BOS`` / ``EOS`` / ``VOCAB`` attributes - a
        ``ClipBpeTokenizer`` built from a checkpoint's vocab.json / merges.txt; default: the vocabulary-free HashTokenizer."""
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
        tok = tokenizer if tokenizer is not None else HashTokenizer(pad_id)
        cfg = CLIPTextConfig(vocab_size=tok.VOCAB, hidden_size=hidden, intermediate_size=intermediate,
                             num_hidden_layers=layers, num_attention_heads=heads, max_position_embeddings=77,
                             hidden_act=act, projection_dim=proj_dim or hidden, bos_token_id=tok.BOS,
                             eos_token_id=tok.EOS, pad_token_id=1)
        torch.manual_seed(seed)
        self.model = (CLIPTextModelWithProjection(cfg) if proj_dim else CLIPTextModel(cfg)).eval()
        if weights is not None:
            if isinstance(weights, str):
                from .weights import load_safetensors_iter
                weights = dict(load_safetensors_iter(weights))
            self.model.load_state_dict(weights)
        self.model.to(device=device, dtype=dtype)
        self.tok = tok
        self.penultimate, self.proj = penultimate, bool(proj_dim)
        self.device = device

    @classmethod
    def from_dir(cls, encoder_dir, tokenizer_dir, penultimate: bool, with_projection: bool, pad_token: Optional[str] = None,
                 device="cpu", dtype=torch.float32):
        """A tower from a diffusers-layout checkpoint: ``encoder_dir`` holds ``config.json`` + ``model.safetensors`` (the
        ``text_encoder`` / ``text_encoder_2`` subfolder), ``tokenizer_dir`` holds ``vocab.json`` + ``merges.txt`` (``tokenizer``
        / ``tokenizer_2``).  SD1.5: ``penultimate=False, with_projection=False``; SDXL: tower 1 ``penultimate=True,
        with_projection=False``, tower 2 ``penultimate=True, with_projection=True, pad_token="!"``
        (reference latent_diffusion.py:92-113, latent_sdxl.py:56-93)."""
        import os
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
        from .weights import load_safetensors_iter
        self = cls.__new__(cls)
        cfg = CLIPTextConfig.from_json_file(os.path.join(str(encoder_dir), "config.json"))
        self.model = (CLIPTextModelWithProjection(cfg) if with_projection else CLIPTextModel(cfg)).eval()
        wfile = next((f for f in (os.path.join(str(encoder_dir), n) for n in ("model.safetensors", "model.fp16.safetensors")) if os.path.exists(f)),
                     os.path.join(str(encoder_dir), "model.safetensors"))       # the fp16 variant is the common SDXL download
        sd = dict(load_safetensors_iter(wfile))
        sd.pop("text_model.embeddings.position_ids", None)         # a buffer older checkpoints still carry
        self.model.load_state_dict(sd)
        self.model.to(device=device, dtype=dtype)
        self.tok = ClipBpeTokenizer(os.path.join(str(tokenizer_dir), "vocab.json"), os.path.join(str(tokenizer_dir), "merges.txt"),
                                    pad_token=pad_token)
        if self.tok.VOCAB > cfg.vocab_size:
            raise ValueError(f"tokenizer has {self.tok.VOCAB} ids but the encoder embeds only {cfg.vocab_size}")
        self.penultimate, self.proj, self.device = penultimate, bool(with_projection), device
        return self

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
