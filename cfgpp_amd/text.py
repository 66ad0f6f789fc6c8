"""CLIP text tower on libcfgpp_hip.so (``csrc/text.hip``): same call contract as ``conditioning.ClipTextTower`` -
``tower(prompts, clip_skip=None) -> (hidden [n,77,D] fp16, pooled [n,P] fp16 | None)`` - without torch ops.

What ``checkpoint.solver_kwargs_from_dir`` (``--model_dir`` of the example CLIs) builds on a GPU device; also usable directly
(``get_solver(..., text_encoder=HipClipTextTower.from_dir(...))``).
``tests/test_gpu_text.py`` compares it with ``transformers.CLIPTextModel(WithProjection)`` on the same weights (CLIP-L and
OpenCLIP-bigG widths, both activations, penultimate / clip_skip outputs, pooled projection).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import List, Optional, Tuple

import torch

from . import _lib
from ._lib import CfgppError, check


class HipClipTextTower:
    def __init__(self, vocab: int, hidden: int, layers: int, heads: int, intermediate: int, act: str, proj_dim: Optional[int],
                 state_dict, tokenizer, penultimate: bool, max_batch: int = 8, device=None):
        """``state_dict``: ``transformers`` CLIPTextModel(WithProjection) tensors (dict / iterable of pairs / safetensors path);
        ``tokenizer``: ``prompts -> LongTensor [n, 77]`` with an ``EOS`` id (``ClipBpeTokenizer`` / ``HashTokenizer``);
        ``penultimate``: SDXL towers return ``hidden_states[-2]``, the SD1.5 tower ``last_hidden_state``."""
        if not torch.cuda.is_available():
            raise CfgppError("HipClipTextTower needs a ROCm GPU; the HIP path has no CPU fallback")
        if act not in ("quick_gelu", "gelu"):
            raise CfgppError(f"HipClipTextTower: activation '{act}' (quick_gelu | gelu)")
        self.lib = _lib.load()
        dev = torch.device(device if device is not None else "cuda")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.hidden, self.layers, self.proj, self.max_batch = int(hidden), int(layers), int(proj_dim or 0), int(max_batch)
        self.tok, self.penultimate = tokenizer, bool(penultimate)
        self._h = self.lib.cfgpp_text_create(int(vocab), self.hidden, self.layers, int(heads), int(intermediate),
                                             0 if act == "quick_gelu" else 1, self.proj, self.max_batch, self.device.index)
        if not self._h:
            raise CfgppError("cfgpp_text_create failed: " + _lib.last_error())
        if isinstance(state_dict, str):
            from .weights import load_safetensors_iter
            state_dict = load_safetensors_iter(state_dict)
        items = state_dict.items() if isinstance(state_dict, dict) else state_dict
        for k, v in items:
            # checkpoint files (and transformers 4.x modules) prefix the tower's tensors with "text_model."; the bare
            # CLIPTextModel of transformers 5.x does not
            if not k.startswith("text_model.") and k != "text_projection.weight":
                k = "text_model." + k
            if k.endswith("position_ids") or (k == "text_projection.weight" and not self.proj):
                continue
            t = v.detach().cpu().contiguous()
            dt = 1 if t.dtype == torch.float16 else 0
            if dt == 0:
                t = t.to(torch.float32)
            shape = (C.c_long * t.dim())(*t.shape)
            check(self.lib.cfgpp_text_load_tensor(self._h, k.encode(), t.data_ptr(), dt, shape, t.dim()), f"cfgpp_text_load_tensor({k})")
        check(self.lib.cfgpp_text_finalize(self._h), "cfgpp_text_finalize")

    @classmethod
    def from_dir(cls, encoder_dir, tokenizer_dir, penultimate: bool, with_projection: bool, pad_token: Optional[str] = None,
                 max_batch: int = 8, device=None):
        """the HIP twin of ``ClipTextTower.from_dir`` (same directory layout and arguments)"""
        from .conditioning import ClipBpeTokenizer
        with open(os.path.join(str(encoder_dir), "config.json"), "r") as f:
            cfg = json.load(f)
        tok = ClipBpeTokenizer(os.path.join(str(tokenizer_dir), "vocab.json"), os.path.join(str(tokenizer_dir), "merges.txt"),
                               pad_token=pad_token)
        return cls(cfg["vocab_size"], cfg["hidden_size"], cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["intermediate_size"],
                   cfg.get("hidden_act", "quick_gelu"), cfg.get("projection_dim") if with_projection else None,
                   next((f for f in (os.path.join(str(encoder_dir), n) for n in ("model.safetensors", "model.fp16.safetensors"))
                         if os.path.exists(f)), os.path.join(str(encoder_dir), "model.safetensors")),
                   tok, penultimate, max_batch=max_batch, device=device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
            self.lib.cfgpp_text_destroy(h)
            self._h = None

    @torch.no_grad()
    def __call__(self, prompts: List[str], clip_skip: Optional[int] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        ids = self.tok(list(prompts)).to(torch.int32)
        n = ids.shape[0]
        # the pooled row: first EOS of every prompt (transformers: ids == eos_token_id, first hit)
        eos = (ids == int(self.tok.EOS)).to(torch.int32).argmax(dim=-1).to(torch.int32)
        if clip_skip is not None:
            if not self.penultimate:
                raise NotImplementedError("clip_skip is defined for the SDXL (penultimate-layer) towers only")
            layer = self.layers - 1 - int(clip_skip)          # hidden_states[-(clip_skip + 2)] of L + 1 states
            if layer < 0:
                raise ValueError(f"clip_skip={clip_skip} reaches below the embeddings of a {self.layers}-layer tower")
        else:
            layer = self.layers - 1 if self.penultimate else -1
        hs = torch.empty((n, 77, self.hidden), dtype=torch.float16, device=self.device)
        pooled = torch.empty((n, self.proj), dtype=torch.float32, device=self.device) if self.proj else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for s in range(0, n, self.max_batch):
            e = min(s + self.max_batch, n)
            idc = ids[s:e].contiguous()
            eoc = eos[s:e].contiguous()
            check(self.lib.cfgpp_text_encode(self._h, C.cast(idc.data_ptr(), C.POINTER(C.c_int)), C.cast(eoc.data_ptr(), C.POINTER(C.c_int)),
                                             e - s, layer, hs[s:e].data_ptr(), None if pooled is None else pooled[s:e].data_ptr(), stream),
                  "cfgpp_text_encode")
        return hs, (None if pooled is None else pooled.to(torch.float16))
