"""CPU: the C-ABI library loads and exports every symbol include/cfgpp.h declares; UNet config
tables reproduce the published parameter totals; FLOP model matches SURVEY.md 8(d)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(path):
    hdr = open(path).read()
    return set(re.findall(r"\b(cfgpp_[a-z0-9_]+)\s*\(", hdr)) - {"cfgpp_unet_config"}


def test_library_exports_every_declared_symbol():
    """the installed header (the drop-in boundary) and the debug header (test hooks, A/B switches) are checked separately:
    every declaration is exported and has a ctypes prototype in ITS table, and the boundary carries no hook or switch"""
    from cfgpp_amd import _lib
    from cfgpp_amd.build import build
    build(verbose=False)
    lib = _lib.load()
    public = _declared(os.path.join(ROOT, "include", "cfgpp.h"))
    debug = _declared(os.path.join(ROOT, "cfgpp_amd", "csrc", "cfgpp_debug.h")) - {"cfgpp_last_error"}
    assert public and debug, "no declarations parsed"
    for name in sorted(public):
        assert hasattr(lib, name), f"{name} declared in include/cfgpp.h but not exported"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
    for name in sorted(debug):
        assert hasattr(lib, name), f"{name} declared in cfgpp_debug.h but not exported"
        assert name in _lib.DEBUG_PROTOTYPES, f"{name} has no ctypes prototype"
    assert set(_lib.PROTOTYPES) == public and set(_lib.DEBUG_PROTOTYPES) == debug
    assert not (public & debug)
    hooks = [n for n in public if n.startswith("cfgpp_op_") or "_set_" in n.replace("cfgpp_unet_set_context", "") or "force" in n or "timeline" in n]
    assert hooks == [], f"test hooks / switches in the public header: {hooks}"
    assert len(public) <= 40
    assert _lib.last_error() == "" or isinstance(_lib.last_error(), str)


def test_shipped_library_is_built_from_these_sources():
    """provenance: the digest inside cfgpp_build_id() is the digest of the kernel sources + flags in THIS tree (build.py names its
    object files by the same digests, so neither a stale object nor a stale library can pass for the current one)"""
    from cfgpp_amd import _lib
    from cfgpp_amd import build as B
    B.build(verbose=False)
    bid = _lib.build_id()
    assert bid == B.library_build_id(), (bid, B.library_build_id())
    tag, digest, head = bid.split(":")
    assert tag == "cfgpp-build" and digest == B.source_digest() and head
    objs = [f for f in os.listdir(B.OBJ) if f.endswith(".o")]
    assert len(objs) == len(B.SOURCES), f"stale or missing objects: {sorted(objs)}"


def test_shipped_kernels_have_no_spills_and_stay_in_their_register_budgets():
    """the gfx950 code objects inside the built library: no kernel spills VGPRs or uses scratch, and the kernels whose occupancy
    DESIGN.md counts on stay inside the register budget that occupancy needs (read from the ELF notes, scripts/kernel_resources.py)"""
    import subprocess
    import sys
    from cfgpp_amd.build import build
    build(verbose=False)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import kernel_resources as KR
    if not os.path.exists(KR.READELF):
        pytest.skip("llvm-readelf not found")
    ks = KR.kernels(os.path.join(ROOT, "cfgpp_amd", "libcfgpp_hip.so"))
    assert len(ks) > 100, f"only {len(ks)} kernels found in the code objects"
    bad = [k["name"] for k in ks if k["spill"] != "0" or k["scratch"] != "0"]
    assert bad == [], f"kernels with VGPR spills or scratch: {bad}"
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in ks), capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for k, name in zip(ks, names):
        v = int(k["vgpr"])                       # gfx950: ONE register file - .vgpr_count is the unified total and already includes .agpr_count
        assert int(k["agpr"]) <= v, (name, k)
        if "::attn64_kernel<" in name:         # four workgroups per CU = four waves per SIMD
            assert v <= 128, (name, v); seen += 1
        if "::tile32_kernel<4, 2, 64, 64, 3, 4" in name:      # 256 x 128 tile, two 8-wave workgroups per CU = four waves per SIMD
            assert v <= 128, (name, v); seen += 1
        if "::big4_kernel<4, 4," in name:          # 256 x 256 on four waves: 256 accumulator AGPRs + the loop's VGPRs, one wave per SIMD
            assert int(k["agpr"]) == 256 and v <= 512, (name, v); seen += 1
        assert v <= 512, (name, v)
    assert seen >= 8, "the occupancy-critical kernels were not found by name"


def test_every_implicit_gemm_kernel_is_counted_in_the_pmc_family():
    """scripts/pmc_summary.py folds rocprofv3 counters onto kernel families by NAME: every kernel of the shipped library that takes
    an IGemmArgs block (the implicit-GEMM family - a new kernel file must not fall out of the roofline's traffic / mfma_util again,
    as big4p_kernel did for one GPU call of round 6) has to map to "igemm" (or to the K-split reduce kernel)"""
    import subprocess
    import sys
    from cfgpp_amd.build import build
    build(verbose=False)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import kernel_resources as KR
    src = open(os.path.join(ROOT, "scripts", "pmc_summary.py")).read()        # (a script: importing it would parse sys.argv)
    ns = {}
    exec(src[src.index("def family(k):"):src.index("def load(d):")], ns)
    PS = type("PS", (), {"family": staticmethod(ns["family"])})
    if not os.path.exists(KR.READELF):
        pytest.skip("llvm-readelf not found")
    ks = KR.kernels(os.path.join(ROOT, "cfgpp_amd", "libcfgpp_hip.so"))
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in ks), capture_output=True, text=True).stdout.split("\n")
    gemm = [n for n in names if "(IGemmArgs" in n]
    assert len(gemm) > 40 and any("big4p_kernel" in n for n in gemm)
    wrong = [n for n in gemm if PS.family(n) not in ("igemm", "igemm_reduce")]
    assert wrong == [], wrong


def test_param_totals_match_published_sizes():
    from cfgpp_amd.unet_config import SD15, SDXL, param_count
    assert param_count(SD15) == 859_520_964       # 859.5 M
    assert param_count(SDXL) == 2_567_463_684     # 2567.5 M


def test_flop_model():
    from cfgpp_amd.unet_config import SD15, SDXL, unet_flops_per_row
    assert abs(unet_flops_per_row(SD15, 64, 64) / 1e12 - 0.800) < 0.01
    assert abs(unet_flops_per_row(SDXL, 128, 128) / 1e12 - 6.71) < 0.05


def test_synthetic_weights_are_deterministic_and_fp16_exact():
    import torch
    from cfgpp_amd.unet_config import TINY_SD, param_shapes
    from cfgpp_amd.weights import synth_state_dict
    a, b = synth_state_dict(TINY_SD, 0), synth_state_dict(TINY_SD, 0)
    assert list(a) == list(param_shapes(TINY_SD))
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], a[k].half().float())
    assert not torch.equal(a["conv_in.weight"], synth_state_dict(TINY_SD, 1)["conv_in.weight"])


def test_oracle_unet_runs_and_is_finite():
    import torch
    from cfgpp_amd.unet_config import TINY_XL as cfg
    from cfgpp_amd.weights import synth_state_dict
    from oracle.unet_ref import UNetRef
    net = UNetRef(cfg, synth_state_dict(cfg))
    out = net(torch.randn(2, 4, 16, 16), 500.0, torch.randn(2, 77, cfg.cross_attention_dim) * 0.5,
              {"text_embeds": torch.randn(1, cfg.addition_pooled_dim), "time_ids": torch.tensor([[128.0, 128, 0, 0, 128, 128]])})["sample"]
    assert out.shape == (2, 4, 16, 16) and torch.isfinite(out).all()


def test_safetensors_loader_streams_diffusers_keys(tmp_path):
    """weights.load_safetensors_iter (the real-checkpoint path of HipEngine / vae_weights=) yields every key
    of a diffusers-layout file unchanged."""
    import torch
    from safetensors.torch import save_file
    from cfgpp_amd.unet_config import TINY_SD, param_shapes
    from cfgpp_amd.weights import load_safetensors_iter, synth_tensor
    shapes = dict(list(param_shapes(TINY_SD).items())[:12])
    sd = {k: synth_tensor(k, s, 0).half() for k, s in shapes.items()}
    path = str(tmp_path / "unet.safetensors")
    save_file(sd, path)
    got = dict(load_safetensors_iter(path))
    assert set(got) == set(sd)
    for k in sd:
        assert got[k].dtype == torch.float16 and torch.equal(got[k], sd[k])


def test_clip_text_towers_shapes_and_determinism():
    """opt-in CLIP text towers (transformers architecture, random init, hash tokenizer): SD1.5 / SDXL contracts"""
    import torch
    from cfgpp_amd.conditioning import ClipTextTower, HashTokenizer
    ids = HashTokenizer()(["a photo of a cat", ""])
    assert ids.shape == (2, 77) and ids[0, 0] == 49406 and ids[1, 1] == 49407 and ids[0, 6] == 49407
    assert (HashTokenizer(pad_id=0)(["x"])[0, 3:] == 0).all()
    enc = ClipTextTower.clip_l(layers=1)
    hs, pooled = enc(["a photo of a cat", "a dog"])
    assert hs.shape == (2, 77, 768) and hs.dtype == torch.float16 and pooled is None and torch.isfinite(hs.float()).all()
    assert torch.equal(hs, ClipTextTower.clip_l(layers=1)(["a photo of a cat", "a dog"])[0])       # seeded init
    l_pen = ClipTextTower.clip_l(layers=2, penultimate=True)
    g = ClipTextTower.open_clip_bigg(layers=2)
    h1, _ = l_pen(["a cat"]); h2, p2 = g(["a cat"])
    assert torch.cat([h1, h2], -1).shape == (1, 77, 2048) and p2.shape == (1, 1280)


def _toy_clip_vocab(n_merges=120):
    """a small CLIP-layout vocabulary: 256 byte characters, the same with </w>, merges learnt by plain BPE on a toy
    corpus, then the two special tokens (the layout of the real 49 408-entry vocab.json / merges.txt)"""
    import collections
    from cfgpp_amd.conditioning import _bytes_to_unicode
    b2u = _bytes_to_unicode()
    corpus = ("a photo of a cat sitting on the sofa . two dogs playing in the park , photorealistic 4k "
              "an astronaut riding a horse on mars ; it's the painter's best work ! low quality jpeg artifacts blurry "
              "caf\u00e9 na\u00efve \u732b the cats' toys don't matter 1920s style ... ") * 3
    words = collections.Counter()
    import regex
    for piece in regex.findall(r"'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", corpus.lower()):
        mapped = [b2u[b] for b in piece.encode("utf-8")]
        mapped[-1] += "</w>"
        words[tuple(mapped)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w[:-1], w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        new_words = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            new_words[tuple(out)] += c
        words = new_words
    base = list(b2u.values())
    tokens = base + [c + "</w>" for c in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
    return {t: i for i, t in enumerate(tokens)}, merges


def test_clip_bpe_tokenizer_matches_transformers(tmp_path):
    """ClipBpeTokenizer (our restatement of CLIP's byte-level BPE) == transformers.CLIPTokenizer on the same vocabulary
    files: ids, truncation to 77, EOS- and "!"-padding (the two SDXL tokenizers)."""
    import json
    import torch
    from transformers import CLIPTokenizer
    from cfgpp_amd.conditioning import ClipBpeTokenizer, ClipTextTower
    vocab, merges = _toy_clip_vocab()
    (tmp_path / "vocab.json").write_text(json.dumps(vocab, ensure_ascii=False), encoding="utf-8")
    (tmp_path / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
    ours = ClipBpeTokenizer(tmp_path / "vocab.json", tmp_path / "merges.txt")
    ours_bang = ClipBpeTokenizer(vocab, [" ".join(m) for m in merges], pad_token="!")
    ref = CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges])
    ref_bang = CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges], pad_token="!")
    prompts = ["a photo of a cat", "", "  Two   DOGS playing,\tin the park!!  ", "it's the painter's best work; don't", "1920s caf\u00e9 na\u00efve \u732b ...",
               "low quality,jpeg artifacts,blurry,poorly drawn,ugly,worst quality,", "an astronaut " * 60, "<|endoftext|> a cat <|startoftext|>",
               "e\u0301 combining accent", "emoji \U0001f600 and symbols #$%&*"]
    want = ref(prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    got = ours(prompts)
    assert got.shape == (len(prompts), 77) and got.dtype == torch.long
    for i, p in enumerate(prompts):
        assert torch.equal(got[i], want[i]), (p, got[i].tolist()[:20], want[i].tolist()[:20])
    want_bang = ref_bang(prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    assert torch.equal(ours_bang(prompts), want_bang)
    assert got[0, 0] == ours.BOS and got[1, 1] == ours.EOS and got[6, 76] == ours.EOS          # empty prompt; truncated prompt ends with EOS
    # plugs into the text tower: vocabulary size and EOS pooling position come from the tokenizer
    tower = ClipTextTower.open_clip_bigg(layers=1, tokenizer=ours_bang)
    hs, pooled = tower(["a photo of a cat", "two dogs"])
    assert hs.shape == (2, 77, 1280) and pooled.shape == (2, 1280) and torch.isfinite(hs.float()).all()
    assert tower.model.config.vocab_size == len(vocab)


def test_clip_tower_from_checkpoint_directory(tmp_path):
    """ClipTextTower.from_dir: config.json + model.safetensors + vocab.json + merges.txt of a (tiny) diffusers-layout text
    encoder give exactly what transformers computes from the same files - last_hidden_state for SD1.5, hidden_states[-2] /
    [-(clip_skip+2)] and the projected pooled output for the SDXL towers."""
    import json
    import torch
    from safetensors.torch import save_file
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection, CLIPTokenizer
    from cfgpp_amd.conditioning import ClipTextTower
    vocab, merges = _toy_clip_vocab()
    tok_dir = tmp_path / "tokenizer"; tok_dir.mkdir()
    (tok_dir / "vocab.json").write_text(json.dumps(vocab, ensure_ascii=False), encoding="utf-8")
    (tok_dir / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
    cfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                         max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=48,
                         bos_token_id=vocab["<|startoftext|>"], eos_token_id=vocab["<|endoftext|>"], pad_token_id=1)
    prompts = ["a photo of a cat", "two dogs playing in the park!"]
    for proj, pad in ((False, None), (True, "!")):
        torch.manual_seed(5)
        ref = (CLIPTextModelWithProjection(cfg) if proj else CLIPTextModel(cfg)).eval()
        enc_dir = tmp_path / f"text_encoder_{int(proj)}"; enc_dir.mkdir()
        cfg.to_json_file(str(enc_dir / "config.json"))
        save_file({k: v.contiguous() for k, v in ref.state_dict().items()}, str(enc_dir / "model.safetensors"))
        ids = CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges], **({"pad_token": pad} if pad else {}))(
            prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        with torch.no_grad():
            out = ref(input_ids=ids, output_hidden_states=True)
        sd15 = ClipTextTower.from_dir(enc_dir, tok_dir, penultimate=False, with_projection=proj, pad_token=pad)
        xl = ClipTextTower.from_dir(enc_dir, tok_dir, penultimate=True, with_projection=proj, pad_token=pad)
        hs, pooled = sd15(prompts)
        assert torch.equal(hs, out.last_hidden_state.half()) and (pooled is None) == (not proj)
        hs2, pooled2 = xl(prompts)
        assert torch.equal(hs2, out.hidden_states[-2].half())
        assert torch.equal(xl(prompts, clip_skip=1)[0], out.hidden_states[-3].half())
        if proj:
            assert torch.equal(pooled2, out.text_embeds.half())


def test_clip_bpe_tokenizer_fuzz_vs_transformers():
    """seeded random strings over letters, digits, punctuation, whitespace kinds, accents, CJK, emoji, contractions and
    in-text special tokens: both padding conventions must agree with transformers id for id"""
    import random
    import torch
    from transformers import CLIPTokenizer
    from cfgpp_amd.conditioning import ClipBpeTokenizer
    vocab, merges = _toy_clip_vocab()
    pairs = [(ClipBpeTokenizer(vocab, [" ".join(m) for m in merges]), CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges])),
             (ClipBpeTokenizer(vocab, [" ".join(m) for m in merges], pad_token="!"),
              CLIPTokenizer(vocab=vocab, merges=[tuple(m) for m in merges], pad_token="!"))]
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCDEFG0123456789 \t\n.,;:!?'\"-_()[]{}<>|/\\@#$%^&*+=~`") + [
        "\u00e9", "\u00ef", "\u00df", "\u03a9", "\u732b", "\u72ac", "\U0001f600", "\u0301", "\u00a0", "\u200b", "\u2019", "\u201c", "\u2026",
        "'s", "'t", "'re", "'ll", "n't", "<|endoftext|>", "<|startoftext|>", "  ", "!!"]
    rng = random.Random(20260926)
    strings = ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 48))) for _ in range(500)]
    strings += ["".join(rng.choice(alphabet) for _ in range(400)) for _ in range(20)]            # longer than 77 tokens: truncation
    for ours, ref in pairs:
        want = ref(strings, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        got = ours(strings)
        bad = [i for i in range(len(strings)) if not torch.equal(got[i], want[i])]
        assert not bad, (repr(strings[bad[0]]), got[bad[0]].tolist()[:16], want[bad[0]].tolist()[:16])
