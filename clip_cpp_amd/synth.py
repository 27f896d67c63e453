"""Synthetic CLIP GGUF files (seeded random weights of a named architecture) written with the PRODUCT tools only:
the converter's GGUF writer (convert_hf_to_gguf.GGUFOut) and libclip's clip_model_quantize.

Used by bench.py for its workload (there is no network for checkpoints; the metric only depends on the shapes) and
usable as a smoke-test model source.  Nothing here touches oracle/: the oracle reads the same FILE when bench.py's
cpu_baseline leg or the tests want a reference result.

Weight statistics follow SURVEY 8d: linear weights ~N(0, 0.02^2) with a few x8 outlier columns, biases ~N(0, 0.01^2),
LayerNorm w ~N(1, 0.05^2), b ~N(0, 0.05^2), embeddings / conv kernel ~N(0, 0.02^2) (conv stored f16 as the reference
converter does).  Tensor names and the fc1->"ffn_down" / fc2->"ffn_up" convention are the reference's (clip.cpp:485-525).
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

from .convert_hf_to_gguf import GGUFOut, OPENAI_CLIP_MEAN, OPENAI_CLIP_STD

N_VOCAB = 49408   # BOS 49406 / EOS 49407 are hard-coded in clip_tokenize (reference clip.cpp:637,671)
FTYPES = {"f32": 0, "f16": 1, "q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8}

# name: vision (S, P, h, L, nh, ff, proj), text (h, L, nh, ff, proj, npos) — the HF configs of the models BASELINE.json names
ARCH = {
    "tiny": dict(v=dict(S=32, P=8, h=64, L=2, nh=2, ff=128, proj=32), t=dict(h=64, L=2, nh=2, ff=128, proj=32, npos=77)),
    "b32": dict(v=dict(S=224, P=32, h=768, L=12, nh=12, ff=3072, proj=512), t=dict(h=512, L=12, nh=8, ff=2048, proj=512, npos=77)),
    "b16": dict(v=dict(S=224, P=16, h=768, L=12, nh=12, ff=3072, proj=512), t=dict(h=512, L=12, nh=8, ff=2048, proj=512, npos=77)),
    "l14": dict(v=dict(S=224, P=14, h=1024, L=24, nh=16, ff=4096, proj=768), t=dict(h=768, L=12, nh=12, ff=3072, proj=768, npos=77)),
    "l14_336": dict(v=dict(S=336, P=14, h=1024, L=24, nh=16, ff=4096, proj=768), t=dict(h=768, L=12, nh=12, ff=3072, proj=768, npos=77)),
    "h14": dict(v=dict(S=224, P=14, h=1280, L=32, nh=16, ff=5120, proj=1024), t=dict(h=1024, L=24, nh=16, ff=4096, proj=1024, npos=77)),
}


def vocab():
    """49408 distinct token strings: printable bytes, their end-of-word forms, fillers, BOS, EOS (shape of the CLIP BPE vocabulary)."""
    toks = [chr(c) for c in range(33, 127)] + [chr(c) + "</w>" for c in range(33, 127)]
    toks += ["w%d</w>" % i for i in range(N_VOCAB - 2 - len(toks))]
    return toks + ["<|startoftext|>", "<|endoftext|>"]


def _block(prefix, i, h, ff):
    b = "%s.blk.%d." % (prefix, i)
    out = []
    for nm in ("attn_k", "attn_v", "attn_q", "attn_out"):
        out += [(b + nm + ".weight", (h, h), "lin"), (b + nm + ".bias", (h,), "bias")]
    out += [(b + "ln1.weight", (h,), "ln_w"), (b + "ln1.bias", (h,), "ln_b"),
            (b + "ffn_down.weight", (ff, h), "lin"), (b + "ffn_down.bias", (ff,), "bias"),     # (sic) fc1: h -> ff
            (b + "ffn_up.weight", (h, ff), "lin"), (b + "ffn_up.bias", (h,), "bias"),          # (sic) fc2: ff -> h
            (b + "ln2.weight", (h,), "ln_w"), (b + "ln2.bias", (h,), "ln_b")]
    return out


def tensor_list(arch, text=True, vision=True):
    specs = []
    if text:
        t = arch["t"]
        specs += [("t.token_embd.weight", (N_VOCAB, t["h"]), "embd"), ("t.position_embd.weight", (t["npos"], t["h"]), "embd")]
        for i in range(t["L"]):
            specs += _block("t", i, t["h"], t["ff"])
        specs += [("t.post_ln.weight", (t["h"],), "ln_w"), ("t.post_ln.bias", (t["h"],), "ln_b")]
    if vision:
        v = arch["v"]
        T = (v["S"] // v["P"]) ** 2 + 1
        specs += [("v.class_embd", (v["h"],), "embd"), ("v.patch_embd.weight", (v["h"], 3, v["P"], v["P"]), "embd"),
                  ("v.position_embd.weight", (T, v["h"]), "embd"), ("v.pre_ln.weight", (v["h"],), "ln_w"), ("v.pre_ln.bias", (v["h"],), "ln_b")]
        for i in range(v["L"]):
            specs += _block("v", i, v["h"], v["ff"])
        specs += [("v.post_ln.weight", (v["h"],), "ln_w"), ("v.post_ln.bias", (v["h"],), "ln_b"),
                  ("visual_projection.weight", (v["proj"], v["h"]), "lin")]
    if text:
        specs.append(("text_projection.weight", (arch["t"]["proj"], arch["t"]["h"]), "lin"))
    return specs


def _tensor(name, shape, kind, seed):
    rng = np.random.default_rng(int.from_bytes(hashlib.sha256(("%d:%s" % (seed, name)).encode()).digest()[:8], "little"))
    if kind == "lin":
        w = rng.standard_normal(shape, dtype=np.float32) * 0.02
        w[:, rng.choice(shape[1], size=max(1, shape[1] // 128), replace=False)] *= 8.0
        return w
    if kind == "embd":
        return rng.standard_normal(shape, dtype=np.float32) * 0.02
    if kind == "bias":
        return rng.standard_normal(shape, dtype=np.float32) * 0.01
    if kind == "ln_w":
        return (1.0 + rng.standard_normal(shape, dtype=np.float32) * 0.05).astype(np.float32)
    return rng.standard_normal(shape, dtype=np.float32) * 0.05      # ln_b


def write_model(path, arch="b32", ftype="q4_0", text=True, vision=True, seed=1234, use_gelu=False, eps=1e-5):
    """Write the f16 (or f32) file with the converter's writer; quantised types go through clip_model_quantize."""
    a = ARCH[arch] if isinstance(arch, str) else arch
    base_ftype = 0 if ftype == "f32" else 1
    g = GGUFOut()
    g.add_str("general.architecture", "clip")
    g.add_bool("clip.has_text_encoder", text)
    g.add_bool("clip.has_vision_encoder", vision)
    g.add_u32("general.file_type", base_ftype)
    g.add_str("general.name", "synthetic-%s" % (arch if isinstance(arch, str) else "custom"))
    g.add_str("general.description", ("two-tower" if text and vision else "text-only" if text else "vision-only") + " CLIP model")
    if text:
        t = a["t"]
        g.add_u32("clip.text.context_length", t["npos"]); g.add_u32("clip.text.embedding_length", t["h"])
        g.add_u32("clip.text.feed_forward_length", t["ff"]); g.add_u32("clip.text.projection_dim", t["proj"])
        g.add_u32("clip.text.attention.head_count", t["nh"]); g.add_f32("clip.text.attention.layer_norm_epsilon", eps)
        g.add_u32("clip.text.block_count", t["L"]); g.add_str_array("tokenizer.ggml.tokens", vocab())
    if vision:
        v = a["v"]
        g.add_u32("clip.vision.image_size", v["S"]); g.add_u32("clip.vision.patch_size", v["P"])
        g.add_u32("clip.vision.embedding_length", v["h"]); g.add_u32("clip.vision.feed_forward_length", v["ff"])
        g.add_u32("clip.vision.projection_dim", v["proj"]); g.add_u32("clip.vision.attention.head_count", v["nh"])
        g.add_f32("clip.vision.attention.layer_norm_epsilon", eps); g.add_u32("clip.vision.block_count", v["L"])
        g.add_f32_array("clip.vision.image_mean", OPENAI_CLIP_MEAN); g.add_f32_array("clip.vision.image_std", OPENAI_CLIP_STD)
    g.add_bool("clip.use_gelu", use_gelu)
    for name, shape, kind in tensor_list(a, text, vision):
        w = _tensor(name, shape, kind, seed)
        if len(shape) == 4 or (base_ftype == 1 and len(shape) == 2 and name.endswith(".weight")):
            w = w.astype(np.float16)
        g.add_tensor(name, w)
    if ftype in ("f32", "f16"):
        g.write(path)
        return path
    tmp = path + ".f16.tmp%d" % os.getpid()
    g.write(tmp)
    try:
        from . import lib
        L = lib()
        # clip_model_quantize reports every tensor on stdout like the reference does (clip.cpp:1741-1800): keep it off the
        # caller's stdout (bench.py prints exactly one JSON line there)
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        saved, devnull = os.dup(1), os.open(os.devnull, os.O_WRONLY)
        try:
            os.dup2(devnull, 1)
            ok = L.clip_model_quantize(os.fsencode(tmp), os.fsencode(path), FTYPES[ftype])
        finally:
            C.CDLL(None).fflush(None)          # the C library buffers stdout when it is a pipe: drain it into /dev/null
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        if not ok:
            raise RuntimeError("clip_model_quantize failed for %s" % path)
    finally:
        os.remove(tmp)
    return path


def cached_model(cache_dir, arch="b32", ftype="q4_0", text=True, vision=True, seed=1234, use_gelu=False):
    """Path of the synthetic model, generating it once (atomic rename: safe when several ranks start together)."""
    os.makedirs(cache_dir, exist_ok=True)
    tag = "synth_%s_%s_%s%s_s%d%s.gguf" % (arch, ftype, "t" if text else "", "v" if vision else "", seed, "_gelu" if use_gelu else "")
    path = os.path.join(cache_dir, tag)
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        write_model(tmp, arch, ftype, text, vision, seed, use_gelu)
        os.replace(tmp, path)
    return path


def token_ids(n_texts, seed=11, min_len=1, max_len=20):
    """Seeded id sequences [BOS, r_1 .. r_n, EOS], n in [min_len, max_len]."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_texts):
        n = int(rng.integers(min_len, max_len + 1))
        out.append(np.concatenate([[49406], rng.integers(0, N_VOCAB - 2, size=n), [49407]]).astype(np.int32))
    return out
