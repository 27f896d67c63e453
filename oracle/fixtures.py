"""Synthetic CLIP GGUF fixtures (TEST INFRASTRUCTURE — no weights ship with the reference).

Writes GGUF files that follow the on-disk contract of the reference's converter
(`models/convert_hf_to_gguf.py:126-206`) and quantizer (`clip.cpp:1661-1844`):
KV keys of `clip.cpp:41-58`, tensor names of `clip.cpp:64-79`, 4-D conv always f16,
quantised files quantise every 2-D tensor whose name matches `.*weight`, 1-D stay f32.

Weights are seeded synthetic: linear ~N(0,0.02^2) with a few x8 outlier
columns, biases ~N(0,0.01^2), LN gain ~N(1,0.05^2), LN bias ~N(0,0.05^2).
"""
import os
import struct
import hashlib

import numpy as np

from . import ref

CONFIGS = {
    # name: vision(S,P,h,L,nh,ff,proj)  text(h,L,nh,ff,proj,npos)
    "tiny": dict(v=dict(S=32, P=8, h=64, L=2, nh=2, ff=128, proj=32), t=dict(h=64, L=2, nh=2, ff=128, proj=32, npos=77)),
    "tiny14": dict(v=dict(S=28, P=14, h=128, L=2, nh=2, ff=256, proj=64), t=dict(h=64, L=2, nh=2, ff=128, proj=64, npos=77)),
    # 336-px geometry of ViT-L/14@336 (T = 24*24 + 1 = 577 tokens, d_head 64) on a small width: long-sequence attention path
    "tiny336": dict(v=dict(S=336, P=14, h=128, L=2, nh=2, ff=256, proj=64), t=dict(h=64, L=2, nh=2, ff=128, proj=64, npos=77)),
    # head sizes 88 / 104 of ViT-g/14 (hidden 1408, 16 heads) and ViT-bigG/14 (hidden 1664, 16 heads) on narrow towers (8 heads)
    "g88": dict(v=dict(S=28, P=14, h=704, L=2, nh=8, ff=1408, proj=64), t=dict(h=64, L=2, nh=2, ff=128, proj=64, npos=77)),
    "g104": dict(v=dict(S=56, P=14, h=832, L=2, nh=8, ff=1664, proj=64), t=dict(h=64, L=2, nh=2, ff=128, proj=64, npos=77)),
    # small widths with the LAYER COUNTS of the base / large models: the reference's loader switches on the tensor count (397 / 200 / 197 for 12 + 12
    # layers, 589 / 392 for 24 + 12, 909 / 520 / 389 for 32 + 24; clip.cpp:261-331) and exits on any other, so these are the small files its own clip.cpp accepts (oracle/ref_graph.py)
    "base12": dict(v=dict(S=32, P=8, h=64, L=12, nh=2, ff=128, proj=32), t=dict(h=64, L=12, nh=2, ff=128, proj=32, npos=77)),
    "large24": dict(v=dict(S=28, P=14, h=128, L=24, nh=4, ff=256, proj=64), t=dict(h=64, L=12, nh=2, ff=128, proj=64, npos=77)),
    "huge32": dict(v=dict(S=28, P=14, h=160, L=32, nh=2, ff=320, proj=64), t=dict(h=64, L=24, nh=2, ff=128, proj=64, npos=77)),   # 909 / 520 / 389 tensors, d_head 80
    "b32": dict(v=dict(S=224, P=32, h=768, L=12, nh=12, ff=3072, proj=512), t=dict(h=512, L=12, nh=8, ff=2048, proj=512, npos=77)),
    "b16": dict(v=dict(S=224, P=16, h=768, L=12, nh=12, ff=3072, proj=512), t=dict(h=512, L=12, nh=8, ff=2048, proj=512, npos=77)),
    "l14": dict(v=dict(S=224, P=14, h=1024, L=24, nh=16, ff=4096, proj=768), t=dict(h=768, L=12, nh=12, ff=3072, proj=768, npos=77)),
    "h14": dict(v=dict(S=224, P=14, h=1280, L=32, nh=16, ff=5120, proj=1024), t=dict(h=1024, L=24, nh=16, ff=4096, proj=1024, npos=77)),
}

N_VOCAB = 49408  # BOS 49406 / EOS 49407 are hard-coded in clip_tokenize (clip.cpp:637,671)

_WORDS = ("a an the of and in on with for to is are cat dog photo picture red green blue apple banana turtle "
          "man woman car tree house bird fish sky sea white black eats sits runs big small two three").split()


def synthetic_vocab():
    """Deterministic CLIP-shaped vocabulary: bytes, bytes</w>, words</w>, sub-word pieces, fillers."""
    toks = []
    seen = set()

    def add(t):
        if t not in seen:
            seen.add(t)
            toks.append(t)

    for c in range(33, 127):
        add(chr(c))
    for c in range(33, 127):
        add(chr(c) + "</w>")
    for w in _WORDS:
        add(w + "</w>")
    letters = "etaoinshrdlu"
    for a in letters:
        for b in letters:
            add(a + b)
            add(a + b + "</w>")
    for a in "0123456789":
        for b in "0123456789":
            add(a + b)
            add(a + b + "</w>")
    for w in ("ing", "tion", "er", "est", "ly", "un", "re"):
        add(w)
        add(w + "</w>")
    i = 0
    while len(toks) < N_VOCAB - 2:
        add("tok%d_" % i)
        i += 1
    toks.append("<|startoftext|>")
    toks.append("<|endoftext|>")
    assert len(toks) == N_VOCAB
    return toks


def tensor_specs(cfg, text=True, vision=True):
    """Ordered (name, shape[numpy order], kind) list. kind in {lin, bias, ln_w, ln_b, embd, conv, cls}."""
    specs = []
    if text:
        t = cfg["t"]
        specs += [("t.token_embd.weight", (N_VOCAB, t["h"]), "embd"), ("t.position_embd.weight", (t["npos"], t["h"]), "embd")]
        for i in range(t["L"]):
            specs += _block("t", i, t["h"], t["ff"])
        specs += [("t.post_ln.weight", (t["h"],), "ln_w"), ("t.post_ln.bias", (t["h"],), "ln_b")]
    if vision:
        v = cfg["v"]
        T = (v["S"] // v["P"]) ** 2 + 1
        specs += [("v.class_embd", (v["h"],), "cls"), ("v.patch_embd.weight", (v["h"], 3, v["P"], v["P"]), "conv"),
                  ("v.position_embd.weight", (T, v["h"]), "embd"),
                  ("v.pre_ln.weight", (v["h"],), "ln_w"), ("v.pre_ln.bias", (v["h"],), "ln_b")]
        for i in range(v["L"]):
            specs += _block("v", i, v["h"], v["ff"])
        specs += [("v.post_ln.weight", (v["h"],), "ln_w"), ("v.post_ln.bias", (v["h"],), "ln_b")]
    if vision:
        specs.append(("visual_projection.weight", (cfg["v"]["proj"], cfg["v"]["h"]), "lin"))
    if text:
        specs.append(("text_projection.weight", (cfg["t"]["proj"], cfg["t"]["h"]), "lin"))
    return specs


def _block(p, i, h, ff):
    b = "%s.blk.%d." % (p, i)
    out = []
    for nm in ("attn_k", "attn_v", "attn_q", "attn_out"):
        out += [(b + nm + ".weight", (h, h), "lin"), (b + nm + ".bias", (h,), "bias")]
    out += [(b + "ln1.weight", (h,), "ln_w"), (b + "ln1.bias", (h,), "ln_b")]
    # (sic) ffn_down is the UP projection h->ff, ffn_up the DOWN projection ff->h (clip.cpp:510-511)
    out += [(b + "ffn_down.weight", (ff, h), "lin"), (b + "ffn_down.bias", (ff,), "bias"),
            (b + "ffn_up.weight", (h, ff), "lin"), (b + "ffn_up.bias", (h,), "bias")]
    out += [(b + "ln2.weight", (h,), "ln_w"), (b + "ln2.bias", (h,), "ln_b")]
    return out


def gen_tensor(name, shape, kind, seed):
    s = int.from_bytes(hashlib.sha256(("%d:%s" % (seed, name)).encode()).digest()[:8], "little")
    rng = np.random.default_rng(s)
    if kind == "lin":
        w = rng.standard_normal(shape, dtype=np.float32) * 0.02
        n_out = max(1, shape[1] // 128)
        cols = rng.choice(shape[1], size=n_out, replace=False)
        w[:, cols] *= 8.0
        return w
    if kind in ("embd", "conv", "cls"):
        return rng.standard_normal(shape, dtype=np.float32) * 0.02
    if kind == "bias":
        return rng.standard_normal(shape, dtype=np.float32) * 0.01
    if kind == "ln_w":
        return (1.0 + rng.standard_normal(shape, dtype=np.float32) * 0.05).astype(np.float32)
    if kind == "ln_b":
        return rng.standard_normal(shape, dtype=np.float32) * 0.05
    raise ValueError(kind)


# ------------------------------------------------------------------ GGUF writer
_GT = {"u8": 0, "i8": 1, "u16": 2, "i16": 3, "u32": 4, "i32": 5, "f32": 6, "bool": 7, "str": 8, "arr": 9, "u64": 10}


def _s(b):
    if isinstance(b, str):
        b = b.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def _kv(key, typ, val):
    out = _s(key) + struct.pack("<I", _GT[typ] if typ != "arr_f32" and typ != "arr_str" else 9)
    if typ == "u32":
        out += struct.pack("<I", val)
    elif typ == "f32":
        out += struct.pack("<f", val)
    elif typ == "bool":
        out += struct.pack("<B", 1 if val else 0)
    elif typ == "str":
        out += _s(val)
    elif typ == "arr_f32":
        out += struct.pack("<IQ", 6, len(val)) + struct.pack("<%df" % len(val), *val)
    elif typ == "arr_str":
        out += struct.pack("<IQ", 8, len(val)) + b"".join(_s(v) for v in val)
    else:
        raise ValueError(typ)
    return out


def write_gguf(path, kvs, tensors, version=2, alignment=32):
    """kvs: list of (key, type, value); tensors: list of (name, numpy_shape, ggml_type_id, raw_bytes ndarray)."""
    head = b"GGUF" + struct.pack("<IQQ", version, len(tensors), len(kvs))
    body = b"".join(_kv(*kv) for kv in kvs)
    infos = b""
    off = 0
    offs = []
    for name, shape, tid, raw in tensors:
        dims = list(reversed(shape))  # ne0 first
        infos += _s(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims)
        infos += struct.pack("<IQ", tid, off)
        offs.append(off)
        off += (raw.size + alignment - 1) // alignment * alignment
    meta = head + body + infos
    pad = (-len(meta)) % alignment
    with open(path, "wb") as f:
        f.write(meta + b"\0" * pad)
        for (name, shape, tid, raw) in tensors:
            f.write(raw.tobytes())
            f.write(b"\0" * ((-raw.size) % alignment))


def file_type_id(ftype):
    return ref.GGML_TYPES[ftype]


def make_model(path, config="tiny", ftype="f32", text=True, vision=True, seed=1234, use_gelu=False,
               image_mean=(0.48145466, 0.4578275, 0.40821073), image_std=(0.26862954, 0.26130258, 0.27577711),
               eps=1e-5, version=2, keep_master=False, dc=0.0, spike=0.0):
    """Write a synthetic CLIP GGUF. Returns dict(name -> f32 master ndarray) if keep_master.
    dc != 0: residual-stream rows with a large common mode (|mean| / std ~ dc and drifting from layer to layer), the statistics the
    seeded N(0, 0.02^2) weights never produce: the vision pre-LN bias gets +dc (rows of std ~1), the text position embedding +0.03 dc
    (rows of std ~0.03), and every out-projection / FFN-down bias a tenth of that — the LayerNorm-fold parity fixtures.
    spike != 0: "massive activation" channels as real CLIP ViT-L / H checkpoints have them — three fixed channels of the same tensors get
    +spike (x the tower's scale) on top, so a handful of residual-stream channels sit tens of sigma away from the rest in every row."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    tid_file = file_type_id(ftype)
    kvs = [("general.architecture", "str", "clip"),
           ("clip.has_text_encoder", "bool", text), ("clip.has_vision_encoder", "bool", vision),
           ("general.file_type", "u32", tid_file),
           ("general.name", "str", "synthetic-%s" % (config if isinstance(config, str) else "custom")),
           ("general.description", "str", ("two-tower" if text and vision else "text-only" if text else "vision-only") + " CLIP model")]
    if tid_file >= 2:
        kvs.append(("general.quantization_version", "u32", 2))
    if text:
        t = cfg["t"]
        kvs += [("clip.text.context_length", "u32", t["npos"]), ("clip.text.embedding_length", "u32", t["h"]),
                ("clip.text.feed_forward_length", "u32", t["ff"]), ("clip.text.projection_dim", "u32", t["proj"]),
                ("clip.text.attention.head_count", "u32", t["nh"]), ("clip.text.attention.layer_norm_epsilon", "f32", eps),
                ("clip.text.block_count", "u32", t["L"]), ("tokenizer.ggml.tokens", "arr_str", synthetic_vocab())]
    if vision:
        v = cfg["v"]
        kvs += [("clip.vision.image_size", "u32", v["S"]), ("clip.vision.patch_size", "u32", v["P"]),
                ("clip.vision.embedding_length", "u32", v["h"]), ("clip.vision.feed_forward_length", "u32", v["ff"]),
                ("clip.vision.projection_dim", "u32", v["proj"]), ("clip.vision.attention.head_count", "u32", v["nh"]),
                ("clip.vision.attention.layer_norm_epsilon", "f32", eps), ("clip.vision.block_count", "u32", v["L"]),
                ("clip.vision.image_mean", "arr_f32", list(image_mean)), ("clip.vision.image_std", "arr_f32", list(image_std))]
    kvs.append(("clip.use_gelu", "bool", use_gelu))

    tensors = []
    master = {}
    for name, shape, kind in tensor_specs(cfg, text, vision):
        w = gen_tensor(name, shape, kind, seed)
        if dc or spike:
            tower_scale = 1.0 if name.startswith("v.") else 0.03
            if name in ("v.pre_ln.bias", "t.position_embd.weight"):
                w = (w + dc * tower_scale).astype(np.float32)
                if spike:
                    w[..., [3, shape[-1] // 2 + 1, shape[-1] - 5]] += spike * tower_scale
            elif name.endswith("attn_out.bias") or name.endswith("ffn_up.bias"):      # (sic: "ffn_up" is fc2, the projection back to h)
                w = (w + 0.1 * dc * tower_scale).astype(np.float32)
        if keep_master:
            master[name] = w
        if len(shape) == 4:
            tid = 1  # conv kernel: always f16 (convert_hf_to_gguf.py:182-186)
        elif len(shape) == 2 and name.endswith("weight"):
            tid = tid_file  # f32 / f16 / quantised (clip.cpp:1711-1739)
        else:
            tid = 0
        if tid == 0:
            raw = np.frombuffer(np.ascontiguousarray(w, dtype=np.float32).tobytes(), dtype=np.uint8)
        elif tid == 1:
            raw = np.frombuffer(w.astype(np.float16).tobytes(), dtype=np.uint8)
        else:
            # the quantiser's source is the f16 (or f32) converter output; we quantise from the f32 master
            raw = ref.quantize(tid, w.reshape(-1, shape[-1]))
        tensors.append((name, shape, tid, raw))
    write_gguf(path, kvs, tensors, version=version)
    return master if keep_master else None


def cached_model(cache_dir, config="tiny", ftype="f32", text=True, vision=True, seed=1234, use_gelu=False, dc=0.0, spike=0.0):
    os.makedirs(cache_dir, exist_ok=True)
    tag = "%s_%s_%s%s_s%d%s%s.gguf" % (config, ftype, "t" if text else "", "v" if vision else "", seed, "_gelu" if use_gelu else "", (("_dc%g" % dc) if dc else "") + (("_spk%g" % spike) if spike else ""))
    path = os.path.join(cache_dir, tag)
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        make_model(tmp, config, ftype, text, vision, seed, use_gelu, dc=dc, spike=spike)
        os.replace(tmp, path)
    return path


def synthetic_images(B, S, seed=7):
    """Seeded N(0,1) f32 HWC images (encoder-only inputs; same bytes go to oracle and GPU)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((B, S, S, 3), dtype=np.float32)


def synthetic_token_ids(n_texts, seed=11, min_len=1, max_len=20):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_texts):
        n = int(rng.integers(min_len, max_len + 1))
        ids = rng.integers(0, N_VOCAB - 2, size=n).astype(np.int32)
        out.append(np.concatenate([[49406], ids, [49407]]).astype(np.int32))
    return out
