"""Hugging Face CLIP checkpoint -> GGUF, as read by clip_model_load (include/clip.h).

Product tool, same command line and output naming as the reference converter
(reference models/convert_hf_to_gguf.py:66-75,120-124) but without its `gguf` pip dependency: the GGUF v3 container is
written by the ~40 lines below (container layout: SURVEY Appendix A).

    python -m clip_cpp_amd.convert_hf_to_gguf -m <hf_model_dir> [--use-f32] [--text-only | --vision-only]
                                              [--image-mean R G B] [--image-std R G B] [-o <out_dir>]

The model directory needs config.json, vocab.json and the weights (safetensors or .bin); image mean / std come from
preprocessor_config.json when present (else the OpenAI CLIP constants) unless overridden.  Tensor naming, dtypes and
metadata keys follow the reference writer so that files are interchangeable:
  * names: the rename chain of reference convert_hf_to_gguf.py:27-35 ("text_model" -> "t", "encoder.layers" -> "blk", ...;
    note its quirk: mlp.fc1 is stored as "ffn_down" and mlp.fc2 as "ffn_up");
  * dtypes (:178-199): 4-D (the patch-embedding conv kernel) always f16; in f16 files every 2-D "*.weight" is f16 and the
    rest f32; in f32 files everything else is f32;
  * skipped (:13-21): logit_scale and the position_ids buffers; --text-only / --vision-only drop the other tower.
Quantised files are produced afterwards with clip_model_quantize (`clip_cpp_amd.Clip.quantize` / reference
models/quantize.cpp).
"""
import argparse
import json
import os
import struct
import sys

import numpy as np

GGML_F32, GGML_F16 = 0, 1
_T_U32, _T_F32, _T_BOOL, _T_STR, _T_ARR = 4, 6, 7, 8, 9
OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]


# ---- GGUF v3 writer (little-endian; key/values, tensor infos, 32-byte aligned data section) ----
def _str(s):
    b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    return struct.pack("<Q", len(b)) + b


class GGUFOut:
    ALIGN = 32

    def __init__(self):
        self.kv = []        # already serialised key/value records
        self.tensors = []   # (name, ndarray, ggml_type)

    def _add(self, key, type_id, payload):
        self.kv.append(_str(key) + struct.pack("<I", type_id) + payload)

    def add_u32(self, key, v): self._add(key, _T_U32, struct.pack("<I", int(v)))
    def add_f32(self, key, v): self._add(key, _T_F32, struct.pack("<f", float(v)))
    def add_bool(self, key, v): self._add(key, _T_BOOL, struct.pack("<B", 1 if v else 0))
    def add_str(self, key, v): self._add(key, _T_STR, _str(v))
    def add_f32_array(self, key, vals): self._add(key, _T_ARR, struct.pack("<IQ", _T_F32, len(vals)) + struct.pack("<%df" % len(vals), *vals))
    def add_str_array(self, key, vals): self._add(key, _T_ARR, struct.pack("<IQ", _T_STR, len(vals)) + b"".join(_str(v) for v in vals))

    def add_tensor(self, name, data):
        data = np.ascontiguousarray(data)
        if data.dtype == np.float16:
            t = GGML_F16
        elif data.dtype == np.float32:
            t = GGML_F32
        else:
            raise ValueError("tensor %s: unsupported dtype %s" % (name, data.dtype))
        self.tensors.append((name, data, t))

    def write(self, path):
        up = lambda n: (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        infos, off = [], 0
        for name, data, t in self.tensors:
            dims = list(reversed(data.shape)) or [1]                  # ne[0] = fastest-varying dimension first
            infos.append(_str(name) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", t, off))
            off += up(data.nbytes)
        meta = b"GGUF" + struct.pack("<IQQ", 3, len(self.tensors), len(self.kv)) + b"".join(self.kv) + b"".join(infos)
        with open(path, "wb") as f:
            f.write(meta + b"\0" * (up(len(meta)) - len(meta)))
            for name, data, t in self.tensors:
                f.write(data.tobytes())
                f.write(b"\0" * (up(data.nbytes) - data.nbytes))


def gguf_inspect(path):
    """(kv dict, [(tensor name, dims ne0-first, ggml type id)]) of a GGUF v2/v3 file — metadata only."""
    sizes = {0: "B", 1: "b", 2: "H", 3: "h", 4: "I", 5: "i", 6: "f", 7: "?", 10: "Q", 11: "q", 12: "d"}
    with open(path, "rb") as f:
        def rd(fmt):
            return struct.unpack("<" + fmt, f.read(struct.calcsize("<" + fmt)))
        def rstr():
            return f.read(rd("Q")[0]).decode("utf-8", errors="replace")
        def rval(t):
            if t == _T_STR:
                return rstr()
            if t == _T_ARR:
                et, n = rd("IQ")
                return [rval(et) for _ in range(n)]
            return rd(sizes[t])[0]
        if f.read(4) != b"GGUF":
            raise ValueError("%s: not a GGUF file" % path)
        version, n_tensors, n_kv = rd("IQQ")
        kv = {}
        for _ in range(n_kv):
            key = rstr()
            kv[key] = rval(rd("I")[0])
        tensors = []
        for _ in range(n_tensors):
            name = rstr()
            nd = rd("I")[0]
            dims = list(rd("%dQ" % nd))
            t, _off = rd("IQ")
            tensors.append((name, dims, t))
    return kv, tensors


# ---- conversion ----
_RENAMES = (("text_model", "t"), ("vision_model", "v"), ("encoder.layers", "blk"), ("embeddings.", ""), ("_proj", ""),
            ("self_attn.", "attn_"), ("layer_norm", "ln"), ("layernorm", "ln"), ("mlp.fc1", "ffn_down"), ("mlp.fc2", "ffn_up"),
            ("embedding", "embd"), ("final", "post"), ("layrnorm", "ln"))
_SKIP = ("logit_scale", "text_model.embeddings.position_ids", "vision_model.embeddings.position_ids")


def gguf_tensor_name(hf_name):
    if "projection" in hf_name:          # text_projection.weight / visual_projection.weight keep their names
        return hf_name
    for old, new in _RENAMES:            # order matters (e.g. "embeddings." before "embedding")
        hf_name = hf_name.replace(old, new)
    return hf_name


def load_state_dict(model_dir):
    """name -> float ndarray, straight from the checkpoint files (safetensors preferred), else through transformers."""
    st = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(st):
        try:
            from safetensors.numpy import load_file
            return dict(load_file(st))
        except Exception:               # e.g. bf16 checkpoints (no numpy dtype): let transformers up-cast
            pass
    from transformers import CLIPModel
    return {k: v.float().numpy() for k, v in CLIPModel.from_pretrained(model_dir).state_dict().items()}


def convert(model_dir, use_f32=False, text_only=False, vision_only=False, image_mean=None, image_std=None, output_dir=None, verbose=True):
    if text_only and vision_only:
        raise SystemExit("--text-only and --vision-only cannot be specified at the same time.")
    with open(os.path.join(model_dir, "config.json"), encoding="utf-8") as f:
        config = json.load(f)
    v_hp, t_hp = config["vision_config"], config["text_config"]
    ftype = 0 if use_f32 else 1
    has_text, has_vision = not vision_only, not text_only
    out_dir = output_dir if output_dir is not None else model_dir
    os.makedirs(out_dir, exist_ok=True)
    middle = "text-" if text_only else "vision-" if vision_only else ""
    prefix = os.path.basename(os.path.normpath(out_dir)).replace("ggml_", "")
    fname_out = os.path.join(out_dir, "%s_ggml-%smodel-%s.gguf" % (prefix, middle, ("f32", "f16")[ftype]))

    g = GGUFOut()
    g.add_str("general.architecture", "clip")
    g.add_bool("clip.has_text_encoder", has_text)
    g.add_bool("clip.has_vision_encoder", has_vision)
    g.add_u32("general.file_type", ftype)
    g.add_str("general.name", config.get("_name_or_path") or os.path.basename(os.path.normpath(model_dir)))
    g.add_str("general.description", "text-only CLIP model" if text_only else "vision-only CLIP model" if vision_only else "two-tower CLIP model")
    if has_text:
        with open(os.path.join(model_dir, "vocab.json"), encoding="utf-8") as f:
            tokens = list(json.load(f))                                # insertion order == id order in HF vocab files
        g.add_u32("clip.text.context_length", t_hp["max_position_embeddings"])
        g.add_u32("clip.text.embedding_length", t_hp["hidden_size"])
        g.add_u32("clip.text.feed_forward_length", t_hp["intermediate_size"])
        g.add_u32("clip.text.projection_dim", t_hp.get("projection_dim", config["projection_dim"]))
        g.add_u32("clip.text.attention.head_count", t_hp["num_attention_heads"])
        g.add_f32("clip.text.attention.layer_norm_epsilon", t_hp["layer_norm_eps"])
        g.add_u32("clip.text.block_count", t_hp["num_hidden_layers"])
        g.add_str_array("tokenizer.ggml.tokens", tokens)
    if has_vision:
        g.add_u32("clip.vision.image_size", v_hp["image_size"])
        g.add_u32("clip.vision.patch_size", v_hp["patch_size"])
        g.add_u32("clip.vision.embedding_length", v_hp["hidden_size"])
        g.add_u32("clip.vision.feed_forward_length", v_hp["intermediate_size"])
        g.add_u32("clip.vision.projection_dim", v_hp.get("projection_dim", config["projection_dim"]))
        g.add_u32("clip.vision.attention.head_count", v_hp["num_attention_heads"])
        g.add_f32("clip.vision.attention.layer_norm_epsilon", v_hp["layer_norm_eps"])
        g.add_u32("clip.vision.block_count", v_hp["num_hidden_layers"])
        pre = {}
        pp = os.path.join(model_dir, "preprocessor_config.json")
        if os.path.exists(pp):
            with open(pp, encoding="utf-8") as f:
                pre = json.load(f)
        g.add_f32_array("clip.vision.image_mean", list(image_mean if image_mean is not None else pre.get("image_mean", OPENAI_CLIP_MEAN)))
        g.add_f32_array("clip.vision.image_std", list(image_std if image_std is not None else pre.get("image_std", OPENAI_CLIP_STD)))
    g.add_bool("clip.use_gelu", v_hp.get("hidden_act", "quick_gelu") == "gelu")

    for name, data in load_state_dict(model_dir).items():
        if name in _SKIP or (text_only and name.startswith("v")) or (vision_only and name.startswith("t")):
            if verbose:
                print("skipping parameter: %s" % name)
            continue
        out_name = gguf_tensor_name(name)
        data = np.squeeze(np.asarray(data))
        if data.ndim == 4:
            data = data.astype(np.float16)                 # conv kernel: always f16 (reference :182-185)
        elif ftype == 1 and out_name.endswith(".weight") and data.ndim == 2:
            data = data.astype(np.float16)
        else:
            data = data.astype(np.float32)
        if verbose:
            print("%s - %s - shape = %s" % (out_name, "f16" if data.dtype == np.float16 else "f32", data.shape))
        g.add_tensor(out_name, data)
    g.write(fname_out)
    if verbose:
        print("Done. Output file: " + fname_out)
    return fname_out


def main(argv=None):
    ap = argparse.ArgumentParser(prog="convert_hf_to_gguf.py")
    ap.add_argument("-m", "--model-dir", required=True, help="Path to model directory cloned from HF Hub")
    ap.add_argument("--use-f32", action="store_true", help="Use f32 instead of f16 (the conv kernel stays f16)")
    ap.add_argument("--text-only", action="store_true", help="Save a text-only model. It can't be used to encode images")
    ap.add_argument("--vision-only", action="store_true", help="Save a vision-only model. It can't be used to encode texts")
    ap.add_argument("--image-mean", nargs=3, type=float, help="Override image mean values")
    ap.add_argument("--image-std", nargs=3, type=float, help="Override image std values")
    ap.add_argument("-o", "--output-dir", default=None, help="Directory to save GGUF files. Default is the original model directory")
    a = ap.parse_args(argv)
    convert(a.model_dir, a.use_f32, a.text_only, a.vision_only, a.image_mean, a.image_std, a.output_dir)


if __name__ == "__main__":
    main(sys.argv[1:])
