"""Independent pin of the oracle's graph wiring: transformers.CLIPModel vs oracle(ideal).

The reference ships no golden vectors for the encoder path and ggml cannot be
built here, so the oracle's network topology / tensor renaming / layouts are
pinned against an INDEPENDENT implementation: Hugging Face `CLIPModel` built
offline from a config, loaded with the same seeded synthetic weights that
`oracle/fixtures.py` writes into the GGUF (through the converter's rename chain,
`models/convert_hf_to_gguf.py:31-35`).  HF runs in f32 with `quick_gelu`
(ggml's `gelu` is the tanh approximation, HF's "gelu" is erf — not comparable).

Run in the dev container (needs torch + transformers):
    python -m oracle.hf_crosscheck            # rewrites tests/golden/hf_*.npz
The test-suite (`tests/test_oracle_golden.py`) regenerates the same GGUF from the
seed, checks the stored weight checksum, runs the oracle and compares.
"""
import hashlib
import os
import sys

import numpy as np

from . import fixtures

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def gguf_name(hf_name):
    """Restatement of the converter's rename chain (convert_hf_to_gguf.py:31-35)."""
    if "projection" in hf_name:
        return hf_name
    n = hf_name
    for a, b in (("text_model", "t"), ("vision_model", "v"), ("encoder.layers", "blk"), ("embeddings.", ""),
                 ("_proj", ""), ("self_attn.", "attn_"), ("layer_norm", "ln"), ("layernorm", "ln"),
                 ("mlp.fc1", "ffn_down"), ("mlp.fc2", "ffn_up"), ("embedding", "embd"), ("final", "post"),
                 ("layrnorm", "ln")):
        n = n.replace(a, b)
    return n


def weights_checksum(master):
    h = hashlib.sha256()
    for k in sorted(master):
        h.update(k.encode())
        h.update(np.ascontiguousarray(master[k]).tobytes())
    return h.hexdigest()


def run(config="tiny", seed=1234, B=3, out_name=None, act="quick_gelu", f16_weights=False):
    """act: HF hidden_act of both towers ("gelu" = exact erf GELU: bounds the gap to ggml's tanh form, which the reference uses
    whenever clip.use_gelu is set, clip.cpp:1410-1414); f16_weights: give HF the weights as an f16 GGUF stores them (2-D "*.weight"
    tensors and the conv kernel rounded to fp16) so that the oracle can be run on the f16 FILE in ggml-faithful mode."""
    import torch
    from transformers import CLIPConfig, CLIPModel

    cfg = fixtures.CONFIGS[config]
    v, t = cfg["v"], cfg["t"]
    hf_cfg = CLIPConfig(
        text_config=dict(vocab_size=fixtures.N_VOCAB, hidden_size=t["h"], intermediate_size=t["ff"],
                         num_hidden_layers=t["L"], num_attention_heads=t["nh"], max_position_embeddings=t["npos"],
                         hidden_act=act, layer_norm_eps=1e-5, projection_dim=t["proj"],
                         bos_token_id=49406, eos_token_id=49407, pad_token_id=1),
        vision_config=dict(hidden_size=v["h"], intermediate_size=v["ff"], num_hidden_layers=v["L"],
                           num_attention_heads=v["nh"], image_size=v["S"], patch_size=v["P"],
                           hidden_act=act, layer_norm_eps=1e-5, projection_dim=v["proj"]),
        projection_dim=v["proj"])
    model = CLIPModel(hf_cfg).eval().float()
    tmp = os.path.join("/tmp", "hfx_%s_%d.gguf" % (config, seed))
    master = fixtures.make_model(tmp, config, "f32", True, True, seed, use_gelu=False, keep_master=True)
    sd = model.state_dict()
    used = set()
    for name in sd:
        if name in ("logit_scale",) or name.endswith("position_ids"):
            continue
        g = gguf_name(name)
        w = master[g]
        used.add(g)
        assert tuple(sd[name].shape) == tuple(w.shape), (name, g, sd[name].shape, w.shape)
        if g == "v.patch_embd.weight" or (f16_weights and w.ndim == 2 and g.endswith("weight")):
            w = w.astype(np.float16).astype(np.float32)  # the GGUF stores the conv kernel (and, in f16 files, every 2-D weight) in f16
        sd[name] = torch.from_numpy(w.copy())
    assert used == set(master), set(master) - used
    model.load_state_dict(sd)

    imgs = fixtures.synthetic_images(B, v["S"], seed=7)
    texts = fixtures.synthetic_token_ids(4, seed=11, min_len=1, max_len=12)
    with torch.no_grad():
        px = torch.from_numpy(imgs).permute(0, 3, 1, 2).contiguous()
        vout = model.vision_model(pixel_values=px)
        img_emb = model.visual_projection(vout.pooler_output).numpy()
        txt_emb = []
        for ids in texts:
            tout = model.text_model(input_ids=torch.from_numpy(ids.astype(np.int64))[None])
            txt_emb.append(model.text_projection(tout.pooler_output)[0].numpy())
    out = dict(images=imgs, image_embeds=img_emb, checksum=np.array(weights_checksum(master)),
               config=np.array(config), seed=np.array(seed), act=np.array(act), f16_weights=np.array(f16_weights))
    for i, (ids, e) in enumerate(zip(texts, txt_emb)):
        out["ids_%d" % i] = ids
        out["text_embeds_%d" % i] = e
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, out_name or ("hf_%s.npz" % config))
    np.savez_compressed(path, **out)
    os.remove(tmp)
    return path


if __name__ == "__main__":
    for c in (sys.argv[1:] or ["tiny", "tiny14"]):
        print("wrote", run(c))
    print("wrote", run("tiny", out_name="hf_tiny_erf_gelu.npz", act="gelu"))            # exact-GELU network (bounds ggml's tanh GELU)
    print("wrote", run("tiny14", out_name="hf_tiny14_f16w.npz", f16_weights=True))      # weights as an f16 file stores them
