"""HF -> GGUF converter (clip_cpp_amd/convert_hf_to_gguf.py; reference models/convert_hf_to_gguf.py) on a randomly
initialised Hugging Face CLIPModel built offline: the produced file must (a) carry the reference's names / dtypes / keys,
(b) reproduce HF's own image and text features through the oracle (f32 file, ideal numerics), (c) load through libclip.so
and survive clip_model_quantize."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import fixtures, ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hf_dir(tmp_path_factory):
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    d = tmp_path_factory.mktemp("hf") / "ggml_tiny-clip"
    cfg = tr.CLIPConfig(
        text_config=dict(vocab_size=fixtures.N_VOCAB, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                         max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=32,
                         bos_token_id=49406, eos_token_id=49407, pad_token_id=1),
        vision_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=32, patch_size=8,
                           hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=32),
        projection_dim=32)
    torch.manual_seed(0)
    model = tr.CLIPModel(cfg).eval().float()
    with torch.no_grad():                      # HF initialises biases / LN to 0 / 1: randomise so that every tensor matters
        for n, p in model.named_parameters():
            if p.ndim == 1 and "logit_scale" not in n:
                p.add_(torch.randn_like(p) * 0.05)
    model.save_pretrained(str(d))
    vocab = {t: i for i, t in enumerate(fixtures.synthetic_vocab())}
    assert len(vocab) == fixtures.N_VOCAB
    (d / "vocab.json").write_text(json.dumps(vocab), encoding="utf-8")
    (d / "preprocessor_config.json").write_text(json.dumps({"image_mean": [0.5, 0.4, 0.3], "image_std": [0.2, 0.25, 0.3]}))
    return str(d), model


def _convert(d, *extra):
    r = subprocess.run([sys.executable, "-m", "clip_cpp_amd.convert_hf_to_gguf", "-m", d, *extra], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = [l for l in r.stdout.splitlines() if l.startswith("Done. Output file: ")]
    assert out, r.stdout[-500:]
    return out[-1].split(": ", 1)[1], r.stdout


def test_f32_file_reproduces_hf_features_through_the_oracle(hf_dir, tmp_path):
    import torch
    d, model = hf_dir
    path, log = _convert(d, "--use-f32", "-o", str(tmp_path / "ggml_out"))
    assert os.path.basename(path) == "out_ggml-model-f32.gguf"           # reference naming: <prefix minus "ggml_">_ggml-model-<ftype>.gguf
    assert "skipping parameter: logit_scale" in log
    orc = ref.OracleModel(path)
    assert orc.info["has_text"] == 1 and orc.info["has_vision"] == 1 and orc.info["use_gelu"] == 0 and orc.info["ftype"] == 0
    imgs = fixtures.synthetic_images(3, 32, seed=5)
    texts = fixtures.synthetic_token_ids(4, seed=6, min_len=1, max_len=12)
    with torch.no_grad():
        want_i = model.visual_projection(model.vision_model(pixel_values=torch.from_numpy(imgs).permute(0, 3, 1, 2).contiguous()).pooler_output).numpy()
        want_t = [model.text_projection(model.text_model(input_ids=torch.from_numpy(t.astype(np.int64))[None]).pooler_output)[0].numpy() for t in texts]
    got_i = orc.image_batch_encode(imgs, normalize=False, mode=ref.MODE_IDEAL)
    # the conv kernel is stored in f16 even in f32 files (reference :182-185): tolerance covers that rounding only
    np.testing.assert_allclose(got_i, want_i, atol=2e-3, rtol=2e-3)
    for t, w in zip(texts, want_t):
        np.testing.assert_allclose(orc.text_encode(t, normalize=False, mode=ref.MODE_IDEAL), w, atol=2e-5, rtol=1e-4)


def test_f16_tower_files_names_dtypes_and_library_load(hf_dir, tmp_path, clip_lib):
    d, model = hf_dir
    from clip_cpp_amd import gguf_inspect
    both, _ = _convert(d, "-o", str(tmp_path / "o1"))
    tonly, _ = _convert(d, "--text-only", "-o", str(tmp_path / "o2"))
    vonly, _ = _convert(d, "--vision-only", "--image-mean", "0.1", "0.2", "0.3", "-o", str(tmp_path / "o3"))
    assert os.path.basename(both) == "o1_ggml-model-f16.gguf" and os.path.basename(tonly) == "o2_ggml-text-model-f16.gguf"
    assert os.path.basename(vonly) == "o3_ggml-vision-model-f16.gguf"
    kv, tensors = gguf_inspect(both)
    assert kv["general.architecture"] == "clip" and kv["general.file_type"] == 1 and kv["clip.use_gelu"] is False
    assert kv["clip.vision.image_mean"] == pytest.approx([0.5, 0.4, 0.3]) and len(kv["tokenizer.ggml.tokens"]) == fixtures.N_VOCAB
    names = {n: (dims, t) for n, dims, t in tensors}
    # the reference's renaming incl. its fc1 -> "ffn_down" quirk, ne0-first dims, f16 only for 2-D weights and the conv
    assert names["v.blk.0.ffn_down.weight"] == ([64, 128], 1) and names["v.blk.0.ffn_up.weight"] == ([128, 64], 1)
    assert names["v.patch_embd.weight"] == ([8, 8, 3, 64], 1) and names["v.class_embd"] == ([64], 0)
    assert names["t.token_embd.weight"] == ([64, fixtures.N_VOCAB], 1) and names["t.blk.1.attn_q.bias"] == ([64], 0)
    assert names["text_projection.weight"] == ([64, 32], 1) and names["v.pre_ln.weight"] == ([64], 0) and "logit_scale" not in names
    assert len(tensors) == 2 * (16 * 2) + 4 + 5 + 2 + 2      # per tower: 16/layer; vision +class,patch,pos,pre/post-LN(4); text +tok,pos,final LN(2); 2 projections
    kt, tt = gguf_inspect(tonly)
    kvv, tv = gguf_inspect(vonly)
    assert kt["clip.has_vision_encoder"] is False and all(not n.startswith("v") for n, _, _ in tt) and "clip.vision.image_size" not in kt
    assert kvv["clip.has_text_encoder"] is False and all(not n.startswith("t") for n, _, _ in tv) and kvv["clip.vision.image_mean"] == pytest.approx([0.1, 0.2, 0.3])
    # through the product library: load (host-only context when there is no GPU) and quantise
    env = dict(os.environ, CLIP_AMD_ALLOW_NO_DEVICE="1")
    q = str(tmp_path / "q.gguf")
    code = ("import clip_cpp_amd as c; L=c.lib(); ctx=c.Clip(%r); assert ctx.vision_config['image_size']==32 and ctx.text_config['n_layer']==2, (ctx.vision_config, ctx.text_config);"
            "assert ctx.tokenize('a photo of a cat')[0]==49406; assert L.clip_model_quantize(%r.encode(), %r.encode(), 2)") % (both, both, q)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert ref.OracleModel(q).info["ftype"] == 2


@pytest.mark.gpu
def test_converted_f16_file_on_the_gpu_matches_hf(hf_dir, tmp_path, clip_lib):
    """HF checkpoint -> convert_hf_to_gguf (f16) -> libclip.so on the MI355X == HF's own features (1 - cos <= 1e-4)."""
    import torch
    d, model = hf_dir
    path, _ = _convert(d, "-o", str(tmp_path / "g"))
    clip = clip_lib.Clip(path, device=0)
    imgs = fixtures.synthetic_images(4, 32, seed=8)
    texts = fixtures.synthetic_token_ids(5, seed=9, min_len=1, max_len=20)
    with torch.no_grad():
        want_i = model.visual_projection(model.vision_model(pixel_values=torch.from_numpy(imgs).permute(0, 3, 1, 2).contiguous()).pooler_output).numpy()
        want_t = np.stack([model.text_projection(model.text_model(input_ids=torch.from_numpy(t.astype(np.int64))[None]).pooler_output)[0].numpy() for t in texts])

    def omc(a, b):
        return 1.0 - (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1)
    assert np.all(omc(clip.encode_images(imgs, normalize=False), want_i) <= 1e-4)
    assert np.all(omc(clip.encode_texts(texts, normalize=False), want_t) <= 1e-4)
