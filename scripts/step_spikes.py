"""Where do one-off stalls sit?  Times every call of a small-batch forward (host clock, stream synchronised per call) and prints the outliers.
usage: python scripts/step_spikes.py [batch] [calls] [vision|text|both]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import clip_cpp_amd
from clip_cpp_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
what = sys.argv[3] if len(sys.argv) > 3 else "both"
path = synth.cached_model(os.environ.get("CLIP_AMD_FIXTURE_CACHE", "/tmp/clip_amd_fixtures"), "b32", "q4_0", text=True, vision=True)
clip = clip_cpp_amd.Clip(path, device=0, verbosity=0)
st = torch.cuda.Stream()
clip.set_stream(st.cuda_stream)
S, proj = clip.vision_config["image_size"], clip.vision_config["projection_dim"]
imgs = torch.randn(B, S, S, 3, device="cuda")
out = torch.empty(B, proj, device="cuda")
texts = synth.token_ids(B, seed=11, min_len=1, max_len=75)
ids = torch.tensor(np.concatenate(texts), dtype=torch.int32, device="cuda")
offs = np.concatenate([[0], np.cumsum([len(t) for t in texts])]).astype(np.int32)
tout = torch.empty(B, proj, device="cuda")
torch.cuda.synchronize()
if os.environ.get("SPIKES_NOGC") == "1":
    import gc
    gc.collect(); gc.disable()
ts = []
for i in range(N):
    t0 = time.perf_counter()
    if what in ("vision", "both"):
        clip.encode_images_device(imgs.data_ptr(), B, out.data_ptr())
    if what in ("text", "both"):
        clip.encode_texts_device(ids.data_ptr(), offs, tout.data_ptr())
    st.synchronize()
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
med = float(np.median(ts))
print("batch %d %s graphs=%s: median %.3f ms, mean %.3f ms over %d calls" % (B, what, os.environ.get("CLIP_AMD_GRAPHS", "1"), med, ts.mean(), N))
for i in np.nonzero(ts > 4 * med + 1.0)[0]:
    print("   call %5d: %.2f ms" % (i, ts[i]))
