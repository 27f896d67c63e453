"""Phase stamps of the small-M (skinny) kernels over ONE ViT-B/32 q4_0 image forward (timing build: scripts/build_sk_timing.sh).
   CLIP_AMD_LIB=clip_cpp_amd/variants/libclip_sktiming.so CLIP_AMD_GRAPHS=0 python scripts/sk_stamps.py
Per launch (launch order: ln1_qkv, out_resid, ln2_ffn_up, ffn_down_resid per layer), for the first and the last workgroup: shader
cycles from kernel entry to [loads requested, LayerNorm prologue done, MFMAs issued, partial sums exchanged, stores issued] and the
workgroup's lifetime on the 100 MHz real-time clock."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import numpy as np  # noqa: E402
import clip_cpp_amd as cc  # noqa: E402
from clip_cpp_amd import synth  # noqa: E402

L = cc.lib()
path = synth.cached_model(os.environ.get("CLIP_AMD_FIXTURE_CACHE", "/tmp/clip_amd_fixtures"), "b32", "q4_0", text=True, vision=True, seed=1234)
clip = cc.Clip(path, device=0)
img = np.random.default_rng(0).standard_normal((1, 224, 224, 3)).astype(np.float32)
for _ in range(5):
    clip.encode_images(img)
buf = (C.c_ulonglong * (512 * 16))()
L.clip_amd_debug_read_sk_stamps.restype = C.c_int
L.clip_amd_debug_read_sk_stamps(buf, 512)                 # reset the launch counter
clip.encode_images(img)
n = L.clip_amd_debug_read_sk_stamps(buf, 512)
names = ["ln1_qkv", "out_resid", "ln2_ffn_up", "ffn_down_resid"]
print("# %d skinny launches in one image forward; cycles from entry: req = first loads requested, ln = LayerNorm prologue done, mfma = MFMAs issued, red = sums exchanged, st = stores issued; life = us on the 100 MHz clock" % n)
acc = {}
for i in range(n):
    s = [int(buf[i * 16 + j]) for j in range(16)]
    nm = names[i % 4] if n % 4 == 0 else "launch"
    for w, o in (("first", 0), ("last", 8)):
        t0 = s[o]
        if not t0:
            continue
        ph = [s[o + j] - t0 for j in range(1, 6)]
        life = (s[o + 7] - s[o + 6]) / 100.0
        acc.setdefault((nm, w), []).append(ph + [life])
        if i < 8:
            print("launch %2d %-15s %-5s WG: req %5d ln %6d mfma %6d red %6d st %6d cycles | life %.2f us" % (i, nm, w, *ph, life))
print("# mean over the layers")
for (nm, w), v in sorted(acc.items()):
    m = np.mean(np.array(v, dtype=np.float64), axis=0)
    print("%-15s %-5s WG: req %5.0f ln %6.0f mfma %6.0f red %6.0f st %6.0f cycles | life %.2f us  (%d launches)" % (nm, w, *m, len(v)))
# spacing of consecutive launches (first workgroup's entry on the real-time clock)
ent = [int(buf[i * 16 + 6]) for i in range(n) if int(buf[i * 16 + 6])]
if len(ent) > 2:
    d = np.diff(np.array(ent, dtype=np.float64)) / 100.0
    print("# entry-to-entry spacing of consecutive skinny launches (us; the attention launch sits inside every 4th gap): mean %.2f min %.2f max %.2f" % (d.mean(), d.min(), d.max()))
    print("# by position in the layer:", " ".join("%s->next %.2f" % (names[k], d[k::4].mean()) for k in range(4)))
