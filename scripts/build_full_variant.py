"""Full library variant for same-box A/B runs: EVERY translation unit rebuilt with extra compiler flags into clip_cpp_amd/variants/NAME/ and linked as
clip_cpp_amd/variants/libclip_NAME.so (load it with CLIP_AMD_LIB=<path>; scripts/gpu_session.sh ab:N:-:CLIP_AMD_LIB=@/clip_cpp_amd/variants/libclip_NAME.so).
build_variant.sh rebuilds the tiled-GEMM units only — enough for tile / schedule ablations, not for switches that every kernel family must share
(-DCLIPAMD_LNAPPLY_R5, -DCLIPAMD_GELU_R5, -DCLIPAMD_DEQUANT_R5, -fno-slp-vectorize).
usage: python scripts/build_full_variant.py NAME [flags ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clip_cpp_amd.build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
b.BUILD = os.path.join(b.HERE, "variants", name)
b.LIB = os.path.join(b.HERE, "variants", "libclip_%s.so" % name)
b.GGML_STUB = os.path.join(b.BUILD, "libggml.so")
b.COMMON = b.COMMON + flags
os.makedirs(b.BUILD, exist_ok=True)
print(b.build(force=False, verbose=False))
