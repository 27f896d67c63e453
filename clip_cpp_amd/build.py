"""Build libclip.so (host C++ + gfx950 HIP kernels) in-tree with hipcc.

    python -m clip_cpp_amd.build [--force]

Objects go to clip_cpp_amd/build/, the shared library to clip_cpp_amd/libclip.so (git-ignored, travels
to the GPU box with the gpurun snapshot).  The GEMM translation unit is compiled once per weight type
(-DCLIPAMD_GEMM_WT=n) so the six instantiation sets build in parallel.  A stub libggml.so that only
exports the ggml_time_* shim is built next to it because the reference's ctypes binding dlopens
"./libggml.so" before "./libclip.so" (reference examples/python_bindings/clip_cpp/clip.py:28-30).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libclip.so")
GGML_STUB = os.path.join(HERE, "libggml.so")
ARCH = "gfx950"

HOST_SOURCES = ["gguf.cpp", "quant.cpp", "load.cpp", "forward.cpp", "tokenizer.cpp", "preprocess.cpp", "image_io.cpp", "image_formats.cpp",
                "jpeg_decode.cpp", "host_pipeline.cpp", "api.cpp"]
HIP_SOURCES = ["k_attn.hip", "k_attn_f32.hip", "k_misc.hip", "k_preproc.hip", "k_gemm.hip", "k_gemm8.hip", "k_gemm4.hip", "k_gemm32.hip", "k_gemm_f32.hip", "k_skinny.hip", "k_gemm_ring.hip", "k_fold.hip"]
GEMM_WTYPES = [0, 1, 2, 3, 4, 5]
# Per-file flags.  The f32-file kernels (a correctness path, not a tuned one) are built without the SLP vectoriser: packed f32 instructions that
# consume a transcendental's result one wait state later are the suspect of the round-6 epilogue hazard (gemm_common.h GELU_SCALAR_FENCE,
# profiles/r06_experiments.txt section 8), and hipcc forms exactly those around the online softmax of k_attn_f32.hip when it may.
EXTRA_FLAGS = {"k_attn_f32.hip": ["-fno-slp-vectorize"], "k_gemm_f32.hip": ["-fno-slp-vectorize"]}

COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-inline-asm", "-Wno-bitwise-instead-of-logical",
          "-I" + os.path.join(os.path.dirname(HERE), "include")]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    cc = hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(os.path.dirname(HERE), "include", f) for f in ("clip.h", "clip_amd.h")]
    jobs = []
    objs = []
    for src in HOST_SOURCES:
        o = os.path.join(BUILD, src + ".o")
        objs.append(o)
        s = os.path.join(CSRC, src)
        if force or _newer(o, [s] + headers):
            # host files include hip_runtime.h for the runtime API only: compile them as HIP too (no kernels inside)
            # -fwrapv: the file decoders run integer transforms over UNTRUSTED coefficients (a corrupt JPEG overflows the 32-bit IDCT
            # products); wrapping is what the reference's decoder does in practice and it keeps the behaviour defined
            jobs.append([cc, "-x", "hip", "--offload-arch=" + ARCH, "-fwrapv"] + COMMON + ["-c", s, "-o", o])
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([cc, "--offload-arch=" + ARCH] + COMMON + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o])
    s = os.path.join(CSRC, "k_gemm.hip")
    for wt in GEMM_WTYPES:
        o = os.path.join(BUILD, "k_gemm_wt%d.o" % wt)
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([cc, "--offload-arch=" + ARCH] + COMMON + ["-DCLIPAMD_GEMM_WT=%d" % wt, "-c", s, "-o", o])
    s = os.path.join(CSRC, "k_skinny.hip")       # same scheme for the small-M kernel: one object per weight type + the dispatcher
    for wt in GEMM_WTYPES:
        o = os.path.join(BUILD, "k_skinny_wt%d.o" % wt)
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([cc, "--offload-arch=" + ARCH] + COMMON + ["-DCLIPAMD_SKINNY_WT=%d" % wt, "-c", s, "-o", o])
    s = os.path.join(CSRC, "k_gemm_ring.hip")    # and the mid-M ring kernel
    for wt in GEMM_WTYPES:
        o = os.path.join(BUILD, "k_gemm_ring_wt%d.o" % wt)
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([cc, "--offload-arch=" + ARCH] + COMMON + ["-DCLIPAMD_RING_WT=%d" % wt, "-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    if force or jobs or _newer(LIB, objs):
        _run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-lpthread", "-ldl"])
    stub_src = os.path.join(CSRC, "ggml_stub.c")
    if force or _newer(GGML_STUB, [stub_src]):
        _run(["gcc", "-O2", "-fPIC", "-shared", "-o", GGML_STUB, stub_src])
    return LIB


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
