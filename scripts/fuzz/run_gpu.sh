#!/bin/bash
# GPU-side companion of run.sh: the batched C-ABI entry points of a two-tower ViT-B/32 q4_0 file under ASan + UBSan and under TSan
# (host code instrumented; kernels are the objects of the normal build).  Meant for `gpurun -- 'bash scripts/fuzz/run_gpu.sh'`.
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$ROOT"
W=${FUZZ_DIR:-$ROOT/scripts/fuzz/_build}; mkdir -p "$W" gpurun_out     # (git-ignored; travels with the gpurun snapshot when pre-built with BUILD_ONLY=1)
CL=/opt/rocm/lib/llvm/bin/clang++
[ -z "$BUILD_ONLY" ] && MODEL=$(python3 -c "from clip_cpp_amd import synth; print(synth.cached_model('/tmp/clip_amd_fixtures', 'b32', 'q4_0', text=True, vision=True, seed=1234))" | tail -1)
for SAN in address,undefined thread; do
    T=${SAN%%,*}; mkdir -p "$W/obj_$T"
    FL="-std=c++17 -g -O1 -fwrapv -fPIC -fsanitize=$SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include -I$ROOT/clip_cpp_amd/csrc"
    if [ ! -x "$W/api_stress_$T" ] || [ scripts/fuzz/api_stress.cpp -nt "$W/api_stress_$T" ] || [ -n "$(find clip_cpp_amd/csrc -newer "$W/api_stress_$T" -name '*.cpp' -o -newer "$W/api_stress_$T" -name '*.h' | head -1)" ]; then
        for f in gguf quant load forward tokenizer preprocess image_io jpeg_decode host_pipeline api; do $CL $FL -c "clip_cpp_amd/csrc/$f.cpp" -o "$W/obj_$T/$f.o" & done; wait
        $CL $FL scripts/fuzz/api_stress.cpp "$W/obj_$T"/*.o $(ls clip_cpp_amd/build/*.o | grep -E "/k_") -L/opt/rocm/lib -lamdhip64 -lz -lpthread -ldl -Wl,-rpath,/opt/rocm/lib -o "$W/api_stress_$T" || exit 1
    fi
    [ -n "$BUILD_ONLY" ] && continue
    echo "== $SAN"
    ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 TSAN_OPTIONS=report_signal_unsafe=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
        timeout 400 "$W/api_stress_$T" "$MODEL" > "gpurun_out/api_stress_$T.log" 2>&1
    echo "rc=$?"
    # reports whose summary line points into the (uninstrumented) HIP / HSA runtime are not this library's
    echo "sanitizer reports outside /opt/rocm: $(grep -E "^SUMMARY: (Thread|Address|UndefinedBehavior)Sanitizer" "gpurun_out/api_stress_$T.log" | grep -vc "/opt/rocm/")"
    grep -E "CHECK failed line|api_stress:" "gpurun_out/api_stress_$T.log" | tail -5
done
