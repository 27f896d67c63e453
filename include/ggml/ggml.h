/* ggml/ggml.h — timing shim.
 *
 * The reference's public header includes "ggml/ggml.h" (reference clip.h:4) and its
 * example programs call ggml_time_init()/ggml_time_us() (examples/main.cpp:7-8,
 * tests/benchmark.cpp:57,63, models/quantize.cpp:30).  This library contains no ggml;
 * the two timing symbols are provided (exported from libclip.so and from the stub
 * libggml.so that the reference's ctypes binding dlopens, clip_cpp/clip.py:28-30) so
 * those callers build unchanged.
 */
#ifndef CLIP_AMD_GGML_SHIM_H
#define CLIP_AMD_GGML_SHIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

void ggml_time_init(void);
int64_t ggml_time_ms(void);
int64_t ggml_time_us(void);

#ifdef __cplusplus
}
#endif

#endif
