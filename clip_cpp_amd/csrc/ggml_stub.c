/* ggml_stub.c — stub libggml.so: only the timing helpers the reference's callers use
 * (include/ggml/ggml.h).  There is no ggml in this build. */
#include <stdint.h>
#include <time.h>

static struct timespec g_t0;

static int64_t now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)(ts.tv_sec - g_t0.tv_sec) * 1000000 + (ts.tv_nsec - g_t0.tv_nsec) / 1000;
}
void ggml_time_init(void) { clock_gettime(CLOCK_MONOTONIC, &g_t0); }
int64_t ggml_time_us(void) { return now_us(); }
int64_t ggml_time_ms(void) { return now_us() / 1000; }
