// Host-side stress of the batched C-ABI entry points ON A GPU under a sanitizer (ASan + UBSan, or TSan): staging rings, packer threads,
// ragged-text metadata ring, sharded replica threads (two replicas over-subscribing one device), two contexts served from two threads.
// Checks only that the calls succeed, agree with themselves across batch sizes and that the sanitizer stays silent; numerics are the
// business of tests/.  Built and run by scripts/fuzz/run_gpu.sh (dev aid).
//   api_stress <two-tower gguf>
#include "clip.h"
#include "clip_amd.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

static int fails = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); fails++; } } while (0)

static float max_abs_diff(const float * a, const float * b, size_t n) { float m = 0; for (size_t i = 0; i < n; i++) m = std::max(m, std::fabs(a[i] - b[i])); return m; }

static void exercise(clip_ctx * ctx, unsigned seed, bool small) {
    std::mt19937 rng(seed);
    const int S = clip_get_vision_hparams(ctx)->image_size, proj = clip_get_vision_hparams(ctx)->projection_dim;
    const size_t per = (size_t)3 * S * S;
    const int NMAX = small ? 70 : 600;
    std::vector<float> pix((size_t)NMAX * per);
    { std::normal_distribution<float> nd(0.f, 1.f); for (size_t i = 0; i < per * 8; i++) pix[i] = nd(rng); for (size_t i = per * 8; i < pix.size(); i++) pix[i] = pix[i % (per * 8)] * (1.f + 0.001f * (float)(i / per)); }
    std::vector<clip_image_f32> imgs(NMAX);
    for (int i = 0; i < NMAX; i++) imgs[i] = clip_image_f32{S, S, pix.data() + (size_t)i * per, per};
    std::vector<float> ref((size_t)NMAX * proj), out((size_t)NMAX * proj);
    clip_image_f32_batch all{imgs.data(), (size_t)NMAX};
    CHECK(clip_image_batch_encode(ctx, 4, &all, ref.data(), true));
    for (int B : {1, 3, 64, NMAX / 2 + 1}) for (int thr : {1, 4}) {
        const int lo = (int)(rng() % (NMAX - B + 1));
        clip_image_f32_batch b{imgs.data() + lo, (size_t)B};
        CHECK(clip_image_batch_encode(ctx, thr, &b, out.data(), true));
        CHECK(max_abs_diff(out.data(), ref.data() + (size_t)lo * proj, (size_t)B * proj) < 2e-3f);
    }
    // raw u8 images of mixed sizes, several staging pieces
    const int NU = small ? 40 : 300;
    std::vector<std::vector<uint8_t>> store(NU);
    std::vector<clip_image_u8> u8(NU);
    for (int i = 0; i < NU; i++) { const int nx = S + (int)(rng() % 200), ny = S + (int)(rng() % 150); store[i].resize((size_t)3 * nx * ny); for (auto & v : store[i]) v = (uint8_t)rng(); u8[i] = clip_image_u8{nx, ny, store[i].data(), store[i].size()}; }
    std::vector<float> uref((size_t)NU * proj), uout((size_t)NU * proj);
    CHECK(clip_amd_image_batch_encode_u8(ctx, u8.data(), NU, uref.data(), true));
    for (int n : {1, 5, NU / 2}) { CHECK(clip_amd_image_batch_encode_u8(ctx, u8.data() + 3, n, uout.data(), true)); CHECK(max_abs_diff(uout.data(), uref.data() + (size_t)3 * proj, (size_t)n * proj) < 2e-3f); }
    // ragged text batches
    const int NT = small ? 60 : 400, nvocab = clip_get_text_hparams(ctx)->n_vocab, npos = clip_get_text_hparams(ctx)->num_positions;
    std::vector<std::vector<int32_t>> ids(NT);
    std::vector<clip_tokens> toks(NT);
    for (int i = 0; i < NT; i++) { const int len = 1 + (int)(rng() % (npos - 2)); ids[i].push_back(nvocab - 2); for (int k = 0; k < len; k++) ids[i].push_back((int32_t)(rng() % (nvocab - 2))); ids[i].push_back(nvocab - 1); toks[i] = clip_tokens{ids[i].data(), ids[i].size()}; }
    std::vector<float> tref((size_t)NT * proj), tout((size_t)NT * proj);
    CHECK(clip_text_batch_encode(ctx, 4, toks.data(), NT, tref.data(), true));
    for (int n : {1, 2, 17, NT / 2}) { const int lo = (int)(rng() % (NT - n + 1)); CHECK(clip_text_batch_encode(ctx, 2, toks.data() + lo, n, tout.data(), true)); CHECK(max_abs_diff(tout.data(), tref.data() + (size_t)lo * proj, (size_t)n * proj) < 2e-3f); }
    CHECK(clip_text_encode(ctx, 1, &toks[0], tout.data(), true));
    CHECK(max_abs_diff(tout.data(), tref.data(), proj) < 2e-3f);
    // zero-shot, batched on the device and the reference's one-image form
    const char * labels[5] = {"cat", "dog", "a red apple", "car", "tree"};
    std::vector<float> sc((size_t)8 * 5); std::vector<int> ix((size_t)8 * 5);
    CHECK(clip_amd_zero_shot_label_images(ctx, u8.data(), 8, labels, 5, sc.data(), ix.data()));
    float s1[5]; int i1[5];
    CHECK(clip_zero_shot_label_image(ctx, 2, &u8[0], labels, 5, s1, i1));
    float score = 0;
    CHECK(clip_compare_text_and_image(ctx, 2, "a photo of a dog", &u8[1], &score));
}

int main(int argc, char ** argv) {
    if (argc < 2) return 2;
    setenv("CLIP_AMD_WEIGHT_CACHE", "0", 1);
    clip_ctx * a = clip_model_load(argv[1], 0);
    CHECK(a != nullptr);
    if (!a) return 1;
    exercise(a, 1, false);
    // two contexts, two host threads (the serving form)
    clip_ctx * b = clip_model_load(argv[1], 0);
    CHECK(b != nullptr);
    if (b) {
        std::thread t1([&] { exercise(a, 2, true); }), t2([&] { exercise(b, 3, true); });
        t1.join(); t2.join();
        clip_free(b);
    }
    clip_free(a);
    // two replicas sharing one device (sharded runner, replica threads, per-replica staging); no collective in this form
    setenv("CLIP_AMD_MULTI_OVERSUBSCRIBE", "1", 1);
    clip_ctx * m = clip_amd_model_load_multi(argv[1], 0, 2);
    CHECK(m != nullptr);
    if (m) { CHECK(clip_amd_ctx_device_count(m) == 2); exercise(m, 4, true); clip_free(m); }
    fprintf(stderr, "api_stress: %d failed checks\n", fails);
    fflush(nullptr);
    _exit(fails ? 1 : 0);       // (not through the exit handlers: ASan's device allocator asserts inside the HIP runtime's finalizer)
}
