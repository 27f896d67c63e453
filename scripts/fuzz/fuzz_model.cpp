// sanitizer + mutation harness for clip_model_load / clip_model_quantize on malformed GGUF files (dev aid; host-only context;
// built and run by scripts/fuzz/run.sh)
//   fuzz_model <iterations per seed> <scratch file> <rng seed> <seed GGUF files...>
#include "clip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <random>
#include <string>
static std::vector<uint8_t> slurp(const char * f) { std::vector<uint8_t> v; FILE * fp = fopen(f, "rb"); if (!fp) return v; fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, fp) != (size_t)n) v.clear(); fclose(fp); return v; }
int main(int argc, char ** argv) {
    const int iters = atoi(argv[1]);
    const std::string tmp = argv[2];
    std::mt19937 rng(atoi(argv[3]));
    setenv("CLIP_AMD_ALLOW_NO_DEVICE", "1", 1);
    setenv("CLIP_AMD_WEIGHT_CACHE", "0", 1);
    long ok = 0, tot = 0;
    for (int a = 4; a < argc; a++) {
        std::vector<uint8_t> seed = slurp(argv[a]);
        if (seed.empty()) continue;
        // header region = everything before the tensor data: find it roughly as the first 1/8 of the file or 64 KB
        for (int it = 0; it < iters; it++) {
            std::vector<uint8_t> d = seed;
            const int kind = rng() % 7;
            const int nmut = 1 + rng() % 4;
            const size_t hdr = std::min<size_t>(d.size(), (rng() & 1) ? 2048 : 700000);
            if (it > 0) for (int m = 0; m < nmut; m++) {
                if (d.empty()) break;
                size_t pos = rng() % hdr;
                switch (kind) {
                case 0: d[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 1: d[pos] = (uint8_t)rng(); break;
                case 2: d[pos] = (rng() & 1) ? 0xFF : 0x00; break;
                case 3: d.resize(rng() % d.size()); break;
                case 4: { size_t n = 1 + rng() % 16; if (pos + n < d.size()) d.erase(d.begin() + pos, d.begin() + pos + n); } break;
                case 5: { size_t n = 1 + rng() % 16; std::vector<uint8_t> ins(n); for (auto & b : ins) b = (uint8_t)rng(); d.insert(d.begin() + pos, ins.begin(), ins.end()); } break;
                case 6: { if (pos + 8 < d.size()) { uint64_t v = (rng() & 1) ? ~0ull : ((uint64_t)rng() << (rng() % 40)); memcpy(&d[pos & ~(size_t)3], &v, 8); } } break;
                }
            }
            FILE * fp = fopen(tmp.c_str(), "wb"); fwrite(d.data(), 1, d.size(), fp); fclose(fp);
            tot++;
            clip_ctx * ctx = clip_model_load(tmp.c_str(), 0);
            if (ctx) {
                ok++;
                clip_tokens t{};
                if (clip_tokenize(ctx, "a photo of a cat, 12 dogs & more!", &t)) delete[] t.data;
                (void)clip_get_text_hparams(ctx); (void)clip_get_vision_hparams(ctx);
                clip_free(ctx);
            }
            if (it % 4 == 0) { std::string out = tmp + ".q"; (void)clip_model_quantize(tmp.c_str(), out.c_str(), 2 + (int)(rng() % 2)); }
        }
    }
    printf("%ld/%ld loaded\n", ok, tot);
    return 0;
}
