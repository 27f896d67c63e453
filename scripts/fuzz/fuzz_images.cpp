// sanitizer + mutation harness for the image decoders (dev aid; built and run by scripts/fuzz/run.sh)
//   fuzz_images <iterations per seed> <scratch file> <rng seed> <seed files...>
#include "clip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <random>
namespace clipamd { bool load_image_file(const char * fname, clip_image_u8 * img); }
static std::vector<uint8_t> slurp(const char * f) { std::vector<uint8_t> v; FILE * fp = fopen(f, "rb"); if (!fp) return v; fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, fp) != (size_t)n) v.clear(); fclose(fp); return v; }
int main(int argc, char ** argv) {
    const int iters = atoi(argv[1]);
    const char * tmp = argv[2];
    std::mt19937 rng(atoi(argv[3]));
    long ok = 0, tot = 0;
    for (int a = 4; a < argc; a++) {
        std::vector<uint8_t> seed = slurp(argv[a]);
        if (seed.empty()) continue;
        for (int it = 0; it < iters; it++) {
            std::vector<uint8_t> d = seed;
            const int kind = rng() % 6;
            const int nmut = 1 + rng() % 8;
            if (it > 0) for (int m = 0; m < nmut; m++) {
                if (d.empty()) break;
                size_t pos = rng() % d.size();
                if (rng() % 3 == 0) pos = rng() % std::min<size_t>(d.size(), 700);   // headers
                switch (kind) {
                case 0: d[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 1: d[pos] = (uint8_t)rng(); break;
                case 2: d[pos] = (rng() & 1) ? 0xFF : 0x00; break;
                case 3: d.resize(pos); break;                                    // truncate
                case 4: { size_t n = 1 + rng() % 16; if (pos + n < d.size()) d.erase(d.begin() + pos, d.begin() + pos + n); } break;
                case 5: { size_t n = 1 + rng() % 16; std::vector<uint8_t> ins(n); for (auto & b : ins) b = (uint8_t)rng(); d.insert(d.begin() + pos, ins.begin(), ins.end()); } break;
                }
            }
            FILE * fp = fopen(tmp, "wb"); fwrite(d.data(), 1, d.size(), fp); fclose(fp);
            clip_image_u8 img{};
            tot++;
            if (clipamd::load_image_file(tmp, &img)) {
                ok++;
                if (img.nx <= 0 || img.ny <= 0 || !img.data) { fprintf(stderr, "bad success\n"); abort(); }
                volatile unsigned s = 0; for (size_t k = 0; k < (size_t)3 * img.nx * img.ny; k++) s += img.data[k];
                delete[] img.data;
            }
        }
    }
    printf("%ld/%ld decoded\n", ok, tot);
    return 0;
}
