// jpeg_decode.cpp — JPEG decoder for clip_image_load_from_file (placeholder: not implemented yet).
#include <cstdint>
#include <string>
#include <vector>

namespace clipamd {

bool decode_jpeg(const uint8_t * data, size_t size, std::vector<uint8_t> & rgb, int & nx, int & ny, std::string & err) {
    (void)rgb; (void)nx; (void)ny;
    if (size >= 2 && data[0] == 0xFF && data[1] == 0xD8) err = "JPEG decoding is not implemented yet (use PNG/BMP/PNM)";
    return false;
}

}  // namespace clipamd
