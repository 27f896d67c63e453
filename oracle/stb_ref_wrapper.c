/* stb_ref_wrapper.c — builds the REFERENCE's own image decoder (vendored stb_image.h, used by
 * clip_image_load_from_file at reference clip.cpp:709-726) into oracle/_ref/libstb_ref.so, from the
 * header where it lies under /root/reference (never copied into this repo).  Test infrastructure:
 * used to check the product's JPEG/PNG decoders against the reference's decoder. */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_STDIO
#include "stb_image.h"

unsigned char * stbref_load_from_memory(const unsigned char * buf, int len, int * x, int * y, int * comp) {
    return stbi_load_from_memory(buf, len, x, y, comp, 3);
}
void stbref_free(void * p) { stbi_image_free(p); }
