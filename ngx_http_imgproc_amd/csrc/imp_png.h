// imp_png.h -- the host side of the PNG front (imp_png.cpp), shared with the kernel file (imp_png.hip)
#pragma once
#include <cstddef>

namespace imp {

constexpr int PNG_MAX_W = 4096;          // what k_png_unfilter's LDS layout holds
constexpr int PNG_MAX_H = 16384;

struct PngHeader {
    int w = 0, h = 0, bpp = 0;
    bool taken = false;                  // within what k_png_unfilter does
};

// IMP_OK with H filled; IMP_ERROR_UNSUPPORTED = not a PNG at all; IMP_ERROR_DECODE_FAILED = a PNG whose IHDR is damaged
int png_header(const unsigned char* blob, size_t size, PngHeader* H);
// the chunk walk (every CRC checked), the zlib stream of the IDAT chunks inflated into dst[h * (w * bpp + 1)], the filter
// bytes checked: IMP_OK or IMP_ERROR_DECODE_FAILED
// rows (optional) is called while the inflate goes, at the ends of deflate blocks: rows [0, complete) of dst are final and their
// filter bytes checked; a non-zero return stops the decode with that code.
typedef int (*png_rows_fn)(void* ctx, int complete);
int png_scanlines(const unsigned char* blob, size_t size, const PngHeader& H, unsigned char* dst, png_rows_fn rows = nullptr, void* ctx = nullptr);

}  // namespace imp
