// imp_png.cpp -- the host side of the PNG front: the file format (PNG specification, 2nd edition, section 5: signature,
// chunk layout, CRC; 11.2.2 IHDR; 10.1 the zlib stream across IDAT chunks; 9.2 filter types).  Host code only -- no HIP
// call -- so that tests/c/fuzz_host.cpp can run it under AddressSanitizer / UBSan on damaged files
// (tests/test_host_sanitizers.py); impgpu_image_decode_png (imp_png.hip) calls png_scanlines with the pinned staging buffer
// as its destination.  The inflate is imp_inflate.cpp's (1.6-1.9 x zlib 1.2.11 on scanlines); there is no device inflate in this
// library.
#include <zlib.h>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/impgpu.h"
#include "imp_inflate.h"
#include "imp_png.h"

namespace imp {

static unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

int png_header(const unsigned char* blob, size_t size, PngHeader* H) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (!blob || size < 8 || std::memcmp(blob, sig, 8) != 0) return IMP_ERROR_UNSUPPORTED;
    if (size < 8 + 25 || be32(blob + 8) != 13 || std::memcmp(blob + 12, "IHDR", 4) != 0) return IMP_ERROR_DECODE_FAILED;
    if (crc32_ieee(blob + 12, 17) != be32(blob + 29)) return IMP_ERROR_DECODE_FAILED;
    const unsigned w = be32(blob + 16), h = be32(blob + 20);
    const int depth = blob[24], colour = blob[25], compression = blob[26], filter = blob[27], interlace = blob[28];
    if (w == 0 || h == 0 || w > 0x7fffffffu || h > 0x7fffffffu || compression != 0 || filter != 0 || interlace > 1)
        return IMP_ERROR_DECODE_FAILED;
    H->w = (int)w; H->h = (int)h;
    H->bpp = colour == 0 ? 1 : colour == 2 ? 3 : colour == 6 ? 4 : 0;
    H->taken = depth == 8 && H->bpp != 0 && interlace == 0 && w <= (unsigned)PNG_MAX_W && h <= (unsigned)PNG_MAX_H;
    return IMP_OK;
}

namespace {
struct RowWatch {
    const PngHeader* H;
    unsigned char* dst;
    size_t rstride;
    int checked = 0;                     // rows whose filter byte has been looked at
    png_rows_fn rows;
    void* ctx;
    int code = IMP_OK;
};

// 9.2: filter types 0..4; rows are final once the inflate has passed their last byte
bool watch_rows(void* p, size_t produced) {
    RowWatch* W = (RowWatch*)p;
    const int complete = (int)(produced / W->rstride);
    for (; W->checked < complete; W->checked++)
        if (W->dst[(size_t)W->checked * W->rstride] > 4) { W->code = IMP_ERROR_DECODE_FAILED; return false; }
    if (W->rows && complete > 0) {
        W->code = W->rows(W->ctx, complete);
        if (W->code) return false;
    }
    return true;
}
}  // namespace

int png_scanlines(const unsigned char* blob, size_t size, const PngHeader& H, unsigned char* dst, png_rows_fn rows, void* ctx) {
    const size_t rstride = (size_t)H.w * H.bpp + 1, raw_bytes = rstride * H.h;
    // the IDAT payloads are ONE zlib stream (10.1): gathered (a copy of the compressed bytes: 0.3 ms per 3 MB) so that the
    // inflate can run over one piece of memory
    static thread_local std::vector<unsigned char> stream;
    stream.clear();
    bool bad = false, seen_idat = false, seen_iend = false;
    for (size_t at = 8 + 25; !bad && !seen_iend;) {
        if (size - at < 12) { bad = true; break; }
        const unsigned len = be32(blob + at);
        const unsigned char* kind = blob + at + 4;
        if (len > 0x7fffffffu || size - at - 12 < len) { bad = true; break; }
        const bool critical = !(kind[0] & 0x20);
        // (libpng's default only warns about a damaged ANCILLARY chunk and skips it; such a file is left to the host decoder)
        if (crc32_ieee(kind, 4 + (size_t)len) != be32(blob + at + 8 + len)) { bad = true; break; }
        if (!std::memcmp(kind, "IDAT", 4)) {
            seen_idat = true;
            stream.insert(stream.end(), blob + at + 8, blob + at + 8 + len);
        } else if (!std::memcmp(kind, "IEND", 4)) {
            seen_iend = true;
        } else if (critical && std::memcmp(kind, "PLTE", 4) != 0) {
            bad = true;                                                      // an unknown critical chunk (5.4)
        }
        at += 12 + (size_t)len;
    }
    if (bad || !seen_idat) return IMP_ERROR_DECODE_FAILED;
    // exactly the image's bytes: a stream that ends early fails, whatever follows the last scanline is not read (libpng's rule)
    RowWatch W{&H, dst, rstride, 0, rows, ctx, IMP_OK};
    // IMPGPU_PNG_INFLATE=zlib (read per call): the audited library instead of this repository's one-shot inflate -- an operator's
    // choice for a worker that decompresses untrusted input; 1.3-1.6 x slower (profiles/r04_png_probe_zlib.json), same bytes
    const char* which = std::getenv("IMPGPU_PNG_INFLATE");
    int rc = IMP_OK;
    if (which && !std::strcmp(which, "zlib")) {
        z_stream z;
        std::memset(&z, 0, sizeof z);
        if (inflateInit(&z) != Z_OK) return IMP_ERROR_MALLOC_FAILED;
        size_t in_at = 0, produced = 0;
        bool ended = false;
        while (produced < raw_bytes && !ended && rc == IMP_OK) {
            // in pieces (the counters are 32-bit; and the rows behind a piece go to the device while the next is inflated)
            const size_t in_piece = stream.size() - in_at < (size_t(1) << 30) ? stream.size() - in_at : (size_t(1) << 30);
            const size_t out_piece = raw_bytes - produced < (size_t(1) << 20) ? raw_bytes - produced : (size_t(1) << 20);
            z.next_in = stream.data() + in_at; z.avail_in = (uInt)in_piece;
            z.next_out = dst + produced; z.avail_out = (uInt)out_piece;
            const int zr = inflate(&z, Z_NO_FLUSH);
            in_at += in_piece - z.avail_in;
            const size_t got = out_piece - z.avail_out;
            produced += got;
            if (zr == Z_STREAM_END) ended = true;
            else if (zr != Z_OK || (got == 0 && in_piece - z.avail_in == 0)) rc = IMP_ERROR_DECODE_FAILED;    // damaged, or no progress: the input ran out
            if (rc == IMP_OK && !watch_rows(&W, produced)) rc = W.code ? W.code : IMP_ERROR_DECODE_FAILED;
        }
        inflateEnd(&z);
        if (rc == IMP_OK && produced < raw_bytes) rc = IMP_ERROR_DECODE_FAILED;
    } else {
        if (inflate_exact(stream.data(), stream.size(), dst, raw_bytes, watch_rows, &W)) rc = W.code ? W.code : IMP_ERROR_DECODE_FAILED;
        else if (!watch_rows(&W, raw_bytes)) rc = W.code;            // the rows of the last block
    }
    // (the gather buffer is per thread and for life: one large file must not leave every worker thread holding its size)
    if (stream.capacity() > (size_t(4) << 20)) std::vector<unsigned char>().swap(stream);
    return rc;
}

}  // namespace imp

using namespace imp;

extern "C" {

int impgpu_png_info(const unsigned char* blob, size_t size, int* width, int* height, int* channels) {
    PngHeader H;
    const int rc = png_header(blob, size, &H);
    if (rc) return rc;
    if (!H.taken) return IMP_ERROR_UNSUPPORTED;
    if (width) *width = H.w;
    if (height) *height = H.h;
    if (channels) *channels = H.bpp;
    return IMP_OK;
}

int impgpu_png_scanlines(const unsigned char* blob, size_t size, unsigned char* out, size_t capacity, size_t* length) {
    PngHeader H;
    int rc = png_header(blob, size, &H);
    if (rc) return rc;
    if (!H.taken) return IMP_ERROR_UNSUPPORTED;
    const size_t raw_bytes = ((size_t)H.w * H.bpp + 1) * H.h;
    if (length) *length = raw_bytes;
    if (raw_bytes / 1032 > size) return IMP_ERROR_DECODE_FAILED;             // (zlib's best ratio: the file cannot hold that much)
    if (!out || capacity < raw_bytes) return IMP_ERROR_MALLOC_FAILED;
    return png_scanlines(blob, size, H, out);
}

}  // extern "C"
