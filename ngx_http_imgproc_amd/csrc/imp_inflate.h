// imp_inflate.h -- a one-shot inflate for the PNG front (imp_png.cpp): RFC 1950 container, RFC 1951 blocks.
#pragma once
#include <cstddef>
#include <cstdint>

namespace imp {

// Inflates the zlib stream in[0, in_size) into out[0, out_size): exactly out_size bytes are wanted (a PNG's scanlines).
// Returns 0 when out is full (whatever follows in the stream is not read: libpng's rule for data past the image) or the stream
// ended having produced exactly out_size bytes; 1 when the stream is damaged, ends early, or refers to data before its start.
// Never reads outside in[], never writes outside out[].  The Adler-32 trailer is not checked (the chunks' CRCs cover the
// compressed bytes; tests/test_inflate.py checks the decoder itself against zlib and the oracle's inflate).
// progress (optional) is called at the end of every deflate block with the bytes produced so far (they are final); returning
// false stops the inflate with 1.
typedef bool (*inflate_progress_fn)(void* ctx, size_t produced);
int inflate_exact(const uint8_t* in, size_t in_size, uint8_t* out, size_t out_size, inflate_progress_fn progress = nullptr, void* ctx = nullptr);

// CRC-32 (IEEE 802.3, the PNG chunk check: = zlib's crc32(0, p, n)), by carry-less multiplication where the CPU has it
// (0.2 ms against 2.8 for the 3 MB of a 1080p file's IDAT chunks), else through zlib.
uint32_t crc32_ieee(const uint8_t* p, size_t n);

}  // namespace imp
