// inflate_test.cpp -- imp::inflate_exact (csrc/imp_inflate.cpp) against zlib, built with AddressSanitizer / UBSan
// (tests/c/Makefile, tests/test_inflate.py): streams of every block type (stored, fixed, dynamic; Z_FIXED, Z_HUFFMAN_ONLY,
// full flushes every few bytes), contents that reach short and far distances and long runs, the exact size / fewer bytes /
// more bytes than the stream holds, truncated streams, flipped bits (zlib's verdict on the same bytes is the reference).
//   inflate_test_asan [iterations]      prints "<cases> cases, <bad> bad" and a timing of both decoders on 6 MB of scanline-like data
#include "imp_inflate.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
using namespace imp;
static std::vector<uint8_t> deflate_with(const std::vector<uint8_t>& src, int level, int strategy, int flush_every) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, 15, 8, strategy);
    std::vector<uint8_t> out(deflateBound(&zs, src.size()) * 2 + src.size() + (flush_every ? src.size() / flush_every * 16 : 0) + 65536);
    zs.next_out = out.data(); zs.avail_out = out.size();
    size_t at = 0;
    while (at < src.size()) {
        size_t n = flush_every ? std::min<size_t>(flush_every, src.size() - at) : src.size() - at;
        zs.next_in = (Bytef*)src.data() + at; zs.avail_in = n;
        deflate(&zs, at + n == src.size() ? Z_FINISH : Z_FULL_FLUSH);
        at += n;
    }
    if (src.empty()) { zs.next_in = (Bytef*)""; zs.avail_in = 0; deflate(&zs, Z_FINISH); }
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}
static unsigned rnd_state = 12345;
static unsigned rnd() { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }
int main(int argc, char** argv) {
    long cases = 0, bad = 0;
    const int iterations = argc > 1 ? atoi(argv[1]) : 3000;
    for (int iter = 0; iter < iterations; iter++) {
        const int kind = rnd() % 6;
        size_t n = kind == 5 ? rnd() % 40 : rnd() % 200000;
        std::vector<uint8_t> src(n);
        for (size_t i = 0; i < n; i++) {
            switch (kind) {
                case 0: src[i] = rnd(); break;
                case 1: src[i] = (uint8_t)(i * 7 + (rnd() % 3)); break;
                case 2: src[i] = i >= 3 && rnd() % 10 ? src[i - 1 - rnd() % 3] : rnd(); break;                 // short distances
                case 3: src[i] = i >= 40000 && rnd() % 50 ? src[i - 32768 + (rnd() % 5)] : (rnd() % 4); break; // far matches
                case 4: src[i] = (rnd() % 100) ? 0 : rnd(); break;                                            // long runs
                default: src[i] = rnd(); break;
            }
        }
        const int level = rnd() % 10, strategy = (rnd() % 4 == 0) ? Z_FIXED : (rnd() % 5 == 0 ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY);
        std::vector<uint8_t> z = deflate_with(src, level, strategy, rnd() % 3 ? 0 : 1 + rnd() % 50000);
        // exact
        {
            std::vector<uint8_t> out(n + 1, 0xEE);
            int rc = inflate_exact(z.data(), z.size(), out.data(), n);
            cases++;
            if (rc != 0 || (n && memcmp(out.data(), src.data(), n) != 0) || out[n] != 0xEE) { bad++; printf("FAIL exact iter %d kind %d n %zu level %d strat %d rc %d\n", iter, kind, n, level, strategy, rc); }
        }
        // fewer bytes wanted than the stream holds: must fill and stop
        if (n > 10) {
            size_t want = 1 + rnd() % (n - 1);
            std::vector<uint8_t> out(want + 1, 0xEE);
            int rc = inflate_exact(z.data(), z.size(), out.data(), want);
            cases++;
            if (rc != 0 || memcmp(out.data(), src.data(), want) != 0 || out[want] != 0xEE) { bad++; printf("FAIL partial iter %d want %zu of %zu rc %d\n", iter, want, n, rc); }
        }
        // more bytes wanted than the stream holds: must fail
        {
            std::vector<uint8_t> out(n + 8, 0xEE);
            int rc = inflate_exact(z.data(), z.size(), out.data(), n + 5);
            cases++;
            if (rc == 0) { bad++; printf("FAIL short-stream accepted iter %d\n", iter); }
        }
        // truncated input: never crash; rc 0 only if output complete and right
        for (int t = 0; t < 4 && z.size() > 3; t++) {
            size_t cut = rnd() % z.size();
            std::vector<uint8_t> zc(z.begin(), z.begin() + cut);
            std::vector<uint8_t> out(n + 1, 0xEE);
            int rc = inflate_exact(zc.data(), zc.size(), out.data(), n);
            cases++;
            if (out[n] != 0xEE || (rc == 0 && (n && memcmp(out.data(), src.data(), n) != 0))) { bad++; printf("FAIL trunc iter %d cut %zu rc %d\n", iter, cut, rc); }
        }
        // corrupted input: compare with zlib's verdict on the bytes it produces
        for (int t = 0; t < 4 && z.size() > 3; t++) {
            std::vector<uint8_t> zc(z);
            zc[2 + rnd() % (zc.size() - 2)] ^= 1u << (rnd() % 8);
            std::vector<uint8_t> out(n + 1, 0xEE), ref(n + 1, 0xEE);
            int rc = inflate_exact(zc.data(), zc.size(), out.data(), n);
            z_stream zs; memset(&zs, 0, sizeof zs); inflateInit(&zs);
            zs.next_in = zc.data(); zs.avail_in = zc.size(); zs.next_out = ref.data(); zs.avail_out = n;
            int zr = inflate(&zs, Z_FINISH);
            size_t got = zs.total_out; inflateEnd(&zs);
            const bool zfull = got == n && (zr == Z_STREAM_END || zr == Z_BUF_ERROR || zr == Z_OK || zr == Z_DATA_ERROR);
            cases++;
            if (out[n] != 0xEE) { bad++; printf("FAIL overrun iter %d\n", iter); }
            // when zlib delivered all n bytes before any error, ours must deliver the same bytes; when zlib stopped early with an error, ours must fail
            if (got == n && zr != Z_DATA_ERROR) { if (rc != 0 || (n && memcmp(out.data(), ref.data(), n) != 0)) { bad++; printf("FAIL corrupt-mismatch iter %d zr %d rc %d\n", iter, zr, rc); } }
            else if (got < n && rc == 0) { bad++; printf("FAIL corrupt-accepted iter %d zr %d got %zu n %zu\n", iter, zr, got, n); }
            (void)zfull;
        }
    }
    // the chunk CRC against zlib's: every length around the 16- and 64-byte steps of the folding loop, unaligned starts
    for (int it = 0; it < 6000; it++) {
        size_t n = it < 400 ? (size_t)it : rnd() % 70000;
        std::vector<uint8_t> b(n + 3);
        for (auto& v : b) v = (uint8_t)rnd();
        const uint8_t* p = b.data() + (it % 3);
        cases++;
        if (crc32_ieee(p, n) != (uint32_t)crc32(0, p, (uInt)n)) { bad++; printf("FAIL crc n %zu\n", n); }
    }
    printf("%ld cases, %ld bad\n", cases, bad);
    // speed
    {
        std::vector<uint8_t> src(6 << 20);
        for (size_t i = 0; i < src.size(); i++) src[i] = (uint8_t)((rnd() % 16 == 0) ? rnd() : (i >= 5761 && rnd() % 4 ? src[i - 5761] + (rnd() % 3) - 1 : (rnd() % 7)));
        std::vector<uint8_t> z = deflate_with(src, 6, Z_DEFAULT_STRATEGY, 0);
        std::vector<uint8_t> out(src.size());
        for (int rep = 0; rep < 3; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            int rc = inflate_exact(z.data(), z.size(), out.data(), out.size());
            auto t1 = std::chrono::steady_clock::now();
            z_stream zs; memset(&zs, 0, sizeof zs); inflateInit(&zs);
            zs.next_in = z.data(); zs.avail_in = z.size(); zs.next_out = out.data(); zs.avail_out = out.size();
            inflate(&zs, Z_FINISH); inflateEnd(&zs);
            auto t2 = std::chrono::steady_clock::now();
            printf("rc %d ratio %.2f  ours %.2f ms  zlib %.2f ms\n", rc, (double)src.size() / z.size(), std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
    }
    return bad != 0;
}
