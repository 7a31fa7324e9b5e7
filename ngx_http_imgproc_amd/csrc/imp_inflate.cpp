// imp_inflate.cpp -- inflate_exact: the host inflate of the PNG front.  zlib 1.2.11's inflate delivered a PNG's scanlines at
// ~350 MB/s and was 80-84 % of a PNG request's time; this is the usual faster shape of the same
// algorithm (RFC 1951), written for one job -- the whole stream and the whole output are in memory and the output size is
// known:  a 64-bit bit buffer refilled eight bytes at a time; one table read per symbol (11-bit primary table for
// literals / lengths, 8-bit for distances, second-level tables for the longer codes) whose entry already holds the literal
// or the base value, the extra-bit count and the code length; literals two at a time; matches copied eight bytes at a
// time.  The inner loop runs while both buffers have a margin (no bounds checks per symbol) and hands the last bytes to a
// careful loop.  Host code only, no HIP call: tests/c/fuzz_host.cpp runs it under AddressSanitizer / UBSan.
#include "imp_inflate.h"
#include <immintrin.h>
#include <zlib.h>                 // crc32: the tail of a buffer and CPUs without PCLMULQDQ
#include <cstring>

namespace imp {
namespace {

constexpr int LL_BITS = 11, D_BITS = 8;
constexpr uint32_t E_LITERAL = 0x8000u, E_LENGTH = 0x4000u, E_EOB = 0x2000u, E_SUB = 0x1000u;
// entry: bits 0..3 code length to consume (for a second-level pointer: the primary bits), 4..7 extra bits (or second-level index bits),
//        8..11 unused, 12..15 kind flags, 16..31 literal / base value / second-level offset.  0 = no such code.

struct Tables {
    uint32_t ll[(1 << LL_BITS) + 286 * 16];            // primary + the second-level tables at their worst (a table of 2^(15 - 11) per long code)
    uint32_t d[(1 << D_BITS) + 30 * 128];
};

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; i++) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// what a decoded symbol means, as a table entry without its length field
inline uint32_t ll_payload(int sym) {
    if (sym < 256) return E_LITERAL | ((uint32_t)sym << 16);
    if (sym == 256) return E_EOB;
    if (sym > 285) return 0;                                         // 286, 287: never valid in a stream
    return E_LENGTH | ((uint32_t)LEN_EXTRA[sym - 257] << 4) | ((uint32_t)LEN_BASE[sym - 257] << 16);
}
inline uint32_t d_payload(int sym) {
    if (sym > 29) return 0;
    return E_LENGTH | ((uint32_t)DIST_EXTRA[sym] << 4) | ((uint32_t)DIST_BASE[sym] << 16);
}

// Canonical Huffman code (RFC 1951 3.2.2) -> lookup table indexed by the next `bits` stream bits (LSB first).  Codes longer
// than `bits` go through a second-level table per distinct `bits`-bit prefix.  Returns false for an over-subscribed code, or
// an incomplete one that zlib would refuse.
bool build_table(const uint8_t* lens, int nsym, int bits, bool is_dist, uint32_t* table, size_t table_cap, bool must_be_complete = false) {
    int count[16] = {0};
    for (int i = 0; i < nsym; i++) count[lens[i]]++;
    int left = 1, used = 0;
    for (int l = 1; l <= 15; l++) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
        used += count[l];
    }
    // zlib's rule (inftrees.c): an incomplete code is an error, except no code at all (a block without matches may send no
    // distance code; using one then fails) and a single code of length 1; the code-length code must be complete
    if (left > 0 && used != 0 && (must_be_complete || !(used == 1 && count[1] == 1))) return false;
    uint32_t next_code[16];
    uint32_t code = 0;
    count[0] = 0;
    for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
    const size_t primary = (size_t)1 << bits;
    for (size_t i = 0; i < primary; i++) table[i] = 0;
    // second-level tables: one per prefix, sized by the longest code under it
    int sub_len[1 << LL_BITS];                                       // longest code length per prefix (0 = none)
    for (size_t i = 0; i < primary; i++) sub_len[i] = 0;
    uint32_t codes[288];
    for (int s = 0; s < nsym; s++) {
        const int l = lens[s];
        if (!l) continue;
        codes[s] = reverse_bits(next_code[l]++, l);
        if (l > bits) {
            const uint32_t prefix = codes[s] & (uint32_t)(primary - 1);
            if (l > sub_len[prefix]) sub_len[prefix] = l;
        }
    }
    size_t next_free = primary;
    for (size_t p = 0; p < primary; p++) {
        if (!sub_len[p]) continue;
        const int sb = sub_len[p] - bits;
        if (next_free + ((size_t)1 << sb) > table_cap) return false;
        table[p] = E_SUB | (uint32_t)bits | ((uint32_t)sb << 4) | ((uint32_t)next_free << 16);
        for (size_t i = 0; i < ((size_t)1 << sb); i++) table[next_free + i] = 0;
        next_free += (size_t)1 << sb;
    }
    for (int s = 0; s < nsym; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t payload = is_dist ? d_payload(s) : ll_payload(s);
        if (l <= bits) {
            if (!payload) continue;                                  // (an invalid symbol keeps its slots empty: decoding it fails)
            for (uint32_t i = codes[s]; i < primary; i += 1u << l) table[i] = payload | (uint32_t)l;
        } else {
            const uint32_t prefix = codes[s] & (uint32_t)(primary - 1);
            const uint32_t e = table[prefix];
            const int sb = (int)((e >> 4) & 15);
            const size_t base = e >> 16;
            if (!payload) continue;
            for (uint32_t i = codes[s] >> bits; i < (1u << sb); i += 1u << (l - bits)) table[base + i] = payload | (uint32_t)(l - bits);
        }
    }
    return true;
}

}  // namespace

int inflate_exact(const uint8_t* in, size_t in_size, uint8_t* out, size_t out_size, inflate_progress_fn progress, void* ctx) {
    if (in_size < 2) return 1;
    const unsigned cmf = in[0], flg = in[1];
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return 1;
    if (out_size == 0) return 0;                                     // (nothing wanted: nothing read)
    const uint8_t* ip = in + 2;
    const uint8_t* const iend = in + in_size;
    uint8_t* op = out;
    uint8_t* const oend = out + out_size;
    uint64_t bb = 0;                                                 // bit buffer: bit 0 is the next stream bit
    int bn = 0;                                                      // bits in bb that came from the stream
    static thread_local Tables T;
    static thread_local bool fixed_built = false;
    static thread_local Tables F;

    // careful refill: never reads past iend; returns false when fewer than `need` bits are left in the stream
    auto need_bits = [&](int need) -> bool {
        while (bn < need) {
            if (ip >= iend) return false;
            bb |= (uint64_t)*ip++ << bn;
            bn += 8;
        }
        return true;
    };
    auto take = [&](int k) -> uint32_t {
        const uint32_t v = (uint32_t)(bb & (((uint64_t)1 << k) - 1));
        bb >>= k;
        bn -= k;
        return v;
    };

    for (;;) {
        if (!need_bits(3)) return 1;
        const uint32_t last = take(1), type = take(2);
        const uint32_t *ll = nullptr, *dt = nullptr;
        if (type == 0) {
            take(bn & 7);                                            // to the byte boundary
            bb &= bn ? (((uint64_t)1 << bn) - 1) : 0;                  // (the fast loop's refill leaves bits of bytes not yet counted above bn: the input pointer is about to jump)
            if (!need_bits(32)) return 1;
            const uint32_t len = take(16), nlen = take(16);
            if ((len ^ 0xffffu) != nlen) return 1;
            // whole bytes still in the bit buffer first, then straight from the input
            size_t want = len;
            while (want && bn >= 8) { if (op == oend) return 0; *op++ = (uint8_t)take(8); want--; }
            if (want) {
                if ((size_t)(iend - ip) < want) return 1;
                const size_t room = (size_t)(oend - op), n = want < room ? want : room;
                std::memcpy(op, ip, n);
                op += n;
                ip += want;
                if (n < want) return 0;                              // the image is complete inside this block
            }
            if (op == oend) return 0;
            if (last) break;
            if (progress && !progress(ctx, (size_t)(op - out))) return 1;
            continue;
        } else if (type == 1) {
            if (!fixed_built) {
                uint8_t lens[288];
                for (int i = 0; i < 144; i++) lens[i] = 8;
                for (int i = 144; i < 256; i++) lens[i] = 9;
                for (int i = 256; i < 280; i++) lens[i] = 7;
                for (int i = 280; i < 288; i++) lens[i] = 8;
                build_table(lens, 288, LL_BITS, false, F.ll, sizeof F.ll / 4);
                for (int i = 0; i < 32; i++) lens[i] = 5;
                build_table(lens, 32, D_BITS, true, F.d, sizeof F.d / 4);
                fixed_built = true;
            }
            ll = F.ll; dt = F.d;
        } else if (type == 2) {
            if (!need_bits(14)) return 1;
            const uint32_t nlen = take(5) + 257, ndist = take(5) + 1, ncode = take(4) + 4;
            if (nlen > 286 || ndist > 30) return 1;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (uint32_t i = 0; i < ncode; i++) {
                if (!need_bits(3)) return 1;
                cl[order[i]] = (uint8_t)take(3);
            }
            uint32_t clt[(1 << 7) + 64];
            if (!build_table(cl, 19, 7, true, clt, sizeof clt / 4, true)) return 1;    // (validates the code; its symbols are read through clsym / cllen below)
            // the code-length code is decoded with its own tiny canonical decoder: rebuild symbols from lengths
            // (the generic table stores distance payloads; read the symbol back from a parallel table instead)
            uint8_t clsym[1 << 7], cllen[1 << 7];
            {
                int count[8] = {0};
                for (int i = 0; i < 19; i++) count[cl[i]]++;
                count[0] = 0;
                uint32_t code = 0, next_code[8];
                for (int l = 1; l <= 7; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
                for (int i = 0; i < 128; i++) { clsym[i] = 0; cllen[i] = 0; }
                for (int s = 0; s < 19; s++) {
                    const int l = cl[s];
                    if (!l) continue;
                    const uint32_t c = reverse_bits(next_code[l]++, l);
                    for (uint32_t i = c; i < 128; i += 1u << l) { clsym[i] = (uint8_t)s; cllen[i] = (uint8_t)l; }
                }
            }
            uint8_t lens[286 + 30 + 138];
            uint32_t i = 0;
            while (i < nlen + ndist) {
                if (!need_bits(7) && bn == 0) return 1;
                const uint32_t idx = (uint32_t)(bb & 127);
                const int l = cllen[idx];
                if (!l || l > bn) return 1;
                take(l);
                const uint32_t sym = clsym[idx];
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return 1;
                    val = lens[i - 1];
                    if (!need_bits(2)) return 1;
                    rep = 3 + take(2);
                } else if (sym == 17) {
                    if (!need_bits(3)) return 1;
                    rep = 3 + take(3);
                } else {
                    if (!need_bits(7)) return 1;
                    rep = 11 + take(7);
                }
                if (i + rep > nlen + ndist) return 1;
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (lens[256] == 0) return 1;                            // no end-of-block code
            if (!build_table(lens, (int)nlen, LL_BITS, false, T.ll, sizeof T.ll / 4)) return 1;
            if (!build_table(lens + nlen, (int)ndist, D_BITS, true, T.d, sizeof T.d / 4)) return 1;
            ll = T.ll; dt = T.d;
        } else {
            return 1;
        }

        // ---- the symbols of a compressed block
        for (;;) {
            // fast loop: while 8 input bytes can be read blindly (two refills per iteration at most ... one here) and a whole
            // match (258) plus the 8-byte copy overshoot fits the output
            while ((size_t)(iend - ip) >= 16 && (size_t)(oend - op) >= 258 + 16) {
                // refill to >= 56 bits
                uint64_t w;
                std::memcpy(&w, ip, 8);
                bb |= w << bn;
                ip += (63 - bn) >> 3;
                bn |= 56;
                uint32_t e = ll[bb & ((1u << LL_BITS) - 1)];
                if (e & E_SUB) { const int pb = (int)(e & 15), sb = (int)((e >> 4) & 15); e = ll[(e >> 16) + ((bb >> pb) & ((1u << sb) - 1))]; bb >>= pb; bn -= pb; }
                if (e & E_LITERAL) {
                    // literals come in runs (a photograph's scanlines are mostly literals): up to three more without a
                    // refill -- a literal found in the primary table takes at most 11 bits, the first one at most 15, so
                    // 56 - 15 - 3 * 11 = 8 bits are left at worst; whatever is not a primary-table literal waits for the next round
                    bb >>= (e & 15); bn -= (int)(e & 15);
                    *op++ = (uint8_t)(e >> 16);
                    uint32_t e2 = ll[bb & ((1u << LL_BITS) - 1)];
                    if (!(e2 & E_LITERAL)) continue;
                    bb >>= (e2 & 15); bn -= (int)(e2 & 15); *op++ = (uint8_t)(e2 >> 16);
                    e2 = ll[bb & ((1u << LL_BITS) - 1)];
                    if (!(e2 & E_LITERAL)) continue;
                    bb >>= (e2 & 15); bn -= (int)(e2 & 15); *op++ = (uint8_t)(e2 >> 16);
                    e2 = ll[bb & ((1u << LL_BITS) - 1)];
                    if (!(e2 & E_LITERAL)) continue;
                    bb >>= (e2 & 15); bn -= (int)(e2 & 15); *op++ = (uint8_t)(e2 >> 16);
                    continue;
                }
                if (!(e & E_LENGTH)) {
                    if (e & E_EOB) { bb >>= (e & 15); bn -= (int)(e & 15); goto block_done; }
                    return 1;                                        // no such code
                }
                bb >>= (e & 15); bn -= (int)(e & 15);
                const int lx = (int)((e >> 4) & 15);
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << lx) - 1));
                bb >>= lx; bn -= lx;                                 // (<= 15 + 15 + 5 = 35 bits used so far: >= 21 left)
                if (bn < 32) {                                       // distance code + extra: up to 15 + 13 bits
                    std::memcpy(&w, ip, 8);
                    bb |= w << bn;
                    ip += (63 - bn) >> 3;
                    bn |= 56;
                }
                uint32_t de = dt[bb & ((1u << D_BITS) - 1)];
                if (de & E_SUB) { const int pb = (int)(de & 15), sb = (int)((de >> 4) & 15); de = dt[(de >> 16) + ((bb >> pb) & ((1u << sb) - 1))]; bb >>= pb; bn -= pb; }
                if (!(de & E_LENGTH)) return 1;
                bb >>= (de & 15); bn -= (int)(de & 15);
                const int dx = (int)((de >> 4) & 15);
                const uint32_t dist = (de >> 16) + (uint32_t)(bb & ((1u << dx) - 1));
                bb >>= dx; bn -= dx;
                if (dist > (size_t)(op - out)) return 1;             // before the start of the output
                const uint8_t* src = op - dist;
                uint8_t* dst = op;
                op += len;
                if (dist >= 8) {
                    do { std::memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < op);       // (may overshoot by 7: inside the margin)
                } else if (dist == 1) {
                    std::memset(dst, *src, len);
                } else {
                    do { *dst++ = *src++; } while (dst < op);
                }
            }
            // careful loop: one symbol, every access checked; goes back to the fast loop when the margins allow
            {
                // (a failed refill is only an error if the symbol needs the missing bits: checked against bn below)
                (void)need_bits(15);
                if (bn == 0) return 1;
                uint32_t e = ll[bb & ((1u << LL_BITS) - 1)];
                int used = 0;
                if (e & E_SUB) {
                    const int pb = (int)(e & 15), sb = (int)((e >> 4) & 15);
                    e = ll[(e >> 16) + ((bb >> pb) & ((1u << sb) - 1))];
                    used = pb;
                }
                if (!(e & (E_LITERAL | E_LENGTH | E_EOB))) return 1;
                used += (int)(e & 15);
                if (used > bn) return 1;                             // the code runs past the end of the stream
                bb >>= used; bn -= used;
                if (e & E_LITERAL) {
                    if (op == oend) return 0;
                    *op++ = (uint8_t)(e >> 16);
                    if (op == oend) return 0;
                    continue;
                }
                if (e & E_EOB) goto block_done;
                const int lx = (int)((e >> 4) & 15);
                if (!need_bits(lx)) return 1;
                const uint32_t len = (e >> 16) + take(lx);
                (void)need_bits(15);
                if (bn == 0) return 1;
                uint32_t de = dt[bb & ((1u << D_BITS) - 1)];
                used = 0;
                if (de & E_SUB) {
                    const int pb = (int)(de & 15), sb = (int)((de >> 4) & 15);
                    de = dt[(de >> 16) + ((bb >> pb) & ((1u << sb) - 1))];
                    used = pb;
                }
                if (!(de & E_LENGTH)) return 1;
                used += (int)(de & 15);
                if (used > bn) return 1;
                bb >>= used; bn -= used;
                const int dx = (int)((de >> 4) & 15);
                if (!need_bits(dx)) return 1;
                const uint32_t dist = (de >> 16) + take(dx);
                if (dist > (size_t)(op - out)) return 1;
                size_t n = len;
                if (n > (size_t)(oend - op)) n = (size_t)(oend - op);
                for (size_t k = 0; k < n; k++) { *op = *(op - dist); op++; }
                if (op == oend) return 0;
            }
        }
    block_done:
        if (op == oend) return 0;
        if (last) break;
        if (progress && !progress(ctx, (size_t)(op - out))) return 1;
    }
    return op == oend ? 0 : 1;
}


// CRC-32 (IEEE 802.3, reflected) of a buffer whose length is a multiple of 16 and at least 64, by carry-less multiplication
// (V. Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009): four 128-bit
// lanes folded 64 bytes at a time, then folded into one, then reduced 128 -> 64 -> 32 bits (Barrett).
namespace {
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t crc, const unsigned char* buf, size_t len) {
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000, 0x0163cd6124);
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = k1k2;
    buf += 64; len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = k3k4;
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = poly;
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
}  // namespace

uint32_t crc32_ieee(const uint8_t* p, size_t n) {
    uint32_t c = 0xffffffffu;
    if (n >= 64 + 16 && __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1")) {
        const size_t body = n & ~(size_t)15;
        c = crc32_clmul(c, p, body);
        p += body; n -= body;
    }
    c = ~c;
    return (uint32_t)crc32(c, p, (uInt)n);                        // tail (and small inputs) through zlib, continuing from c
}

}  // namespace imp
