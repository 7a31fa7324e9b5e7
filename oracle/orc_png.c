/*
 * orc_png.c -- CPU restatement of the PNG decode the reference performs on the host at bridge.c:545-552
 * (cvDecodeImage(&rawencoded, -1) for a SIG_PNG blob, bridge.c:376-378): OpenCV 2.4's PngDecoder drives libpng and, with
 * the "unchanged" flag, delivers an 8-bit gray file as 1 channel, RGB as B,G,R (png_set_bgr) and RGBA as B,G,R,A; a tRNS
 * chunk is not expanded.
 *
 * TEST INFRASTRUCTURE ONLY (see imp_oracle.h).
 *
 * libpng and zlib are third-party dependencies that are absent from /root/reference (config:5 links opencv_highgui, which
 * links the system's libpng).  What is restated here is the published format:
 *   PNG specification (W3C, 2nd edition)  5.2-5.4 signature / chunk layout / CRC, 11.2.2 IHDR, 9.2-9.4 the five scanline
 *                                          filters and the Paeth predictor, 10.1 the zlib stream across IDAT chunks
 *   RFC 1950 (zlib container: CMF / FLG / Adler-32), RFC 1951 (deflate: stored, fixed and dynamic Huffman blocks)
 * -- the inflate below is written from the RFC, bit by bit and slow, so that the checker shares no code with the zlib the
 * product links for its host inflate.
 * PINNED against third-party C: tests/test_oracle_png.py compares it byte for byte with Pillow's decoder (libpng / zlib
 * inside Pillow 12.2.0) on tests/golden/png/ (files whose rows use every filter type, written by that directory's
 * generator, and files libpng wrote itself), and on files made on the fly.
 *
 * Scope = the product's (everything else returns ORC_ERROR_UNSUPPORTED and the product hands such files to the host
 * decoder): bit depth 8, colour type 0 / 2 / 6, no interlace.  Damaged files return ORC_ERROR_DECODE_FAILED.
 */
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

#define ORC_ERROR_DECODE_FAILED 3

/* ---------------------------------------------------------------- CRC (PNG specification, annex D) */
static unsigned png_crc(const unsigned char* p, long n) {
    static unsigned table[256];
    static int made = 0;
    unsigned c = 0xffffffffu;
    long i;
    if (!made) {
        unsigned k, j;
        for (k = 0; k < 256; k++) {
            unsigned v = k;
            for (j = 0; j < 8; j++) v = (v & 1) ? 0xedb88320u ^ (v >> 1) : v >> 1;
            table[k] = v;
        }
        made = 1;
    }
    for (i = 0; i < n; i++) c = table[(c ^ p[i]) & 255] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

static unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3]; }

/* ---------------------------------------------------------------- inflate (RFC 1951) */
typedef struct {
    const unsigned char* in;
    long size, at;
    unsigned hold;
    int nhold;
    unsigned char* out;
    long cap, produced;      /* bytes past `cap` are counted but not stored (data beyond the image is ignored) */
    unsigned char window[32768];
    long wpos;
} inflater;

static int bits(inflater* z, int n, unsigned* v) {          /* LSB first (RFC 1951 3.1.1) */
    while (z->nhold < n) {
        if (z->at >= z->size) return -1;
        z->hold |= (unsigned)z->in[z->at++] << z->nhold;
        z->nhold += 8;
    }
    *v = n ? (z->hold & ((1u << n) - 1)) : 0;
    z->hold >>= n;
    z->nhold -= n;
    return 0;
}

static void emit(inflater* z, unsigned char b) {
    if (z->produced < z->cap) z->out[z->produced] = b;
    z->window[z->wpos & 32767] = b;
    z->wpos++;
    z->produced++;
}

typedef struct {
    short count[16];
    short symbol[288];
} huffman;

/* canonical code from lengths (RFC 1951 3.2.2); -1 for an over-subscribed set */
static int build(huffman* h, const short* len, int n) {
    short offs[16];
    int i, left = 1;
    for (i = 0; i < 16; i++) h->count[i] = 0;
    for (i = 0; i < n; i++) h->count[len[i]]++;
    for (i = 1; i < 16; i++) {
        left <<= 1;
        left -= h->count[i];
        if (left < 0) return -1;
    }
    offs[1] = 0;
    for (i = 1; i < 15; i++) offs[i + 1] = (short)(offs[i] + h->count[i]);
    for (i = 0; i < n; i++)
        if (len[i]) h->symbol[offs[len[i]]++] = (short)i;
    return left;
}

static int decode(inflater* z, const huffman* h) {          /* one bit at a time, MSB of the code first */
    int code = 0, first = 0, index = 0, len;
    for (len = 1; len < 16; len++) {
        unsigned b;
        int count;
        if (bits(z, 1, &b)) return -1;
        code |= (int)b;
        count = h->count[len];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

static int block(inflater* z, const huffman* lit, const huffman* dist) {
    static const short lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const short lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const short dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const short dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        int sym = decode(z, lit);
        if (sym < 0) return -1;
        if (sym < 256) {
            emit(z, (unsigned char)sym);
        } else if (sym == 256) {
            return 0;
        } else {
            unsigned extra;
            int len, d, ds;
            long k;
            sym -= 257;
            if (sym >= 29) return -1;
            if (bits(z, lext[sym], &extra)) return -1;
            len = lbase[sym] + (int)extra;
            ds = decode(z, dist);
            if (ds < 0 || ds >= 30) return -1;
            if (bits(z, dext[ds], &extra)) return -1;
            d = dbase[ds] + (int)extra;
            if (d > z->wpos) return -1;
            for (k = 0; k < len; k++) emit(z, z->window[(z->wpos - d) & 32767]);
        }
    }
}

/* the whole zlib stream (RFC 1950) -> 0, or -1 when it is damaged or ends early */
static int inflate_all(inflater* z) {
    unsigned cmf, flg, last, type;
    if (z->size < 2) return -1;
    cmf = z->in[0]; flg = z->in[1];
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 || (flg & 0x20)) return -1;
    z->at = 2;
    do {
        if (bits(z, 1, &last) || bits(z, 2, &type)) return -1;
        if (type == 0) {
            unsigned len, nlen;
            z->hold = 0; z->nhold = 0;
            if (z->at + 4 > z->size) return -1;
            len = z->in[z->at] | ((unsigned)z->in[z->at + 1] << 8);
            nlen = z->in[z->at + 2] | ((unsigned)z->in[z->at + 3] << 8);
            z->at += 4;
            if ((len ^ 0xffffu) != nlen || z->at + (long)len > z->size) return -1;
            while (len--) emit(z, z->in[z->at++]);
        } else if (type == 1) {
            huffman lit, dist;
            short len[288];
            int i;
            for (i = 0; i < 144; i++) len[i] = 8;
            for (; i < 256; i++) len[i] = 9;
            for (; i < 280; i++) len[i] = 7;
            for (; i < 288; i++) len[i] = 8;
            build(&lit, len, 288);
            for (i = 0; i < 30; i++) len[i] = 5;
            build(&dist, len, 30);
            if (block(z, &lit, &dist)) return -1;
        } else if (type == 2) {
            static const short order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            huffman lencode, lit, dist;
            short len[320];
            unsigned nlen, ndist, ncode, v;
            int i, r;
            if (bits(z, 5, &nlen) || bits(z, 5, &ndist) || bits(z, 4, &ncode)) return -1;
            nlen += 257; ndist += 1; ncode += 4;
            if (nlen > 286 || ndist > 30) return -1;
            for (i = 0; i < 19; i++) len[i] = 0;
            for (i = 0; i < (int)ncode; i++) {
                if (bits(z, 3, &v)) return -1;
                len[order[i]] = (short)v;
            }
            if (build(&lencode, len, 19) != 0) return -1;
            for (i = 0; i < (int)(nlen + ndist);) {
                int sym = decode(z, &lencode);
                if (sym < 0) return -1;
                if (sym < 16) {
                    len[i++] = (short)sym;
                } else {
                    int prev = 0, rep;
                    if (sym == 16) {
                        if (i == 0) return -1;
                        prev = len[i - 1];
                        if (bits(z, 2, &v)) return -1;
                        rep = 3 + (int)v;
                    } else if (sym == 17) {
                        if (bits(z, 3, &v)) return -1;
                        rep = 3 + (int)v;
                    } else {
                        if (bits(z, 7, &v)) return -1;
                        rep = 11 + (int)v;
                    }
                    if (i + rep > (int)(nlen + ndist)) return -1;
                    while (rep--) len[i++] = (short)prev;
                }
            }
            if (len[256] == 0) return -1;
            r = build(&lit, len, (int)nlen);
            if (r < 0 || (r > 0 && (int)nlen - lit.count[0] != 1)) return -1;
            r = build(&dist, len + nlen, (int)ndist);
            if (r < 0 || (r > 0 && (int)ndist - dist.count[0] != 1)) return -1;
            if (block(z, &lit, &dist)) return -1;
        } else {
            return -1;
        }
        if (z->produced >= z->cap) return 0;               /* the image is complete: what follows is not read (libpng: "too much image data") */
    } while (!last);
    return 0;
}

/* ---------------------------------------------------------------- the file */
int orc_png_decode(const unsigned char* blob, long size, orc_image** out) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    unsigned w, h;
    int depth, colour, bpp, y, x, ch;
    long at, raw_bytes, stride, zsize = 0, zcap = 0;
    unsigned char *zdata = NULL, *raw = NULL, *prev_row;
    int seen_iend = 0, bad = 0;
    inflater* z;
    orc_image* im;
    *out = NULL;
    if (!blob || size < 8 || memcmp(blob, sig, 8) != 0) return ORC_ERROR_UNSUPPORTED;
    if (size < 33 || be32(blob + 8) != 13 || memcmp(blob + 12, "IHDR", 4) != 0) return ORC_ERROR_DECODE_FAILED;
    if (png_crc(blob + 12, 17) != be32(blob + 29)) return ORC_ERROR_DECODE_FAILED;
    w = be32(blob + 16); h = be32(blob + 20);
    depth = blob[24]; colour = blob[25];
    if (!w || !h || w > 0x7fffffffu || h > 0x7fffffffu || blob[26] || blob[27] || blob[28] > 1) return ORC_ERROR_DECODE_FAILED;
    bpp = colour == 0 ? 1 : colour == 2 ? 3 : colour == 6 ? 4 : 0;
    if (depth != 8 || !bpp || blob[28] != 0 || w > 4096 || h > 16384) return ORC_ERROR_UNSUPPORTED;
    /* the chunks: IDAT payloads are one zlib stream (10.1) */
    for (at = 33; !bad && !seen_iend;) {
        unsigned len;
        const unsigned char* kind;
        int critical;
        if (size - at < 12) { bad = 1; break; }
        len = be32(blob + at);
        kind = blob + at + 4;
        if (len > 0x7fffffffu || size - at - 12 < (long)len) { bad = 1; break; }
        critical = !(kind[0] & 0x20);
        if (png_crc(kind, 4 + (long)len) != be32(blob + at + 8 + len)) { bad = 1; break; }      /* (ancillary chunks too: see the product) */
        if (!memcmp(kind, "IDAT", 4)) {
            if (zsize + (long)len > zcap) {
                zcap = (zsize + (long)len) * 2 + 64;
                zdata = (unsigned char*)realloc(zdata, (size_t)zcap);
            }
            memcpy(zdata + zsize, blob + at + 8, len);
            zsize += (long)len;
        } else if (!memcmp(kind, "IEND", 4)) {
            seen_iend = 1;
        } else if (critical && memcmp(kind, "PLTE", 4) != 0) {
            bad = 1;
        }
        at += 12 + (long)len;
    }
    if (bad || !zdata) { free(zdata); return ORC_ERROR_DECODE_FAILED; }
    stride = (long)w * bpp;
    raw_bytes = (stride + 1) * (long)h;
    raw = (unsigned char*)malloc((size_t)raw_bytes);
    z = (inflater*)calloc(1, sizeof(inflater));
    z->in = zdata; z->size = zsize; z->out = raw; z->cap = raw_bytes;
    if (inflate_all(z) || z->produced < raw_bytes) { free(z); free(zdata); free(raw); return ORC_ERROR_DECODE_FAILED; }
    free(z); free(zdata);
    /* 9.2: Recon(x) = Filt(x) + predictor over a = the byte bpp to the left, b = the byte above, c = above-left; all
     * zero outside the image.  In place, top to bottom. */
    prev_row = NULL;
    for (y = 0; y < (int)h; y++) {
        unsigned char* row = raw + (long)y * (stride + 1);
        const int type = row[0];
        unsigned char* r = row + 1;
        if (type > 4) { free(raw); return ORC_ERROR_DECODE_FAILED; }
        for (x = 0; x < stride; x++) {
            const int a = x >= bpp ? r[x - bpp] : 0, b = prev_row ? prev_row[x] : 0, c = (prev_row && x >= bpp) ? prev_row[x - bpp] : 0;
            int pred = 0;
            if (type == 1) pred = a;
            else if (type == 2) pred = b;
            else if (type == 3) pred = (a + b) / 2;
            else if (type == 4) {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            r[x] = (unsigned char)(r[x] + pred);
        }
        prev_row = r;
    }
    im = orc_image_create((int)w, (int)h, bpp);
    for (y = 0; y < (int)h; y++) {
        const unsigned char* r = raw + (long)y * (stride + 1) + 1;
        unsigned char* d = im->data + (long)y * im->step;
        for (x = 0; x < (int)w; x++)
            for (ch = 0; ch < bpp; ch++) {
                const int from = (bpp >= 3 && ch < 3) ? 2 - ch : ch;           /* R,G,B(,A) -> B,G,R(,A) */
                d[x * bpp + ch] = r[x * bpp + from];
            }
    }
    free(raw);
    *out = im;
    return ORC_OK;
}
