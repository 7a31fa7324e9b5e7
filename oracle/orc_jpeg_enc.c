/*
 * orc_jpeg_enc.c -- CPU restatement of the JPEG encode the reference performs on the host at
 * bridge.c:704 (cvEncodeImage(".jpg", image, {CV_IMWRITE_JPEG_QUALITY, q}), q from bridge.c:474-486):
 * OpenCV 2.4.9's JpegEncoder drives libjpeg with jpeg_set_defaults + jpeg_set_quality(q, TRUE) and
 * feeds it R,G,B rows made from the B,G,R(,A) frame (alpha dropped), or the gray rows of a
 * 1-channel frame.  So: baseline sequential Huffman, YCbCr 4:2:0 (luma 2x2, chroma 1x1) or one
 * gray component, the Annex K quantisation tables scaled by q and Annex K Huffman tables, ISLOW
 * forward DCT, no restart markers, a JFIF 1.01 APP0 with density 1:1.
 *
 * TEST INFRASTRUCTURE ONLY (see imp_oracle.h).
 *
 * libjpeg is a third-party dependency that is absent from /root/reference; its published algorithm
 * is restated here from the libjpeg-turbo sources (API level 6.2):
 *   parameters ............ jcparam.c  (jpeg_set_defaults, jpeg_quality_scaling, jpeg_add_quant_table, std_huff_tables)
 *   colour conversion ..... jccolor.c  (rgb_ycc_start's tables, rgb_ycc_convert; SCALEBITS 16)
 *   edge padding .......... jcprepct.c (expand_bottom_edge on the colour buffer and on every component),
 *                           jcsample.c (expand_right_edge)
 *   chroma downsampling ... jcsample.c (h2v2_downsample: 2x2 box with the alternating 1,2 bias; fullsize_downsample)
 *   forward DCT ........... jfdctint.c (jpeg_fdct_islow: CONST_BITS 13, PASS1_BITS 2, output scaled by 8)
 *   quantisation .......... jcdctmgr.c (divisor = 8 q; round half away from zero by magnitude)
 *   dummy blocks .......... jccoefct.c (compress_data: blocks of the last MCU column / row that lie beyond the
 *                           component's own blocks are zero with the DC of their predecessor)
 *   entropy coding ........ jchuff.c   (encode_one_block, emit_bits' FF00 stuffing, flush_bits' 1-fill)
 *   markers ............... jcmarker.c (SOI, APP0, DQT per table, SOF0, DHT per table in component order, SOS, EOI)
 * PINNED against third-party C: tests/test_oracle_jpeg_enc.py compares the produced FILE byte for byte with
 * Pillow's encoder (libjpeg-turbo inside Pillow 12.2.0: save(format="JPEG", quality=q, subsampling=2)) over sizes
 * that exercise every padding and dummy-block rule, qualities 0..100 and gray frames; tests/golden/jpeg_enc/ holds
 * committed inputs with Pillow's files.
 */
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

/* ITU-T T.81 Annex K.1, in zigzag order (as DQT stores them) */
static const unsigned char std_q_luma[64] = {
    16, 11, 12, 14, 12, 10, 16, 14, 13, 14, 18, 17, 16, 19, 24, 40,
    26, 24, 22, 22, 24, 49, 35, 37, 29, 40, 58, 51, 61, 60, 57, 51,
    56, 55, 64, 72, 92, 78, 64, 68, 87, 69, 55, 56, 80, 109, 81, 87,
    95, 98, 103, 104, 103, 62, 77, 113, 121, 112, 100, 120, 92, 101, 103, 99,
};
static const unsigned char std_q_chroma[64] = {
    17, 18, 18, 24, 21, 24, 47, 26, 26, 47, 99, 66, 56, 66, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
};
/* Annex K.3 - K.6 */
static const unsigned char bits_dc_luma[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const unsigned char bits_dc_chroma[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const unsigned char vals_dc[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const unsigned char bits_ac_luma[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125};
static const unsigned char vals_ac_luma[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa,
};
static const unsigned char bits_ac_chroma[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119};
static const unsigned char vals_ac_chroma[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa,
};
/* jutils.c jpeg_natural_order: position in the 8x8 block of the k-th zigzag coefficient */
static const unsigned char zz[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};

typedef struct {
    unsigned int code[256];
    unsigned char size[256];
} ehuff;

/* jchuff.c jpeg_make_c_derived_tbl: canonical codes in order of increasing length */
static void derive(ehuff* t, const unsigned char* bits, const unsigned char* vals) {
    memset(t, 0, sizeof(*t));
    unsigned int code = 0;
    int p = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++, p++) {
            t->code[vals[p]] = code++;
            t->size[vals[p]] = (unsigned char)l;
        }
        code <<= 1;
    }
}

typedef struct {
    unsigned char* out;
    size_t cap, len;
    int overflow;
    unsigned long long acc;     /* put buffer, MSB first */
    int nacc;
} sink;

static void put(sink* s, int b) {
    if (s->len < s->cap) s->out[s->len] = (unsigned char)b;
    else s->overflow = 1;
    s->len++;
}
static void put2(sink* s, int v) { put(s, v >> 8); put(s, v & 255); }

/* jchuff.c emit_bits: bytes leave the buffer MSB first, FF is followed by a stuffed 00 */
static void emit(sink* s, unsigned int code, int size) {
    s->acc = (s->acc << size) | (code & ((1u << size) - 1));
    s->nacc += size;
    while (s->nacc >= 8) {
        int c = (int)((s->acc >> (s->nacc - 8)) & 255);
        put(s, c);
        if (c == 255) put(s, 0);
        s->nacc -= 8;
    }
}

static int nbits_of(int v) {
    int n = 0;
    while (v) { n++; v >>= 1; }
    return n;
}

/* jchuff.c encode_one_block on a block in natural order */
static void encode_block(sink* s, const short* blk, int last_dc, const ehuff* dc, const ehuff* ac) {
    int t = blk[0] - last_dc, t2 = t;
    if (t < 0) { t = -t; t2--; }
    int n = nbits_of(t);
    emit(s, dc->code[n], dc->size[n]);
    if (n) emit(s, (unsigned int)t2, n);
    int r = 0;
    for (int k = 1; k < 64; k++) {
        t = blk[zz[k]];
        if (t == 0) { r++; continue; }
        while (r > 15) { emit(s, ac->code[0xF0], ac->size[0xF0]); r -= 16; }
        t2 = t;
        if (t < 0) { t = -t; t2--; }
        n = nbits_of(t);
        const int sym = (r << 4) + n;
        emit(s, ac->code[sym], ac->size[sym]);
        emit(s, (unsigned int)t2, n);
        r = 0;
    }
    if (r > 0) emit(s, ac->code[0], ac->size[0]);
}

#define FIXC(x) ((int)((x) * 65536.0 + 0.5))
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* jfdctint.c jpeg_fdct_islow, in place on 64 ints (samples already level-shifted) */
static void fdct_islow(int* d) {
    for (int pass = 0; pass < 2; pass++) {
        const int st = pass ? 8 : 1, adv = pass ? 1 : 8;
        for (int i = 0; i < 8; i++) {
            int* p = d + i * adv;
            const int t0 = p[0] + p[7 * st], t7 = p[0] - p[7 * st], t1 = p[st] + p[6 * st], t6 = p[st] - p[6 * st];
            const int t2 = p[2 * st] + p[5 * st], t5 = p[2 * st] - p[5 * st], t3 = p[3 * st] + p[4 * st], t4 = p[3 * st] - p[4 * st];
            const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            const int sh = pass ? 13 + 2 : 13 - 2;
            if (!pass) { p[0] = (t10 + t11) << 2; p[4 * st] = (t10 - t11) << 2; }
            else { p[0] = DESCALE(t10 + t11, 2); p[4 * st] = DESCALE(t10 - t11, 2); }
            int z1 = (t12 + t13) * 4433;
            p[2 * st] = DESCALE(z1 + t13 * 6270, sh);
            p[6 * st] = DESCALE(z1 + t12 * -15137, sh);
            z1 = t4 + t7;
            int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
            const int z5 = (z3 + z4) * 9633;
            const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
            z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
            z3 += z5; z4 += z5;
            p[7 * st] = DESCALE(a4 + z1 + z3, sh);
            p[5 * st] = DESCALE(a5 + z2 + z4, sh);
            p[3 * st] = DESCALE(a6 + z2 + z3, sh);
            p[st] = DESCALE(a7 + z1 + z4, sh);
        }
    }
}

typedef struct {
    int bw, bh;             /* compptr->width_in_blocks / height_in_blocks: the component's own blocks */
    int pw, ph;             /* padded plane: bw*8 x (rows up to whole iMCU rows) */
    unsigned char* plane;
    short* coef;            /* bw*bh real blocks, natural order */
    unsigned short q[64];   /* natural order */
} ecomp;

/* forward_DCT + quantize of jcdctmgr.c for the block at (bx, by) */
static void make_block(const ecomp* c, int bx, int by, short* out) {
    int ws[64];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) ws[y * 8 + x] = (int)c->plane[(size_t)(by * 8 + y) * c->pw + bx * 8 + x] - 128;
    fdct_islow(ws);
    for (int i = 0; i < 64; i++) {
        const int q = c->q[i] << 3;
        int t = ws[i];
        if (t < 0) { t = -t; t += q >> 1; t = t >= q ? t / q : 0; t = -t; }
        else { t += q >> 1; t = t >= q ? t / q : 0; }
        out[i] = (short)t;
    }
}

static void quant_table(int quality, const unsigned char* std, unsigned short* natural, unsigned char* zigzag) {
    /* jcparam.c jpeg_quality_scaling + jpeg_add_quant_table(force_baseline = TRUE) */
    int q = quality <= 0 ? 1 : quality > 100 ? 100 : quality;
    q = q < 50 ? 5000 / q : 200 - q * 2;
    for (int i = 0; i < 64; i++) {
        long t = ((long)std[i] * q + 50L) / 100L;
        if (t <= 0) t = 1;
        if (t > 255) t = 255;
        zigzag[i] = (unsigned char)t;
        natural[zz[i]] = (unsigned short)t;
    }
}

long orc_jpeg_encode_bound(int w, int h, int c) {
    const long mw = (w + 15) / 16, mh = (h + 15) / 16;
    return 1024 + mw * mh * (c == 1 ? 4 : 6) * 256L;       /* a block never needs more than 64 * 27 bits, stuffed */
}

/* -> ORC_OK and *len = file size; ORC_ERROR_INVALID_ARGS; ORC_ERROR_MALLOC_FAILED when cap is too small (*len = needed) */
int orc_jpeg_encode(const unsigned char* px, int w, int h, int c, int step, int quality, unsigned char* out, long cap, long* len) {
    if (!px || !out || !len || w <= 0 || h <= 0 || w > 65500 || h > 65500 || (c != 1 && c != 3 && c != 4) || step < w * c)
        return ORC_ERROR_INVALID_ARGS;
    const int nc = c == 1 ? 1 : 3;
    const int hs = nc == 3 ? 2 : 1;                         /* luma sampling factors (both axes) */
    const int mcuw = (w + 8 * hs - 1) / (8 * hs), mcuh = (h + 8 * hs - 1) / (8 * hs);
    ecomp comp[3];
    memset(comp, 0, sizeof(comp));
    unsigned char qz[2][64];
    quant_table(quality, std_q_luma, comp[0].q, qz[0]);
    if (nc == 3) {
        quant_table(quality, std_q_chroma, comp[1].q, qz[1]);
        memcpy(comp[2].q, comp[1].q, sizeof(comp[1].q));
    }
    int rc = ORC_OK;
    /* ---- component planes: colour conversion, edge expansion, downsampling */
    {
        const int cw = (w + 1) / 2, chh = (h + 1) / 2;       /* jdiv_round_up(image_width * h_samp, max_h_samp) for chroma */
        comp[0].bw = (w + 7) / 8; comp[0].bh = (h + 7) / 8;
        comp[0].pw = comp[0].bw * 8; comp[0].ph = mcuh * 8 * hs;
        for (int k = 1; k < nc; k++) {
            comp[k].bw = (cw + 7) / 8; comp[k].bh = (chh + 7) / 8;
            comp[k].pw = comp[k].bw * 8; comp[k].ph = mcuh * 8;
        }
        for (int k = 0; k < nc; k++) {
            comp[k].plane = (unsigned char*)malloc((size_t)comp[k].pw * comp[k].ph);
            comp[k].coef = (short*)malloc((size_t)comp[k].bw * comp[k].bh * 64 * sizeof(short));
            if (!comp[k].plane || !comp[k].coef) rc = ORC_ERROR_MALLOC_FAILED;
        }
        if (rc) goto done;
        if (nc == 1) {
            for (int y = 0; y < comp[0].ph; y++) {
                const unsigned char* s = px + (size_t)(y < h ? y : h - 1) * step;      /* expand_bottom_edge */
                unsigned char* d = comp[0].plane + (size_t)y * comp[0].pw;
                for (int x = 0; x < comp[0].pw; x++) d[x] = s[x < w ? x : w - 1];        /* expand_right_edge */
            }
        } else {
            /* full-resolution Y, Cb, Cr rows, the colour buffer padded at the right to 2 * chroma plane width and at the
             * bottom (by row replication) to whole row groups, then downsampled */
            const int fw = comp[1].pw * 2 > comp[0].pw ? comp[1].pw * 2 : comp[0].pw;
            const int fh = comp[0].ph;
            unsigned char* ycc = (unsigned char*)malloc((size_t)fw * fh * 3);
            if (!ycc) { rc = ORC_ERROR_MALLOC_FAILED; goto done; }
            for (int y = 0; y < fh; y++) {
                const unsigned char* s = px + (size_t)(y < h ? y : h - 1) * step;
                unsigned char* d = ycc + (size_t)y * fw * 3;
                for (int x = 0; x < fw; x++) {
                    const unsigned char* p = s + (size_t)(x < w ? x : w - 1) * c;
                    const int b = p[0], g = p[1], r = p[2];
                    /* jccolor.c: Y = 0.299 R + 0.587 G + 0.114 B etc. on 16-bit fixed point tables */
                    d[x * 3 + 0] = (unsigned char)((FIXC(0.29900) * r + FIXC(0.58700) * g + FIXC(0.11400) * b + 32768) >> 16);
                    d[x * 3 + 1] = (unsigned char)((-FIXC(0.16874) * r - FIXC(0.33126) * g + FIXC(0.50000) * b + (128 << 16) + 32768 - 1) >> 16);
                    d[x * 3 + 2] = (unsigned char)((FIXC(0.50000) * r - FIXC(0.41869) * g - FIXC(0.08131) * b + (128 << 16) + 32768 - 1) >> 16);
                }
            }
            for (int y = 0; y < comp[0].ph; y++)
                for (int x = 0; x < comp[0].pw; x++) comp[0].plane[(size_t)y * comp[0].pw + x] = ycc[((size_t)y * fw + x) * 3];
            for (int k = 1; k < 3; k++)
                for (int y = 0; y < comp[k].ph; y++) {
                    int bias = 1;                                                        /* 1, 2, 1, 2, ... along the row */
                    /* rows past the last real chroma row repeat THAT row (expand_bottom_edge runs on the downsampled
                     * component): for an even height this is the mean of the last two source rows, not of the last one */
                    const int ys = y < chh ? y : chh - 1;
                    for (int x = 0; x < comp[k].pw; x++) {
                        const unsigned char* r0 = ycc + ((size_t)(2 * ys) * fw + 2 * x) * 3 + k;
                        const unsigned char* r1 = r0 + (size_t)fw * 3;
                        comp[k].plane[(size_t)y * comp[k].pw + x] = (unsigned char)((r0[0] + r0[3] + r1[0] + r1[3] + bias) >> 2);
                        bias ^= 3;
                    }
                }
            free(ycc);
        }
    }
    /* ---- real blocks */
    for (int k = 0; k < nc; k++)
        for (int by = 0; by < comp[k].bh; by++)
            for (int bx = 0; bx < comp[k].bw; bx++) make_block(&comp[k], bx, by, comp[k].coef + ((size_t)by * comp[k].bw + bx) * 64);
    /* ---- the file */
    {
        sink s;
        memset(&s, 0, sizeof(s));
        s.out = out; s.cap = cap > 0 ? (size_t)cap : 0;
        put2(&s, 0xFFD8);
        put2(&s, 0xFFE0); put2(&s, 16); put(&s, 'J'); put(&s, 'F'); put(&s, 'I'); put(&s, 'F'); put(&s, 0);
        put(&s, 1); put(&s, 1); put(&s, 0); put2(&s, 1); put2(&s, 1); put(&s, 0); put(&s, 0);
        for (int t = 0; t < (nc == 3 ? 2 : 1); t++) {
            put2(&s, 0xFFDB); put2(&s, 67); put(&s, t);
            for (int i = 0; i < 64; i++) put(&s, qz[t][i]);
        }
        put2(&s, 0xFFC0); put2(&s, 8 + 3 * nc); put(&s, 8); put2(&s, h); put2(&s, w); put(&s, nc);
        for (int k = 0; k < nc; k++) { put(&s, k + 1); put(&s, k == 0 ? (hs << 4) | hs : 0x11); put(&s, k ? 1 : 0); }
        ehuff dc[2], ac[2];
        derive(&dc[0], bits_dc_luma, vals_dc); derive(&ac[0], bits_ac_luma, vals_ac_luma);
        derive(&dc[1], bits_dc_chroma, vals_dc); derive(&ac[1], bits_ac_chroma, vals_ac_chroma);
        for (int t = 0; t < (nc == 3 ? 2 : 1); t++) {
            const unsigned char* bits[2] = {t ? bits_dc_chroma : bits_dc_luma, t ? bits_ac_chroma : bits_ac_luma};
            const unsigned char* vals[2] = {vals_dc, t ? vals_ac_chroma : vals_ac_luma};
            for (int cls = 0; cls < 2; cls++) {
                int n = 0;
                for (int i = 0; i < 16; i++) n += bits[cls][i];
                put2(&s, 0xFFC4); put2(&s, 2 + 1 + 16 + n); put(&s, (cls << 4) | t);
                for (int i = 0; i < 16; i++) put(&s, bits[cls][i]);
                for (int i = 0; i < n; i++) put(&s, vals[cls][i]);
            }
        }
        put2(&s, 0xFFDA); put2(&s, 6 + 2 * nc); put(&s, nc);
        for (int k = 0; k < nc; k++) { put(&s, k + 1); put(&s, k ? 0x11 : 0x00); }
        put(&s, 0); put(&s, 63); put(&s, 0);
        /* jccoefct.c compress_data: MCUs in raster order; in each, every component's MCU_width x MCU_height blocks */
        int last_dc[3] = {0, 0, 0};
        short dummy[64];
        for (int my = 0; my < mcuh; my++)
            for (int mx = 0; mx < mcuw; mx++)
                for (int k = 0; k < nc; k++) {
                    const int n = k == 0 ? hs : 1;
                    int mcu_prev = 0;               /* DC of the block before this one in the MCU buffer */
                    for (int vy = 0; vy < n; vy++)
                        for (int vx = 0; vx < n; vx++) {
                            const int bx = mx * n + vx, by = my * n + vy;
                            const short* blk;
                            if (by < comp[k].bh && bx < comp[k].bw) blk = comp[k].coef + ((size_t)by * comp[k].bw + bx) * 64;
                            else {
                                /* a dummy block: zero AC and the DC of its predecessor in the MCU buffer -- at the right
                                 * edge the block to its left, in a dummy row at the bottom MCU_buffer[blkn - 1] for the
                                 * whole row, which is the same value block after block */
                                memset(dummy, 0, sizeof(dummy));
                                dummy[0] = (short)mcu_prev;
                                blk = dummy;
                            }
                            encode_block(&s, blk, last_dc[k], &dc[k ? 1 : 0], &ac[k ? 1 : 0]);
                            last_dc[k] = blk[0];
                            mcu_prev = blk[0];
                        }
                }
        /* flush_bits: fill the last byte with ones */
        if (s.nacc) emit(&s, 0x7F, 7), s.acc = 0, s.nacc = 0;
        put2(&s, 0xFFD9);
        *len = (long)s.len;
        if (s.overflow) rc = ORC_ERROR_MALLOC_FAILED;
    }
done:
    for (int k = 0; k < 3; k++) { free(comp[k].plane); free(comp[k].coef); }
    return rc;
}
