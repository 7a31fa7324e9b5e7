/*
 * orc_jpeg.c -- CPU restatement of the JPEG decode the reference performs on the host at
 * bridge.c:545-552 (cvDecodeImage(&rawencoded, -1)): OpenCV 2.4.9's JpegDecoder drives libjpeg
 * with its default decompression parameters (dct_method = JDCT_ISLOW, do_fancy_upsampling = TRUE,
 * out_color_space = JCS_RGB for colour / JCS_GRAYSCALE for 1-component files) and swaps R and B
 * per scanline, so a colour JPEG arrives in the operator chain as a 3-channel B,G,R frame and a
 * gray one as a 1-channel frame (promoted at bridge.c:613-618).
 *
 * TEST INFRASTRUCTURE ONLY (see imp_oracle.h).
 *
 * libjpeg is a third-party dependency that is absent from /root/reference (config:5 links
 * opencv_highgui, which links the system's libjpeg).  Its published algorithm is restated here
 * from the libjpeg-turbo sources (the libjpeg every current distribution ships; API level 6.2):
 *   marker parsing ........ jdmarker.c (get_sof, get_sos, get_dht, get_dqt, get_dri, APP0/APP14)
 *   entropy decoding ...... jdhuff.c   (decode_mcu_slow, HUFF_EXTEND, jpeg_make_d_derived_tbl's checks)
 *   dequantise + IDCT ..... jidctint.c (jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2)
 *   chroma upsampling ..... jdsample.c (h2v1_fancy_upsample, h2v2_fancy_upsample, h1v2_fancy_upsample,
 *                                       the replicating upsamplers when downsampled_width <= 2),
 *                           jdmainct.c (context rows: first / last real sample row replicated)
 *   colour conversion ..... jdcolor.c  (build_ycc_rgb_table, ycc_rgb_convert; SCALEBITS 16)
 * UNLIKE the rest of the oracle this file IS PINNED against third-party C: tests/test_oracle_jpeg.py
 * compares it bit for bit with Pillow's decoder (libjpeg-turbo 3.1.4.1 inside Pillow 12.2.0) over 4:4:4, 4:2:2,
 * 4:2:0, 4:4:0 and gray files, odd sizes, with and without restart intervals, and tests/golden/jpeg/ holds
 * the committed files with Pillow's pixels.
 *
 * Scope (everything else returns ORC_ERROR_UNSUPPORTED, and the product hands such files to the
 * host decoder unchanged): 8-bit baseline / extended-sequential Huffman (SOF0, SOF1), one
 * interleaved scan, 1 component, or 3 components with luma sampling 1x1, 2x1, 1x2 or 2x2 and
 * chroma 1x1.  Anything malformed returns ORC_ERROR_DECODE_FAILED where libjpeg would warn and
 * carry on with zeros -- the product refuses the same files, so the difference never reaches a pixel.
 */
#include <stdlib.h>
#include <string.h>
#include "orc_internal.h"

#define ORC_ERROR_DECODE_FAILED 3

typedef struct {
    int present;
    unsigned char bits[17];
    unsigned char vals[256];
    int nvals;
    int mincode[17], maxcode[18], valptr[17];
} jhuff;

typedef struct {
    int id, h, v, tq, td, ta;
    int bw, bh;             /* blocks per row / column of the padded plane (whole MCUs) */
    int dsw, dsh;           /* compptr->downsampled_width / _height: the real samples */
    short* coef;            /* bw*bh blocks of 64, natural order */
    unsigned char* plane;   /* (bw*8) x (bh*8) samples after the IDCT */
} jcomp;

typedef struct {
    const unsigned char* p;
    long size, pos;
    int width, height, ncomp, precision;
    int hmax, vmax, mcux, mcuy;
    jcomp comp[4];
    unsigned short qt[4][64];   /* natural order */
    int qt_present[4];
    jhuff dc[4], ac[4];
    int restart_interval;
    int saw_jfif, saw_adobe, adobe_transform;
    int progressive;
    long scan_begin;            /* first entropy-coded byte */
} jdec;

/* jutils.c jpeg_natural_order */
static const unsigned char zigzag_to_natural[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static int rd16(const jdec* d, long at) { return (d->p[at] << 8) | d->p[at + 1]; }

/* jdhuff.c jpeg_make_d_derived_tbl: code lengths -> canonical codes; rejects over-subscribed tables and,
 * for DC tables, symbols above 15 */
static int build_huff(jhuff* h, int is_dc) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        code += h->bits[l];
        k += h->bits[l];
        h->maxcode[l] = h->bits[l] ? code - 1 : -1;
        if (code > (1 << l)) return ORC_ERROR_DECODE_FAILED;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
    if (is_dc)
        for (int i = 0; i < h->nvals; i++)
            if (h->vals[i] > 15) return ORC_ERROR_DECODE_FAILED;
    return ORC_OK;
}

/* jdmarker.c read_markers up to and including the first SOS */
static int parse_headers(jdec* d) {
    if (d->size < 4 || d->p[0] != 0xFF || d->p[1] != 0xD8) return ORC_ERROR_UNSUPPORTED;
    long pos = 2;
    int have_sof = 0;
    for (;;) {
        if (pos + 4 > d->size) return ORC_ERROR_DECODE_FAILED;
        if (d->p[pos] != 0xFF) return ORC_ERROR_DECODE_FAILED;
        while (pos < d->size && d->p[pos] == 0xFF) pos++;      /* fill bytes */
        if (pos >= d->size) return ORC_ERROR_DECODE_FAILED;
        const int m = d->p[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;   /* parameterless */
        if (m == 0xD9) return ORC_ERROR_DECODE_FAILED;          /* EOI before any scan */
        if (pos + 2 > d->size) return ORC_ERROR_DECODE_FAILED;
        const int len = rd16(d, pos);
        if (len < 2 || pos + len > d->size) return ORC_ERROR_DECODE_FAILED;
        const unsigned char* s = d->p + pos + 2;
        const int n = len - 2;
        switch (m) {
        case 0xC0: case 0xC1: {                                  /* get_sof */
            if (have_sof) return ORC_ERROR_DECODE_FAILED;
            if (n < 6) return ORC_ERROR_DECODE_FAILED;
            d->precision = s[0];
            d->height = (s[1] << 8) | s[2];
            d->width = (s[3] << 8) | s[4];
            d->ncomp = s[5];
            if (d->precision != 8) return ORC_ERROR_UNSUPPORTED;
            if (d->height == 0 || d->width == 0) return ORC_ERROR_UNSUPPORTED;   /* DNL-defined height */
            if (d->ncomp != 1 && d->ncomp != 3) return ORC_ERROR_UNSUPPORTED;
            if (n != 6 + 3 * d->ncomp) return ORC_ERROR_DECODE_FAILED;
            for (int i = 0; i < d->ncomp; i++) {
                jcomp* c = &d->comp[i];
                c->id = s[6 + 3 * i];
                c->h = s[7 + 3 * i] >> 4;
                c->v = s[7 + 3 * i] & 15;
                c->tq = s[8 + 3 * i];
                if (c->h < 1 || c->h > 4 || c->v < 1 || c->v > 4 || c->tq > 3) return ORC_ERROR_DECODE_FAILED;
            }
            have_sof = 1;
            break;
        }
        case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            return ORC_ERROR_UNSUPPORTED;                        /* progressive, lossless, arithmetic, hierarchical */
        case 0xC4: {                                             /* get_dht */
            int at = 0;
            while (at < n) {
                if (at + 17 > n) return ORC_ERROR_DECODE_FAILED;
                const int tc = s[at] >> 4, th = s[at] & 15;
                if (tc > 1 || th > 3) return ORC_ERROR_DECODE_FAILED;
                jhuff* h = tc ? &d->ac[th] : &d->dc[th];
                memset(h, 0, sizeof *h);
                int count = 0;
                for (int l = 1; l <= 16; l++) { h->bits[l] = s[at + l]; count += s[at + l]; }
                at += 17;
                if (count > 256 || at + count > n) return ORC_ERROR_DECODE_FAILED;
                memcpy(h->vals, s + at, (size_t)count);
                h->nvals = count;
                at += count;
                const int rc = build_huff(h, tc == 0);
                if (rc) return rc;
                h->present = 1;
            }
            break;
        }
        case 0xDB: {                                             /* get_dqt */
            int at = 0;
            while (at < n) {
                const int pq = s[at] >> 4, tq = s[at] & 15;
                if (pq > 1 || tq > 3) return ORC_ERROR_DECODE_FAILED;
                at++;
                if (at + 64 * (pq + 1) > n) return ORC_ERROR_DECODE_FAILED;
                for (int i = 0; i < 64; i++) {
                    const int q = pq ? ((s[at + 2 * i] << 8) | s[at + 2 * i + 1]) : s[at + i];
                    d->qt[tq][zigzag_to_natural[i]] = (unsigned short)q;
                }
                at += 64 * (pq + 1);
                d->qt_present[tq] = 1;
            }
            break;
        }
        case 0xDD:                                               /* get_dri */
            if (n != 2) return ORC_ERROR_DECODE_FAILED;
            d->restart_interval = (s[0] << 8) | s[1];
            break;
        case 0xE0:                                               /* examine_app0 */
            if (n >= 5 && s[0] == 'J' && s[1] == 'F' && s[2] == 'I' && s[3] == 'F' && s[4] == 0) d->saw_jfif = 1;
            break;
        case 0xEE:                                               /* examine_app14 */
            if (n >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') {
                d->saw_adobe = 1;
                d->adobe_transform = s[11];
            }
            break;
        case 0xDA: {                                             /* get_sos */
            if (!have_sof) return ORC_ERROR_DECODE_FAILED;
            if (n < 1) return ORC_ERROR_DECODE_FAILED;
            const int ns = s[0];
            if (n != 4 + 2 * ns) return ORC_ERROR_DECODE_FAILED;
            if (ns != d->ncomp) return ORC_ERROR_UNSUPPORTED;    /* one scan per component */
            for (int i = 0; i < ns; i++) {
                if (s[1 + 2 * i] != d->comp[i].id) return ORC_ERROR_UNSUPPORTED;   /* scan order = frame order */
                d->comp[i].td = s[2 + 2 * i] >> 4;
                d->comp[i].ta = s[2 + 2 * i] & 15;
                if (d->comp[i].td > 3 || d->comp[i].ta > 3) return ORC_ERROR_DECODE_FAILED;
            }
            if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return ORC_ERROR_UNSUPPORTED;
            d->scan_begin = pos + len;
            return ORC_OK;
        }
        default:
            break;                                               /* COM, other APPn: skipped */
        }
        pos += len;
    }
}

/* jdmaster.c / jdinput.c initial_setup + per_scan_setup */
static int setup_geometry(jdec* d) {
    if (d->ncomp == 1) {                /* a single-component scan is never interleaved: one block per MCU */
        d->comp[0].h = d->comp[0].v = 1;
    } else {
        const int h0 = d->comp[0].h, v0 = d->comp[0].v;
        if (d->comp[1].h != 1 || d->comp[1].v != 1 || d->comp[2].h != 1 || d->comp[2].v != 1) return ORC_ERROR_UNSUPPORTED;
        if (h0 > 2 || v0 > 2) return ORC_ERROR_UNSUPPORTED;
    }
    d->hmax = d->comp[0].h;
    d->vmax = d->comp[0].v;
    d->mcux = (d->width + 8 * d->hmax - 1) / (8 * d->hmax);
    d->mcuy = (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
    for (int i = 0; i < d->ncomp; i++) {
        jcomp* c = &d->comp[i];
        c->bw = d->mcux * c->h;
        c->bh = d->mcuy * c->v;
        c->dsw = (d->width * c->h + d->hmax - 1) / d->hmax;
        c->dsh = (d->height * c->v + d->vmax - 1) / d->vmax;
        if (!d->qt_present[c->tq] || !d->dc[c->td].present || !d->ac[c->ta].present) return ORC_ERROR_DECODE_FAILED;
        c->coef = (short*)calloc((size_t)c->bw * c->bh * 64, sizeof(short));
        c->plane = (unsigned char*)malloc((size_t)c->bw * c->bh * 64);
        if (!c->coef || !c->plane) return ORC_ERROR_MALLOC_FAILED;
    }
    return ORC_OK;
}

/* ---- entropy decoding: jdhuff.c ---- */
typedef struct {
    const unsigned char* p;
    long pos, end;
    unsigned int acc;    /* bit accumulator, `nbits` valid low bits */
    int nbits;
    int hit_marker;      /* the marker byte that stopped the byte feed (0 = none) */
    int padded;          /* zero bits supplied after the feed stopped */
    int malformed;
} jbits;

/* jdhuff.c jpeg_fill_bit_buffer: FF 00 -> FF, fill FFs swallowed, any other FF xx stops the feed and zero bytes
 * are supplied from then on (`padded` counts those bits: consuming one of them means the data ended early) */
static void fill(jbits* b, int need) {
    while (b->nbits < need) {
        int byte = 0;
        if (!b->hit_marker) {
            if (b->pos >= b->end) b->hit_marker = 0xD9;         /* ran off the end: like a premature EOI */
            else {
                byte = b->p[b->pos++];
                if (byte == 0xFF) {
                    int next, fills = -1;
                    do { next = b->pos < b->end ? b->p[b->pos++] : 0xD9; fills++; } while (next == 0xFF);
                    if (next != 0) { b->hit_marker = next; byte = 0; }
                    else if (fills) b->malformed = 1;           /* FF FF 00: libjpeg reads a data FF; refused here */
                }
            }
        }
        if (b->hit_marker) b->padded += 8;
        b->acc = (b->acc << 8) | (unsigned)byte;
        b->nbits += 8;
    }
}
static int getbits(jbits* b, int n) {
    if (n == 0) return 0;
    fill(b, n);
    b->nbits -= n;
    return (int)((b->acc >> b->nbits) & ((1u << n) - 1));
}
/* jdhuff.c jpeg_huff_decode (the bit-at-a-time form) */
static int decode_symbol(jbits* b, const jhuff* h) {
    int code = getbits(b, 1), l = 1;
    while (l <= 16 && code > h->maxcode[l]) {
        code = (code << 1) | getbits(b, 1);
        l++;
    }
    if (l > 16) return -1;
    return h->vals[h->valptr[l] + code - h->mincode[l]];
}
/* jdhuff.c HUFF_EXTEND */
static int extend(int r, int s) { return r < (1 << (s - 1)) ? r + (int)((~0u) << s) + 1 : r; }

/* jdhuff.c decode_mcu_slow for every MCU of the scan, process_restart between intervals */
static int decode_scan(jdec* d) {
    jbits b;
    memset(&b, 0, sizeof b);
    b.p = d->p;
    b.pos = d->scan_begin;
    b.end = d->size;
    int last_dc[4] = {0, 0, 0, 0};
    const long total = (long)d->mcux * d->mcuy;
    int next_rst = 0;
    for (long m = 0; m < total; m++) {
        if (d->restart_interval && m && m % d->restart_interval == 0) {
            /* process_restart: drop the partial byte, expect RSTn exactly here */
            b.nbits = 0;
            b.acc = 0;
            if (!b.hit_marker) {
                /* the encoder pads with 1-bits to the byte boundary and then writes the marker */
                if (b.pos + 2 > b.end || b.p[b.pos] != 0xFF) return ORC_ERROR_DECODE_FAILED;
                long q = b.pos;
                while (q < b.end && b.p[q] == 0xFF) q++;
                if (q >= b.end) return ORC_ERROR_DECODE_FAILED;
                b.hit_marker = b.p[q];
                b.pos = q + 1;
            }
            if (b.hit_marker != 0xD0 + next_rst) return ORC_ERROR_DECODE_FAILED;
            b.hit_marker = 0;
            b.padded = 0;
            next_rst = (next_rst + 1) & 7;
            last_dc[0] = last_dc[1] = last_dc[2] = last_dc[3] = 0;
        }
        const int mx = (int)(m % d->mcux), my = (int)(m / d->mcux);
        for (int ci = 0; ci < d->ncomp; ci++) {
            jcomp* c = &d->comp[ci];
            for (int by = 0; by < c->v; by++)
                for (int bx = 0; bx < c->h; bx++) {
                    short* blk = c->coef + ((size_t)(my * c->v + by) * c->bw + (mx * c->h + bx)) * 64;
                    int s = decode_symbol(&b, &d->dc[c->td]);
                    if (s < 0) return ORC_ERROR_DECODE_FAILED;
                    if (s) {
                        const int r = getbits(&b, s);
                        s = extend(r, s);
                    }
                    last_dc[ci] += s;
                    blk[0] = (short)last_dc[ci];
                    for (int k = 1; k < 64; k++) {
                        int rs = decode_symbol(&b, &d->ac[c->ta]);
                        if (rs < 0) return ORC_ERROR_DECODE_FAILED;
                        const int r = rs >> 4;
                        s = rs & 15;
                        if (s) {
                            k += r;
                            if (k > 63) return ORC_ERROR_DECODE_FAILED;
                            const int v = getbits(&b, s);
                            blk[zigzag_to_natural[k]] = (short)extend(v, s);
                        } else {
                            if (r != 15) break;
                            if (k + 15 > 63) return ORC_ERROR_DECODE_FAILED;   /* sixteen zeros that leave the block */
                            k += 15;
                        }
                    }
                }
        }
        /* the supplied zeros are the youngest bits of the accumulator: more of them than bits left = one was decoded */
        if (b.padded > b.nbits || b.malformed) return ORC_ERROR_DECODE_FAILED;
    }
    /* Stricter than libjpeg, which skips "extraneous bytes before marker" with a warning: what follows the last MCU must be
     * the (at most 7) padding bits and then a marker or the end of the file -- the same rule process_restart applies above */
    if (!b.hit_marker && b.pos < b.end && (b.p[b.pos] != 0xFF || (b.pos + 1 < b.end && b.p[b.pos + 1] == 0x00)))
        return ORC_ERROR_DECODE_FAILED;
    return ORC_OK;
}

/* ---- jidctint.c jpeg_idct_islow ---- */
#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static void idct_1d(const int in[8], int out[8], int shift) {
    int z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
    /* even part */
    z2 = in[2];
    z3 = in[6];
    z1 = (z2 + z3) * FIX_0_541196100;
    tmp2 = z1 + z3 * (-FIX_1_847759065);
    tmp3 = z1 + z2 * FIX_0_765366865;
    z2 = in[0];
    z3 = in[4];
    tmp0 = (int)((unsigned)(z2 + z3) << CONST_BITS);
    tmp1 = (int)((unsigned)(z2 - z3) << CONST_BITS);
    tmp10 = tmp0 + tmp3;
    tmp13 = tmp0 - tmp3;
    tmp11 = tmp1 + tmp2;
    tmp12 = tmp1 - tmp2;
    /* odd part */
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    z4 = tmp1 + tmp3;
    z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336;
    tmp1 *= FIX_2_053119869;
    tmp2 *= FIX_3_072711026;
    tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223;
    z2 *= -FIX_2_562915447;
    z3 *= -FIX_1_961570560;
    z4 *= -FIX_0_390180644;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    out[0] = DESCALE(tmp10 + tmp3, shift);
    out[7] = DESCALE(tmp10 - tmp3, shift);
    out[1] = DESCALE(tmp11 + tmp2, shift);
    out[6] = DESCALE(tmp11 - tmp2, shift);
    out[2] = DESCALE(tmp12 + tmp1, shift);
    out[5] = DESCALE(tmp12 - tmp1, shift);
    out[3] = DESCALE(tmp13 + tmp0, shift);
    out[4] = DESCALE(tmp13 - tmp0, shift);
}

static void idct_block(const short* coef, const unsigned short* q, unsigned char* out, int pitch) {
    int ws[64], in[8], o[8];
    for (int x = 0; x < 8; x++) {                /* pass 1: columns, results scaled up by 2^PASS1_BITS */
        for (int k = 0; k < 8; k++) in[k] = coef[8 * k + x] * q[8 * k + x];
        idct_1d(in, o, CONST_BITS - PASS1_BITS);
        for (int k = 0; k < 8; k++) ws[8 * k + x] = o[k];
    }
    for (int y = 0; y < 8; y++) {                /* pass 2: rows; the SIMD builds saturate instead of masking */
        idct_1d(ws + 8 * y, o, CONST_BITS + PASS1_BITS + 3);
        for (int k = 0; k < 8; k++) out[(size_t)y * pitch + k] = orc_sat_u8(o[k] + 128);
    }
}

static void idct_planes(jdec* d) {
    for (int ci = 0; ci < d->ncomp; ci++) {
        jcomp* c = &d->comp[ci];
        const int pitch = c->bw * 8;
        for (int by = 0; by < c->bh; by++)
            for (int bx = 0; bx < c->bw; bx++)
                idct_block(c->coef + ((size_t)by * c->bw + bx) * 64, d->qt[c->tq], c->plane + (size_t)by * 8 * pitch + bx * 8, pitch);
    }
}

/* ---- jdsample.c: one full-resolution chroma sample.  Written per output sample instead of per row pair; the
 * neighbour clamps reproduce the special first / last columns of the row loops and jdmainct.c's replicated
 * context rows (the sample above the first row is the first row, the one below the last REAL row is that row) ---- */
static int chroma_at(const jcomp* c, int hs, int vs, int x, int y) {
    const int pitch = c->bw * 8;
    const unsigned char* p = c->plane;
    if (hs == 1 && vs == 1) return p[(size_t)y * pitch + x];                    /* fullsize_upsample */
    const int fancy_h = hs == 2 && c->dsw > 2;                                  /* jinit_upsampler's conditions */
    const int cx = hs == 2 ? x >> 1 : x, cy = vs == 2 ? y >> 1 : y;
    if (hs == 2 && vs == 1) {
        if (!fancy_h) return p[(size_t)cy * pitch + cx];                        /* h2v1_upsample */
        /* h2v1_fancy_upsample */
        int nx = (x & 1) ? cx + 1 : cx - 1;
        nx = nx < 0 ? 0 : nx > c->dsw - 1 ? c->dsw - 1 : nx;
        return (3 * p[(size_t)cy * pitch + cx] + p[(size_t)cy * pitch + nx] + ((x & 1) ? 2 : 1)) >> 2;
    }
    int ny = (y & 1) ? cy + 1 : cy - 1;
    ny = ny < 0 ? 0 : ny > c->dsh - 1 ? c->dsh - 1 : ny;
    if (hs == 1) {                                                              /* h1v2_fancy_upsample */
        return (3 * p[(size_t)cy * pitch + cx] + p[(size_t)ny * pitch + cx] + ((y & 1) ? 2 : 1)) >> 2;
    }
    if (!fancy_h) return p[(size_t)cy * pitch + cx];                            /* h2v2_upsample */
    /* h2v2_fancy_upsample */
    int nx = (x & 1) ? cx + 1 : cx - 1;
    nx = nx < 0 ? 0 : nx > c->dsw - 1 ? c->dsw - 1 : nx;
    const int thiscol = 3 * p[(size_t)cy * pitch + cx] + p[(size_t)ny * pitch + cx];
    const int nextcol = 3 * p[(size_t)cy * pitch + nx] + p[(size_t)ny * pitch + nx];
    return (3 * thiscol + nextcol + ((x & 1) ? 7 : 8)) >> 4;
}

/* ---- jdcolor.c build_ycc_rgb_table + ycc_rgb_convert ---- */
#define SCALEBITS 16
#define ONE_HALF (1 << (SCALEBITS - 1))
#define FIX16(x) ((int)((x) * (1L << SCALEBITS) + 0.5))

static int output_image(jdec* d, orc_image** out) {
    orc_image* img = orc_image_create(d->width, d->height, d->ncomp == 1 ? 1 : 3);
    if (!img) return ORC_ERROR_MALLOC_FAILED;
    const jcomp* c0 = &d->comp[0];
    const int p0 = c0->bw * 8;
    if (d->ncomp == 1) {
        for (int y = 0; y < d->height; y++) memcpy(img->data + (size_t)y * img->step, c0->plane + (size_t)y * p0, (size_t)d->width);
        *out = img;
        return ORC_OK;
    }
    /* jdapimin.c default_decompress_parms: which colour space three components mean */
    int ycc = 1;
    if (d->saw_jfif) ycc = 1;
    else if (d->saw_adobe) ycc = d->adobe_transform != 0;
    else if (d->comp[0].id == 'R' && d->comp[1].id == 'G' && d->comp[2].id == 'B') ycc = 0;
    int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
    for (int i = 0; i < 256; i++) {
        const int x = i - 128;
        cr_r[i] = (FIX16(1.40200) * x + ONE_HALF) >> SCALEBITS;
        cb_b[i] = (FIX16(1.77200) * x + ONE_HALF) >> SCALEBITS;
        cr_g[i] = (-FIX16(0.71414)) * x;
        cb_g[i] = (-FIX16(0.34414)) * x + ONE_HALF;
    }
    const int hs = d->hmax, vs = d->vmax;
    for (int y = 0; y < d->height; y++) {
        unsigned char* row = img->data + (size_t)y * img->step;
        for (int x = 0; x < d->width; x++) {
            const int Y = c0->plane[(size_t)y * p0 + x];
            const int cb = chroma_at(&d->comp[1], hs, vs, x, y);
            const int cr = chroma_at(&d->comp[2], hs, vs, x, y);
            int r, g, b;
            if (ycc) {
                r = orc_sat_u8(Y + cr_r[cr]);
                g = orc_sat_u8(Y + ((cb_g[cb] + cr_g[cr]) >> SCALEBITS));
                b = orc_sat_u8(Y + cb_b[cb]);
            } else {
                r = Y; g = cb; b = cr;
            }
            row[3 * x + 0] = (unsigned char)b;      /* OpenCV grfmt_jpeg.cpp: icvCvt_RGB2BGR_8u_C3R per scanline */
            row[3 * x + 1] = (unsigned char)g;
            row[3 * x + 2] = (unsigned char)r;
        }
    }
    *out = img;
    return ORC_OK;
}

static void jdec_free(jdec* d) {
    for (int i = 0; i < 4; i++) {
        free(d->comp[i].coef);
        free(d->comp[i].plane);
    }
}

static int jdec_open(jdec* d, const unsigned char* blob, long size) {
    memset(d, 0, sizeof *d);
    if (!blob || size <= 0) return ORC_ERROR_INVALID_ARGS;
    d->p = blob;
    d->size = size;
    d->adobe_transform = 1;
    int rc = parse_headers(d);
    if (rc) return rc;
    rc = setup_geometry(d);
    if (rc) return rc;
    return decode_scan(d);
}

int orc_jpeg_decode(const unsigned char* blob, long size, orc_image** out) {
    jdec d;
    int rc = jdec_open(&d, blob, size);
    if (!rc) {
        idct_planes(&d);
        rc = output_image(&d, out);
    }
    jdec_free(&d);
    return rc;
}

/* Geometry of a file the decoder accepts: info[0..7] = width, height, components, luma h, luma v,
 * restart interval, MCUs per row, MCU rows. */
int orc_jpeg_info(const unsigned char* blob, long size, int* info) {
    jdec d;
    memset(&d, 0, sizeof d);
    if (!blob || size <= 0) return ORC_ERROR_INVALID_ARGS;
    d.p = blob;
    d.size = size;
    int rc = parse_headers(&d);
    if (!rc) rc = setup_geometry(&d);
    if (!rc) {
        info[0] = d.width; info[1] = d.height; info[2] = d.ncomp; info[3] = d.hmax; info[4] = d.vmax;
        info[5] = d.restart_interval; info[6] = d.mcux; info[7] = d.mcuy;
    }
    jdec_free(&d);
    return rc;
}

/* The quantised coefficients of component `ci` after entropy decoding (what jpeg_read_coefficients returns):
 * blocks in raster order over the MCU-padded plane, 64 shorts each, natural order.  `cap` in shorts. */
int orc_jpeg_coefficients(const unsigned char* blob, long size, int ci, short* out, long cap, int* blocks_w, int* blocks_h) {
    jdec d;
    int rc = jdec_open(&d, blob, size);
    if (!rc) {
        if (ci < 0 || ci >= d.ncomp) rc = ORC_ERROR_INVALID_ARGS;
        else {
            const long n = (long)d.comp[ci].bw * d.comp[ci].bh * 64;
            *blocks_w = d.comp[ci].bw;
            *blocks_h = d.comp[ci].bh;
            if (n > cap) rc = ORC_ERROR_INVALID_ARGS;
            else memcpy(out, d.comp[ci].coef, (size_t)n * sizeof(short));
        }
    }
    jdec_free(&d);
    return rc;
}
