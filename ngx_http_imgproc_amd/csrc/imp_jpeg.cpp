// imp_jpeg.cpp -- host side of the JPEG front (see imp_jpeg.h): marker parser, Huffman table builder, preparation of the
// entropy-coded segment for the device, the host entropy decoder (A/B path) and the lane-by-lane host model of the device's
// entropy stage.  No HIP runtime calls in this file: it also builds with g++ for the sanitizer fuzz (tests/c/).
//
// Stands where the reference calls cvDecodeImage(&rawencoded, -1) (bridge.c:545-552): OpenCV's JpegDecoder over libjpeg
// with default parameters (ISLOW IDCT, fancy upsampling, JCS_RGB swapped to B,G,R).  Only the byte-serial parts stay on
// the host -- reading the marker segments (a few hundred bytes) and removing the FF00 stuffing while the scan is copied into
// pinned memory; Huffman decoding, dequantisation, IDCT, upsampling and colour conversion run on the device.
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include "imp_jpeg_core.h"

namespace imp {

namespace {

// zig-zag position -> row-major position inside the 8x8 block (ITU T.81 figure A.6)
const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Segment {            // one marker segment's payload
    const uint8_t* p;
    size_t n;
    bool has(size_t at, size_t len) const { return at <= n && len <= n - at; }
    int u8(size_t at) const { return p[at]; }
    int u16(size_t at) const { return (p[at] << 8) | p[at + 1]; }
};

int take_sof(const Segment& s, JpegHeader* H) {
    if (!s.has(0, 6)) return IMP_ERROR_DECODE_FAILED;
    if (s.u8(0) != 8) { H->why = JPEG_WHY_12BIT; return IMP_ERROR_UNSUPPORTED; }   // 12-bit samples
    H->height = s.u16(1);
    H->width = s.u16(3);
    H->ncomp = s.u8(5);
    if (H->height == 0 || H->width == 0) { H->why = JPEG_WHY_OTHER; return IMP_ERROR_UNSUPPORTED; }   // height deferred to a DNL marker
    if (H->ncomp != 1 && H->ncomp != 3) { H->why = JPEG_WHY_COMPONENTS; return IMP_ERROR_UNSUPPORTED; }    // CMYK / YCCK
    if (s.n != size_t(6 + 3 * H->ncomp)) return IMP_ERROR_DECODE_FAILED;
    for (int i = 0; i < H->ncomp; i++) {
        JpegComp& c = H->comp[i];
        c.id = s.u8(6 + 3 * i);
        c.h = s.u8(7 + 3 * i) >> 4;
        c.v = s.u8(7 + 3 * i) & 15;
        c.tq = s.u8(8 + 3 * i);
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return IMP_ERROR_DECODE_FAILED;
    }
    return IMP_OK;
}

int take_dht(const Segment& s, JpegHeader* H) {
    size_t at = 0;
    while (at < s.n) {
        if (!s.has(at, 17)) return IMP_ERROR_DECODE_FAILED;
        const int cls = s.u8(at) >> 4, slot = s.u8(at) & 15;
        if (cls > 1 || slot > 3) return IMP_ERROR_DECODE_FAILED;
        JpegHuffSpec t;
        int total = 0;
        for (int l = 1; l <= 16; l++) {
            t.bits[l] = (uint8_t)s.u8(at + l);
            total += t.bits[l];
        }
        at += 17;
        if (total > 256 || !s.has(at, (size_t)total)) return IMP_ERROR_DECODE_FAILED;
        std::memcpy(t.vals, s.p + at, (size_t)total);
        t.nvals = total;
        t.present = true;
        at += (size_t)total;
        (cls ? H->ac : H->dc)[slot] = t;
    }
    return IMP_OK;
}

int take_dqt(const Segment& s, JpegHeader* H) {
    size_t at = 0;
    while (at < s.n) {
        const int wide = s.u8(at) >> 4, slot = s.u8(at) & 15;
        if (wide > 1 || slot > 3) return IMP_ERROR_DECODE_FAILED;
        at++;
        if (!s.has(at, size_t(64) << wide)) return IMP_ERROR_DECODE_FAILED;
        for (int k = 0; k < 64; k++) H->qt[slot][kNatural[k]] = (uint16_t)(wide ? s.u16(at + 2 * k) : s.u8(at + k));
        at += size_t(64) << wide;
        H->qt_present[slot] = true;
    }
    return IMP_OK;
}

}  // namespace

int jpeg_parse(const uint8_t* blob, size_t size, JpegHeader* H) {
    H->why = JPEG_WHY_NONE;
    if (!blob || size < 4 || blob[0] != 0xFF || blob[1] != 0xD8) { H->why = JPEG_WHY_OTHER; return IMP_ERROR_UNSUPPORTED; }
    size_t at = 2;
    bool have_frame = false, jfif = false, adobe = false;
    int adobe_transform = 1;
    for (;;) {
        if (at + 2 > size || blob[at] != 0xFF) return IMP_ERROR_DECODE_FAILED;
        while (at < size && blob[at] == 0xFF) at++;                   // any number of fill bytes may precede a marker
        if (at >= size) return IMP_ERROR_DECODE_FAILED;
        const int marker = blob[at++];
        if (marker == 0xD8 || marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;   // no payload
        if (marker == 0xD9) return IMP_ERROR_DECODE_FAILED;           // the file ends before a scan
        if (at + 2 > size) return IMP_ERROR_DECODE_FAILED;
        const size_t len = (size_t(blob[at]) << 8) | blob[at + 1];
        if (len < 2 || len > size - at) return IMP_ERROR_DECODE_FAILED;
        const Segment s{blob + at + 2, len - 2};
        int rc = IMP_OK;
        switch (marker) {
        case 0xC0: case 0xC1:                                         // baseline / extended sequential, Huffman
            if (have_frame) return IMP_ERROR_DECODE_FAILED;
            rc = take_sof(s, H);
            have_frame = true;
            break;
        case 0xC2: case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            H->why = marker == 0xC2 ? JPEG_WHY_PROGRESSIVE : JPEG_WHY_PROCESS;
            return IMP_ERROR_UNSUPPORTED;                             // progressive, lossless, hierarchical, arithmetic
        case 0xC4: rc = take_dht(s, H); break;
        case 0xDB: rc = take_dqt(s, H); break;
        case 0xDD:
            if (s.n != 2) return IMP_ERROR_DECODE_FAILED;
            H->restart_interval = s.u16(0);
            break;
        case 0xE0: jfif = jfif || (s.n >= 5 && !std::memcmp(s.p, "JFIF\0", 5)); break;
        case 0xEE:
            if (s.n >= 12 && !std::memcmp(s.p, "Adobe", 5)) { adobe = true; adobe_transform = s.u8(11); }
            break;
        case 0xDA: {
            if (!have_frame || s.n < 1) return IMP_ERROR_DECODE_FAILED;
            const int ns = s.u8(0);
            if (s.n != size_t(4 + 2 * ns)) return IMP_ERROR_DECODE_FAILED;
            if (ns != H->ncomp) { H->why = JPEG_WHY_SCANS; return IMP_ERROR_UNSUPPORTED; }         // one scan per component
            for (int i = 0; i < ns; i++) {
                if (s.u8(1 + 2 * i) != H->comp[i].id) { H->why = JPEG_WHY_SCANS; return IMP_ERROR_UNSUPPORTED; }
                H->comp[i].td = s.u8(2 + 2 * i) >> 4;
                H->comp[i].ta = s.u8(2 + 2 * i) & 15;
                if (H->comp[i].td > 3 || H->comp[i].ta > 3) return IMP_ERROR_DECODE_FAILED;
            }
            if (s.u8(1 + 2 * ns) != 0 || s.u8(2 + 2 * ns) != 63 || s.u8(3 + 2 * ns) != 0) { H->why = JPEG_WHY_SCANS; return IMP_ERROR_UNSUPPORTED; }
            H->scan_begin = at + len;
            // which colour space three components mean (libjpeg's default_decompress_parms)
            H->ycc = true;
            if (H->ncomp == 3 && !jfif) {
                if (adobe) H->ycc = adobe_transform != 0;
                else if (H->comp[0].id == 'R' && H->comp[1].id == 'G' && H->comp[2].id == 'B') H->ycc = false;
            }
            // geometry
            if (H->ncomp == 1) {
                H->comp[0].h = H->comp[0].v = 1;                      // a one-component scan is never interleaved
            } else {
                if (H->comp[1].h != 1 || H->comp[1].v != 1 || H->comp[2].h != 1 || H->comp[2].v != 1) { H->why = JPEG_WHY_SAMPLING; return IMP_ERROR_UNSUPPORTED; }
                if (H->comp[0].h > 2 || H->comp[0].v > 2) { H->why = JPEG_WHY_SAMPLING; return IMP_ERROR_UNSUPPORTED; }
            }
            H->hs = H->comp[0].h;
            H->vs = H->comp[0].v;
            H->mcux = (H->width + 8 * H->hs - 1) / (8 * H->hs);
            H->mcuy = (H->height + 8 * H->vs - 1) / (8 * H->vs);
            H->bpm = 0;
            for (int i = 0; i < H->ncomp; i++) {
                JpegComp& c = H->comp[i];
                c.bw = H->mcux * c.h;
                c.bh = H->mcuy * c.v;
                c.dsw = (H->width * c.h + H->hs - 1) / H->hs;
                c.dsh = (H->height * c.v + H->vs - 1) / H->vs;
                H->bpm += c.h * c.v;
                if (!H->qt_present[c.tq] || !H->dc[c.td].present || !H->ac[c.ta].present) return IMP_ERROR_DECODE_FAILED;
            }
            return IMP_OK;
        }
        default: break;                                               // comments, other application segments
        }
        if (rc) return rc;
        at += len;
    }
}

// Canonical code assignment (ITU T.81 annex C); refuses over-subscribed tables and DC categories above 15 like
// libjpeg's jpeg_make_d_derived_tbl.
int jpeg_build_table(const JpegHuffSpec& spec, bool is_dc, JpegHuffDev* out) {
    std::memset(out, 0, sizeof *out);
    std::memcpy(out->vals, spec.vals, sizeof out->vals);
    unsigned code = 0;
    int first = 0;                                                    // index of the first symbol of this length
    for (int l = 1; l <= 16; l++) {
        const unsigned n = spec.bits[l];
        if (code + n > (1u << l)) return IMP_ERROR_DECODE_FAILED;
        out->offs[l] = first - (int)code;
        if (l <= JPEG_LOOKBITS)
            for (unsigned k = 0; k < n; k++) {
                const unsigned lo = (code + k) << (JPEG_LOOKBITS - l);
                for (unsigned f = 0; f < (1u << (JPEG_LOOKBITS - l)); f++) out->lut[lo + f] = (uint16_t)jpeg_lut_entry((uint32_t)l, spec.vals[first + (int)k], is_dc);
            }
        code += n;
        first += (int)n;
        out->limit[l] = code << (16 - l);
        code <<= 1;
    }
    out->limit[0] = 0;
    out->limit[17] = 0x10000;
    // Second level: the codes longer than the first level's index, grouped by their first JPEG_LOOKBITS bits; a group's
    // entry in the first level says where its sub-table starts and how many more bits index it.  (A long code met by ANY lane
    // of a wave sends the whole wave through the long-code path: with the limit walk that path was most of the walk's
    // instructions.)  A table whose long codes need more room than there is keeps the limit walk.
    {
        struct Long { unsigned code; int len; uint8_t sym; };
        std::vector<Long> longs;
        unsigned cd = 0;
        int idx = 0;
        for (int l = 1; l <= 16; l++) {
            for (unsigned k = 0; k < spec.bits[l]; k++, idx++)
                if (l > JPEG_LOOKBITS) longs.push_back(Long{cd + k, l, spec.vals[idx]});
            cd = (cd + spec.bits[l]) << 1;
        }
        unsigned next = 0;
        bool fits = true;
        std::vector<std::pair<unsigned, unsigned>> pointers;         // (first-level index, entry)
        for (size_t a = 0; a < longs.size() && fits;) {
            const unsigned prefix = longs[a].code >> (longs[a].len - JPEG_LOOKBITS);
            size_t b = a;
            int maxlen = 0;
            while (b < longs.size() && (longs[b].code >> (longs[b].len - JPEG_LOOKBITS)) == prefix) { maxlen = std::max(maxlen, longs[b].len); b++; }
            const unsigned nb = (unsigned)(maxlen - JPEG_LOOKBITS);
            if (next + (1u << nb) > (unsigned)JPEG_SUB_ENTRIES || nb > 7) { fits = false; break; }
            for (size_t i = a; i < b; i++) {
                const unsigned rem = (unsigned)(longs[i].len - JPEG_LOOKBITS);
                const unsigned lo = (longs[i].code & ((1u << rem) - 1)) << (nb - rem);
                for (unsigned f = 0; f < (1u << (nb - rem)); f++) out->sub[next + lo + f] = (uint16_t)jpeg_lut_entry((uint32_t)longs[i].len, longs[i].sym, is_dc);
            }
            pointers.push_back({prefix, 0x8000u | (nb << 12) | ((next >> 1) << 5)});
            next += std::max(2u, 1u << nb);
            a = b;
        }
        if (fits) for (auto& pr : pointers) out->lut[pr.first] = (uint16_t)pr.second;
        else std::memset(out->sub, 0, sizeof out->sub);
    }
    if (is_dc)
        for (int i = 0; i < spec.nvals; i++)
            if (spec.vals[i] > 15) return IMP_ERROR_DECODE_FAILED;
    return IMP_OK;
}

size_t jpeg_scan_capacity(size_t scan_bytes, size_t nsegs) {
    return scan_bytes + (nsegs + 2) * JPEG_CHUNK_BYTES_MAX;
}

// Copies the entropy-coded bytes out of the file: FF 00 becomes FF, fill FFs go, an RSTn marker closes the interval (the
// rest of its chunk is filled with 1-bits, exactly what the encoder pads the last byte with) and the next one starts on a
// chunk boundary.  Stops at EOI, at any other marker, or at the end of the file.
size_t jpeg_chunk_bytes_for(size_t file_bytes, size_t launch_bytes, bool busy) {
    if (const char* s = ab_env("IMPGPU_JPEG_CHUNK_WORDS")) {
        const int w = std::atoi(s);
        if (w == 8 || w == 16 || w == 32 || w == 64 || w == 128) return (size_t)w * 4;
    }
    // measured, one request at a time (profiles/r03_request_latency.txt; 256 / 512 / 1024 bits): 640x480 0.77 / 0.76 / 0.89 ms,
    // 720p 0.83 / 0.84 / 0.94, 1080p 0.90 / 0.84 / 0.95, 4K 1.45 / 1.24 / 1.24; a queue's worth of files (28 MB) is a little
    // faster on 1024
    if (launch_bytes > (size_t(4) << 20)) return 256;                     // (a walk's overlap weighs half as much on 2048 bits as on 1024)
    // Round 5: `busy` = other decode groups of this process are in flight (a broker's lanes under load).  Short chunks buy a lone
    // file short chains with vector work the idle device has to spare -- every walk decodes its five-block overlap (about 750
    // bits) on top of its chunk, 3.9 x the bits at 256-bit chunks against 1.7 x at 1024 -- but under load that work is what
    // the lanes slow each other with.  One broker, four lanes, mixed pool, requests/s at 16 / 32 workers with every launch forced to
    // 32 / 64 / 128 / 256-byte chunks: 12.7 / 15.4 k, 15.3 / 20.3 k, 15.5 / 22.7 k, 13.2 / 21.0 k (the rule below without `busy`:
    // 14.2 / 20.0 k; a lone worker: 2.32 / 2.33 / 2.03 / 1.63 k).
    if (busy && launch_bytes >= (size_t(2560) << 10)) return 128;
    if (busy && launch_bytes >= (size_t(512) << 10)) return 64;
    // round 4, the stage without rounds, one file at a time (profiles/r04_jpeg_chunk_ab.txt, ms at 256 / 512 / 1024 / 2048 bits):
    // 640x480 0.25 / 0.30 / 0.41 / 0.62, 1080p 4:2:0 0.32 / 0.36 / 0.46 / 0.67, 4:4:4 0.28 / 0.33 / 0.45 / 0.67, 4K 0.58 / 0.52 / 0.61 / 0.81
    // -- every kernel's time is the length of a chunk's walk until the chunks no longer fit the device at once
    // (round 5, after the write walk and the launches got shorter: 128 / 256 / 512 bits 640x480 0.213 / 0.204 / 0.245, 1080p 0.341 / 0.290 / 0.321,
    // 4K 0.786 / 0.553 / 0.492 -- 128-bit chunks lose everywhere: the walks' overlap is five blocks whatever the chunk)
    return file_bytes <= (size_t(1200) << 10) ? 32 : 64;
}

unsigned jpeg_overlap_bits_for(unsigned chunk_bits, size_t scan_bytes, size_t total_blocks) {
    if (const char* s = ab_env("IMPGPU_JPEG_OVERLAP")) {
        const int v = std::atoi(s);
        if (v >= 0 && v <= 1 << 20) return (unsigned)v;
    }
    // tools/jpeg_sync_probe.py: with five blocks' worth of bits in front of the chunk 97.5-99.5 % of a photograph's chunks find
    // their true entry state among their walks' (quality 50-90, 1-2.5 bits per pixel); with two or three, four in five
    (void)chunk_bits;
    const size_t bits_per_block = total_blocks ? scan_bytes * 8 / total_blocks : 64;
    const size_t want = (5 * bits_per_block + 63) / 64 * 64;
    return (unsigned)(want < 256 ? 256 : want > 1024 ? 1024 : want);
}

int jpeg_prepare_scan(const uint8_t* blob, size_t size, const JpegHeader& H, uint8_t* out, size_t cap, JpegScan* scan) {
    const size_t CBY = scan->chunk_bytes;                                // 256, 128, 64 or 32
    if (CBY != 512 && CBY != 256 && CBY != 128 && CBY != 64 && CBY != 32) return IMP_ERROR_INVALID_ARGS;
    scan->seg_first_chunk.clear();
    scan->seg_bits.clear();
    const size_t total_mcus = (size_t)H.mcux * H.mcuy;
    const size_t want_segs = H.restart_interval ? (total_mcus + H.restart_interval - 1) / H.restart_interval : 1;
    size_t at = H.scan_begin, o = 0, seg_begin = 0;
    int expect_rst = 0;
    bool done = false, closed = false;
    auto close_segment = [&]() -> bool {
        if (o == seg_begin) return false;                             // an interval with no data
        scan->seg_first_chunk.push_back((uint32_t)(seg_begin / CBY));
        scan->seg_bits.push_back((uint32_t)((o - seg_begin) * 8));
        const size_t padded = (o + CBY - 1) / CBY * CBY;
        std::memset(out + o, 0xFF, padded - o);
        o = seg_begin = padded;
        return true;
    };
    while (!done) {
        const uint8_t* ff = at < size ? (const uint8_t*)std::memchr(blob + at, 0xFF, size - at) : nullptr;
        const size_t run = (ff ? (size_t)(ff - blob) : size) - at;
        if (o + run + 2 * JPEG_CHUNK_BYTES_MAX > cap) return IMP_ERROR_DECODE_FAILED;
        std::memcpy(out + o, blob + at, run);
        o += run;
        at += run;
        if (!ff) break;                                               // no EOI: the MCU count decides whether it was complete
        size_t m = at + 1;
        while (m < size && blob[m] == 0xFF) m++;                      // fill bytes
        if (m >= size) break;
        const int code = blob[m];
        if (code == 0x00 && m == at + 1) {
            out[o++] = 0xFF;                                          // a stuffed data byte
            at = m + 1;
        } else if (code >= 0xD0 && code <= 0xD7) {
            if (!H.restart_interval || code != 0xD0 + expect_rst) return IMP_ERROR_DECODE_FAILED;
            if (!close_segment()) return IMP_ERROR_DECODE_FAILED;
            if (scan->seg_first_chunk.size() >= want_segs) { closed = true; break; }   // every MCU is accounted for: like libjpeg, ignore what follows
            expect_rst = (expect_rst + 1) & 7;
            at = m + 1;
        } else if (code == 0x00) {
            return IMP_ERROR_DECODE_FAILED;                           // FF FF 00: not a sequence an encoder writes
        } else {
            done = true;                                              // EOI or whatever follows the scan
        }
    }
    if (!closed && !close_segment()) return IMP_ERROR_DECODE_FAILED;
    if (scan->seg_first_chunk.size() != want_segs) return IMP_ERROR_DECODE_FAILED;
    scan->nchunks = o / CBY;
    if (o + JPEG_CHUNK_BYTES > cap) return IMP_ERROR_DECODE_FAILED;
    std::memset(out + o, 0xFF, JPEG_CHUNK_BYTES);                     // guard chunk: a lane may look 64 bits past its own
    if ((uint64_t)scan->nchunks * CBY * 8 >= (1ull << 32)) return IMP_ERROR_UNSUPPORTED;   // bit positions are 32-bit
    return IMP_OK;
}

// ---- host entropy decoder: the A/B path (IMPGPU_JPEG_HUFF=host).  Same table format as the device, same strictness.
namespace {
struct BitFeed {
    const uint8_t* p;
    size_t at, end;
    uint64_t acc = 0;
    int have = 0;        // valid bits in acc (left-aligned reads take the top)
    int fed_past = 0;    // bits supplied after the data ended (1-bits, like the encoder's padding)
    void fill() {
        while (have <= 56) {
            uint64_t byte = 0xFF;
            if (at < end) byte = p[at++]; else fed_past += 8;
            acc |= byte << (56 - have);
            have += 8;
        }
    }
    unsigned peek16() { if (have < 16) fill(); return (unsigned)(acc >> 48); }
    void drop(int n) { acc <<= n; have -= n; }
    int take(int n) {
        if (n == 0) return 0;
        if (have < n) fill();
        const int v = (int)(acc >> (64 - n));
        drop(n);
        return v;
    }
    bool overran() const { return fed_past > have; }
};

inline int huff_symbol(BitFeed& b, const JpegHuffDev& t) {
    const unsigned peek = b.peek16();
    const unsigned e = t.lut[peek >> (16 - JPEG_LOOKBITS)];
    if (e & 31) {                                                     // (length, value bits, run, end-of-block) -> the symbol byte
        b.drop((int)(e & 31));
        return ((e >> 13) & 1) ? 0 : (int)((((e >> 9) & 15) << 4) | ((e >> 5) & 15));
    }
    for (int l = JPEG_LOOKBITS + 1; l <= 16; l++)
        if (peek < t.limit[l]) { b.drop(l); return t.vals[t.offs[l] + (int)(peek >> (16 - l))]; }
    return -1;
}
inline int extend_sign(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }
}  // namespace

int jpeg_host_entropy(const uint8_t* blob, size_t size, const JpegHeader& H, int16_t* coef, const JpegFrame& F) {
    // prepared (unstuffed, interval-aligned) copy first: the bit reader then never meets a marker
    std::vector<uint8_t> buf(jpeg_scan_capacity(size - H.scan_begin, H.restart_interval ? (size_t)H.mcux * H.mcuy / H.restart_interval + 1 : 1));
    JpegScan scan;
    if (int rc = jpeg_prepare_scan(blob, size, H, buf.data(), buf.size(), &scan)) return rc;
    JpegHuffDev dct[4], act[4];
    bool built_dc[4] = {}, built_ac[4] = {};
    for (int i = 0; i < H.ncomp; i++) {
        const JpegComp& c = H.comp[i];
        if (!built_dc[c.td]) { if (int rc = jpeg_build_table(H.dc[c.td], true, &dct[c.td])) return rc; built_dc[c.td] = true; }
        if (!built_ac[c.ta]) { if (int rc = jpeg_build_table(H.ac[c.ta], false, &act[c.ta])) return rc; built_ac[c.ta] = true; }
    }
    const size_t total_mcus = (size_t)H.mcux * H.mcuy;
    const size_t per_seg = H.restart_interval ? (size_t)H.restart_interval : total_mcus;
    for (size_t sg = 0; sg < scan.seg_first_chunk.size(); sg++) {
        BitFeed b{buf.data(), (size_t)scan.seg_first_chunk[sg] * scan.chunk_bytes, 0};
        b.end = b.at + scan.seg_bits[sg] / 8;
        int pred[3] = {0, 0, 0};
        const size_t m0 = sg * per_seg, m1 = m0 + per_seg < total_mcus ? m0 + per_seg : total_mcus;
        for (size_t m = m0; m < m1; m++) {
            const int mx = (int)(m % H.mcux), my = (int)(m / H.mcux);
            for (int ci = 0; ci < H.ncomp; ci++) {
                const JpegComp& c = H.comp[ci];
                for (int by = 0; by < c.v; by++)
                    for (int bx = 0; bx < c.h; bx++) {
                        int16_t* blk = coef + F.coef_off[ci] + ((size_t)(my * c.v + by) * c.bw + (size_t)(mx * c.h + bx)) * 64;
                        int s = huff_symbol(b, dct[c.td]);
                        if (s < 0) return IMP_ERROR_DECODE_FAILED;
                        if (s) pred[ci] += extend_sign(b.take(s), s);
                        blk[0] = (int16_t)pred[ci];
                        int z = 1;
                        while (z < 64) {
                            const int rs = huff_symbol(b, act[c.ta]);
                            if (rs < 0) return IMP_ERROR_DECODE_FAILED;
                            const int run = rs >> 4, bits = rs & 15;
                            if (bits == 0) {
                                if (run != 15) break;                         // end of block
                                if (z + 16 > 64) return IMP_ERROR_DECODE_FAILED;
                                z += 16;
                            } else {
                                z += run;
                                if (z > 63) return IMP_ERROR_DECODE_FAILED;
                                blk[kNatural[z]] = (int16_t)extend_sign(b.take(bits), bits);
                                z++;
                            }
                        }
                        if (b.overran()) return IMP_ERROR_DECODE_FAILED;
                    }
            }
        }
        // what is left of the interval must be the encoder's padding: fewer than 8 bits
        const long long consumed = (long long)(b.at - (size_t)scan.seg_first_chunk[sg] * scan.chunk_bytes) * 8 + b.fed_past - b.have;
        if ((long long)scan.seg_bits[sg] - consumed >= 8) return IMP_ERROR_DECODE_FAILED;
    }
    return IMP_OK;
}

// What the kernels need to know about the file; the two DC and two AC table slots the components share.
int jpeg_frame_setup(const JpegHeader& H, JpegFrame* F, int dc_ids[2], int ac_ids[2]) {
    std::memset(F, 0, sizeof *F);
    F->width = H.width; F->height = H.height; F->ncomp = H.ncomp; F->hs = H.hs; F->vs = H.vs;
    F->mcux = H.mcux; F->mcuy = H.mcuy; F->bpm = H.bpm; F->ycc = H.ycc ? 1 : 0;
    uint64_t off = 0;
    for (int i = 0; i < H.ncomp; i++) {
        F->bw[i] = H.comp[i].bw; F->bh[i] = H.comp[i].bh; F->dsw[i] = H.comp[i].dsw; F->dsh[i] = H.comp[i].dsh;
        F->coef_off[i] = (unsigned)off;
        off += (uint64_t)H.comp[i].bw * (uint64_t)H.comp[i].bh * 64u;
    }
    if (off >= (1ull << 31)) return IMP_ERROR_UNSUPPORTED;           // slot numbers are 32-bit
    F->total_slots = (unsigned)off;
    const uint64_t total_mcus = (uint64_t)H.mcux * (uint64_t)H.mcuy;
    const uint64_t per_seg = (H.restart_interval ? (uint64_t)H.restart_interval : total_mcus) * (uint64_t)H.bpm * 64u;
    F->slots_per_seg = (int)std::min<uint64_t>(per_seg, off);        // (an interval longer than the scan is the whole scan)
    dc_ids[0] = dc_ids[1] = ac_ids[0] = ac_ids[1] = -1;
    for (int i = 0; i < H.ncomp; i++) {
        int k;
        for (k = 0; k < 2; k++) { if (dc_ids[k] < 0) dc_ids[k] = H.comp[i].td; if (dc_ids[k] == H.comp[i].td) break; }
        if (k == 2) return IMP_ERROR_UNSUPPORTED;                     // three different tables: rare, left to the host decoder
        F->dctab[i] = k;
        for (k = 0; k < 2; k++) { if (ac_ids[k] < 0) ac_ids[k] = H.comp[i].ta; if (ac_ids[k] == H.comp[i].ta) break; }
        if (k == 2) return IMP_ERROR_UNSUPPORTED;
        F->actab[i] = k;
    }
    return IMP_OK;
}

// A request stream's files come from few encoders, and most encoders write the standard tables (T.81 annex K): the last few
// tables a thread built are kept (compared by their whole DHT content, not by a hash) and copied -- 0.2 us against the 2-3 us
// a table's lookup arrays take to build, 20 us of a broker lane's launch of three files.
int jpeg_build_tables(const JpegHeader& H, const int dc_ids[2], const int ac_ids[2], JpegHuffDev tabs[4]) {
    struct Kept { JpegHuffSpec spec; bool is_dc = false; bool used = false; JpegHuffDev dev; };
    static thread_local Kept kept[8];
    static thread_local unsigned next = 0;
    auto build = [&](const JpegHuffSpec& spec, bool is_dc, JpegHuffDev* out) -> int {
        for (const Kept& k : kept)
            if (k.used && k.is_dc == is_dc && k.spec.nvals == spec.nvals && !std::memcmp(k.spec.bits, spec.bits, sizeof spec.bits) &&
                !std::memcmp(k.spec.vals, spec.vals, (size_t)spec.nvals)) { *out = k.dev; return IMP_OK; }
        if (int rc = jpeg_build_table(spec, is_dc, out)) return rc;
        Kept& k = kept[next++ & 7u];
        k.spec = spec; k.is_dc = is_dc; k.used = true; k.dev = *out;
        return IMP_OK;
    };
    for (int k = 0; k < 2; k++) {
        if (dc_ids[k] >= 0) if (int rc = build(H.dc[dc_ids[k]], true, &tabs[k])) return rc;
        if (ac_ids[k] >= 0) if (int rc = build(H.ac[ac_ids[k]], false, &tabs[2 + k])) return rc;
    }
    return IMP_OK;
}

// The per-chunk interval numbers followed by the two per-interval arrays: the kernel's small side input.
void jpeg_scan_meta(const JpegScan& scan, std::vector<uint32_t>* meta) {
    const size_t ns = scan.seg_first_chunk.size();
    meta->assign(scan.nchunks + 2 * ns, 0);
    for (size_t sg = 0; sg < ns; sg++) {
        const size_t c0 = scan.seg_first_chunk[sg], c1 = sg + 1 < ns ? scan.seg_first_chunk[sg + 1] : scan.nchunks;
        for (size_t c = c0; c < c1; c++) (*meta)[c] = (uint32_t)sg;
        (*meta)[scan.nchunks + sg] = scan.seg_first_chunk[sg];
        (*meta)[scan.nchunks + ns + sg] = scan.seg_bits[sg];
    }
}

// ---- the device's entropy stage, lane by lane on the host (diagnostics and CPU tests only; see imp_jpeg_core.h).
// The same steps as the kernels (imp_jpeg.hip), with the very same walks:
//   k_jpeg_walks / k_jpeg_mend / k_jpeg_select   one jpeg_span_walk per (chunk, block of the MCU) from `overlap` bits in front of the chunk; a chunk's true
//                 entry state is SELECTED among its walks' `in` states by comparing them with its predecessor's true exit --
//                 on the device a scan over per-chunk maps, here the plain recurrence -- and only a chunk none of whose
//                 walks had fallen into step by its first bit ("miss") is walked again from the true state
//   k_jpeg_write  one full walk per chunk from its true entry: coefficients to their places, DC terms summed up inside the
//                 chunk only
//   k_jpeg_dcfix  the DC predictors at the chunk's entry (a prefix sum over the interval's chunks) added to the DC terms of
//                 the blocks that begin in the chunk
// *rounds receives the number of misses.
static thread_local int g_sync_stats[8];

int jpeg_emulate_entropy(const uint8_t* blob, size_t size, const JpegHeader& H, const JpegFrame& F0, const int dc_ids[2],
                         const int ac_ids[2], int16_t* coef, unsigned* status, int* rounds) {
    JpegFrame F = F0;
    const size_t total_mcus = (size_t)H.mcux * H.mcuy;
    const size_t nsegs = H.restart_interval ? (total_mcus + H.restart_interval - 1) / H.restart_interval : 1;
    std::vector<uint8_t> buf(jpeg_scan_capacity(size - H.scan_begin, nsegs));
    JpegScan scan;
    scan.chunk_bytes = jpeg_chunk_bytes_for(size - H.scan_begin, size - H.scan_begin);      // as a lone request would be cut
    if (int rc = jpeg_prepare_scan(blob, size, H, buf.data(), buf.size(), &scan)) return rc;
    std::vector<JpegHuffDev> tabs(4);
    if (int rc = jpeg_build_tables(H, dc_ids, ac_ids, tabs.data())) return rc;
    std::vector<JpegHuffTabs> Lv(1);
    JpegHuffTabs& L = Lv[0];
    for (int k = 0; k < 4; k++) {
        for (size_t i = 0; i < sizeof tabs[k].lut / sizeof tabs[k].lut[0]; i++) L.lut[k][i] = jpeg_lut_expand(tabs[k].lut[i]);
        for (size_t i = 0; i < (size_t)JPEG_SUB_ENTRIES; i++) L.sub[k][i] = jpeg_lut_expand(tabs[k].sub[i]);
        std::memcpy(L.limit[k], tabs[k].limit, sizeof L.limit[k]);
        std::memcpy(L.offs[k], tabs[k].offs, sizeof L.offs[k]);
        std::memcpy(L.vals[k], tabs[k].vals, sizeof L.vals[k]);
    }
    JpegBlockTabs K;
    std::memcpy(K.natural, kNatural, 64);
    for (int k = 0; k < F.bpm; k++) jpeg_block_steps(F, k, &K.blk_base[k], &K.blk_dx[k], &K.blk_dy[k]);
    F.nchunks = (unsigned)scan.nchunks;
    F.nsegs = (unsigned)scan.seg_first_chunk.size();
    F.chunk_bits = (unsigned)scan.chunk_bytes * 8;
    F.overlap_bits = jpeg_overlap_bits_for(F.chunk_bits, size - H.scan_begin, (size_t)F.total_slots / 64);
    const uint8_t* bytes = buf.data();
    auto word1 = [bytes](uint32_t i) -> uint32_t {
        const uint8_t* q = bytes + (size_t)i * 4;
        return ((uint32_t)q[3] << 24) | ((uint32_t)q[2] << 16) | ((uint32_t)q[1] << 8) | q[0];      // as a little-endian load
    };
    const uint32_t CB = F.chunk_bits, B = (uint32_t)F.bpm;
    const size_t n = scan.nchunks;
    std::vector<uint32_t> meta;
    jpeg_scan_meta(scan, &meta);
    // ---- k_jpeg_walks, k_jpeg_mend, k_jpeg_select
    std::vector<uint64_t> entry(n), middle(n, JPEG_STATE_NONE);      // middle: the true walk's state at the chunk's middle (k_jpeg_write's second lane)
    std::vector<uint32_t> slots(n), slots_mid(n, 0), seg_end(n), limit(n);
    std::vector<char> origin(n);
    struct Cand {
        uint64_t in[6], out[6], rep_out[6], mid[6], rep_mid[6];
        uint32_t n[6], rep_n[6], nmid[6], rep_nmid[6];
        uint32_t map, via;                                           // via: bit k = the map's entry k comes from a repair walk
        bool exact;
    };
    std::vector<Cand> cand(n);
    // A. the walks
    for (size_t g = 0; g < n; g++) {
        const uint32_t sg = meta[g], first = scan.seg_first_chunk[sg];
        origin[g] = first == g;
        seg_end[g] = first * CB + scan.seg_bits[sg];
        limit[g] = std::min<uint32_t>((uint32_t)(g + 1) * CB, seg_end[g]);
        const uint32_t start = (uint32_t)g * CB, seg_start = first * CB;
        const uint32_t p0 = start - seg_start > F.overlap_bits ? start - F.overlap_bits : seg_start;
        Cand& c = cand[g];
        c.exact = p0 == seg_start;                                   // the walk starts where the interval does: no guess
        for (uint32_t k = 0; k < 6; k++) { c.in[k] = c.out[k] = c.rep_out[k] = c.mid[k] = c.rep_mid[k] = JPEG_STATE_NONE; c.n[k] = c.rep_n[k] = c.nmid[k] = c.rep_nmid[k] = 0; }
        for (uint32_t k = 0; k < (c.exact ? 1u : B); k++) {
            const JpegSpan sp = jpeg_span_walk(L, word1, jpeg_pack_state(p0, k, 0, 0), start, std::min(start + CB / 2, limit[g]), limit[g], seg_end[g], F);
            c.in[k] = sp.in; c.out[k] = sp.out; c.n[k] = sp.n; c.mid[k] = sp.mid; c.nmid[k] = sp.nmid;
        }
    }
    // B. the maps: a predecessor's exit that is one of the chunk's `in` states selects that walk; one that is not is decoded
    // on from (a "repair" walk) and, almost always, joins one of the chunk's walks before the chunk ends
    int repairs = 0, chases = 0, misses = 0;
    for (size_t g = 0; g < n; g++) {
        Cand& c = cand[g];
        c.map = jpeg_map_const(0);
        c.via = 0;
        if (c.exact) continue;
        const Cand& pr = cand[g - 1];
        c.map = 0;
        for (uint32_t k = 0; k < 6; k++) {
            const uint64_t E = pr.out[k];
            uint32_t v = JPEG_MAP_FAIL, via = 0;
            uint32_t twin = 6;
            for (uint32_t k2 = 0; k2 < k && twin == 6; k2++) if (pr.out[k2] == E) twin = k2;
            if (E == JPEG_STATE_NONE) {
            } else if (twin < 6) {
                v = jpeg_map_at(c.map, twin); via = (c.via >> twin) & 1;
                c.rep_out[k] = c.rep_out[twin]; c.rep_n[k] = c.rep_n[twin]; c.rep_mid[k] = c.rep_mid[twin]; c.rep_nmid[k] = c.rep_nmid[twin];
            } else {
                for (uint32_t k1 = 0; k1 < B && v == JPEG_MAP_FAIL; k1++) if (c.in[k1] == E) v = k1;
                if (v == JPEG_MAP_FAIL) {
                    const JpegSpan sp = jpeg_span_walk(L, word1, E, (uint32_t)g * CB, std::min((uint32_t)g * CB + CB / 2, limit[g]), limit[g], seg_end[g], F);
                    c.rep_out[k] = sp.out; c.rep_n[k] = sp.n; c.rep_mid[k] = sp.mid; c.rep_nmid[k] = sp.nmid;
                    via = 1;
                    repairs++;
                    for (uint32_t k1 = 0; k1 < B && v == JPEG_MAP_FAIL; k1++) if (c.out[k1] == sp.out) v = k1;
                }
            }
            c.map |= v << (4 * k);
            c.via |= via << k;
        }
    }
    // C. along the chain (on the device: a scan over the maps inside a workgroup, a look-back between workgroups).  Where
    // even the repair walk joined nothing, its exit is carried on as an explicit state (a "chase") until it joins.
    {
        bool explicit_state = false;
        uint64_t S = 0;
        uint32_t idx = 0;                                            // which of the predecessor's exits is the true one
        for (size_t g = 0; g < n; g++) {
            Cand& c = cand[g];
            if (c.exact) { entry[g] = c.in[0]; slots[g] = c.n[0]; middle[g] = c.mid[0]; slots_mid[g] = c.nmid[0]; idx = 0; explicit_state = false; continue; }
            if (explicit_state) {
                chases++;
                uint32_t k1 = 6;
                for (uint32_t k = 0; k < B && k1 == 6; k++) if (c.in[k] == S) k1 = k;
                if (k1 < 6) { entry[g] = S; slots[g] = c.n[k1]; idx = k1; explicit_state = false; continue; }
                const JpegSpan sp = jpeg_span_walk(L, word1, S, (uint32_t)g * CB, limit[g], limit[g], seg_end[g], F);    // (a chunk reached by the chase: no middle state, one lane decodes it)
                entry[g] = S; slots[g] = sp.n;
                for (uint32_t k = 0; k < B && k1 == 6; k++) if (c.out[k] == sp.out) k1 = k;
                if (k1 < 6) { idx = k1; explicit_state = false; } else S = sp.out;
                continue;
            }
            const uint32_t v = jpeg_map_at(c.map, idx);
            if (!((c.via >> idx) & 1)) {
                if (v == JPEG_MAP_FAIL) return IMP_ERROR_DEVICE;     // (the true exit of a chunk is never "no candidate")
                entry[g] = c.in[v]; slots[g] = c.n[v]; middle[g] = c.mid[v]; slots_mid[g] = c.nmid[v]; idx = v;
            } else {
                misses++;
                entry[g] = cand[g - 1].out[idx]; slots[g] = c.rep_n[idx]; middle[g] = c.rep_mid[idx]; slots_mid[g] = c.rep_nmid[idx];
                if (v != JPEG_MAP_FAIL) idx = v;
                else { S = c.rep_out[idx]; explicit_state = true; }
            }
        }
    }
    if (const char* tr = std::getenv("IMPGPU_JPEG_TRACE"); tr && !std::strcmp(tr, "3"))
        for (size_t g = 0; g < n; g++) std::fprintf(stderr, "ce %zu %016llx %u\n", g, (unsigned long long)entry[g], slots[g]);
    g_sync_stats[0] = (int)n; g_sync_stats[1] = misses; g_sync_stats[2] = repairs; g_sync_stats[3] = chases;
    g_sync_stats[4] = (int)CB; g_sync_stats[5] = (int)F.overlap_bits; g_sync_stats[6] = (int)B; g_sync_stats[7] = 0;
    if (rounds) *rounds = misses;
    *status = 0;
    // ---- k_jpeg_write: two lanes (units) per chunk where the chunk's middle state is known -- the first decodes the symbols
    // that start before the middle, the second enters there; everything (slot prefix, budget, verdicts, DC sums) is per unit
    struct Unit { size_t g; uint64_t entry; uint32_t limit, own_n; bool origin, tail; };
    std::vector<Unit> units;
    for (size_t g = 0; g < n; g++) {
        const bool have = (uint32_t)(middle[g] >> 48) == 0 && (uint32_t)middle[g] < limit[g];     // (as write_body decides it)
        if (have) {
            units.push_back(Unit{g, entry[g], std::min((uint32_t)g * CB + CB / 2, limit[g]), slots_mid[g], origin[g] != 0, false});
            units.push_back(Unit{g, middle[g], limit[g], slots[g] - slots_mid[g], false, true});
        } else units.push_back(Unit{g, entry[g], limit[g], slots[g], origin[g] != 0, true});
    }
    g_sync_stats[7] = (int)(units.size() - n);                       // chunks decoded by two lanes
    const size_t nu = units.size();
    std::vector<JpegDecoded> dec(nu);
    std::vector<uint32_t> slot0(nu);
    uint32_t run_n = 0;
    for (size_t u = 0; u < nu; u++) {
        const Unit& U = units[u];
        const size_t g = U.g;
        const uint32_t sg = meta[g];
        if (U.origin) run_n = 0;
        JpegWriteCtx W;
        W.coef = coef;
        W.slot0 = slot0[u] = sg * (uint32_t)F.slots_per_seg + run_n;
        W.dc0[0] = W.dc0[1] = W.dc0[2] = 0;                          // DC terms relative to the unit's entry; k_jpeg_dcfix adds the rest
        W.status = status;
        const uint32_t base_n = run_n;
        run_n += U.own_n;
        const size_t last = (sg + 1 < F.nsegs ? scan.seg_first_chunk[sg + 1] : n) - 1;
        const uint32_t want = std::min<uint32_t>((uint32_t)F.slots_per_seg, F.total_slots - sg * (uint32_t)F.slots_per_seg);
        const bool closes = g == last && U.tail;
        dec[u] = JpegDecoded{};
        if (!closes && run_n > want) { *status |= JPEG_ST_OVERRUN; continue; }
        // the last unit of an interval walks with the interval's remaining slots as a budget and gives the verdict
        const uint32_t budget = closes ? (want >= base_n ? want - base_n : 0u) : 0xffffffffu;
        dec[u] = jpeg_write_chunk(L, K, word1, U.entry, U.limit, seg_end[g], F, &W, budget);
        if (closes) {
            const uint32_t pe = (uint32_t)dec[u].exit, fle = (uint32_t)(dec[u].exit >> 48);
            if ((fle & JPEG_FL_INVALID) || pe > seg_end[g] || seg_end[g] - pe >= 8) *status |= JPEG_ST_BAD_CODE;
            if (base_n + dec[u].n != want) *status |= JPEG_ST_BAD_COUNT;
        } else if (dec[u].n != U.own_n) *status |= JPEG_ST_BAD_COUNT;      // the lean walk and the full walk disagree: cannot happen
    }
    // ---- k_jpeg_dcfix
    int run_dc[3] = {0, 0, 0};
    for (size_t u = 0; u < nu; u++) {
        if (units[u].origin) run_dc[0] = run_dc[1] = run_dc[2] = 0;
        jpeg_dc_fixup(K, F, coef, slot0[u], units[u].entry, dec[u].ndc, run_dc);
        for (int k = 0; k < 3; k++) run_dc[k] += dec[u].dc[k];
    }
    return IMP_OK;
}

}  // namespace imp

using namespace imp;

extern "C" {

int impgpu_jpeg_info(const unsigned char* blob, size_t size, int* width, int* height, int* channels) {
    JpegHeader H;
    if (int rc = jpeg_parse(blob, size, &H)) return rc;
    if (width) *width = H.width;
    if (height) *height = H.height;
    if (channels) *channels = H.ncomp;
    return IMP_OK;
}

int impgpu_jpeg_classify(const unsigned char* blob, size_t size) {
    JpegHeader H;
    const int rc = jpeg_parse(blob, size, &H);
    if (rc == IMP_OK) return 0;
    if (rc == IMP_ERROR_UNSUPPORTED) return H.why > 0 && H.why < JPEG_WHY_COUNT ? H.why : JPEG_WHY_OTHER;
    return JPEG_WHY_COUNT;                                           // damaged before the first scan
}

void impgpu_jpeg_sync_stats(int stats[8]) {
    if (stats) for (int i = 0; i < 8; i++) stats[i] = g_sync_stats[i];
}

int impgpu_jpeg_coefficients(const unsigned char* blob, size_t size, int how, short* out, size_t capacity, int* info) {
    if (!blob || !out) return IMP_ERROR_INVALID_ARGS;
    JpegHeader H;
    if (int rc = jpeg_parse(blob, size, &H)) return rc;
    JpegFrame F;
    int dc_ids[2], ac_ids[2];
    if (int rc = jpeg_frame_setup(H, &F, dc_ids, ac_ids)) return rc;
    if ((size_t)F.total_slots > capacity) return IMP_ERROR_INVALID_ARGS;
    // (how = 1: the planes start out as garbage -- the write walk has to put every coefficient of every block there itself)
    std::memset(out, how == 0 ? 0 : 0x5a, (size_t)F.total_slots * sizeof(short));
    unsigned status = 0;
    int rounds = 0;
    int rc;
    if (how == 0) rc = jpeg_host_entropy(blob, size, H, out, F);
    else rc = jpeg_emulate_entropy(blob, size, H, F, dc_ids, ac_ids, out, &status, &rounds);
    if (info) {
        info[0] = (int)F.total_slots; info[1] = (int)status; info[2] = rounds;
        for (int i = 0; i < 3; i++) { info[3 + i] = (int)F.coef_off[i]; info[6 + i] = F.bw[i]; info[9 + i] = F.bh[i]; }
    }
    if (!rc && status) rc = IMP_ERROR_DECODE_FAILED;
    return rc;
}

}  // extern "C"
