// imp_jpeg_core.h -- the chunk decoder of the device's entropy stage, written once for both sides: the kernels of
// imp_jpeg.hip run it per lane, and the host runs the very same function lane by lane in jpeg_emulate_entropy
// (imp_jpeg.cpp) so that the self-synchronising scheme can be checked on a machine without a GPU
// (tests/test_jpeg_host.py) and under AddressSanitizer.  The product path never decodes on the host.
#pragma once
#include "imp_jpeg.h"

#if defined(__HIPCC__)
#define IMP_HD __host__ __device__
#else
#define IMP_HD
#endif

namespace imp {

// decoder state between two symbols: bit position, block within the MCU, next coefficient index (0 = the DC term)
constexpr uint32_t JPEG_FL_END = 1u, JPEG_FL_INVALID = 2u;
IMP_HD inline uint64_t jpeg_pack_state(uint32_t p, uint32_t c, uint32_t z, uint32_t fl) {
    return (uint64_t)p | ((uint64_t)(c | (z << 8) | (fl << 16)) << 32);
}

// What a decoder lane needs to know about a symbol, packed: code length (5 bits), value bits to read (4), zero run before the
// coefficient (4), end-of-block (1).  DC symbols (the category IS the number of value bits) and AC symbols (run << 4 | size;
// 0x00 = end of block, 0xF0 = sixteen zeros) look the same from there on.
IMP_HD inline uint32_t jpeg_lut_entry(uint32_t len, uint32_t sym, bool is_dc) {
    // (an AC symbol with no value bits is the end of the block unless its run is 15 -- libjpeg's reading of the values T.81
    // leaves undefined, 0x10 .. 0xE0, which only a damaged table holds)
    const uint32_t size = sym & 15, run = is_dc ? 0u : sym >> 4, eob = (!is_dc && size == 0 && run != 15) ? 1u : 0u;
    return len | (size << 5) | (run << 9) | (eob << 13);
}

// The same entry as the lanes read it, with the two sums the resynchronising rounds live on worked out once: bits 16..20 =
// code length + value bits (what the symbol takes from the stream), bits 21..27 = how far the coefficient index moves
// (run + 1; 64 for an end of block: "to the block's end" and "past it" are the same thing to the index).  0 stays 0.
IMP_HD inline uint32_t jpeg_lut_expand(uint32_t e) {
    if ((e & 31) == 0) return e;                                    // no code of this length: 0, or the pointer to the second level
    const uint32_t total = (e & 31) + ((e >> 5) & 15), adv = ((e >> 13) & 1) ? 64u : ((e >> 9) & 15) + 1;
    return e | (total << 16) | (adv << 21);
}

struct JpegHuffTabs {                       // the four tables as a decoder lane reads them (LDS on the device)
    uint32_t lut[4][1 << JPEG_LOOKBITS];    // jpeg_lut_expand(JpegHuffDev::lut)
    uint32_t sub[4][JPEG_SUB_ENTRIES];      // jpeg_lut_expand(JpegHuffDev::sub)
    uint32_t limit[4][18];
    int32_t offs[4][18];
    uint8_t vals[4][256];
};
struct JpegBlockTabs {                      // where the coefficients go (k_jpeg_write, k_jpeg_dcfix)
    uint8_t natural[64];                    // zig-zag position -> position in the 8x8 block
    uint32_t blk_base[8];                   // per block of the MCU: its first coefficient in MCU (0,0), in shorts
    uint32_t blk_dx[8], blk_dy[8];          // ... and how far the same block is in the next MCU / the next MCU row
};

// How a walk reads its tables: the expanded form above (LDS of the kernels that walk a lot), or the four tables as the host
// built them, widened entry by entry (k_jpeg_select: 12 KB instead of 22, for the few walks of a chase).
struct JpegHuffCompact { JpegHuffDev t[4]; };
IMP_HD inline uint32_t jpeg_tab_first(const JpegHuffTabs& L, uint32_t tab, uint32_t i) { return L.lut[tab][i]; }
IMP_HD inline uint32_t jpeg_tab_second(const JpegHuffTabs& L, uint32_t tab, uint32_t i) { return L.sub[tab][i]; }
IMP_HD inline uint32_t jpeg_tab_limit(const JpegHuffTabs& L, uint32_t tab, int l) { return L.limit[tab][l]; }
IMP_HD inline int32_t jpeg_tab_offs(const JpegHuffTabs& L, uint32_t tab, uint32_t l) { return L.offs[tab][l]; }
IMP_HD inline uint32_t jpeg_tab_val(const JpegHuffTabs& L, uint32_t tab, uint32_t i) { return L.vals[tab][i]; }
IMP_HD inline uint32_t jpeg_tab_first_raw(const JpegHuffTabs& L, uint32_t tab, uint32_t i) { return L.lut[tab][i]; }
IMP_HD inline uint32_t jpeg_tab_second_raw(const JpegHuffTabs& L, uint32_t tab, uint32_t i) { return L.sub[tab][i]; }
IMP_HD inline uint32_t jpeg_tab_first_raw(const JpegHuffCompact& L, uint32_t tab, uint32_t i) { return L.t[tab].lut[i]; }
IMP_HD inline uint32_t jpeg_tab_second_raw(const JpegHuffCompact& L, uint32_t tab, uint32_t i) { return L.t[tab].sub[i]; }
IMP_HD inline uint32_t jpeg_tab_first(const JpegHuffCompact& L, uint32_t tab, uint32_t i) { return jpeg_lut_expand(L.t[tab].lut[i]); }
IMP_HD inline uint32_t jpeg_tab_second(const JpegHuffCompact& L, uint32_t tab, uint32_t i) { return jpeg_lut_expand(L.t[tab].sub[i]); }
IMP_HD inline uint32_t jpeg_tab_limit(const JpegHuffCompact& L, uint32_t tab, int l) { return L.t[tab].limit[l]; }
IMP_HD inline int32_t jpeg_tab_offs(const JpegHuffCompact& L, uint32_t tab, uint32_t l) { return L.t[tab].offs[l]; }
IMP_HD inline uint32_t jpeg_tab_val(const JpegHuffCompact& L, uint32_t tab, uint32_t i) { return L.t[tab].vals[i]; }

// blk_base / blk_dx / blk_dy of JpegBlockTabs from the frame geometry (luma blocks first, then Cb, Cr)
IMP_HD inline void jpeg_block_steps(const JpegFrame& F, int k, uint32_t* base, uint32_t* dx, uint32_t* dy) {
    const int nluma = F.bpm == 1 ? 1 : F.bpm - 2;
    int ci = 0, bx = 0, by = 0;
    if (k < nluma) { bx = k % F.hs; by = k / F.hs; } else { ci = k - nluma + 1; }
    const int h = ci ? 1 : F.hs, v = ci ? 1 : F.vs;
    const unsigned off = ci == 0 ? F.coef_off[0] : ci == 1 ? F.coef_off[1] : F.coef_off[2];
    const int bw = ci == 0 ? F.bw[0] : ci == 1 ? F.bw[1] : F.bw[2];
    *base = off + (uint32_t)(by * bw + bx) * 64u;
    *dx = (uint32_t)h * 64u;
    *dy = (uint32_t)(v * bw) * 64u;
}

struct JpegDecoded {
    uint64_t exit;
    uint32_t n;          // coefficient slots passed
    int dc[3];           // sum of the DC differences met, per component
    uint32_t ndc;        // DC symbols met = blocks that begin in the chunk
};

struct JpegWriteCtx {
    int16_t* coef;
    uint32_t slot0;      // absolute slot (within the scan) of the chunk's first symbol
    int dc0[3];          // DC predictors at the chunk's entry
    uint32_t* status;
    // device only: where the lane builds the block it is decoding (64 zeroed shorts in LDS) and where its wave lists the
    // blocks that are complete (64 x {LDS byte address, first short in the planes}); byte addresses in LDS
    uint32_t stage, list;
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef int16_t __attribute__((address_space(3))) * JpegLdsShort;
typedef uint32_t __attribute__((address_space(3))) * JpegLdsWord;
typedef int __attribute__((ext_vector_type(4))) JpegV4;
typedef JpegV4 __attribute__((address_space(3))) * JpegLdsV4;
typedef JpegV4 __attribute__((address_space(1))) * JpegGlobalV4;
// The complete blocks of a wave leave together: eight lanes per block, 16 bytes each -- one store instruction writes eight
// whole 128-byte lines, where a 2-byte store per coefficient and lane was 64 partial lines per instruction (70 % of the
// kernel's time), and the buffers go back to zero on the way.  Every lane of the wave must call this together.
__device__ __forceinline__ void jpeg_flush_blocks(const JpegWriteCtx* W, bool ready, uint32_t at) {
    const uint64_t mask = __ballot(ready);
    if (mask == 0) return;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    JpegLdsWord list = (JpegLdsWord)(uintptr_t)W->list;
    if (ready) { list[2 * rank] = W->stage; list[2 * rank + 1] = at; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint32_t nready = (uint32_t)__popcll(mask);
    const JpegV4 zero = {0, 0, 0, 0};
    for (uint32_t base = 0; base < nready; base += 8) {
        const uint32_t i = base + (lane >> 3), part = lane & 7;
        if (i < nready) {
            const uint32_t src = list[2 * i], dst = list[2 * i + 1];
            JpegLdsV4 q = (JpegLdsV4)(uintptr_t)(src + part * 16);
            const JpegV4 v = *q;
            *q = zero;
            ((JpegGlobalV4)(uintptr_t)(W->coef + dst))[part] = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
#endif

IMP_HD inline void jpeg_flag(uint32_t* status, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(status, bits);
#else
    *status |= bits;
#endif
}

// a coefficient store; on the device a GLOBAL one (through a pointer out of a table it would be a flat store, which counts
// as an LDS access as well and makes every wait for a table read wait for the stores too)
IMP_HD inline void jpeg_put_coef(int16_t* base, uint32_t at, int16_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef int16_t __attribute__((address_space(1))) * GlobalCoef;
    ((GlobalCoef)(uintptr_t)base)[at] = v;
#else
    base[at] = v;
#endif
}

// the i-th 32-bit word of the stream as memory holds it (first byte in bits 0..7) -> first bit of the stream in bit 31
IMP_HD inline uint32_t jpeg_be(uint32_t raw) { return __builtin_bswap32(raw); }

// The decoder's view of the stream: the next 32 bits, out of two consecutive words (first bit of the stream in bit 31) of which
// 32 - sh bits of w0 are used up; sh = 0 .. 31, 0 = all of w0, the view is w1.  One v_alignbit_b32 on the device (a 64-bit
// shift register was a quarter-rate v_lshlrev_b64 per symbol).  Consuming n bits: sh -= n, and below zero the words move up.
IMP_HD inline uint32_t jpeg_window(uint32_t w0, uint32_t w1, int sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(w0, w1, (uint32_t)sh);
#else
    return sh ? (w0 << (32 - sh)) | (w1 >> sh) : w1;
#endif
}

// a whole block of zeros (the block's owner writes them just before its coefficients: no fill of the planes beforehand, and
// every 128-byte line leaves the cache completely written -- 2-byte stores into lines zeroed long before cost a
// read-modify-write each in memory, 70 % of the write kernel's time when it was built that way)
IMP_HD inline void jpeg_zero_block(int16_t* base, uint32_t at) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef int __attribute__((ext_vector_type(4))) v4i;
    typedef v4i __attribute__((address_space(1))) * GlobalV4;
    GlobalV4 q = (GlobalV4)(uintptr_t)(base + at);
    const v4i zero = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = zero;
#else
    for (int i = 0; i < 64; i++) base[at + i] = 0;
#endif
}

// The stream as a decoder lane reads it: one word per load (`word(i)` = the i-th word as loaded), the next one requested a
// refill ahead.  Measured against sixteen bytes per load with a four-word queue in registers (same box, 64 files): k_jpeg_write
// 427 against 517 us, k_jpeg_walks 460 against 512 -- the walks are bound by the instructions they issue, and the queue costs
// more of those than the wider loads save.  What matters is that the loads are PLAIN ones: a lane comes back to its 128-byte
// line for the next word, and a non-temporal load (round 3's) does not keep the line -- k_jpeg_walks 318 against 465 us.
template <class WordFn>
struct JpegBitReader1 {
    WordFn word;
    uint32_t w0, w1, ahead, nxt;
    int sh;
    IMP_HD explicit JpegBitReader1(WordFn f) : word(f) {}
    IMP_HD void start(uint32_t p) {
        const uint32_t widx = p >> 5, bit = p & 31;
        w0 = bit ? jpeg_be(word(widx)) : 0u;
        w1 = jpeg_be(word(widx + (bit ? 1u : 0u)));
        nxt = widx + (bit ? 2u : 1u);
        sh = bit ? 32 - (int)bit : 0;
        ahead = word(nxt);
    }
    IMP_HD uint32_t window() const { return jpeg_window(w0, w1, sh); }
    IMP_HD void take(uint32_t bits) {
        sh -= (int)bits;
        if (sh < 0) {
            w0 = w1;
            w1 = jpeg_be(ahead);
            sh += 32;
            nxt++;
            ahead = word(nxt);
        }
    }
};

// One chunk, decoded for good: every symbol that STARTS before `limit`, from the packed state `entry` (its true one, found
// by k_jpeg_select).  `word(i)` returns the i-th 32-bit word of the unstuffed stream AS LOADED (little-endian: jpeg_be turns it
// round where it is used, not where it is loaded, so that the load has a whole refill's worth of symbols to arrive).
// `max_slots` ends the walk once that many coefficient slots have been passed: the last chunk of an interval stops after the
// interval's last MCU like a sequential decoder does, whatever the (up to seven) padding bits behind it look like.
//
// A BLOCK BELONGS TO THE CHUNK IT BEGINS IN: that lane zeroes its 64 coefficients, writes its non-zero ones and, if the block
// runs past `limit`, decodes on to the block's end; the lane of the next chunk decodes those symbols too (it has to get
// past them) but stores nothing until a block begins.  So every block of the planes is written by exactly one lane, whole.
// Returned: slots and DC symbols / sums of the symbols that start before `limit` only.  DC terms are written relative to
// the chunk's entry (dc0 = 0: jpeg_dc_fixup adds the predictors).
//
// Written for a SIMT lane: one loop, one table read per symbol, selects instead of branches, no array indexed by a run-time
// value (those live in scratch memory on the device), the next stream word always one step ahead in a register.  A
// symbol is (code length, value bits, zero run, end-of-block) straight out of the table; DC and AC symbols, ZRL and EOB all
// take the same few instructions: the coefficient index moves by run + 1, or to the block's end.
template <class Tabs, class WordFn>
IMP_HD inline JpegDecoded jpeg_write_chunk(const Tabs& L, const JpegBlockTabs& K, WordFn word, uint64_t entry, uint32_t limit, uint32_t seg_end,
                                           const JpegFrame& F, const JpegWriteCtx* W, uint32_t max_slots = 0xffffffffu, bool active = true) {
    JpegDecoded r;
    r.exit = entry;
    r.n = 0;
    r.ndc = 0;
    r.dc[0] = r.dc[1] = r.dc[2] = 0;
    uint32_t p = (uint32_t)entry;
    uint32_t c = (uint32_t)(entry >> 32) & 0xff, z = (uint32_t)(entry >> 40) & 0xff;
    // nothing can follow an invalid / ended predecessor.  (No early return: on the device every lane of the wave has to reach
    // the loop below, where finished blocks are written out by the lanes together -- `active` = this lane has a chunk at all.)
    bool ok = active && (uint32_t)(entry >> 48) == 0 && p < limit;
    p = ok ? p : 0u;
    c = ok ? c : 0u;
    const uint32_t bpm = (uint32_t)F.bpm, nluma = bpm == 1 ? 1u : bpm - 2;
    // per block of the MCU: its component (two bits each) and whether it takes the second DC / AC table (a bit each)
    uint32_t comp_of = 0, dc_sel = 0, ac_sel = 0;
    for (uint32_t k = 0; k < 6; k++) {
        const uint32_t ci = k < nluma ? 0u : (k - nluma + 1 < 3 ? k - nluma + 1 : 2u);
        comp_of |= ci << (2 * k);
        dc_sel |= (uint32_t)(ci == 0 ? F.dctab[0] : ci == 1 ? F.dctab[1] : F.dctab[2]) << k;
        ac_sel |= (uint32_t)(ci == 0 ? F.actab[0] : ci == 1 ? F.actab[1] : F.actab[2]) << k;
    }
    JpegBitReader1<WordFn> bits(word);
    bits.start(p);
    uint32_t fl = 0, n = 0, damaged = 0, ndc = 0;
    int dcs0 = 0, dcs1 = 0, dcs2 = 0;
    // where the coefficients go: MCU coordinates are carried along, a block's address is base + mx*dx + my*dy
    uint32_t mx = 0, my = 0, blk = 0;
    bool blk_ok = false;
    bool own = false;                                               // the block being decoded began in this chunk
    if (ok) {
        const uint32_t gb = W->slot0 >> 6, mcu = gb / bpm;
        my = mcu / (uint32_t)F.mcux;
        mx = mcu - my * (uint32_t)F.mcux;
        if (gb - mcu * bpm != c || (W->slot0 & 63) != z) { jpeg_flag(W->status, JPEG_ST_BAD_COUNT); ok = false; }
        blk_ok = my < (uint32_t)F.mcuy;
        blk = K.blk_base[c] + mx * K.blk_dx[c] + my * K.blk_dy[c];
    }
    // (one way out of the loop, at its bottom: an ending skips the rest of the body instead of jumping out -- every extra exit
    // of a divergent loop costs a handful of scalar instructions per symbol for its execution-mask bookkeeping)
    bool go = ok && n < max_slots;
#if defined(__HIP_DEVICE_COMPILE__)
    // on the device the whole wave stays in the loop until its last lane is done: finished blocks leave cooperatively
    JpegLdsShort stage = (JpegLdsShort)(uintptr_t)W->stage;
    bool dirty = false;                                             // the block buffer holds coefficients not yet written out
    while (__any(go)) {
        bool ready = false;
        uint32_t ready_at = 0;
        if (go) {
#else
    while (go) {
        {
#endif
        const uint32_t ci = (comp_of >> (2 * c)) & 3;
        const bool isdc = z == 0;
        const uint32_t tab = isdc ? ((dc_sel >> c) & 1) : 2 + ((ac_sel >> c) & 1);
        const uint32_t win = bits.window(), peek = win >> 16;
        // (only the entry as the host built it is read here -- length, value bits, run, end-of-block: its low 16 bits)
        uint32_t e = jpeg_tab_first_raw(L, tab, peek >> (16 - JPEG_LOOKBITS));
        if ((e & 31) == 0) {                                        // a code longer than the table's index
            bool none;
            if (e & 0x8000u) {                                      // ... in the second level, indexed by the bits that follow
                const uint32_t nb = (e >> 12) & 7, off = ((e >> 5) & 127) << 1;
                e = jpeg_tab_second_raw(L, tab, off + ((win >> (32 - JPEG_LOOKBITS - nb)) & ((1u << nb) - 1)));
                none = e == 0;
            } else {                                                // ... or (a table with too many of them) its length from the
                uint32_t len = JPEG_LOOKBITS + 1;                   // canonical limits, its symbol from the value list
                for (int l = JPEG_LOOKBITS + 1; l < 16; l++) len += peek >= jpeg_tab_limit(L, tab, l) ? 1u : 0u;
                none = peek >= jpeg_tab_limit(L, tab, 16);          // no code starts with these 16 bits
                const uint32_t sym = jpeg_tab_val(L, tab, (uint32_t)(jpeg_tab_offs(L, tab, len) + (int)(peek >> (16 - len))) & 255);
                e = none ? 0u : jpeg_lut_entry(len, sym, isdc);
            }
            fl = none ? ((seg_end - p < 16) ? JPEG_FL_END : JPEG_FL_INVALID) : fl;   // (the 1-bits that pad an interval are no code either)
        }
        const uint32_t len = e & 31, size = (e >> 5) & 15, run = (e >> 9) & 15, eob = (e >> 13) & 1;
        const uint32_t total = len + size;
        if (fl == 0 && p + total > seg_end) fl = JPEG_FL_END;       // the interval's padding, not a symbol
        if (fl == 0) {
            // the value: `size` bits after the code, negative when its first bit is 0 (T.81 F.2.2.1)
            const uint32_t vb = win << len;
            const int v = (int)((vb >> 1) >> (31 - size)) + ((int)(~vb) >> 31 & (1 - (1 << size)));
            const bool counted = p < limit;                         // (not so for the symbols that finish a block behind the chunk's end)
            p += total;
            bits.take(total);
            uint32_t adv = eob ? 64 - z : run + 1;
            const bool over = z + adv > 64;                         // a run that leaves the block: damaged
            adv = over ? 64 - z : adv;
            damaged |= over ? 1u : 0u;
            if (isdc) {
                own = true;
                const int dcv = (ci == 0 ? W->dc0[0] + dcs0 : ci == 1 ? W->dc0[1] + dcs1 : W->dc0[2] + dcs2) + v;
#if defined(__HIP_DEVICE_COMPILE__)
                if (blk_ok) { stage[0] = (int16_t)dcv; dirty = true; }
#else
                if (blk_ok) {
                    jpeg_zero_block(W->coef, blk);
                    jpeg_put_coef(W->coef, blk, (int16_t)dcv);
                }
#endif
            } else if (size && !over && blk_ok && own) {
#if defined(__HIP_DEVICE_COMPILE__)
                stage[K.natural[z + run]] = (int16_t)v;
#else
                jpeg_put_coef(W->coef, blk + K.natural[z + run], (int16_t)v);
#endif
            }
            ndc += isdc ? 1u : 0u;
            dcs0 += (isdc && ci == 0) ? v : 0;
            dcs1 += (isdc && ci == 1) ? v : 0;
            dcs2 += (isdc && ci == 2) ? v : 0;
            z += adv;
            n += counted ? adv : 0u;
            if (z >= 64) {                                          // next block
#if defined(__HIP_DEVICE_COMPILE__)
                ready = dirty;
                ready_at = blk;
                dirty = false;
#endif
                z = 0;
                c++;
                if (c == bpm) { mx++; if (mx == (uint32_t)F.mcux) { mx = 0; my++; } }
                c = c == bpm ? 0 : c;
                blk_ok = my < (uint32_t)F.mcuy;
                blk = K.blk_base[c] + mx * K.blk_dx[c] + my * K.blk_dy[c];
            }
        }
        go = fl == 0 && n < max_slots && (p < limit || (own && z != 0));
        }
#if defined(__HIP_DEVICE_COMPILE__)
        jpeg_flush_blocks(W, ready, ready_at);
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    jpeg_flush_blocks(W, dirty, blk);                               // (a block cut short by an ending: what there is of it)
#endif
    if (damaged || (fl & JPEG_FL_INVALID)) jpeg_flag(W->status, JPEG_ST_BAD_CODE);
    if (ok) r.exit = jpeg_pack_state(p, c, z, fl);
    r.n = n;
    r.ndc = ndc;
    r.dc[0] = dcs0;
    r.dc[1] = dcs1;
    r.dc[2] = dcs2;
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// The same walk as the kernel runs it (round 5).  jpeg_write_chunk above spent 273 instructions per symbol on the device
// (profiles/r04_jpeg_sq_counters.txt: 101 k per wave of 2048-bit chunks) against 26 in jpeg_span_walk, because with 64
// lanes in a wave SOME lane ends a block in nine iterations out of ten: the block bookkeeping (next block of the MCU, MCU
// coordinates, the block's address: table reads and multiplies) and the cooperative flush (a ballot, a list in LDS, a
// store loop) ran for one or two blocks at a time in almost every iteration.  Here a lane that ends a block STEPS ASIDE
// (`pend`) and the symbol loop goes on with the others until an eighth of the wave's working lanes (at most eight: one
// full store instruction) is waiting, or nobody is left to decode; then the waiting lanes do their bookkeeping together
// and their blocks leave in one flush.  A waiting lane loses a couple of iterations per block of about 27; the wave's
// instruction count per symbol falls to what the decode itself needs.  States, counts, flags and the bytes written are
// those of jpeg_write_chunk (DC terms relative to the chunk's entry: W->dc0 is not read).
template <class Tabs, class WordFn>
__device__ inline JpegDecoded jpeg_write_chunk_dev(const Tabs& L, const JpegBlockTabs& K, WordFn word, uint64_t entry, uint32_t limit, uint32_t seg_end,
                                                   const JpegFrame& F, const JpegWriteCtx* W, uint32_t max_slots, bool active) {
    JpegDecoded r;
    r.exit = entry;
    uint32_t p = (uint32_t)entry;
    uint32_t c = (uint32_t)(entry >> 32) & 0xff, z = (uint32_t)(entry >> 40) & 0xff;
    bool ok = active && (uint32_t)(entry >> 48) == 0 && p < limit;
    p = ok ? p : 0u;
    c = ok ? c : 0u;
    const uint32_t bpm = (uint32_t)F.bpm, nluma = bpm == 1 ? 1u : bpm - 2;
    // per block of the MCU: its component (two bits each); four bits each: the table of its DC symbol (0 / 1) and, two bits
    // up, of its AC symbols (2 / 3)
    uint32_t comp_of = 0, tabsel = 0;
    for (uint32_t k = 0; k < 6; k++) {
        const uint32_t ci = k < nluma ? 0u : (k - nluma + 1 < 3 ? k - nluma + 1 : 2u);
        const uint32_t dct = (uint32_t)(ci == 0 ? F.dctab[0] : ci == 1 ? F.dctab[1] : F.dctab[2]);
        const uint32_t act = 2u + (uint32_t)(ci == 0 ? F.actab[0] : ci == 1 ? F.actab[1] : F.actab[2]);
        comp_of |= ci << (2 * k);
        tabsel |= (dct | (act << 2)) << (4 * k);
    }
    JpegBitReader1<WordFn> bits(word);
    bits.start(p);
    uint32_t fl = 0, n = 0, damaged = 0, ndc = 0;
    int dcs0 = 0, dcs1 = 0, dcs2 = 0;
    uint32_t mx = 0, my = 0, blk = 0;
    bool blk_ok = false;
    if (ok) {
        const uint32_t gb = W->slot0 >> 6, mcu = gb / bpm;
        my = mcu / (uint32_t)F.mcux;
        mx = mcu - my * (uint32_t)F.mcux;
        if (gb - mcu * bpm != c || (W->slot0 & 63) != z) { jpeg_flag(W->status, JPEG_ST_BAD_COUNT); ok = false; }
        blk_ok = my < (uint32_t)F.mcuy;
        blk = K.blk_base[c] + mx * K.blk_dx[c] + my * K.blk_dy[c];
    }
    JpegLdsShort stage = (JpegLdsShort)(uintptr_t)W->stage;
    // The lane's flags in ONE register (as separate bools the compiler keeps each as a wave-wide mask in scalar registers and
    // spends three scalar instructions on every update, a dozen per iteration just carrying them round the loop):
    // GO = more symbols to decode; PEND = its block has ended, it waits for the next flush; OWN = the block being decoded
    // began in this chunk; DIRTY = the block buffer holds coefficients not yet written out; BLKOK = the block lies inside the frame
    constexpr uint32_t GO = 1, PEND = 2, OWN = 4, DIRTY = 8, BLKOK = 16;
    uint32_t st = ((ok && n < max_slots) ? GO : 0u) | (blk_ok ? BLKOK : 0u);
    for (;;) {
        const uint64_t working = __ballot((st & (GO | PEND)) != 0);
        if (working == 0) break;
        const uint32_t eighth = (uint32_t)__popcll(working) >> 3;
        const uint32_t enough = eighth < 1 ? 1u : eighth > 8 ? 8u : eighth;
        for (;;) {
            const bool act = (st & (GO | PEND)) == GO;
            if (!__any(act) || (uint32_t)__popcll(__ballot((st & PEND) != 0)) >= enough) break;
            if (act) {
                const bool isdc = z == 0;
                const uint32_t tab = (tabsel >> (4 * c + (isdc ? 0u : 2u))) & 3;
                const uint32_t win = bits.window(), peek = win >> 16;
                uint32_t e = jpeg_tab_first_raw(L, tab, peek >> (16 - JPEG_LOOKBITS));
                if ((e & 31) == 0) {                                // a code longer than the table's index
                    bool none;
                    if (e & 0x8000u) {                              // ... in the second level, indexed by the bits that follow
                        const uint32_t nb = (e >> 12) & 7, off = ((e >> 5) & 127) << 1;
                        e = jpeg_tab_second_raw(L, tab, off + ((win >> (32 - JPEG_LOOKBITS - nb)) & ((1u << nb) - 1)));
                        none = e == 0;
                    } else {                                        // ... or (a table with too many of them) by the canonical limits
                        uint32_t len = JPEG_LOOKBITS + 1;
                        for (int l = JPEG_LOOKBITS + 1; l < 16; l++) len += peek >= jpeg_tab_limit(L, tab, l) ? 1u : 0u;
                        none = peek >= jpeg_tab_limit(L, tab, 16);
                        const uint32_t sym = jpeg_tab_val(L, tab, (uint32_t)(jpeg_tab_offs(L, tab, len) + (int)(peek >> (16 - len))) & 255);
                        e = none ? 0u : jpeg_lut_entry(len, sym, isdc);
                    }
                    fl = none ? ((seg_end - p < 16) ? JPEG_FL_END : JPEG_FL_INVALID) : fl;
                }
                const uint32_t len = e & 31, size = (e >> 5) & 15, run = (e >> 9) & 15, eob = (e >> 13) & 1;
                const uint32_t total = len + size;
                if (fl == 0 && p + total > seg_end) fl = JPEG_FL_END;   // the interval's padding, not a symbol
                if (fl == 0) {
                    const uint32_t vb = win << len;
                    const int v = (int)((vb >> 1) >> (31 - size)) + ((int)(~vb) >> 31 & (1 - (1 << size)));
                    const bool counted = p < limit;
                    p += total;
                    bits.take(total);
                    uint32_t adv = eob ? 64 - z : run + 1;
                    const bool over = z + adv > 64;                 // a run that leaves the block: damaged
                    adv = over ? 64 - z : adv;
                    damaged |= over ? 1u : 0u;
                    // DC and AC symbols take the same few instructions: a DC symbol has z = 0 and run = 0, so its place in the
                    // block is natural[0] = 0 by itself; its value is the running sum of its component, and it makes the block
                    // this lane's own
                    const uint32_t ci = (comp_of >> (2 * c)) & 3;
                    const int dsum = ci == 0 ? dcs0 : ci == 1 ? dcs1 : dcs2;
                    st |= isdc ? (OWN | ((st & BLKOK) ? DIRTY : 0u)) : 0u;
                    const bool store = (st & BLKOK) && (isdc || (size && !over && (st & OWN)));
                    if (store) stage[K.natural[z + run]] = (int16_t)(v + (isdc ? dsum : 0));
                    ndc += isdc ? 1u : 0u;
                    dcs0 += (isdc && ci == 0) ? v : 0;
                    dcs1 += (isdc && ci == 1) ? v : 0;
                    dcs2 += (isdc && ci == 2) ? v : 0;
                    z += adv;
                    n += counted ? adv : 0u;
                    st |= z >= 64 ? PEND : 0u;
                }
                // more to do?  (a block that began here is decoded to its end, past the chunk's)
                const bool more = fl == 0 && n < max_slots && (p < limit || (st & (OWN | PEND)) == OWN);
                st = more ? st : st & ~GO;
            }
        }
        // the lanes whose block has ended: hand the block over, move on to the next block of the scan
        bool ready = false;
        uint32_t ready_at = 0;
        if (st & PEND) {
            ready = (st & DIRTY) != 0;
            ready_at = blk;
            z = 0;
            c++;
            if (c == bpm) { mx++; if (mx == (uint32_t)F.mcux) { mx = 0; my++; } }
            c = c == bpm ? 0 : c;
            st = (st & ~(PEND | DIRTY | BLKOK)) | (my < (uint32_t)F.mcuy ? BLKOK : 0u);
            blk = K.blk_base[c] + mx * K.blk_dx[c] + my * K.blk_dy[c];
        }
        jpeg_flush_blocks(W, ready, ready_at);
    }
    jpeg_flush_blocks(W, (st & DIRTY) != 0, blk);                   // (a block cut short by an ending: what there is of it)
    if (damaged || (fl & JPEG_FL_INVALID)) jpeg_flag(W->status, JPEG_ST_BAD_CODE);
    if (ok) r.exit = jpeg_pack_state(p, c, z, fl);
    r.n = n;
    r.ndc = ndc;
    r.dc[0] = dcs0;
    r.dc[1] = dcs1;
    r.dc[2] = dcs2;
    return r;
}
#endif

// The write walk (jpeg_write_chunk with dc0 = 0) leaves every DC term relative to its chunk's entry; this adds the
// predictors at the entry (`base`, per component) to the `ndc` blocks that begin in the chunk.  Same block walk as the
// decoder's: MCU coordinates carried along, a block's address is base + mx*dx + my*dy.
IMP_HD inline void jpeg_dc_fixup(const JpegBlockTabs& K, const JpegFrame& F, int16_t* coef, uint32_t slot0, uint64_t entry, uint32_t ndc, const int base[3]) {
    const uint32_t z = (uint32_t)(entry >> 40) & 0xff;
    const uint32_t bpm = (uint32_t)F.bpm, nluma = bpm == 1 ? 1u : bpm - 2;
    const uint32_t gb = (slot0 >> 6) + (z ? 1u : 0u);               // a chunk entered in mid-block begins with the next block's DC term
    const uint32_t mcu = gb / bpm;
    uint32_t c = gb - mcu * bpm, my = mcu / (uint32_t)F.mcux, mx = mcu - my * (uint32_t)F.mcux;
    for (uint32_t j = 0; j < ndc; j++) {
        const int add = c < nluma ? base[0] : (c - nluma == 0 ? base[1] : base[2]);
        if (my < (uint32_t)F.mcuy && add) {
            const uint32_t at = K.blk_base[c] + mx * K.blk_dx[c] + my * K.blk_dy[c];
            coef[at] = (int16_t)(coef[at] + add);
        }
        c++;
        if (c == bpm) { c = 0; mx++; if (mx == (uint32_t)F.mcux) { mx = 0; my++; } }
    }
}

// Round 4: the walk of the phase-parallel scheme (k_jpeg_walks, and the repair / chase walks of k_jpeg_mend / k_jpeg_select).  It starts `overlap` bits BEFORE the chunk it belongs to, in
// a guessed state (that bit, start of block k of the MCU), and reports two states: `in`, the first state it reaches at or
// behind bit `cross` (the chunk's first bit), and `out`, the first at or behind `limit` -- plus the coefficient slots passed
// by the symbols that start in [cross, limit).  A state is everything the decoder's future depends on, so whenever a
// predecessor's `out` equals a walk's `in` the walk's `out` IS what decoding on from the predecessor's state gives: the
// successor is SELECTED, not decoded again.  With one walk per block of the MCU (six for 4:2:0) the true state is almost
// always among the `in`s once the overlap holds a block end or two -- bit position and coefficient index fall into step by
// themselves, and every block phase is being tried.  
// Round 5, second half: the walk also reports `mid`, the first state at or behind bit `half` (cross <= half <= limit; the
// chunk's middle), and the slots passed up to there: k_jpeg_write then decodes the chunk with TWO lanes, the second entering
// at `mid` -- its chain half as long, the synchronising phases none the longer (the walk passes that bit anyway).
struct JpegSpan {
    uint64_t in, out, mid;
    uint32_t n, nmid;
};
template <class Tabs, class WordFn>
IMP_HD inline JpegSpan jpeg_span_walk(const Tabs& L, WordFn word, uint64_t entry, uint32_t cross, uint32_t half, uint32_t limit, uint32_t seg_end, const JpegFrame& F) {
    JpegSpan r;
    r.in = r.out = r.mid = entry;
    r.n = r.nmid = 0;
    uint32_t p = (uint32_t)entry;
    uint32_t c = (uint32_t)(entry >> 32) & 0xff, z = (uint32_t)(entry >> 40) & 0xff;
    if ((uint32_t)(entry >> 48)) return r;                          // a dead state stays what it is
    (void)seg_end;
    const uint32_t bpm = (uint32_t)F.bpm, nluma = bpm == 1 ? 1u : bpm - 2;
    // per block of the MCU, four bits: the table its DC symbol is read with (0 / 1) and, two bits up, the table of its AC
    // symbols (2 / 3) -- one shift and one mask per symbol pick the table; and, four bits each again, the block that follows
    uint32_t tabsel = 0, nextc = 0;
    for (uint32_t k = 0; k < 6; k++) {
        const uint32_t ci = k < nluma ? 0u : (k - nluma + 1 < 3 ? k - nluma + 1 : 2u);
        const uint32_t dct = (uint32_t)(ci == 0 ? F.dctab[0] : ci == 1 ? F.dctab[1] : F.dctab[2]);
        const uint32_t act = 2u + (uint32_t)(ci == 0 ? F.actab[0] : ci == 1 ? F.actab[1] : F.actab[2]);
        tabsel |= (dct | (act << 2)) << (4 * k);
        nextc |= (k + 1 == bpm ? 0u : k + 1) << (4 * k);
    }
    JpegBitReader1<WordFn> bits(word);
    bits.start(p);
    uint32_t fl = 0, blocks = 0, z0 = z, passed = 0;
    // No look at the interval's end here (the write walk does that): only an interval's LAST chunk could meet the padding,
    // and its exit state and slot count are never used -- the chunk behind it starts a new interval.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int stage = 0; stage < 3; stage++) {
        uint32_t target = fl ? 0u : (stage == 0 ? cross : stage == 1 ? half : limit);
        blocks = 0;
        z0 = z;
        while (p < target) {                                        // (an ending pulls `target` down to zero: one way out)
            const uint32_t win = bits.window();
            const uint32_t tab = (tabsel >> (4 * c + (z == 0 ? 0u : 2u))) & 3;
            uint32_t e = jpeg_tab_first(L, tab, win >> (32 - JPEG_LOOKBITS));
            if ((e >> 16) == 0) {                                   // a code longer than the table's index
                bool none;
                if (e & 0x8000u) {                                  // ... in the second level, indexed by the bits that follow
                    const uint32_t nb = (e >> 12) & 7, off = ((e >> 5) & 127) << 1;
                    e = jpeg_tab_second(L, tab, off + ((win >> (32 - JPEG_LOOKBITS - nb)) & ((1u << nb) - 1)));
                    none = e == 0;
                } else {                                            // ... or (a table with too many of them) by the canonical limits
                    const uint32_t peek = win >> 16;
                    uint32_t len = JPEG_LOOKBITS + 1;
                    for (int l = JPEG_LOOKBITS + 1; l < 16; l++) len += peek >= jpeg_tab_limit(L, tab, l) ? 1u : 0u;
                    none = peek >= jpeg_tab_limit(L, tab, 16);                // no code starts with these 16 bits
                    const uint32_t sym = jpeg_tab_val(L, tab, (uint32_t)(jpeg_tab_offs(L, tab, len) + (int)(peek >> (16 - len))) & 255);
                    e = none ? 0u : jpeg_lut_expand(jpeg_lut_entry(len, sym, z == 0));
                }
                fl = none ? JPEG_FL_INVALID : fl;
                target = none ? 0u : target;
            }
            const uint32_t total = (e >> 16) & 31, adv = e >> 21;
            p += total;
            bits.take(total);
            z += adv;
            const bool ended = z >= 64;
            const uint32_t cn = (nextc >> (4 * c)) & 15;
            blocks += ended ? 1u : 0u;
            z = ended ? 0u : z;
            c = ended ? cn : c;
        }
        if (stage == 0) r.in = jpeg_pack_state(p, c, z, fl);
        else passed += 64u * blocks + z - z0;                       // (every block passed counts 64, whatever its last run claimed)
        if (stage == 1) { r.mid = jpeg_pack_state(p, c, z, fl); r.nmid = passed; }
    }
    r.out = jpeg_pack_state(p, c, z, fl);
    r.n = passed;
    return r;
}

// ---- maps: which of a chunk's six exit candidates follows from each of its predecessor's six (a nibble each; FAIL = not
// known).  Composing them along the chain is the scan that replaces the rounds of re-decoding.
constexpr uint32_t JPEG_MAP_FAIL = 15u;
constexpr uint64_t JPEG_STATE_NONE = ~0ull;                         // "no candidate here" (its flag bits are set: a dead state)
IMP_HD inline uint32_t jpeg_map_at(uint32_t m, uint32_t k) { return (m >> (4 * k)) & 15u; }
IMP_HD inline uint32_t jpeg_map_const(uint32_t k) { return k * 0x111111u; }
IMP_HD inline uint32_t jpeg_map_then(uint32_t first, uint32_t second) {      // `first`, then `second`
    uint32_t r = 0;
    for (uint32_t k = 0; k < 6; k++) {
        const uint32_t v = jpeg_map_at(first, k);
        const uint32_t w = (second >> (4 * (v & 7))) & 15u;         // (v & 7: a FAIL must not shift by 60)
        r |= (v == JPEG_MAP_FAIL ? JPEG_MAP_FAIL : w) << (4 * k);
    }
    return r;
}
// every input leads to the same known output: what comes before the chunk no longer matters
IMP_HD inline bool jpeg_map_is_const(uint32_t m) { return (m & 0xffffffu) == jpeg_map_const(m & 15u) && (m & 15u) != JPEG_MAP_FAIL; }

}  // namespace imp
