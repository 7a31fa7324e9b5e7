// imp_jpeg.hip -- device side of the JPEG front (imp_jpeg.h): what libjpeg does under cvDecodeImage at bridge.c:545-552.
//
//   k_jpeg_entropy   Huffman decoding of a whole scan in ONE launch.  The unstuffed stream is cut into 1024-bit chunks, one
//                    per lane.  Only the first chunk of a restart interval starts at a known decoder state; every other lane
//                    starts at its chunk's first bit in a guessed state and relies on the self-synchronisation of Huffman
//                    codes: lane t takes over the state lane t-1 leaves behind and re-decodes its chunk until nothing in the
//                    workgroup changes (a fixed point: every chunk's entry state is its predecessor's exit state, and the
//                    first chunk of an interval is exact, so by induction all are).  Workgroups take tickets, so a
//                    workgroup may wait for its predecessor's exit state and running totals (coefficient slots, DC sums)
//                    -- a chained scan.  With the totals known every lane decodes its chunk once more and scatters the
//                    non-zero coefficients into the (zeroed) planes, DC terms already integrated.
//   k_jpeg_pixels    dequantisation, libjpeg's ISLOW 8x8 IDCT (jidctint.c), fancy chroma upsampling (jdsample.c) and
//                    YCbCr -> B,G,R (jdcolor.c) for a 256x64 pixel tile per workgroup; the planes live only in LDS.
// Integer arithmetic throughout; tests/test_gpu_jpeg.py compares with the Pillow-pinned oracle bit for bit.
#include "imp_jpeg_core.h"

namespace imp {

namespace {

constexpr int HB = JPEG_HUFF_BLOCK;

constexpr int CTL_REC = JPEG_CTL_REC;

__constant__ uint8_t c_natural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                      30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

__device__ __forceinline__ void st_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// Wait for a flag another workgroup raises.  Workgroups are numbered by ticket, so the one waited for is already running
// (or done); the bound only keeps a bug from hanging the device.  The poll is a RELAXED load: an acquire load invalidates
// the XCD's whole L2 every time round, and hundreds of waiting waves doing that every few hundred cycles took every
// kernel on the device down with them (a request stream got slower with every thread added).  One acquire fence after
// the flag has been seen orders the payload reads behind it.
#ifndef JPEG_POLL_SLEEP
#define JPEG_POLL_SLEEP 2      // x 64 cycles between two looks at a flag of the chain (16: 4K lone file 13 % slower; A/B with -DJPEG_POLL_SLEEP=)
#endif
__device__ bool wait_flag(const uint32_t* flag) {
    for (int spin = 0; spin < (1 << 21); spin++) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return true;
        }
        __builtin_amdgcn_s_sleep(JPEG_POLL_SLEEP);
    }
    return false;
}

__global__ __launch_bounds__(HB) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_jpeg_entropy(const JpegJob* __restrict__ jobs, const JpegMapEntry* __restrict__ block_map,
                                                      uint32_t* __restrict__ launch_ticket) {
    __shared__ JpegHuffTabs L;
    __shared__ uint64_t s_exit[HB], s_entry[HB];
    __shared__ uint32_t s_segend[HB];
    __shared__ uint8_t s_queue[HB];
    __shared__ uint32_t s_queued[2];            // the round's queue length; two, so that one barrier less per round is needed
    __shared__ uint32_t s_n[HB];
    __shared__ int s_dc[3][HB];
    __shared__ uint8_t s_head[HB];              // 1 = an interval starts in or before this chunk (inside the workgroup)
    __shared__ uint32_t s_ticket;
    __shared__ uint64_t s_pred;
    __shared__ uint32_t s_carry[4];
    const int t = threadIdx.x;
    // workgroups are numbered in the order they START (a ticket), never by blockIdx: the one a workgroup waits for is then
    // always running already
    if (t == 0) s_ticket = atomicAdd(launch_ticket, 1u);
    __syncthreads();
    const JpegMapEntry me = block_map[__builtin_amdgcn_readfirstlane(s_ticket)];
    const JpegJob& J = jobs[__builtin_amdgcn_readfirstlane(me.job)];
    const JpegFrame& F = J.F;           // (a reference: scalar loads from the table; a copy of the struct would live in scratch memory)
    struct { const uint32_t *words, *chunk_seg, *seg_first_chunk, *seg_bits; const JpegHuffDev* tables; int16_t* coef; uint32_t *header, *records; } A =
        {J.words, J.chunk_seg, J.seg_first_chunk, J.seg_bits, J.tables, J.coef, J.header, J.records};
    // tables
    for (int i = t; i < 4 * (1 << JPEG_LOOKBITS); i += HB) L.lut[i >> JPEG_LOOKBITS][i & ((1 << JPEG_LOOKBITS) - 1)] = jpeg_lut_expand(A.tables[i >> JPEG_LOOKBITS].lut[i & ((1 << JPEG_LOOKBITS) - 1)]);
    for (int i = t; i < 4 * 18; i += HB) { L.limit[i / 18][i % 18] = A.tables[i / 18].limit[i % 18]; L.offs[i / 18][i % 18] = A.tables[i / 18].offs[i % 18]; }
    for (int i = t; i < 4 * 256; i += HB) L.vals[i >> 8][i & 255] = A.tables[i >> 8].vals[i & 255];
    if (t < 64) L.natural[t] = c_natural[t];
    if (t < F.bpm) jpeg_block_steps(F, t, &L.blk_base[t], &L.blk_dx[t], &L.blk_dy[t]);
    const uint32_t CHUNK_BITS = F.chunk_bits;                       // 1024, 512 or 256 (jpeg_chunk_bytes_for)
    const uint32_t b = __builtin_amdgcn_readfirstlane(me.local);
    const uint32_t nblocks = (F.nchunks + HB - 1) / HB;
    const uint32_t g0 = b * HB, g = g0 + (uint32_t)t;
    const bool live = g < F.nchunks;
    uint32_t seg = 0, first = 0, seg_end = 0, limit = 0;
    bool origin = false;
    if (live) {
        seg = A.chunk_seg[g];
        first = A.seg_first_chunk[seg];
        origin = first == g;
        seg_end = first * CHUNK_BITS + A.seg_bits[seg];
        limit = min((g + 1) * CHUNK_BITS, seg_end);
    }
    // The stream is read straight from memory: a lane walks its own 128-byte line, and the decoder keeps the next word a
    // whole refill ahead in a register, so the latency is off the critical path; staging the workgroup's 33 KB in LDS bought
    // nothing and cost two of every three resident workgroups.  (bit 31 of a word = the first bit of the stream: byte swap)
    const uint32_t* __restrict__ words = A.words;
    // (a GLOBAL load: the pointer comes out of a table in memory, which makes it a flat one to the compiler, and a flat load
    // counts as an LDS access too -- every wait for a table read would then wait for the stream word as well)
    typedef const uint32_t __attribute__((address_space(1))) * GlobalWords;
    const GlobalWords gwords = (GlobalWords)(uintptr_t)words;
    auto word = [&](uint32_t i) -> uint32_t { return __builtin_nontemporal_load(gwords + i); };
    uint32_t* rec = A.records + (size_t)b * CTL_REC;
    const uint32_t* prec = rec - CTL_REC;
    __syncthreads();
    auto stamp = [&](int k) { if (t == 0) rec[12 + k] = (uint32_t)wall_clock64(); };     // 100 MHz
    stamp(0);

    // ---- 1./2. Every lane decodes its chunk from a guessed state (its first bit, start of a block) -- exact only for the
    // first chunk of an interval -- and then pulls its predecessor's exit state, re-decoding whenever that differs from the
    // state it started from, until nothing changes in the workgroup.  Lane 0's predecessor is the previous workgroup's last
    // lane.  A workgroup publishes its last exit state twice: TENTATIVELY as soon as its own lanes agree (Huffman codes
    // resynchronise within a few chunks, so that state is almost always the true one already, whatever the workgroup's own
    // entry state turns out to be), which lets all workgroups run their fix-up at the same time instead of one after the
    // other, and FINALLY once its own entry state is final.  The chain of final states is then a flag and a compare per
    // workgroup; only a workgroup whose tentative input was wrong converges once more.
    // The tentative state is not published once but KEPT UP TO DATE: the last lane stores its exit state (one 64-bit word,
    // 0 = nothing yet) after every round that changed it, and lane 0 of the next workgroup looks at it at the top of every
    // round -- so the correction a workgroup's first chunks need from its predecessor happens during the rounds its slowest
    // chunks need anyway, not in a phase of its own after them.
    uint64_t* const tent = (uint64_t*)(rec + 0);
    const uint64_t* const ptent = (const uint64_t*)(prec + 0);
    uint64_t published = 0;                                         // (lane HB - 1)
    const bool chained = origin || !live;                           // (read by lane 0 only) nothing to wait for
    const uint64_t guess = jpeg_pack_state(g * CHUNK_BITS, 0, 0, 0);
    const uint64_t none = ~0ull;                                    // "no state": its flag bits are set
    uint64_t entry = none;
    s_exit[t] = none;
    s_entry[t] = none;
    s_segend[t] = seg_end;
    s_n[t] = 0;
    s_dc[0][t] = s_dc[1][t] = s_dc[2][t] = 0;
    if (t == 0) { s_pred = none; s_queued[0] = s_queued[1] = 0; }
    __syncthreads();
    for (int phase = 0; phase < 3; phase++) {
        stamp(1 + 2 * phase);                                       // 1, 3, 5: the phase's rounds start (after the wait: 2, 4 below)
        if (phase == 1 && t == HB - 1 && b + 1 < nblocks && !published) {
            // its rounds are over and the last lane never had a state worth handing on (an undecodable pattern): say so, the
            // next workgroup then starts from its own guess instead of waiting
            published = s_exit[t];
            __hip_atomic_store(tent, published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (phase == 1 && t == 0 && !chained) {                     // a predecessor that has not said anything yet: wait for its first word
            uint64_t v = 0;
            for (int spin = 0; spin < (1 << 21) && !v; spin++) {
                v = __hip_atomic_load(ptent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!v) __builtin_amdgcn_s_sleep(16);
            }
            if (v) s_pred = v;
            else atomicOr(&A.header[1], JPEG_ST_CHAIN_TIMEOUT);
        }
        if (phase == 2 && t == 0 && !chained) {                     // the predecessor's FINAL state
            if (wait_flag(prec + 3)) s_pred = (uint64_t)prec[4] | ((uint64_t)prec[5] << 32);
            else atomicOr(&A.header[1], JPEG_ST_CHAIN_TIMEOUT);
        }
        if (phase > 0) stamp(2 * phase + 2);                        // 4, 6: the predecessor's state has arrived
        for (int round = 0; round <= HB + 1; round++) {
            if (phase < 2 && t == 0 && !chained) {
                const uint64_t v = __hip_atomic_load(ptent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v) s_pred = v;
            }
            // (every exit state of the round before is in place: its closing barrier -- or the one in front of the phases --
            // has been passed; two barriers per round, not three: at 256-bit chunks a round is short enough for that to show)
            const int q = round & 1;
            uint64_t want = t > 0 ? s_exit[t - 1] : s_pred;
            // a predecessor with nothing to hand on (none yet, or it ran into an undecodable pattern -- a wrong guess,
            // normally): the lane's own guess.  Otherwise that dead state would travel a chunk per round to the interval's end.
            if (origin || (want >> 48)) want = guess;
            // The chunks whose entry state changed are queued and the queue is worked off by the first lanes of the
            // workgroup: after the first round or two only a few chunks per workgroup still move (the ones that take long to
            // synchronise), and they should keep one wave busy, not four.
            if (live && want != entry) {
                entry = want;
                s_entry[t] = want;
                s_queue[atomicAdd(&s_queued[q], 1u)] = (uint8_t)t;
            }
            if (t == 0) s_queued[q ^ 1] = 0;                        // (last read before the barrier that closed the round before)
            __syncthreads();
            const uint32_t queued = s_queued[q];
            if (queued == 0) { if (t == 0) atomicMax(&A.header[phase == 0 ? 2 : 3], (uint32_t)round); break; }
            if ((uint32_t)t < queued) {
                const uint32_t k = s_queue[t];
                s_exit[k] = jpeg_sync_chunk(L, word, s_entry[k], min((g0 + k + 1) * CHUNK_BITS, s_segend[k]), s_segend[k], F);
            }
            __syncthreads();
            if (t == HB - 1 && b + 1 < nblocks) {
                const uint64_t mine = s_exit[t];
                if (mine != published && !(mine >> 48)) {
                    __hip_atomic_store(tent, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    published = mine;
                }
            }
        }
    }
    __syncthreads();
    if (t == HB - 1 && b + 1 < nblocks) {                           // the final exit state
        rec[4] = (uint32_t)s_exit[t];
        rec[5] = (uint32_t)(s_exit[t] >> 32);
        st_release(rec + 3, 1u);
    }
    stamp(7);
    // ---- 3. every entry state is final: one full walk per chunk for what the rounds above left out -- the slots it passes
    // and its DC differences
    if (live) {
        const JpegDecoded cnt = jpeg_decode_chunk<false>(L, word, entry, limit, seg_end, F, nullptr);
        s_n[t] = cnt.n;
        s_dc[0][t] = cnt.dc[0];
        s_dc[1][t] = cnt.dc[1];
        s_dc[2][t] = cnt.dc[2];
    }
    __syncthreads();
    stamp(8);
    // ---- 4. running totals inside each interval: segmented inclusive scan over (n, dc0, dc1, dc2)
    struct { uint32_t n; int dc[3]; } d = {s_n[t], {s_dc[0][t], s_dc[1][t], s_dc[2][t]}};     // this chunk's own
    s_head[t] = origin ? 1 : 0;
    __syncthreads();
    for (int ofs = 1; ofs < HB; ofs <<= 1) {
        uint32_t n2 = 0, h2 = 0;
        int a0 = 0, a1 = 0, a2 = 0;
        const bool take = t >= ofs;
        if (take) { n2 = s_n[t - ofs]; a0 = s_dc[0][t - ofs]; a1 = s_dc[1][t - ofs]; a2 = s_dc[2][t - ofs]; h2 = s_head[t - ofs]; }
        const uint32_t myh = s_head[t];
        __syncthreads();
        if (take) {
            if (!myh) { s_n[t] += n2; s_dc[0][t] += a0; s_dc[1][t] += a1; s_dc[2][t] += a2; }
            s_head[t] = (uint8_t)(myh | h2);
        }
        __syncthreads();
    }
    // the previous workgroup's totals for the interval that runs into this one
    if (t == 0) {
        s_carry[0] = s_carry[1] = s_carry[2] = s_carry[3] = 0;
        if (!chained) {
            if (wait_flag(prec + 6)) { s_carry[0] = prec[7]; s_carry[1] = prec[8]; s_carry[2] = prec[9]; s_carry[3] = prec[10]; }
            else atomicOr(&A.header[1], JPEG_ST_CHAIN_TIMEOUT);
        }
    }
    __syncthreads();
    const bool open = !s_head[t];                                   // still in the interval that began in an earlier workgroup
    const uint32_t incl_n = s_n[t] + (open ? s_carry[0] : 0u);
    const int incl_dc0 = s_dc[0][t] + (open ? (int)s_carry[1] : 0), incl_dc1 = s_dc[1][t] + (open ? (int)s_carry[2] : 0),
              incl_dc2 = s_dc[2][t] + (open ? (int)s_carry[3] : 0);
    if (t == HB - 1 && b + 1 < nblocks) {
        rec[7] = incl_n; rec[8] = (uint32_t)incl_dc0; rec[9] = (uint32_t)incl_dc1; rec[10] = (uint32_t)incl_dc2;
        st_release(rec + 6, 1u);
    }
    stamp(9);
    if (!live) return;
    // ---- 5. decode once more, now knowing where every coefficient goes.  The last chunk of an interval walks with the
    // interval's remaining slots as a budget -- a sequential decoder stops after the last MCU and never looks at the padding
    // bits, which in a damaged file need not be the 1-bits an encoder writes -- and gives the verdict: it must end inside
    // the padding with exactly the interval's slots decoded.
    const uint32_t last_chunk_of_seg = (seg + 1 < F.nsegs ? A.seg_first_chunk[seg + 1] : F.nchunks) - 1;
    const uint32_t base_n = incl_n - d.n;
    const uint32_t slots_here = min((uint32_t)F.slots_per_seg, F.total_slots - seg * (uint32_t)F.slots_per_seg);
    const bool closes = g == last_chunk_of_seg;
    JpegWriteCtx W;
    W.coef = A.coef;
    W.slot0 = seg * (uint32_t)F.slots_per_seg + base_n;
    W.dc0[0] = incl_dc0 - d.dc[0];
    W.dc0[1] = incl_dc1 - d.dc[1];
    W.dc0[2] = incl_dc2 - d.dc[2];
    W.status = &A.header[1];
    if (!closes && incl_n > slots_here) { atomicOr(&A.header[1], JPEG_ST_OVERRUN); return; }   // would write outside the interval
    const uint32_t budget = closes ? (slots_here >= base_n ? slots_here - base_n : 0u) : 0xffffffffu;
    const JpegDecoded e = jpeg_decode_chunk<true>(L, word, entry, limit, seg_end, F, &W, budget);
    if (closes) {
        const uint32_t pe = (uint32_t)e.exit, fle = (uint32_t)(e.exit >> 48);
        if ((fle & JPEG_FL_INVALID) || pe > seg_end || seg_end - pe >= 8) atomicOr(&A.header[1], JPEG_ST_BAD_CODE);
        if (base_n + e.n != slots_here) atomicOr(&A.header[1], JPEG_ST_BAD_COUNT);
    }
    stamp(10);
}

// ---------------------------------------------------------------- pixels
constexpr int TILE_W = JPEG_TILE_W, TILE_H = JPEG_TILE_H;
constexpr int YP = TILE_W;                       // luma LDS pitch


// jidctint.c jpeg_idct_islow, one dimension: CONST_BITS 13; the caller picks the descale shift of its pass
__device__ __forceinline__ void idct8(const int in[8], int out[8], const int shift) {
    constexpr int C0_298 = 2446, C0_390 = 3196, C0_541 = 4433, C0_765 = 6270, C0_899 = 7373, C1_175 = 9633, C1_501 = 12299,
                  C1_847 = 15137, C1_961 = 16069, C2_053 = 16819, C2_562 = 20995, C3_072 = 25172;
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * C0_541;
    int tmp2 = z1 + z3 * (-C1_847);
    int tmp3 = z1 + z2 * C0_765;
    z2 = in[0];
    z3 = in[4];
    int tmp0 = (int)((unsigned)(z2 + z3) << 13);
    int tmp1 = (int)((unsigned)(z2 - z3) << 13);
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * C1_175;
    tmp0 *= C0_298;
    tmp1 *= C2_053;
    tmp2 *= C3_072;
    tmp3 *= C1_501;
    z1 *= -C0_899;
    z2 *= -C2_562;
    z3 *= -C1_961;
    z4 *= -C0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    const int half = 1 << (shift - 1);
    out[0] = (tmp10 + tmp3 + half) >> shift;
    out[7] = (tmp10 - tmp3 + half) >> shift;
    out[1] = (tmp11 + tmp2 + half) >> shift;
    out[6] = (tmp11 - tmp2 + half) >> shift;
    out[2] = (tmp12 + tmp1 + half) >> shift;
    out[5] = (tmp12 - tmp1 + half) >> shift;
    out[3] = (tmp13 + tmp0 + half) >> shift;
    out[4] = (tmp13 - tmp0 + half) >> shift;
}

__device__ __forceinline__ uint32_t sat_u8(int v) { return (uint32_t)min(max(v, 0), 255); }

// one block: 64 coefficients at `blk` (natural order) -> 8 rows of 8 samples at out[0..7][0..7] in LDS
__device__ void idct_block_to_lds(const int16_t* blk, const uint16_t* q, uint8_t* out, int pitch) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    int ws[64];
    v4i raw[8];
#pragma unroll
    for (int k = 0; k < 8; k++) raw[k] = *(const v4i*)(blk + 8 * k);     // a row: eight shorts
#pragma unroll
    for (int x = 0; x < 8; x++) {
        int in[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int pair = raw[k][x >> 1];
            const int cv = (x & 1) ? (pair >> 16) : (int)(short)(pair & 0xffff);
            in[k] = cv * (int)q[8 * k + x];
        }
        idct8(in, o, 13 - 2);
#pragma unroll
        for (int k = 0; k < 8; k++) ws[8 * k + x] = o[k];
    }
#pragma unroll
    for (int y = 0; y < 8; y++) {
        int o[8];
        idct8(ws + 8 * y, o, 13 + 2 + 3);
        const uint32_t lo = sat_u8(o[0] + 128) | (sat_u8(o[1] + 128) << 8) | (sat_u8(o[2] + 128) << 16) | (sat_u8(o[3] + 128) << 24);
        const uint32_t hi = sat_u8(o[4] + 128) | (sat_u8(o[5] + 128) << 8) | (sat_u8(o[6] + 128) << 16) | (sat_u8(o[7] + 128) << 24);
        *(uint2*)(out + y * pitch) = make_uint2(lo, hi);
    }
}

template <int HS, int VS, int NC>
__global__ __launch_bounds__(256) void k_jpeg_pixels(const JpegJob* __restrict__ jobs, const JpegMapEntry* __restrict__ tile_map) {
    constexpr int CBW = TILE_W / 8 / HS + (HS == 2 ? 2 : 0);        // chroma blocks per tile row, halo included
    constexpr int CBH = TILE_H / 8 / VS + (VS == 2 ? 2 : 0);
    constexpr int CP = CBW * 8;                                     // chroma LDS pitch
    __shared__ __attribute__((aligned(16))) uint8_t s_y[TILE_H * YP];
    __shared__ __attribute__((aligned(16))) uint8_t s_c[NC == 3 ? 2 : 1][NC == 3 ? CP * CBH * 8 : 16];
    __shared__ uint16_t s_q[3][64];
    const int t = threadIdx.x;
    const JpegMapEntry me = tile_map[blockIdx.x];
    const JpegJob& J = jobs[__builtin_amdgcn_readfirstlane(me.job)];
    const JpegFrame& F = J.F;
    const int16_t* __restrict__ coef = J.coef;
    uint8_t* __restrict__ dst = J.dst;
    const int dstep = J.dstep;
    if (t < 64 * NC) s_q[t >> 6][t & 63] = J.qt[t];
    __syncthreads();
    const int tiles_x = (F.width + TILE_W - 1) / TILE_W;
    const int tile_y = (int)__builtin_amdgcn_readfirstlane(me.local) / tiles_x, tile_x = (int)__builtin_amdgcn_readfirstlane(me.local) - tile_y * tiles_x;
    {   // luma: one block per lane
        const int bx = tile_x * (TILE_W / 8) + (t & 31), by = tile_y * (TILE_H / 8) + (t >> 5);
        if (bx < F.bw[0] && by < F.bh[0])
            idct_block_to_lds(coef + F.coef_off[0] + ((size_t)by * F.bw[0] + bx) * 64, s_q[0], s_y + (t >> 5) * 8 * YP + (t & 31) * 8, YP);
    }
    const int cbx0 = tile_x * (TILE_W / 8 / HS) - (HS == 2 ? 1 : 0), cby0 = tile_y * (TILE_H / 8 / VS) - (VS == 2 ? 1 : 0);
    if constexpr (NC == 3) {
        for (int i = t; i < 2 * CBW * CBH; i += 256) {
            const int ci = i >= CBW * CBH ? 2 : 1, j = i - (ci - 1) * CBW * CBH;
            const int lx = j % CBW, ly = j / CBW, bx = cbx0 + lx, by = cby0 + ly;
            if (bx >= 0 && by >= 0 && bx < F.bw[ci] && by < F.bh[ci])
                idct_block_to_lds(coef + F.coef_off[ci] + ((size_t)by * F.bw[ci] + bx) * 64, s_q[ci], s_c[ci - 1] + ly * 8 * CP + lx * 8, CP);
        }
    }
    __syncthreads();
    // pixels: a wave takes a tile row at a time, four pixels per lane
    const int lane = t & 63, wv = t >> 6;
    const int X0 = tile_x * TILE_W + lane * 4;
    if (X0 >= F.width) return;
    for (int r = wv; r < TILE_H; r += 4) {
        const int Y = tile_y * TILE_H + r;
        if (Y >= F.height) break;
        const uint32_t y4 = *(const uint32_t*)(s_y + r * YP + lane * 4);
        uint8_t* row = dst + (size_t)Y * dstep;
        if constexpr (NC == 1) {
            if (X0 + 3 < F.width) *(uint32_t*)(row + X0) = y4;
            else for (int k = 0; X0 + k < F.width; k++) row[X0 + k] = (uint8_t)(y4 >> (8 * k));
        } else {
            int cbv[4], crv[4];
            const int dsw = F.dsw[1], dsh = F.dsh[1];
            if constexpr (HS == 1 && VS == 1) {
                const uint32_t b4 = *(const uint32_t*)(s_c[0] + r * CP + lane * 4), r4 = *(const uint32_t*)(s_c[1] + r * CP + lane * 4);
#pragma unroll
                for (int k = 0; k < 4; k++) { cbv[k] = (b4 >> (8 * k)) & 255; crv[k] = (r4 >> (8 * k)) & 255; }
            } else {
                // chroma sample coordinates (absolute), then into the tile's halo plane
                const int cy = VS == 2 ? Y >> 1 : Y;
                int ny = cy;
                if constexpr (VS == 2) ny = min(max((Y & 1) ? cy + 1 : cy - 1, 0), dsh - 1);
                const int ly = cy - cby0 * 8, lny = ny - cby0 * 8;
                const bool fancy_h = HS == 2 && dsw > 2;                 // jinit_upsampler: the replicating forms otherwise
#pragma unroll
                for (int comp = 0; comp < 2; comp++) {
                    const uint8_t* P = s_c[comp];
                    int* outv = comp ? crv : cbv;
                    if constexpr (HS == 2) {
                        const int cx = X0 >> 1;
                        if (!fancy_h) {
#pragma unroll
                            for (int k = 0; k < 4; k++) outv[k] = P[ly * CP + (min(cx + (k >> 1), dsw - 1) - cbx0 * 8)];
                        } else {
                            int col[4];                                  // columns cx-1 .. cx+2, clamped like the row loops' ends
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int lx = min(max(cx - 1 + j, 0), dsw - 1) - cbx0 * 8;
                                if constexpr (VS == 2) col[j] = 3 * P[ly * CP + lx] + P[lny * CP + lx];
                                else col[j] = P[ly * CP + lx];
                            }
                            if constexpr (VS == 2) {                     // h2v2_fancy_upsample
                                outv[0] = (3 * col[1] + col[0] + 8) >> 4;
                                outv[1] = (3 * col[1] + col[2] + 7) >> 4;
                                outv[2] = (3 * col[2] + col[1] + 8) >> 4;
                                outv[3] = (3 * col[2] + col[3] + 7) >> 4;
                            } else {                                     // h2v1_fancy_upsample
                                outv[0] = (3 * col[1] + col[0] + 1) >> 2;
                                outv[1] = (3 * col[1] + col[2] + 2) >> 2;
                                outv[2] = (3 * col[2] + col[1] + 1) >> 2;
                                outv[3] = (3 * col[2] + col[3] + 2) >> 2;
                            }
                        }
                    } else {                                             // h1v2_fancy_upsample
                        const int bias = (Y & 1) ? 2 : 1;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int lx = min(X0 + k, dsw - 1) - cbx0 * 8;
                            outv[k] = (3 * P[ly * CP + lx] + P[lny * CP + lx] + bias) >> 2;
                        }
                    }
                }
            }
            uint32_t px[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int yy = (y4 >> (8 * k)) & 255;
                int rr, gg, bb;
                if (F.ycc) {                                             // jdcolor.c build_ycc_rgb_table, SCALEBITS 16
                    const int cb = cbv[k] - 128, cr = crv[k] - 128;
                    rr = yy + ((91881 * cr + 32768) >> 16);
                    bb = yy + ((116130 * cb + 32768) >> 16);
                    gg = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
                } else { rr = yy; gg = cbv[k]; bb = crv[k]; }
                px[k] = sat_u8(bb) | (sat_u8(gg) << 8) | (sat_u8(rr) << 16);
            }
            if (X0 + 3 < F.width) {
                uint32_t* o = (uint32_t*)(row + (size_t)X0 * 3);
                o[0] = px[0] | (px[1] << 24);
                o[1] = (px[1] >> 8) | (px[2] << 16);
                o[2] = (px[2] >> 16) | (px[3] << 8);
            } else {
                for (int k = 0; X0 + k < F.width; k++) {
                    row[(X0 + k) * 3 + 0] = (uint8_t)px[k];
                    row[(X0 + k) * 3 + 1] = (uint8_t)(px[k] >> 8);
                    row[(X0 + k) * 3 + 2] = (uint8_t)(px[k] >> 16);
                }
            }
        }
    }
}

}  // namespace

int launch_jpeg_entropy(const JpegJob* jobs, const JpegMapEntry* block_map, unsigned total_blocks, uint32_t* ticket, hipStream_t s) {
    if (total_blocks == 0) return IMP_OK;
    hipLaunchKernelGGL(k_jpeg_entropy, dim3(total_blocks), dim3(HB), 0, s, jobs, block_map, ticket);
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

int launch_jpeg_pixels(int hs, int vs, int ncomp, const JpegJob* jobs, const JpegMapEntry* tile_map, unsigned total_tiles, hipStream_t s) {
    if (total_tiles == 0) return IMP_OK;
    const dim3 grid(total_tiles), block(256);
    if (ncomp == 1) hipLaunchKernelGGL((k_jpeg_pixels<1, 1, 1>), grid, block, 0, s, jobs, tile_map);
    else if (hs == 1 && vs == 1) hipLaunchKernelGGL((k_jpeg_pixels<1, 1, 3>), grid, block, 0, s, jobs, tile_map);
    else if (hs == 2 && vs == 1) hipLaunchKernelGGL((k_jpeg_pixels<2, 1, 3>), grid, block, 0, s, jobs, tile_map);
    else if (hs == 1 && vs == 2) hipLaunchKernelGGL((k_jpeg_pixels<1, 2, 3>), grid, block, 0, s, jobs, tile_map);
    else if (hs == 2 && vs == 2) hipLaunchKernelGGL((k_jpeg_pixels<2, 2, 3>), grid, block, 0, s, jobs, tile_map);
    else return IMP_ERROR_UNSUPPORTED;
    IMP_HIP(hipGetLastError());
    return IMP_OK;
}

}  // namespace imp
