// imp_jpeg.hip -- device side of the JPEG front (imp_jpeg.h): what libjpeg does under cvDecodeImage at bridge.c:545-552.
//
// Huffman decoding of any number of scans in five launches, none of them a round that is repeated.  The unstuffed stream of a
// file is cut into chunks of 256 ... 2048 bits.  A Huffman stream has no random access, but a decoder started at a wrong bit
// falls into step with the true one after a few symbols; what does NOT fall into step by itself is the block within the MCU
// (luma or chroma tables?), which a guessing decoder only learns by derailing at the next luma/chroma boundary.  So:
//   k_jpeg_walks   a lane per (chunk, block of the MCU): it starts `overlap` bits in FRONT of its chunk as if block k of an MCU
//                  began there, and notes the state it is in at the chunk's first bit ("in") and at its last ("out")
//   k_jpeg_mend    a lane per (chunk, predecessor candidate): the predecessor's exit either equals one of the chunk's `in`
//                  states (then the chunk's MAP sends that candidate there) or no walk arrived in it; those few are decoded on
//                  from ("repair": they almost always join a walk before the chunk ends), and what even that leaves open is
//                  carried forward speculatively for a few chunks as explicit states ("ext records")
//   k_jpeg_select  the true entry state of a chunk is SELECTED, not decoded: the maps compose, so a scan inside the workgroup
//                  and a look-back between workgroups (which take tickets, so the one waited for is always running already)
//                  turn "my predecessor's candidate" into "my candidate"; where a composition is open one lane follows the
//                  explicit state ("chase") until it joins a candidate again.  Round 3 re-decoded in rounds instead, one chunk
//                  of progress per round: 9-14 + 2-6 rounds for a 4:2:0 file.
//   k_jpeg_write   a lane per chunk decodes it once from its true entry state; a block belongs to the chunk it begins in, whose
//                  lane writes all 64 of its coefficients (zeros, then the others), so nothing has to be cleared
//                  beforehand; DC terms are summed up inside the chunk only
//   k_jpeg_dcfix   adds the DC predictors at a chunk's entry (a prefix sum over the interval's chunks) to its blocks
//   k_jpeg_pixels  dequantisation, libjpeg's ISLOW 8x8 IDCT (jidctint.c), fancy chroma upsampling (jdsample.c) and
//                  YCbCr -> B,G,R (jdcolor.c) for a 256x64 pixel tile per workgroup; the planes live only in LDS.
// Integer arithmetic throughout; tests/test_gpu_jpeg.py compares with the Pillow-pinned oracle bit for bit.
#include "imp_jpeg_core.h"

namespace imp {

namespace {

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
__device__ uint32_t wait_flag_ge(const uint32_t* flag, uint32_t want) {
    for (int spin = 0; spin < (1 << 21); spin++) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= want) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return v;
        }
        __builtin_amdgcn_s_sleep(JPEG_POLL_SLEEP);
    }
    return 0;
}

// The chain of k_jpeg_select's workgroups speaks in ONE word per workgroup: flag << 30 | kind << 24 | map -- flag 1 = the map of
// the workgroup's range is there, 2 = its final word too (kind: the true candidate of its last chunk, 0..5; 13 / 15 = a record /
// a state to go on with, in words [4] [5] of the record).  Flag and payload in one atomic word: a reader needs ONE trip to
// memory per workgroup it looks at and no fence (the workgroups sit on eight XCDs with an L2 each; a flag, an acquire fence and
// a second read for the payload were two trips and an invalidate, 5 us a hop in a chain of workgroups that wait for each
// other's final word -- profiles/r05_jpeg_select_chain.txt).  Only kinds 13 and 15 carry more, released and acquired as before.
__device__ __forceinline__ uint32_t chain_word(uint32_t flag, uint32_t kind, uint32_t map) { return (flag << 30) | (kind << 24) | (map & 0xffffffu); }
__device__ __forceinline__ void st_relaxed(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ uint32_t wait_word_ge(const uint32_t* word, uint32_t want) {       // the word once its flag is >= want (0: never came); no fence
    for (int spin = 0; spin < (1 << 21); spin++) {
        const uint32_t v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 30) >= want) return v;
        __builtin_amdgcn_s_sleep(JPEG_POLL_SLEEP);
    }
    return 0;
}

// the Huffman tables of a job into LDS as a decoder lane reads them
__device__ __forceinline__ void load_tables(JpegHuffTabs& L, const JpegHuffDev* tables, int t, int nthreads) {
    for (int i = t; i < 4 * (1 << JPEG_LOOKBITS); i += nthreads) L.lut[i >> JPEG_LOOKBITS][i & ((1 << JPEG_LOOKBITS) - 1)] = jpeg_lut_expand(tables[i >> JPEG_LOOKBITS].lut[i & ((1 << JPEG_LOOKBITS) - 1)]);
    for (int i = t; i < 4 * JPEG_SUB_ENTRIES; i += nthreads) L.sub[i / JPEG_SUB_ENTRIES][i % JPEG_SUB_ENTRIES] = jpeg_lut_expand(tables[i / JPEG_SUB_ENTRIES].sub[i % JPEG_SUB_ENTRIES]);
    for (int i = t; i < 4 * 18; i += nthreads) { L.limit[i / 18][i % 18] = tables[i / 18].limit[i % 18]; L.offs[i / 18][i % 18] = tables[i / 18].offs[i % 18]; }
    for (int i = t; i < 4 * 256; i += nthreads) L.vals[i >> 8][i & 255] = tables[i >> 8].vals[i & 255];
}
__device__ __forceinline__ void load_block_tables(JpegBlockTabs& K, const JpegFrame& F, int t) {
    if (t < 64) K.natural[t] = c_natural[t];
    if (t < F.bpm) jpeg_block_steps(F, t, &K.blk_base[t], &K.blk_dx[t], &K.blk_dy[t]);
}

constexpr int SB = JPEG_SYNC_BLOCK;

// The walks, and nothing else: a lane per (chunk, block of the MCU), results to memory.  No barrier, no other workgroup is
// waited for -- the instruction stream of a table-driven decoder saturates the vector unit as long as nothing else holds the
// workgroup's slots (as ONE kernel with the selection the workgroups of a compute unit went through their latency-bound
// phases together, and the unit idled a third of the time).
// (the kernel's body as a function: k_jpeg_walks is this alone, k_jpeg_entropy_small runs it as the first of five phases)
__device__ __forceinline__ void walks_body(JpegHuffTabs& L, const JpegJob& J, const uint32_t b, const bool publish_tabs) {
    const int t = threadIdx.x;
    const JpegFrame& F = J.F;           // (a reference: scalar loads from the table; a copy of the struct would live in scratch memory)
    load_tables(L, J.tables, t, SB);
    const uint32_t CHUNK_BITS = F.chunk_bits, OVERLAP = F.overlap_bits;
    const uint32_t B = (uint32_t)F.bpm, CPW = (uint32_t)SB / B;     // walks per chunk, chunks per workgroup
    const uint32_t g0 = b * CPW;
    const uint32_t nlive = min(CPW, F.nchunks - g0);
    const uint32_t k = (uint32_t)t / CPW, j = (uint32_t)t - k * CPW;
    const uint32_t g = g0 + j;
    // The stream is read straight from memory: a lane walks its own lines, and the decoder keeps the next word a whole refill
    // ahead in a register.  (bit 31 of a word = the first bit of the stream: byte swap.)  A GLOBAL load: the pointer comes out
    // of a table in memory, which makes it a flat one to the compiler, and a flat load counts as an LDS access too -- every
    // wait for a table read would then wait for the stream word as well.
    typedef const uint32_t __attribute__((address_space(1))) * GlobalWords;
    const GlobalWords gwords = (GlobalWords)(uintptr_t)J.words;
    auto word = [&](uint32_t i) -> uint32_t { return gwords[i]; };      // (a plain load: the lane comes back to the line for its next word, which a non-temporal load does not keep -- 313 against 465 us per 64 files in k_jpeg_walks)
    __syncthreads();
    // (the tables as the lanes read them, for k_jpeg_mend)
    if (publish_tabs && b == 0) for (int i = t; i < (int)(sizeof(JpegHuffTabs) / 4); i += SB) ((uint32_t*)J.tabs)[i] = ((const uint32_t*)&L)[i];
    if (k >= B || j >= nlive) return;
    const uint32_t seg = J.chunk_seg[g], first = J.seg_first_chunk[seg];
    const uint32_t seg_start = first * CHUNK_BITS, seg_end = seg_start + J.seg_bits[seg];
    const uint32_t start = g * CHUNK_BITS, limit = min(start + CHUNK_BITS, seg_end);
    const uint32_t p0 = start - seg_start > OVERLAP ? start - OVERLAP : seg_start;
    const bool exact = p0 == seg_start;                             // the walk starts where the interval does: nothing to guess
    JpegSpan sp;
    sp.in = sp.out = sp.mid = JPEG_STATE_NONE;
    sp.n = sp.nmid = 0;
    if (!exact || k == 0) sp = jpeg_span_walk(L, word, jpeg_pack_state(p0, k, 0, 0), start, min(start + CHUNK_BITS / 2, limit), limit, seg_end, F);
    const size_t at = (size_t)k * F.nchunks + g;                    // [k][chunk]: a wave's stores are neighbours
    J.cand_in[at] = sp.in;
    J.cand_out[at] = sp.out;
    J.cand_n[at] = sp.n;
    if (F.wsplit > 1) { J.cand_mid[at] = sp.mid; J.cand_nmid[at] = sp.nmid; }
}

__global__ __launch_bounds__(SB) void k_jpeg_walks(const JpegJob* __restrict__ jobs, const JpegMapEntry* __restrict__ block_map) {
    __shared__ JpegHuffTabs L;
    const JpegMapEntry me = block_map[blockIdx.x];
    const JpegJob& J = jobs[__builtin_amdgcn_readfirstlane(me.job)];
    walks_body(L, J, __builtin_amdgcn_readfirstlane(me.local), true);
}

// Which walk of a chunk does each exit candidate of the chunk before it lead into?  A lane per (chunk, candidate k): the same
// state as an earlier candidate -> that one's answer ("twin"); one of the chunk's `in` states -> that walk; otherwise the
// lane decodes the chunk once more from the candidate state (a "repair" walk -- one chunk in twenty needs one for some
// candidate) and looks which of the chunk's walks it has joined by the end.  No barrier, nothing waited for: the few lanes
// that walk are latency-bound and alone in their waves, so the kernel is as long as one walk however many there are.
struct MendShared {
    uint32_t count;
    uint16_t items[SB];                                             // the lanes (k * CPW + j) whose candidate needs a repair walk
};
// FUSED = as a phase of k_jpeg_entropy_small: the tables are in L already, and no record of speculative steps is left (the
// chunks ahead belong to workgroups that may not have walked yet: k_jpeg_select's chase decodes on from the state instead)
template <bool FUSED>
__device__ __forceinline__ void mend_body(JpegHuffTabs& L, MendShared& M, const JpegJob& J, const uint32_t b) {
    uint32_t& s_count = M.count;
    uint16_t* s_items = M.items;
    const int t = threadIdx.x;
    const JpegFrame& F = J.F;
    if (!FUSED) for (int i = t; i < (int)(sizeof(JpegHuffTabs) / 4); i += SB) ((uint32_t*)&L)[i] = ((const uint32_t*)J.tabs)[i];
    if (t == 0) s_count = 0;
    const uint32_t CHUNK_BITS = F.chunk_bits, OVERLAP = F.overlap_bits;
    const uint32_t B = (uint32_t)F.bpm, CPW = (uint32_t)SB / B;
    const uint32_t g0 = b * CPW;
    const uint32_t nlive = min(CPW, F.nchunks - g0);
    typedef const uint32_t __attribute__((address_space(1))) * GlobalWords;
    const GlobalWords gwords = (GlobalWords)(uintptr_t)J.words;
    auto word = [&](uint32_t i) -> uint32_t { return gwords[i]; };      // (a plain load: the lane comes back to the line for its next word, which a non-temporal load does not keep -- 313 against 465 us per 64 files in k_jpeg_walks)
    const size_t N = F.nchunks;
    __syncthreads();
    // ---- A. every (chunk, predecessor candidate): answered by a compare, or put on the workgroup's list of walks.  (Round 4
    // walked right here: one candidate in ten needs a walk, so every wave ran the whole walk loop with 6 of its 64 lanes
    // active -- profiles/r04_jpeg_sq_counters.txt -- and the kernel was bound by the instructions all those waves issued.)
    {
        const uint32_t k = (uint32_t)t / CPW, j = (uint32_t)t - k * CPW;
        if (k < B && j < nlive) {
            const uint32_t g = g0 + j;
            const uint32_t seg = J.chunk_seg[g], first = J.seg_first_chunk[seg];
            const uint32_t seg_start = first * CHUNK_BITS, start = g * CHUNK_BITS;
            const size_t at = (size_t)k * N + g;
            uint32_t nib = JPEG_MAP_FAIL | 64u;                     // bit 6: the chunk's one walk started with its interval -- nothing to select
            bool walk = false;
            if (start - seg_start > OVERLAP) {
                nib = JPEG_MAP_FAIL;
                const uint64_t E = J.cand_out[at - 1];
                // (all the states it is compared with are loaded at once: asked for one after the other, each only if the one
                // before did not match, they were up to eleven trips to memory in a row)
                uint64_t po[6], ci[6];
#pragma unroll
                for (uint32_t i = 0; i < 6; i++) {
                    po[i] = i < k ? J.cand_out[(size_t)i * N + g - 1] : JPEG_STATE_NONE;
                    ci[i] = i < B ? J.cand_in[(size_t)i * N + g] : JPEG_STATE_NONE;
                }
                int twin = -1;
#pragma unroll
                for (int i = 5; i >= 0; i--) twin = ((uint32_t)i < k && po[i] == E) ? i : twin;          // (the first that matches)
                if (E == JPEG_STATE_NONE) {
                } else if (twin >= 0) nib = 32u | (uint32_t)twin;   // bit 5: the answer is candidate `twin`'s
                else {
#pragma unroll
                    for (int i = 5; i >= 0; i--) nib = ((uint32_t)i < B && ci[i] == E) ? (uint32_t)i : nib;
                    walk = nib == JPEG_MAP_FAIL;
                }
            }
            if (walk) s_items[atomicAdd(&s_count, 1u)] = (uint16_t)t;
            else J.cand_nib[at] = (uint8_t)nib;
        }
    }
    __syncthreads();
    // ---- B. the walks, packed: lane i takes the i-th item, so the lanes that walk sit side by side in as few waves as it takes
    const uint32_t count = s_count;
    for (uint32_t it = (uint32_t)t; it < count; it += SB) {
        const uint32_t t2 = s_items[it];
        const uint32_t k = t2 / CPW, j = t2 - k * CPW;
        const uint32_t g = g0 + j;
        const uint32_t seg = J.chunk_seg[g], first = J.seg_first_chunk[seg];
        const uint32_t seg_start = first * CHUNK_BITS, seg_end = seg_start + J.seg_bits[seg];
        const uint32_t start = g * CHUNK_BITS, limit = min(start + CHUNK_BITS, seg_end);
        const size_t at = (size_t)k * N + g;
        const uint64_t E = J.cand_out[at - 1];
        uint32_t nib = JPEG_MAP_FAIL;
        uint64_t co[6];                                             // (asked for before the walk, there when it is over)
#pragma unroll
        for (uint32_t i = 0; i < 6; i++) co[i] = i < B ? J.cand_out[(size_t)i * N + g] : JPEG_STATE_NONE;
        const JpegSpan sp = jpeg_span_walk(L, word, E, start, min(start + CHUNK_BITS / 2, limit), limit, seg_end, F);
#pragma unroll
        for (int i = 5; i >= 0; i--) nib = ((uint32_t)i < B && co[i] == sp.out) ? (uint32_t)i : nib;
        J.rep_out[at] = sp.out;
        J.rep_n[at] = sp.n;
        if (F.wsplit > 1) { J.rep_mid[at] = sp.mid; J.rep_nmid[at] = sp.nmid; }
        atomicAdd(&J.header[2], 1u);
        if (nib == JPEG_MAP_FAIL) {
            // It has joined none of the chunk's walks: they are all out of step here, and those of the next chunk tend
            // to be too.  Decode on, chunk by chunk, until the state IS one of a chunk's candidates, and leave what was
            // found on the way -- every chunk's entry state and slot count -- in a record k_jpeg_select can follow
            // without decoding anything (it used to: one lane, a chunk at a time, its successors waiting).
            const uint32_t r = FUSED ? 0xffffffffu : atomicAdd(J.ext_count, 1u);
            uint32_t* R = (!FUSED && r < J.ext_cap) ? J.ext + (size_t)r * JPEG_EXT_WORDS : nullptr;
            J.ext_idx[at] = R ? r : 0xffffffffu;
            if (R) {
                uint64_t S = sp.out;
                uint32_t cnt = 0, joined = 15;
                for (uint32_t m = 0; m < (uint32_t)JPEG_EXT_STEPS; m++) {
                    const uint32_t gc = g + 1 + m;
                    if (gc >= F.nchunks) { joined = 14; break; }
                    const uint32_t seg2 = J.chunk_seg[gc], first2 = J.seg_first_chunk[seg2];
                    const uint32_t sstart2 = first2 * CHUNK_BITS, send2 = sstart2 + J.seg_bits[seg2];
                    const uint32_t start2 = gc * CHUNK_BITS, limit2 = min(start2 + CHUNK_BITS, send2);
                    if (start2 - sstart2 <= OVERLAP) { joined = 14; break; }          // a chunk that selects nothing: the chain ends here
                    uint32_t k1 = JPEG_MAP_FAIL, nn = 0;
                    for (uint32_t i = 0; i < B; i++) if (k1 == JPEG_MAP_FAIL && J.cand_in[(size_t)i * N + gc] == S) { k1 = i; nn = J.cand_n[(size_t)i * N + gc]; }
                    uint64_t out = S;
                    if (k1 == JPEG_MAP_FAIL) {
                        const JpegSpan s2 = jpeg_span_walk(L, word, S, start2, limit2, limit2, send2, F);
                        nn = s2.n;
                        out = s2.out;
                        for (uint32_t i = 0; i < B; i++) if (k1 == JPEG_MAP_FAIL && J.cand_out[(size_t)i * N + gc] == out) k1 = i;
                    }
                    R[4 + 3 * cnt] = (uint32_t)S; R[5 + 3 * cnt] = (uint32_t)(S >> 32); R[6 + 3 * cnt] = nn;
                    cnt++;
                    if (k1 != JPEG_MAP_FAIL) { joined = k1; break; }
                    S = out;
                }
                R[0] = g; R[1] = k; R[2] = cnt; R[3] = joined;
                R[4 + 3 * JPEG_EXT_STEPS] = (uint32_t)S; R[5 + 3 * JPEG_EXT_STEPS] = (uint32_t)(S >> 32);   // (not joined: where it stands)
            }
        }
        J.cand_nib[at] = (uint8_t)(nib | 16u);                      // bit 4: this answer comes from a repair walk
    }
}

__global__ __launch_bounds__(SB) void k_jpeg_mend(const JpegJob* __restrict__ jobs, const JpegMapEntry* __restrict__ block_map) {
    __shared__ JpegHuffTabs L;
    __shared__ MendShared M;
    const JpegMapEntry me = block_map[blockIdx.x];
    const JpegJob& J = jobs[__builtin_amdgcn_readfirstlane(me.job)];
    mend_body<false>(L, M, J, __builtin_amdgcn_readfirstlane(me.local));
}

constexpr int JPEG_SEL_RECS = 8;                                    // (a 1080p photograph leaves one record to two workgroups)
struct SelShared {
    uint64_t in[SB], out[SB], rep_out[SB];                          // [k * CPW + j]: walk k of the workgroup's j-th chunk
    uint64_t pout[6];                                               // the exit candidates of the chunk in front of the workgroup
    uint32_t n[SB], rep_n[SB];
    uint32_t res_n[SB];                                             // [j]
    uint32_t map[SB], scan[2][SB];                                  // [j]: the chunk's map; the scan's two buffers
    uint8_t nib[SB], exact[SB], own[SB];                            // own: the candidate whose answer this one shares (a twin's), else itself
    uint32_t jf, mode, idx0, fin, chases, ext, ei, from;
    uint32_t guessed;                                               // 1 + the candidate of the chunk in front that a CONSTANT map (no final word) named: checked at the end
    uint64_t S;
    uint32_t early;                                                 // the final word went out before the picking (below)
    uint32_t extidx[SB];                                            // [k * CPW + j]: the record k_jpeg_mend left for a repair walk that joined nothing (else none)
    uint32_t recbuf[JPEG_EXT_WORDS];                                // the record being followed, fetched whole
    uint8_t midnone[SB];                                            // [j]: the chunk was reached by a chase (no middle state: one lane of k_jpeg_write decodes all of it)
    uint32_t nrec, recno[JPEG_SEL_RECS];                            // records fetched ahead (phase A), by number
    uint32_t recs[JPEG_SEL_RECS][JPEG_EXT_WORDS];
};
// (workgroup `b` of job J, numbered by TICKET: every workgroup it looks back at has started)
// Lw: the workgroup's tables in LDS where it has them (k_jpeg_entropy_small) -- a chase walks with those; else with the file's tables in memory
__device__ __forceinline__ void select_body(SelShared& Z, const JpegJob& J, const uint32_t b, const JpegHuffTabs* Lw = nullptr) {
    auto& s_in = Z.in; auto& s_out = Z.out; auto& s_rep_out = Z.rep_out; auto& s_pout = Z.pout;
    auto& s_n = Z.n; auto& s_rep_n = Z.rep_n; auto& s_res_n = Z.res_n; auto& s_map = Z.map; auto& s_scan = Z.scan;
    auto& s_nib = Z.nib; auto& s_exact = Z.exact; auto& s_own = Z.own;
    uint32_t &s_jf = Z.jf, &s_mode = Z.mode, &s_idx0 = Z.idx0, &s_fin = Z.fin, &s_chases = Z.chases, &s_ext = Z.ext, &s_ei = Z.ei, &s_from = Z.from;
    uint64_t& s_S = Z.S;
    uint32_t& s_guessed = Z.guessed;
    const int t = threadIdx.x;
    const JpegFrame& F = J.F;
    struct { const uint32_t *words, *chunk_seg, *seg_first_chunk, *seg_bits; const JpegHuffDev* tables; uint32_t *header, *records; uint64_t* chunk_entry; uint32_t* chunk_n; } A =
        {J.words, J.chunk_seg, J.seg_first_chunk, J.seg_bits, J.tables, J.header, J.records, J.chunk_entry, J.chunk_n};
    const uint32_t CHUNK_BITS = F.chunk_bits;
    const uint32_t B = (uint32_t)F.bpm, CPW = (uint32_t)SB / B;     // walks per chunk, chunks per workgroup
    const uint32_t nblocks = (F.nchunks + CPW - 1) / CPW;
    const uint32_t g0 = b * CPW;
    const uint32_t nlive = min(CPW, F.nchunks - g0), jl = nlive - 1;
    const uint32_t k = (uint32_t)t / CPW, j = (uint32_t)t - k * CPW;
    const bool lane_ok = k < B && j < nlive;
    const uint32_t g = g0 + j;
    uint32_t* rec = A.records + (size_t)b * JPEG_CTL_REC;
    const uint32_t* prec = rec - JPEG_CTL_REC;
    auto stamp = [&](int i) { if (t == 0) rec[20 + i] = (uint32_t)wall_clock64(); };     // 100 MHz
    stamp(0);
    typedef const uint32_t __attribute__((address_space(1))) * GlobalWords;
    const GlobalWords gwords = (GlobalWords)(uintptr_t)A.words;
    auto word = [&](uint32_t i) -> uint32_t { return gwords[i]; };      // (a plain load: the lane comes back to the line for its next word, which a non-temporal load does not keep -- 313 against 465 us per 64 files in k_jpeg_walks)
    if (t == 0) { s_ext = 0xffffffffu; s_ei = 0; s_from = 0; s_jf = 0xffffffffu; s_mode = 0; s_idx0 = 0; s_fin = 0; s_chases = 0; s_S = 0; s_guessed = 0; Z.early = 0; Z.nrec = 0; }
    __syncthreads();                                                // (Z.nrec is counted up in phase A)

    // ---- A. what the walks found (k_jpeg_walks) and where each of the predecessor's candidates leads (k_jpeg_mend), this
    // workgroup's chunks and the one in front of them
    uint64_t my_in = JPEG_STATE_NONE, my_out = JPEG_STATE_NONE, my_rep_out = JPEG_STATE_NONE;
    uint32_t my_n = 0, my_rep_n = 0, my_nib = JPEG_MAP_FAIL, my_ext = 0xffffffffu;
    if (lane_ok) {
        const size_t at = (size_t)k * F.nchunks + g;
        my_in = J.cand_in[at]; my_out = J.cand_out[at]; my_n = J.cand_n[at];
        my_nib = J.cand_nib[at];
        if (k == 0) s_exact[j] = (my_nib & 64u) ? 1 : 0;
        if (my_nib & 16u) { my_rep_out = J.rep_out[at]; my_rep_n = J.rep_n[at]; }
        // (asked for here, with everything else: the lane that follows a record later finds its number in LDS)
        if ((my_nib & 16u) && (my_nib & 15u) == JPEG_MAP_FAIL) my_ext = J.ext_idx[at];
    }
    Z.extidx[t] = my_ext;
    Z.midnone[t] = 0;
    if (my_ext < J.ext_cap) {                                       // ... and the record itself, while everybody is loading: the chase finds it in LDS
        const uint32_t slot = atomicAdd(&Z.nrec, 1u);
        if (slot < (uint32_t)JPEG_SEL_RECS) {
            const uint32_t* G = J.ext + (size_t)my_ext * JPEG_EXT_WORDS;
            uint32_t w[JPEG_EXT_WORDS];
#pragma unroll
            for (int i = 0; i < JPEG_EXT_WORDS; i++) w[i] = G[i];
#pragma unroll
            for (int i = 0; i < JPEG_EXT_WORDS; i++) Z.recs[slot][i] = w[i];
            Z.recno[slot] = my_ext;
        }
    }
    s_in[t] = my_in;
    s_out[t] = my_out;
    s_n[t] = my_n;
    s_rep_out[t] = my_rep_out;
    s_rep_n[t] = my_rep_n;
    s_nib[t] = (uint8_t)my_nib;
    s_own[t] = (uint8_t)((my_nib & 32u) ? (my_nib & 7u) : k);
    if (t < 6) s_pout[t] = (g0 > 0 && (uint32_t)t < B) ? J.cand_out[(size_t)t * F.nchunks + g0 - 1] : JPEG_STATE_NONE;
    __syncthreads();
    stamp(1);
    if (lane_ok && (my_nib & 32u)) {                                // a twin: the answer of the earlier candidate with the same state
        const uint32_t tw = (my_nib & 7u) * CPW + j;
        s_nib[t] = s_nib[tw];
        s_rep_out[t] = s_rep_out[tw];
        s_rep_n[t] = s_rep_n[tw];
    }
    __syncthreads();
    stamp(2);
    if ((uint32_t)t < nlive) {                                      // (t = j from here on)
        uint32_t m = 0;
        // (a file with fewer than six blocks per MCU: the unused inputs repeat input 0, so that "every input leads to the same
        // output" can be seen from the whole word)
        for (uint32_t i = 0; i < 6; i++) m |= ((uint32_t)s_nib[(i < B ? i : 0u) * CPW + t] & 15u) << (4 * i);
        m = s_exact[t] ? jpeg_map_const(0) : m;
        s_map[t] = m;
        s_scan[0][t] = m;
    }
    __syncthreads();
    stamp(3);
    // ---- C. inclusive scan: P[j] = which candidate of chunk j follows from each candidate of the chunk in front of the workgroup.
    // Up to 64 chunks (42 of a 4:2:0 file) are ONE wave's: the maps go from lane to lane in registers, no barrier between the
    // steps -- the scan is redone after every chase, with the workgroups that wait for this one's final word behind it.
    int cur = 0;
    auto scan_maps = [&]() {                                        // s_scan[0][0 .. nlive) -> its inclusive scan in s_scan[cur]
        cur = 0;
        if (nlive <= 64u) {
            if (t < 64) {
                uint32_t m = (uint32_t)t < nlive ? s_scan[0][t] : 0u;
                for (uint32_t ofs = 1; ofs < nlive; ofs <<= 1) {
                    const uint32_t before = __shfl_up(m, ofs, 64);
                    m = (uint32_t)t >= ofs ? jpeg_map_then(before, m) : m;
                }
                if ((uint32_t)t < nlive) s_scan[0][t] = m;
            }
            __syncthreads();
            return;
        }
        for (uint32_t ofs = 1; ofs < nlive; ofs <<= 1) {
            if ((uint32_t)t < nlive) {
                const uint32_t mine = s_scan[cur][t];
                s_scan[cur ^ 1][t] = (uint32_t)t >= ofs ? jpeg_map_then(s_scan[cur][t - ofs], mine) : mine;
            }
            cur ^= 1;
            __syncthreads();
        }
    };
    scan_maps();
    if (t == 0) {
        if (b + 1 < nblocks) st_relaxed(rec + 1, chain_word(1u, 0u, s_scan[cur][jl]));
        // ---- which of the predecessor's candidates is the true one?  Look back over the maps of the workgroups before, nearest
        // first, until what comes before no longer matters (a constant map: the usual case after ONE) or a final word is met.
        if (!s_exact[0]) {
            uint32_t acc = 0x543210u, idx0 = 0, mode = 2;           // mode 2 = not known yet
            for (int w = (int)b - 1; w >= 0 && mode == 2; w--) {
                const uint32_t* r = A.records + (size_t)w * JPEG_CTL_REC;
                const uint32_t W = wait_word_ge(r + 1, 1u);
                if (!W) break;
                if ((W >> 30) == 2u) {
                    const uint32_t kind = (W >> 24) & 15u;
                    if (kind < 6) { const uint32_t x = jpeg_map_at(acc, kind); if (x != JPEG_MAP_FAIL) { idx0 = x; mode = 0; } }
                    else if (w == (int)b - 1) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        if (kind == 13) { s_ext = r[4]; s_ei = r[5]; mode = 3; }
                        else { s_S = (uint64_t)r[4] | ((uint64_t)r[5] << 32); mode = 1; }
                    }
                    break;                                          // nothing to learn behind a final word
                }
                acc = jpeg_map_then(W & 0xffffffu, acc);
                // "whatever candidate comes in, this one goes out": taken for the true chain too, which holds when the true state
                // in front of workgroup w is one of the candidates or falls into step with them inside these workgroups -- always,
                // in a photograph; in a stream of dense blocks it may do neither.  So the guess is CHECKED against the
                // predecessor's final word once this workgroup is done (below): a wrong one refuses the file.
                if (jpeg_map_is_const(acc)) { idx0 = acc & 15u; mode = 0; s_guessed = 1u + idx0; }
            }
            if (mode == 2) {                                        // the maps do not say: the predecessor's final word, then
                if (const uint32_t W = wait_word_ge(prec + 1, 2u)) {
                    const uint32_t kind = (W >> 24) & 15u;
                    if (kind < 6) { idx0 = kind; mode = 0; }
                    else {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        if (kind == 13) { s_ext = prec[4]; s_ei = prec[5]; mode = 3; }
                        else { s_S = (uint64_t)prec[4] | ((uint64_t)prec[5] << 32); mode = 1; }
                    }
                } else {
                    atomicOr(&A.header[1], JPEG_ST_CHAIN_TIMEOUT);
                    s_S = JPEG_STATE_NONE;                          // a dead state: every chunk behind it stays empty, the verdict is "refused"
                    mode = 1;
                }
            }
            s_idx0 = idx0;
            s_mode = mode;
        }
        // The FINAL word, as early as it is known: with the candidate in front of the workgroup in hand, the scan names the
        // true candidate of the workgroup's last chunk unless a chunk on the way (and no interval's start behind it) left the
        // candidates -- and a workgroup behind this one whose maps do not settle the matter waits for exactly this word.
        // Published after the picking and the counting, every workgroup of such a chain added all of that to the wait of
        // the next (profiles/r05_jpeg_select_chain.txt: 2.4 us a hop; chains of ten and more in a 1080p file).
        if (s_mode == 0 && b + 1 < nblocks) {
            const uint32_t x = jpeg_map_at(s_scan[cur][jl], s_idx0);
            if (x != JPEG_MAP_FAIL) { st_relaxed(rec + 1, chain_word(2u, x, s_scan[cur][jl])); Z.early = 1u + x; }
        }
    }
    __syncthreads();
    stamp(4);
    // ---- D / E.  Every chunk picks its true walk from the scan; where the chain leaves the candidates (a repair walk that joined
    // nothing, or a predecessor that hands on a record or a state), ONE lane follows it up to the chunk where the state is a
    // candidate again, and the chunks behind that are scanned and picked once more by all.
    uint32_t mode = s_mode, idx0 = s_idx0, from = 0;                // chunks below `from` are settled
    // (where the chunk's middle state is to be copied from -- the walk, or the repair walk, that turned out to be the true one:
    // remembered here and copied ONCE behind the rounds; copied inside them, the load in front of the store was a trip to
    // memory per round with the workgroups that wait for this one's final word behind it)
    const uint64_t* mid_from = nullptr;
    const uint32_t* nmid_from = nullptr;
    size_t mid_at = 0;
    uint32_t rounds = 0;                                            // (IMPGPU_JPEG_TRACE=2)
    for (;;) {
        rounds++;
        if ((uint32_t)t >= from && (uint32_t)t < nlive) {
            uint32_t idx_in = JPEG_MAP_FAIL;                        // which candidate of the chunk before is the true one
            if (mode == 0) idx_in = t == 0 ? idx0 : jpeg_map_at(s_scan[cur][t - 1], idx0);
            uint64_t ent = JPEG_STATE_NONE;
            uint32_t n = 0;
            bool known = false;
            if (s_exact[t]) { ent = s_in[t]; n = s_n[t]; known = true; mid_from = J.cand_mid; nmid_from = J.cand_nmid; mid_at = (size_t)g0 + t; }
            else if (idx_in != JPEG_MAP_FAIL) {
                const uint32_t nb = s_nib[idx_in * CPW + t], v = nb & 15u;
                if (!(nb & 16u)) {
                    if (v != JPEG_MAP_FAIL) {
                        ent = s_in[v * CPW + t]; n = s_n[v * CPW + t]; known = true;
                        mid_from = J.cand_mid; nmid_from = J.cand_nmid; mid_at = (size_t)v * F.nchunks + g0 + t;
                    } else atomicMin(&s_jf, (uint32_t)t > 0 ? (uint32_t)t - 1 : 0u);   // (cannot happen: a true exit state is never "no candidate")
                } else {
                    ent = t > 0 ? s_out[idx_in * CPW + t - 1] : s_pout[idx_in];
                    n = s_rep_n[idx_in * CPW + t];
                    known = true;
                    mid_from = J.rep_mid; nmid_from = J.rep_nmid; mid_at = (size_t)s_own[idx_in * CPW + t] * F.nchunks + g0 + t;   // (a twin's walk was made by the candidate it shares its state with)
                    if (v == JPEG_MAP_FAIL) atomicMin(&s_jf, (uint32_t)t);            // its exit joined no walk: follow it from here
                }
            }
            if (known) {
                A.chunk_entry[g0 + t] = ent; s_res_n[t] = n;
            }
        }
        __syncthreads();
        const uint32_t jf = s_jf;
        if (mode == 0 && jf == 0xffffffffu) {                       // everything picked
            if (t == 0) {                                           // (a constant map when an interval starts inside the workgroup)
                s_fin = jpeg_map_at(s_scan[cur][jl], idx0); s_S = 0;
                // (the word for the workgroup behind, before the counting below: it may be waiting for it)
                if (!Z.early && b + 1 < nblocks && s_fin < 6u) { st_relaxed(rec + 1, chain_word(2u, s_fin, 0u)); Z.early = 1u + s_fin; }
            }
            break;
        }
        if (t == 0) {
            // A candidate whose repair walk joined nothing has a record (k_jpeg_mend) of the chunks behind it: their entry states
            // and slot counts, up to the chunk where the state is a candidate again.  Only where there is no record (or it ends
            // unjoined) is an explicit state decoded on from, with the tables read from memory.
            const JpegHuffCompact& L = *(const JpegHuffCompact*)A.tables;
            uint32_t jn = 0, idx = 0, chases = 0, ei = 0, rcount = 0, rjoined = 15;
            bool expl = true;
            uint64_t S = JPEG_STATE_NONE;
            // The record being followed is fetched WHOLE, every word asked for at once, and read out of LDS from there on:
            // word by word out of memory -- its number, its header, a step at a time, its last state -- a record was four to
            // eight trips in a row by one lane with the workgroup (and those that wait for its final word) behind it.
            const uint32_t* R = nullptr;
            uint32_t Rno = 0xffffffffu;
            auto fetch_record = [&](uint32_t r) {
                const uint32_t have = min(Z.nrec, (uint32_t)JPEG_SEL_RECS);
                R = nullptr;
                for (uint32_t i = 0; i < have; i++) if (Z.recno[i] == r) R = Z.recs[i];      // fetched ahead
                if (!R) {
                    const uint32_t* G = J.ext + (size_t)r * JPEG_EXT_WORDS;
                    uint32_t w[JPEG_EXT_WORDS];
#pragma unroll
                    for (int i = 0; i < JPEG_EXT_WORDS; i++) w[i] = G[i];
#pragma unroll
                    for (int i = 0; i < JPEG_EXT_WORDS; i++) Z.recbuf[i] = w[i];
                    R = Z.recbuf;
                }
                Rno = r; rcount = R[2]; rjoined = R[3];
            };
            auto open_record = [&](uint32_t kk, uint32_t jj) {      // candidate kk of the chunk before jj left a record?
                const uint32_t r = Z.extidx[(uint32_t)s_own[kk * CPW + jj] * CPW + jj];
                if (r < J.ext_cap) { fetch_record(r); ei = 0; }
            };
            auto close_record = [&]() {
                if (rjoined < 6) { idx = rjoined; expl = false; }
                else { S = (uint64_t)R[4 + 3 * JPEG_EXT_STEPS] | ((uint64_t)R[5 + 3 * JPEG_EXT_STEPS] << 32); expl = true; }
                R = nullptr;
            };
            if (mode == 1) S = s_S;
            else if (mode == 3) { fetch_record(s_ext); ei = s_ei; }
            else {
                const uint32_t idx_in = jf == 0 ? idx0 : jpeg_map_at(s_scan[cur][jf - 1], idx0);
                S = s_rep_out[idx_in * CPW + jf];
                open_record(idx_in, jf);
                jn = jf + 1;
            }
            uint32_t jj = jn;
            for (; jj < nlive; jj++) {
                if (s_exact[jj]) { expl = false; idx = 0; R = nullptr; break; }   // an interval begins: the scan has the rest
                if (R && ei >= rcount) close_record();
                if (!R && !expl) break;                             // the state is candidate `idx` of chunk jj - 1 again
                uint64_t ent;
                uint32_t n;
                chases++;
                if (R) {
                    ent = (uint64_t)R[4 + 3 * ei] | ((uint64_t)R[5 + 3 * ei] << 32);
                    n = R[6 + 3 * ei];
                    ei++;
                } else {
                    ent = S;
                    uint32_t k1 = JPEG_MAP_FAIL;
                    for (uint32_t i = 0; i < B; i++) if (k1 == JPEG_MAP_FAIL && s_in[i * CPW + jj] == S) k1 = i;
                    if (k1 != JPEG_MAP_FAIL) { n = s_n[k1 * CPW + jj]; idx = k1; expl = false; }
                    else {
                        const uint32_t gg = g0 + jj, seg = A.chunk_seg[gg];
                        const uint32_t seg_end = A.seg_first_chunk[seg] * CHUNK_BITS + A.seg_bits[seg];
                        const uint32_t lim = min((gg + 1) * CHUNK_BITS, seg_end);
                        // (one lane, a chunk after the other, everybody behind it waiting: with the tables in memory a chunk took 8-9 us)
                        const JpegSpan sp = Lw ? jpeg_span_walk(*Lw, word, S, gg * CHUNK_BITS, lim, lim, seg_end, F)
                                               : jpeg_span_walk(L, word, S, gg * CHUNK_BITS, lim, lim, seg_end, F);
                        atomicAdd(&A.header[0], 1u);                // (diagnostics: walks this kernel had to do itself)
                        n = sp.n;
                        for (uint32_t i = 0; i < B; i++) if (k1 == JPEG_MAP_FAIL && s_out[i * CPW + jj] == sp.out) k1 = i;
                        if (k1 != JPEG_MAP_FAIL) { idx = k1; expl = false; } else S = sp.out;
                    }
                }
                A.chunk_entry[g0 + jj] = ent;
                s_res_n[jj] = n;
                Z.midnone[jj] = 1;                                  // (a chunk reached by the chase: one lane decodes all of it)
            }
            if (jj >= nlive) {                                      // the workgroup's end: what the next one starts from --
                if (R && ei >= rcount) close_record();
                if (R) { s_fin = 13u; s_S = (uint64_t)Rno | ((uint64_t)ei << 32); }   // a record to go on with,
                else { s_fin = expl ? 15u : idx; s_S = S; }                            // a state, or a candidate of the last chunk
            }
            s_from = jj;
            s_idx0 = idx;
            s_jf = 0xffffffffu;
            s_chases += chases;
        }
        __syncthreads();
        from = s_from;
        if (from >= nlive) break;
        // scan the rest again: in front of chunk `from` stands candidate s_idx0 of the chunk before it, whatever came before
        idx0 = s_idx0;
        mode = 0;
        if ((uint32_t)t < nlive) s_scan[0][t] = (uint32_t)t < from ? jpeg_map_const(idx0) : s_map[t];
        __syncthreads();
        scan_maps();
    }
    __syncthreads();
    // the final word for whoever looks back this far, the slot counts for k_jpeg_write
    uint32_t total = 0;
    if ((uint32_t)t < nlive) { const uint32_t n = s_res_n[t]; A.chunk_n[g0 + t] = n; total = n; }
    if ((uint32_t)t < nlive && F.wsplit > 1) {
        if (Z.midnone[t]) J.chunk_mid[g0 + t] = JPEG_STATE_NONE;
        else if (mid_from) { J.chunk_mid[g0 + t] = mid_from[mid_at]; J.chunk_nmid[g0 + t] = nmid_from[mid_at]; }
    }
    for (int o = 32; o > 0; o >>= 1) total += __shfl_down(total, o, 64);
    if ((t & 63) == 0) s_scan[0][t >> 6] = total;
    __syncthreads();
    if (t == 0) {
        uint32_t all = 0;
        for (int w = 0; w < SB / 64; w++) all += s_scan[0][w];
        rec[6] = all;
        if (b + 1 < nblocks && !Z.early) {
            if (s_fin < 6u) st_relaxed(rec + 1, chain_word(2u, s_fin, 0u));        // (nobody reads a map behind a final word)
            else {
                rec[4] = (uint32_t)s_S;
                rec[5] = (uint32_t)(s_S >> 32);
                st_release(rec + 1, chain_word(2u, s_fin, 0u));
            }
        }
        if (Z.early && (Z.early - 1u != s_fin || s_S != 0)) atomicOr(&A.header[1], JPEG_ST_CHAIN_GUESS);   // (cannot happen: the early word IS the late one; a file it happened to is refused)
        if (s_chases) atomicAdd(&A.header[3], s_chases);
        // the guess of the look-back against what the workgroup in front really ended in (it has published by now, or is about
        // to: nothing it waits for comes after this workgroup)
        if (s_guessed) {
            const uint64_t took = s_pout[s_guessed - 1u];
            bool held = false;
            if (const uint32_t W = wait_word_ge(prec + 1, 2u)) {
                const uint32_t kind = (W >> 24) & 15u;
                if (kind < 6) held = s_pout[kind] == took;          // (the same STATE: two candidates may be twins)
                else if (kind == 15) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); held = ((uint64_t)prec[4] | ((uint64_t)prec[5] << 32)) == took; }
            } else atomicOr(&A.header[1], JPEG_ST_CHAIN_TIMEOUT);
            if (!held) atomicOr(&A.header[1], JPEG_ST_CHAIN_GUESS);
        }
    }
    stamp(5);
    if (t == 0) { rec[29] = rounds; rec[30] = s_chases; }             // (IMPGPU_JPEG_TRACE=2: rounds of picking, chunks this workgroup chased)
}

__global__ __launch_bounds__(SB) void k_jpeg_select(const JpegJob* __restrict__ jobs, const JpegMapEntry* __restrict__ block_map, uint32_t* __restrict__ launch_ticket) {
    __shared__ SelShared Z;
    __shared__ uint32_t s_ticket;
    // workgroups are numbered in the order they START (a ticket), never by blockIdx: the one a workgroup waits for is then
    // always running already
    if (threadIdx.x == 0) s_ticket = atomicAdd(launch_ticket, 1u);
    __syncthreads();
    const JpegMapEntry me = block_map[__builtin_amdgcn_readfirstlane(s_ticket)];
    const JpegJob& J = jobs[__builtin_amdgcn_readfirstlane(me.job)];
    select_body(Z, J, __builtin_amdgcn_readfirstlane(me.local));
}

constexpr int HB = JPEG_HUFF_BLOCK;

// Sum of x[c] over the chunks c in [first, g0) -- g0 = the workgroup's first chunk, first = where the interval it belongs to
// began -- by all lanes: the chunks of whole groups of `per` come as the group totals `tot[w * stride]`, the ragged ends one by one.
template <class T>
__device__ T sum_before(const T* x, int xstride, const T* tot, int stride, uint32_t per, uint32_t first, uint32_t g0, int t, T* s_acc) {
    if (t == 0) *s_acc = 0;
    __syncthreads();
    T part = 0;
    const uint32_t wf = first / per, w0 = g0 / per;
    if (wf == w0) {
        for (uint32_t c = first + (uint32_t)t; c < g0; c += HB) part += x[(size_t)c * xstride];
    } else {
        for (uint32_t c = first + (uint32_t)t; c < (wf + 1) * per; c += HB) part += x[(size_t)c * xstride];
        for (uint32_t w = wf + 1 + (uint32_t)t; w < w0; w += HB) part += tot[(size_t)w * stride];
        for (uint32_t c = w0 * per + (uint32_t)t; c < g0; c += HB) part += x[(size_t)c * xstride];
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((t & 63) == 0 && part) atomicAdd(s_acc, part);
    __syncthreads();
    return *s_acc;
}

struct WriteShared {
    JpegBlockTabs K;
    uint32_t n[HB];
    uint8_t head[HB];                                               // 1 = an interval starts in or before this chunk (inside the workgroup)
    uint32_t carry;
    int tot[4];
    // per lane: the block it is decoding.  144 bytes apart, not 128: with a power-of-two pitch every lane's coefficient k sits
    // in one of two banks and a wave's scattered 2-byte stores queue up behind each other (31 % of the LDS-active cycles were
    // bank conflicts, profiles/r04_jpeg_sq_counters.txt); 36 words apart the same k of sixteen lanes falls into sixteen banks
    __attribute__((aligned(16))) int16_t stage[HB][72];
    uint32_t list[HB / 64][128];                                    // per wave: its complete blocks {where in LDS, where in the planes}
};
// The workgroup's UNITS are g0 .. g0 + own - 1, a lane each: F.wsplit units per chunk -- unit u is chunk u / wsplit, and with
// two the second one enters the chunk at its middle, in the state the chunk's true walk passed there (JpegJob::chunk_mid;
// where that is not known the first unit decodes the whole chunk and the second has nothing to do).  Everything below --
// slot prefix, budget, verdicts, DC sums -- is per unit; k_jpeg_select's per-chunk counts come apart into the two.
// L: the tables, in either form.  wg_dc: where the workgroup's DC sums go.
template <class Tabs>
__device__ __forceinline__ void write_body(const Tabs& L, WriteShared& Y, const JpegJob& J, const uint32_t g0, const uint32_t own, int* wg_dc) {
    JpegBlockTabs& K = Y.K;
    auto& s_n = Y.n; auto& s_head = Y.head; uint32_t& s_carry = Y.carry; auto& s_tot = Y.tot; auto& s_stage = Y.stage; auto& s_list = Y.list;
    const int t = threadIdx.x;
    for (int i = t; i < HB * 72 / 8; i += HB) ((uint4*)&s_stage[0][0])[i] = make_uint4(0, 0, 0, 0);
    const JpegFrame& F = J.F;
    struct { const uint32_t *words, *chunk_seg, *seg_first_chunk, *seg_bits; const JpegHuffDev* tables; int16_t* coef; uint32_t *header, *records; } A =
        {J.words, J.chunk_seg, J.seg_first_chunk, J.seg_bits, J.tables, J.coef, J.header, J.records};
    load_block_tables(K, F, t);
    if (t < 4) s_tot[t] = 0;
    const uint32_t CHUNK_BITS = F.chunk_bits;
    const uint32_t WS = F.wsplit > 1 ? 2u : 1u;
    const uint32_t g = g0 + (uint32_t)t;                            // the unit
    const uint32_t gc = WS == 2 ? g >> 1 : g, half = WS == 2 ? g & 1u : 0u;      // its chunk, which part of it
    const bool live = (uint32_t)t < own && gc < F.nchunks;
    uint32_t seg = 0, first = 0, seg_end = 0, limit = 0, own_n = 0;
    uint64_t entry = JPEG_STATE_NONE;
    bool origin = false, tail = true, idle = false;                 // tail: the chunk's last unit with anything to decode; idle: a second unit with nothing to decode
    if (live) {
        seg = A.chunk_seg[gc];
        first = A.seg_first_chunk[seg];
        origin = first == gc && half == 0;
        seg_end = first * CHUNK_BITS + A.seg_bits[seg];
        limit = min((gc + 1) * CHUNK_BITS, seg_end);
        entry = J.chunk_entry[gc];
        own_n = J.chunk_n[gc];
        if (WS == 2) {
            const uint64_t mid = J.chunk_mid[gc];
            // (a dead state: not known, or the walk ended before it got there; at or behind the chunk's end -- an interval's short
            // last chunk, a symbol that spans the second half: nothing there for a second lane, and the first must close the interval)
            const bool have = (uint32_t)(mid >> 48) == 0 && (uint32_t)mid < limit;
            const uint32_t nmid = have ? J.chunk_nmid[gc] : 0u;
            if (half == 0) { if (have) { limit = min(gc * CHUNK_BITS + CHUNK_BITS / 2, limit); own_n = nmid; tail = false; } }
            else { entry = have ? mid : JPEG_STATE_NONE; own_n = have ? own_n - nmid : 0u; tail = have; idle = !have; }
        }
    }
    typedef const uint32_t __attribute__((address_space(1))) * GlobalWords;
    const GlobalWords gwords = (GlobalWords)(uintptr_t)A.words;
    auto word = [&](uint32_t i) -> uint32_t { return gwords[i]; };
    // ---- the slot of every chunk's first symbol: running totals inside each interval (a segmented scan), plus what the
    // interval had passed before this workgroup's first chunk (k_jpeg_select left its workgroups' totals in their records)
    s_n[t] = own_n;
    s_head[t] = origin ? 1 : 0;
    __syncthreads();
    for (int ofs = 1; ofs < HB; ofs <<= 1) {
        uint32_t n2 = 0, h2 = 0;
        const bool take = t >= ofs;
        if (take) { n2 = s_n[t - ofs]; h2 = s_head[t - ofs]; }
        const uint32_t myh = s_head[t];
        __syncthreads();
        if (take) {
            if (!myh) s_n[t] += n2;
            s_head[t] = (uint8_t)(myh | h2);
        }
        __syncthreads();
    }
    const uint32_t c0 = WS == 2 ? g0 >> 1 : g0;                      // (a workgroup begins with a chunk's first unit: HB is even)
    const uint32_t first0 = A.seg_first_chunk[A.chunk_seg[c0]];
    const uint32_t carry = sum_before<uint32_t>(J.chunk_n, 1, A.records + 6, JPEG_CTL_REC, (uint32_t)SB / (uint32_t)F.bpm, first0, c0, t, &s_carry);
    const bool open = !s_head[t];                                   // still in the interval that began in an earlier workgroup
    const uint32_t incl_n = s_n[t] + (open ? carry : 0u);
    // ---- decode, now knowing where every coefficient goes.  The last chunk of an interval walks with the interval's
    // remaining slots as a budget -- a sequential decoder stops after the last MCU and never looks at the padding bits, which
    // in a damaged file need not be the 1-bits an encoder writes -- and gives the verdict: it must end inside the padding
    // with exactly the interval's slots decoded.
    JpegDecoded e;
    uint32_t slot0 = 0;
    {
        const uint32_t last_chunk_of_seg = live ? (seg + 1 < F.nsegs ? A.seg_first_chunk[seg + 1] : F.nchunks) - 1 : 0u;
        const uint32_t base_n = incl_n - own_n;
        const uint32_t slots_here = min((uint32_t)F.slots_per_seg, F.total_slots - seg * (uint32_t)F.slots_per_seg);
        const bool closes = gc == last_chunk_of_seg && tail;
        JpegWriteCtx W;
        W.coef = A.coef;
        W.slot0 = slot0 = seg * (uint32_t)F.slots_per_seg + base_n;
        W.dc0[0] = W.dc0[1] = W.dc0[2] = 0;                         // relative to the chunk's entry: k_jpeg_dcfix adds the rest
        W.status = &A.header[1];
        W.stage = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&s_stage[t][0];
        W.list = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&s_list[t >> 6][0];
        // (an idle second unit stands behind the unit that closes its interval, whose count of the padding bits need not fit)
        const bool overrun = live && !idle && !closes && incl_n > slots_here;     // would write outside the interval
        if (overrun) atomicOr(&A.header[1], JPEG_ST_OVERRUN);
        const uint32_t budget = closes ? (slots_here >= base_n ? slots_here - base_n : 0u) : 0xffffffffu;
        // (every lane of the wave takes part: finished blocks are written out by the lanes together)
        if (t == 0) wg_dc[6] = (int)(uint32_t)wall_clock64();        // (IMPGPU_JPEG_TRACE=2: the walk begins ... and is over)
#if defined(__HIP_DEVICE_COMPILE__)
        e = jpeg_write_chunk_dev(L, K, word, entry, limit, seg_end, F, &W, budget, live && !overrun);
#else
        e = jpeg_write_chunk(L, K, word, entry, limit, seg_end, F, &W, budget, live && !overrun);    // (the host pass only parses the kernel)
#endif
        if (t == 0) wg_dc[7] = (int)(uint32_t)wall_clock64();
        if (live && !overrun) {
            if (closes) {
                const uint32_t pe = (uint32_t)e.exit, fle = (uint32_t)(e.exit >> 48);
                if ((fle & JPEG_FL_INVALID) || pe > seg_end || seg_end - pe >= 8) atomicOr(&A.header[1], JPEG_ST_BAD_CODE);
                if (base_n + e.n != slots_here) atomicOr(&A.header[1], JPEG_ST_BAD_COUNT);
            } else if (e.n != own_n) atomicOr(&A.header[1], JPEG_ST_BAD_COUNT);   // the two walks disagree: cannot happen
        }
    }
    if (live) {
        J.chunk_slot0[g] = slot0;
        int* d = J.chunk_dc + (size_t)g * 4;
        d[0] = e.dc[0]; d[1] = e.dc[1]; d[2] = e.dc[2]; d[3] = (int)e.ndc;
    }
    // the workgroup's DC sums, for the look-back of k_jpeg_dcfix
    int d0 = e.dc[0], d1 = e.dc[1], d2 = e.dc[2];
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_down(d0, o, 64); d1 += __shfl_down(d1, o, 64); d2 += __shfl_down(d2, o, 64); }
    if ((t & 63) == 0) { atomicAdd(&s_tot[0], d0); atomicAdd(&s_tot[1], d1); atomicAdd(&s_tot[2], d2); }
    __syncthreads();
    if (t < 3) wg_dc[t] = s_tot[t];
}

__global__ __launch_bounds__(HB) void k_jpeg_write(const JpegJob* __restrict__ jobs, const JpegMapEntry* __restrict__ block_map) {
    __shared__ JpegHuffCompact L;                                   // (the tables as the host built them: this walk reads nothing the widened form adds)
    __shared__ WriteShared Y;
    const int t = threadIdx.x;
    const uint32_t clock0 = (uint32_t)wall_clock64();
    const JpegMapEntry me = block_map[blockIdx.x];
    const JpegJob& J = jobs[__builtin_amdgcn_readfirstlane(me.job)];
    for (int i = t; i < (int)(sizeof(JpegHuffCompact) / 4); i += HB) ((uint32_t*)&L)[i] = ((const uint32_t*)J.tables)[i];
    const uint32_t b = __builtin_amdgcn_readfirstlane(me.local);
    write_body(L, Y, J, b * HB, (uint32_t)HB, J.wg_dc + (size_t)b * 8);
    if (t == 0) { J.wg_dc[(size_t)b * 8 + 4] = (int)clock0; J.wg_dc[(size_t)b * 8 + 5] = (int)(uint32_t)wall_clock64(); }   // (IMPGPU_JPEG_TRACE=2)
}

struct DcShared {
    int dc[3][HB];
    uint8_t head[HB];
    int carry[3];
};
// tot / tot_stride / per: the DC sums of whole groups of `per` chunks (k_jpeg_write's workgroups, or k_jpeg_select's in the fused kernel)
__device__ __forceinline__ void dcfix_body(DcShared& D, const JpegJob& J, const uint32_t g0, const uint32_t owned, const int* tot, int tot_stride, uint32_t per) {
    auto& s_dc = D.dc; auto& s_head = D.head; auto& s_carry = D.carry;
    const int t = threadIdx.x;
    const JpegFrame& F = J.F;
    const uint32_t WS = F.wsplit > 1 ? 2u : 1u;
    const uint32_t g = g0 + (uint32_t)t;                            // the unit of k_jpeg_write (write_body)
    const uint32_t gc = WS == 2 ? g >> 1 : g, half = WS == 2 ? g & 1u : 0u;
    const bool live = (uint32_t)t < owned && gc < F.nchunks;
    int own[3] = {0, 0, 0};
    uint32_t ndc = 0;
    bool origin = false;
    if (live) {
        origin = J.seg_first_chunk[J.chunk_seg[gc]] == gc && half == 0;
        const int* d = J.chunk_dc + (size_t)g * 4;
        own[0] = d[0]; own[1] = d[1]; own[2] = d[2]; ndc = (uint32_t)d[3];
    }
    s_dc[0][t] = own[0]; s_dc[1][t] = own[1]; s_dc[2][t] = own[2];
    s_head[t] = origin ? 1 : 0;
    __syncthreads();
    for (int ofs = 1; ofs < HB; ofs <<= 1) {
        int a0 = 0, a1 = 0, a2 = 0;
        uint32_t h2 = 0;
        const bool take = t >= ofs;
        if (take) { a0 = s_dc[0][t - ofs]; a1 = s_dc[1][t - ofs]; a2 = s_dc[2][t - ofs]; h2 = s_head[t - ofs]; }
        const uint32_t myh = s_head[t];
        __syncthreads();
        if (take) {
            if (!myh) { s_dc[0][t] += a0; s_dc[1][t] += a1; s_dc[2][t] += a2; }
            s_head[t] = (uint8_t)(myh | h2);
        }
        __syncthreads();
    }
    const uint32_t first0 = J.seg_first_chunk[J.chunk_seg[WS == 2 ? g0 >> 1 : g0]] * WS;      // (in units, like g0 and the DC sums)
    int carry[3];
    for (int i = 0; i < 3; i++) carry[i] = sum_before<int>(J.chunk_dc + i, 4, tot + i, tot_stride, per, first0, g0, t, &s_carry[i]);
    if (!live || ndc == 0) return;
    const bool open = !s_head[t];
    int base[3];
    for (int i = 0; i < 3; i++) base[i] = s_dc[i][t] - own[i] + (open ? carry[i] : 0);
    // the blocks that begin in the chunk are neighbours in the scan: their predictors go to a side array indexed by the
    // block's number there (k_jpeg_pixels adds them) -- a short each, side by side, where adding them to the DC terms in the
    // planes was a 2-byte read-modify-write into a line of its own per block
    const uint32_t z = (uint32_t)((half ? J.chunk_mid[gc] : J.chunk_entry[gc]) >> 40) & 0xff;      // (a second unit with DC terms entered at the middle state)
    const uint32_t bpm = (uint32_t)F.bpm, nluma = bpm == 1 ? 1u : bpm - 2, nblocks = F.total_slots >> 6;
    const uint32_t gb = (J.chunk_slot0[g] >> 6) + (z ? 1u : 0u);
    uint32_t c = gb % bpm;
    for (uint32_t i = 0; i < ndc && gb + i < nblocks; i++) {
        J.dcadd[gb + i] = (int16_t)(c < nluma ? base[0] : (c - nluma == 0 ? base[1] : base[2]));
        c = c + 1 == bpm ? 0 : c + 1;
    }
}

__global__ __launch_bounds__(HB) void k_jpeg_dcfix(const JpegJob* __restrict__ jobs, const JpegMapEntry* __restrict__ block_map) {
    __shared__ DcShared D;
    const JpegMapEntry me = block_map[blockIdx.x];
    const JpegJob& J = jobs[__builtin_amdgcn_readfirstlane(me.job)];
    // the file's verdict (status, counters: final since k_jpeg_write ended) straight into the host's pinned words -- a copy
    // command between this kernel and k_jpeg_pixels cost the stream 8-10 us per launch
    if (me.local == 0 && threadIdx.x < 4 && J.verdict) J.verdict[threadIdx.x] = J.header[threadIdx.x];
    dcfix_body(D, J, __builtin_amdgcn_readfirstlane(me.local) * (uint32_t)HB, (uint32_t)HB, J.wg_dc, 8, (uint32_t)HB);
}

// ---------------------------------------------------------------- walks, mend and select in ONE launch (round 5)
// A lone file -- and a broker's batch of a few -- spends its time in launches, not in work: each of the kernels above is a
// latency-bound chain of 10-30 us with 15-25 us of launch, table load and tail around it.  Below the size where the kernels
// fill the device (the same 4 MB at which the chunks grow to 256 bytes) the three phases that share a workgroup shape run
// as ONE kernel: a workgroup owns its chunks through all three, numbered by ticket, and between phases it waits only for the
// workgroup BEFORE it in its file -- the candidates of the chunk in front of its own (mend); select's look-back is the
// chain it always was.  A workgroup never waits for a later one, later ones never start before earlier ones (tickets), the
// wait is bounded and counted like k_jpeg_select's: no grid barrier, no co-residency assumed.  What a workgroup's lanes leave
// for the next workgroup (its last chunk's candidates) is fenced by every lane before the flag that announces it.
// k_jpeg_mend's speculative records look at chunks AHEAD and are not kept here (k_jpeg_select's chase decodes on from the
// state instead).  k_jpeg_write and k_jpeg_dcfix stay launches of their own: they need the slot totals and DC sums of EVERY
// workgroup back to the interval's first chunk -- a prefix over the whole file where there are no restart markers -- and with
// them in the same kernel (measured: all five phases in one launch, the predecessors' totals waited for by the lanes in
// parallel) a lone 1080p file took 0.52 ms against 0.31 and a 4K one 1.09 against 0.49: hundreds of workgroups polling flags
// across the XCDs slow each other and the walks; a kernel boundary is the cheaper barrier there.  Only the 640 x 480 file
// gained (0.220 against 0.234).
struct FusedShared {
    JpegHuffTabs L;
    union { MendShared M; SelShared Z; } u;
    uint32_t ticket;
};

__global__ __launch_bounds__(SB) void k_jpeg_entropy_small(const JpegJob* __restrict__ jobs, const JpegMapEntry* __restrict__ block_map, uint32_t* __restrict__ launch_ticket) {
    __shared__ FusedShared X;
    const int t = threadIdx.x;
    if (t == 0) X.ticket = atomicAdd(launch_ticket, 1u);
    __syncthreads();
    const JpegMapEntry me = block_map[__builtin_amdgcn_readfirstlane(X.ticket)];
    const JpegJob& J = jobs[__builtin_amdgcn_readfirstlane(me.job)];
    const uint32_t b = __builtin_amdgcn_readfirstlane(me.local);
    uint32_t* rec = J.records + (size_t)b * JPEG_CTL_REC;
    if (t == 0) rec[26] = (uint32_t)wall_clock64();                 // (IMPGPU_JPEG_TRACE=2: the workgroup's start, its walks done, its mend done)
    // ---- 1. the walks of this workgroup's chunks
    walks_body(X.L, J, b, false);
    __threadfence();
    __syncthreads();
    if (t == 0) { rec[27] = (uint32_t)wall_clock64(); st_release(rec + 7, 1u); }
    // ---- 2. which walk does each of the predecessor chunk's candidates lead into (the chunk in front of the workgroup is
    // the workgroup's before it)
    if (b > 0 && t == 0 && !wait_flag_ge(rec - JPEG_CTL_REC + 7, 1u)) atomicOr(&J.header[1], JPEG_ST_CHAIN_TIMEOUT);
    __syncthreads();
    mend_body<true>(X.L, X.u.M, J, b);
    __syncthreads();
    if (t == 0) rec[28] = (uint32_t)wall_clock64();
    // ---- 3. every chunk's true entry state
    select_body(X.u.Z, J, b, &X.L);
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
// (`dcadd`: what k_jpeg_dcfix found the block's DC term to lack -- the planes hold it relative to its chunk's entry)
__device__ void idct_block_to_lds(const int16_t* blk, const uint16_t* q, uint8_t* out, int pitch, int dcadd) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    int ws[64];
    v4i raw[8];
#pragma unroll
    for (int k = 0; k < 8; k++) raw[k] = *(const v4i*)(blk + 8 * k);     // a row: eight shorts
    raw[0][0] = (raw[0][0] & (int)0xffff0000) | ((raw[0][0] + dcadd) & 0xffff);   // (16-bit, like the JCOEF libjpeg stores)
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
    const int16_t* __restrict__ dcadd = J.dcadd;                    // (null when the host decoded the entropy stage: absolute DC terms)
    uint8_t* __restrict__ dst = J.dst;
    const int dstep = J.dstep;
    if (t < 64 * NC) s_q[t >> 6][t & 63] = J.qt[t];
    __syncthreads();
    const int tiles_x = (F.width + TILE_W - 1) / TILE_W;
    const int tile_y = (int)__builtin_amdgcn_readfirstlane(me.local) / tiles_x, tile_x = (int)__builtin_amdgcn_readfirstlane(me.local) - tile_y * tiles_x;
    {   // luma: one block per lane
        const int bx = tile_x * (TILE_W / 8) + (t & 31), by = tile_y * (TILE_H / 8) + (t >> 5);
        if (bx < F.bw[0] && by < F.bh[0]) {
            // the block's number in the scan: MCU by MCU, the luma blocks of an MCU row by row, then Cb, Cr
            const int gb = ((by / VS) * F.mcux + bx / HS) * F.bpm + (by % VS) * HS + bx % HS;
            idct_block_to_lds(coef + F.coef_off[0] + ((size_t)by * F.bw[0] + bx) * 64, s_q[0], s_y + (t >> 5) * 8 * YP + (t & 31) * 8, YP, dcadd ? dcadd[gb] : 0);
        }
    }
    const int cbx0 = tile_x * (TILE_W / 8 / HS) - (HS == 2 ? 1 : 0), cby0 = tile_y * (TILE_H / 8 / VS) - (VS == 2 ? 1 : 0);
    if constexpr (NC == 3) {
        for (int i = t; i < 2 * CBW * CBH; i += 256) {
            const int ci = i >= CBW * CBH ? 2 : 1, j = i - (ci - 1) * CBW * CBH;
            const int lx = j % CBW, ly = j / CBW, bx = cbx0 + lx, by = cby0 + ly;
            if (bx >= 0 && by >= 0 && bx < F.bw[ci] && by < F.bh[ci]) {
                const int gb = (by * F.mcux + bx) * F.bpm + HS * VS + ci - 1;
                idct_block_to_lds(coef + F.coef_off[ci] + ((size_t)by * F.bw[ci] + bx) * 64, s_q[ci], s_c[ci - 1] + ly * 8 * CP + lx * 8, CP, dcadd ? dcadd[gb] : 0);
            }
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

int launch_jpeg_entropy(const JpegJob* jobs, const JpegMapEntry* sync_map, unsigned sync_blocks, const JpegMapEntry* chunk_map,
                        unsigned chunk_blocks, uint32_t* ticket, hipStream_t s, hipEvent_t* marks, bool small) {
    if (sync_blocks == 0) return IMP_OK;
    auto mark = [&](int i) { if (marks) (void)hipEventRecord(marks[i], s); };
    if (small) {
        hipLaunchKernelGGL(k_jpeg_entropy_small, dim3(sync_blocks), dim3(SB), 0, s, jobs, sync_map, ticket);
        for (int i = 0; i < 3; i++) mark(i);                        // (the stage profile's slots of walks, mend and select: all under the first)
        hipLaunchKernelGGL(k_jpeg_write, dim3(chunk_blocks), dim3(HB), 0, s, jobs, chunk_map);
        mark(3);
        hipLaunchKernelGGL(k_jpeg_dcfix, dim3(chunk_blocks), dim3(HB), 0, s, jobs, chunk_map);
        mark(4);
        IMP_HIP(hipGetLastError());
        return IMP_OK;
    }
    hipLaunchKernelGGL(k_jpeg_walks, dim3(sync_blocks), dim3(SB), 0, s, jobs, sync_map);
    mark(0);
    hipLaunchKernelGGL(k_jpeg_mend, dim3(sync_blocks), dim3(SB), 0, s, jobs, sync_map);
    mark(1);
    hipLaunchKernelGGL(k_jpeg_select, dim3(sync_blocks), dim3(SB), 0, s, jobs, sync_map, ticket);
    mark(2);
    hipLaunchKernelGGL(k_jpeg_write, dim3(chunk_blocks), dim3(HB), 0, s, jobs, chunk_map);
    mark(3);
    hipLaunchKernelGGL(k_jpeg_dcfix, dim3(chunk_blocks), dim3(HB), 0, s, jobs, chunk_map);
    mark(4);
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
