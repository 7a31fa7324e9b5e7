// imp_jpeg_enc.hip -- the encoder's side of the JPEG front: cvEncodeImage(".jpg", image, {CV_IMWRITE_JPEG_QUALITY, q})
// at bridge.c:704 (q from bridge.c:474-486) for the frame the operator chain leaves in HBM, so that the compressed file
// crosses the link instead of the pixels.  What OpenCV 2.4.9's JpegEncoder asks libjpeg for: jpeg_set_defaults +
// jpeg_set_quality(q, TRUE) -- baseline Huffman with the Annex K tables, YCbCr 4:2:0 (one gray component for 1-channel
// frames), ISLOW forward DCT, no restart markers, JFIF 1.01 -- and the output is the same FILE, byte for byte
// (oracle/orc_jpeg_enc.c is pinned against Pillow's libjpeg-turbo; tests/test_gpu_jpeg_enc.py compares with both).
//
//   host      headers (jcmarker.c order: SOI, APP0, DQT per table, SOF0, DHT per table, SOS), the quantisation tables of
//             jpeg_quality_scaling, the derived Huffman code tables (built once)
//   k_jpeg_enc_blocks   eight lanes per 8x8 block slot of the scan (a row, then a column each; round 3: one lane per block), in
//             MCU order: colour conversion (jccolor.c's fixed-point tables) on the fly, edge replication (jcsample.c expand_right_edge, jcprepct.c expand_bottom_edge -- the
//             chroma rows past the last real one repeat THAT row), the 2x2 chroma box with its alternating 1,2 bias
//             (h2v2_downsample), jfdctint.c's ISLOW DCT in registers, jcdctmgr.c's quantisation; the block goes to HBM as
//             64 shorts in zigzag order.  Block slots beyond a component's own blocks (jccoefct.c's dummy blocks) are
//             zero; their DC is their predecessor's and is resolved where it is read.
//   k_jpeg_enc_huff     one workgroup per image walks its blocks 256 at a time: every lane sizes its block's code
//             (jchuff.c encode_one_block), a block scan turns the sizes into bit offsets, the lanes OR their bits into an
//             LDS window (32 bits per ds_or), and the window's whole bytes leave with FF00 stuffing (a second scan over the
//             FF counts); the last partial byte carries into the next strip, flush_bits' 1-fill and EOI end the file.
// An image's scan is one serial bit stream, but only the offsets are serial: the two scans.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "imp_internal.h"
#include "imp_jpeg_std.h"

namespace imp {

namespace {

struct EncJob {
    const uint8_t* src;
    int w, h, c, step;
    int mcuw, mcuh, bpm, nblocks;       // MCUs across / down, blocks per MCU (6: Y Y Y Y Cb Cr, or 1), block slots in the scan
    int lbw, lbh, chh;                  // luma's own blocks (width_in_blocks / height_in_blocks), real chroma rows
    int out_cap;                        // bytes of `out`
    short* coef;                        // nblocks x 64, zigzag order
    uint8_t* out;                       // entropy-coded segment + EOI
    // large frames only (k_jpeg_enc_pack / k_jpeg_enc_stuff): the unstuffed bit stream (zeroed), one (sum, flag) record per
    // 256-block segment and one per 16 KB chunk of the stream (zeroed), [0] = the stream's bytes once it is complete
    uint32_t* ustream;
    uint32_t* seg_rec;
    uint32_t* chunk_rec;
    uint32_t* ubytes;
    int nseg, nchunk;
    // frames of 257 .. 2048 block slots (k_jpeg_enc_huff_seg): hseg workgroups of hper blocks each, and their records --
    // three (value, flag) pairs per workgroup, zeroed: [0] its bits, [2 hseg] the FF bytes among the file's bytes it owns,
    // [4 hseg] the unfinished byte it leaves to the next one
    uint32_t* hrec;
    int hseg, hper;
};
struct EncMap { int job, local; };      // workgroup of k_jpeg_enc_blocks -> (image, first block slot)
struct EncTables {
    uint16_t q[2][64];                  // natural order
    uint32_t huff[4][256];              // code << 8 | size : DC luma, AC luma, DC chroma, AC chroma
};

#define ENC_WIN_WORDS 14336             // 256 blocks x 1658 bits at most, plus the carried byte
#define ENC_BLOCK_BYTES 432             // what a block can need in the file: 1658 bits, every byte stuffed

// ---------------------------------------------------------------- k_jpeg_enc_blocks
__device__ __forceinline__ int enc_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c, one 1-D pass over eight values; PASS = 0 rows (results scaled up by 2^PASS1_BITS), 1 columns
template <int PASS>
__device__ __forceinline__ void enc_fdct8(int& d0, int& d1, int& d2, int& d3, int& d4, int& d5, int& d6, int& d7) {
    const int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6, t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    constexpr int SH = PASS ? 13 + 2 : 13 - 2;
    if (PASS == 0) { d0 = (t10 + t11) << 2; d4 = (t10 - t11) << 2; }
    else { d0 = enc_descale(t10 + t11, 2); d4 = enc_descale(t10 - t11, 2); }
    int z1 = (t12 + t13) * 4433;
    d2 = enc_descale(z1 + t13 * 6270, SH);
    d6 = enc_descale(z1 + t12 * -15137, SH);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * 9633;
    const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    d7 = enc_descale(a4 + z1 + z3, SH);
    d5 = enc_descale(a5 + z2 + z4, SH);
    d3 = enc_descale(a6 + z2 + z3, SH);
    d1 = enc_descale(a7 + z1 + z4, SH);
}

// jcdctmgr.c: magnitude + half the divisor, integer division, sign back (divisor = 8 q: the DCT's output is scaled by 8)
__device__ __forceinline__ int enc_quant(int v, int q8) {
    const int t = abs(v) + (q8 >> 1);
    int r = (int)((float)t / (float)q8);            // t < 2^17, q8 <= 2040: the quotient is off by one at most
    if (r * q8 > t) r--;
    if ((r + 1) * q8 <= t) r++;
    return v < 0 ? -r : r;
}

// which component / block a slot of the scan is: false for a dummy slot
__device__ __forceinline__ bool enc_slot(const EncJob& J, int mcu, int j, int* comp, int* bx, int* by) {
    const int my = mcu / J.mcuw, mx = mcu - my * J.mcuw;
    if (J.bpm == 1) { *comp = 0; *bx = mx; *by = my; return true; }
    if (j >= 4) { *comp = j - 3; *bx = mx; *by = my; return true; }
    *comp = 0; *bx = mx * 2 + (j & 1); *by = my * 2 + (j >> 1);
    return *bx < J.lbw && *by < J.lbh;
}

// jccolor.c rgb_ycc_convert on a pixel packed B | G << 8 | R << 16 (a gray sample in the low byte): which = 0 Y, 1 Cb, 2 Cr
template <int CN>
__device__ __forceinline__ int enc_sample_px(uint32_t px, int which) {
    if (CN == 1) return (int)(px & 255u);
    const int b = (int)(px & 255u), g = (int)((px >> 8) & 255u), r = (int)((px >> 16) & 255u);
    if (which == 0) return (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
    if (which == 1) return (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
    return (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
}

// N pixels of a row from column x0 on, packed as enc_sample_px reads them; columns past the frame's last repeat it
// (jcsample.c expand_right_edge).  Inside the frame and with 4-byte aligned rows the bytes come as dwords (a quarter of the
// loads: a wave's 64 lanes read 64 different rows, so every load instruction is 64 separate accesses whatever its width).
template <int CN, int N>
__device__ __forceinline__ void enc_load_px(const uint8_t* __restrict__ row, int x0, int w, bool aligned, uint32_t* px) {
    if (aligned && x0 + N <= w) {
        constexpr int NW = (N * CN + 3) / 4;
        uint32_t wd[NW + 1];
        const uint32_t* p = (const uint32_t*)(row + (size_t)x0 * CN);      // x0 is a multiple of 8: x0 * CN is a multiple of 4
#pragma unroll
        for (int i = 0; i < NW; i++) wd[i] = p[i];
        wd[NW] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (CN == 4) px[i] = wd[i];
            else if (CN == 1) px[i] = (wd[i >> 2] >> (8 * (i & 3))) & 255u;
            else {
                constexpr int dummy = 0; (void)dummy;
                const int byte = 3 * i, k = byte >> 2, off = byte & 3;
                px[i] = off == 0 ? wd[k] : off == 1 ? wd[k] >> 8 : (uint32_t)((((uint64_t)wd[k + 1] << 32) | wd[k]) >> (8 * off));
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint8_t* q = row + (size_t)min(x0 + i, w - 1) * CN;
            px[i] = CN == 1 ? (uint32_t)q[0] : (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
        }
    }
}

// zigzag position of the coefficient at natural index n (the inverse of jpeg_natural_order)
__constant__ uint8_t c_enc_zzinv[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                                         10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// Round 5: EIGHT lanes per block slot, a row each.  With a lane per block (round 3) a thumbnail's 882 blocks were 14 waves
// whose lanes each made 192 (luma) or 768 (chroma) single-byte loads, two 8-point DCT passes over 64 registers and 64
// quantisations in a row -- 44 us for a 224 x 168 answer, every request's fixed cost whatever the batch.  Here a lane loads
// ONE row of its block (two source rows of sixteen pixels for chroma) as dwords, runs the row pass, hands its eight
// results to the lane that owns the matching column through LDS, runs the column pass, quantises its eight coefficients
// and stores them at their zigzag places: the same jccolor / jcsample / jfdctint / jcdctmgr arithmetic on an eighth of the
// chain.  32 block slots per workgroup.
template <int CN>
__device__ __forceinline__ void enc_row_pass(const EncJob& J, int comp, int bx, int by, int r, bool aligned, int* d) {
    if (comp == 0) {
        const uint8_t* row = J.src + (size_t)min(by * 8 + r, J.h - 1) * J.step;
        uint32_t px[8];
        enc_load_px<CN, 8>(row, bx * 8, J.w, aligned, px);
#pragma unroll
        for (int i = 0; i < 8; i++) d[i] = enc_sample_px<CN>(px[i], 0) - 128;
    } else {
        const int cy = min(by * 8 + r, J.chh - 1);              // rows past the last real chroma row repeat that row
        const uint8_t* r0 = J.src + (size_t)min(2 * cy, J.h - 1) * J.step;
        const uint8_t* r1 = J.src + (size_t)min(2 * cy + 1, J.h - 1) * J.step;
        uint32_t p0[16], p1[16];
        enc_load_px<CN, 16>(r0, bx * 16, J.w, aligned, p0);
        enc_load_px<CN, 16>(r1, bx * 16, J.w, aligned, p1);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int s = enc_sample_px<CN>(p0[2 * i], comp) + enc_sample_px<CN>(p0[2 * i + 1], comp) + enc_sample_px<CN>(p1[2 * i], comp) +
                          enc_sample_px<CN>(p1[2 * i + 1], comp) + 1 + (i & 1);     // bias 1, 2, 1, 2, ... along the row
            d[i] = (s >> 2) - 128;
        }
    }
    enc_fdct8<0>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
}

constexpr int ENC_BLOCKS_PER_WG = 32;

// (its first workgroup also clears the launch's result words -- verdicts the later kernels OR into, the compact area's
// cursor: a fill command of its own on the stream was 4-5 us of a thumbnail's 80)
__global__ __launch_bounds__(256) void k_jpeg_enc_blocks(const EncJob* __restrict__ jobs, const EncMap* __restrict__ map, const EncTables* __restrict__ tabs,
                                                         uint32_t* __restrict__ result, int result_words, uint32_t* __restrict__ records, int record_words) {
    __shared__ int s_t[ENC_BLOCKS_PER_WG][8][9];                    // [block][row][column]: the row pass's results (+1: the column reads of a wave spread over the banks)
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < result_words; i += 256) result[i] = 0u;
    for (int i = (int)blockIdx.x * 256 + (int)threadIdx.x; i < record_words; i += (int)gridDim.x * 256) records[i] = 0u;   // (k_jpeg_enc_huff_seg's ticket and records)
    const EncMap m = map[blockIdx.x];
    const EncJob& J = jobs[m.job];
    const int tid = threadIdx.x, r = tid & 7, bl = tid >> 3;
    const int b = m.local + bl;
    const bool live = b < J.nblocks;
    int comp = 0, bx = 0, by = 0;
    bool real = false;
    if (live) {
        const int mcu = b / J.bpm, j = b - mcu * J.bpm;
        real = enc_slot(J, mcu, j, &comp, &bx, &by);
    }
    const bool aligned = !(((uintptr_t)J.src | (uintptr_t)J.step) & 3);
    int d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (real) {
        if (J.c == 1) enc_row_pass<1>(J, comp, bx, by, r, aligned, d);
        else if (J.c == 3) enc_row_pass<3>(J, comp, bx, by, r, aligned, d);
        else enc_row_pass<4>(J, comp, bx, by, r, aligned, d);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) s_t[bl][r][i] = d[i];
    __syncthreads();
    if (!live) return;
    short* out = J.coef + (size_t)b * 64;
    if (!real) {                                                    // a dummy slot (jccoefct.c): zeros; its DC is resolved where it is read
        ((int4*)out)[r] = int4{0, 0, 0, 0};
        return;
    }
    int v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = s_t[bl][k][r];               // column r of the block
    enc_fdct8<1>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    const uint16_t* q = &tabs->q[0][0] + (comp ? 64 : 0);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int n = k * 8 + r;
        out[c_enc_zzinv[n]] = (short)enc_quant(v[k], (int)q[n] << 3);
    }
}

// ---------------------------------------------------------------- k_jpeg_enc_huff
// DC of block slot (mcu, j): a dummy slot has the DC of its predecessor in the MCU (jccoefct.c: the block to its left at
// the right edge, MCU_buffer[blkn - 1] for a whole dummy row at the bottom), which ends at a real block: slot 0 always is
__device__ __forceinline__ int enc_dc_of(const EncJob& J, int mcu, int j) {
    if (J.bpm == 6 && j < 4) {
        const int my = mcu / J.mcuw, mx = mcu - my * J.mcuw;
        while (j > 0 && !(mx * 2 + (j & 1) < J.lbw && my * 2 + (j >> 1) < J.lbh)) j--;
    }
    return J.coef[((size_t)mcu * J.bpm + j) * 64];
}

__device__ __forceinline__ int enc_nbits(int v) { return 32 - __clz(abs(v)); }      // 0 for 0

// inclusive scan of one int per thread over the NT threads of the block; s_part: NT / 64 words of scratch.  Returns the
// thread's inclusive prefix, *total = the block's sum.
template <int NT>
__device__ __forceinline__ int enc_block_scan(int v, int* s_part, int* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int p = __shfl_up(v, d);
        if (lane >= d) v += p;
    }
    __syncthreads();
    if (lane == 63) s_part[wv] = v;
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) {
        const int p = s_part[k];
        if (k < wv) pre += p;
        tot += p;
    }
    *total = tot;
    return v + pre;
}

struct EncPut {                          // a lane's write head into the LDS window: bits are ORed in, 32 at a time
    uint32_t* win;
    uint64_t acc;
    int nacc, wpos;
    __device__ __forceinline__ void start(uint32_t* w, int bit) { win = w; acc = 0; nacc = bit & 31; wpos = bit >> 5; }
    __device__ __forceinline__ void put(uint32_t code, int len) {      // len <= 32 - 6, code < 2^len (nacc < 32 on entry: they fit the 64)
        if (len == 0) return;                                           // (a symbol without a code: cannot occur for 8-bit data)
        acc |= (uint64_t)code << (64 - nacc - len);
        nacc += len;
        if (nacc >= 32) {
            atomicOr(&win[wpos], (uint32_t)(acc >> 32));
            acc <<= 32;
            nacc -= 32;
            wpos++;
        }
    }
    __device__ __forceinline__ void finish() { if (nacc > 0) atomicOr(&win[wpos], (uint32_t)(acc >> 32)); }
};

// jchuff.c encode_one_block over a block in zigzag order, held in registers (32 dwords of two coefficients).  EMIT = false:
// only the number of bits.  `tab` = the workgroup's copy of the component's two code tables in LDS ([0..255] DC, [256..511] AC,
// entries (code << 8) | length): a walk makes up to 64 dependent lookups, and out of memory they were most of its time
// (round 5: 224 x 168, one workgroup, 74 -> see DESIGN 4b).  A symbol's code and its extra bits leave as one put (<= 26 bits).
template <bool EMIT>
__device__ __forceinline__ int enc_code_block(const uint32_t (&cw)[32], int last_dc, int dc, const uint32_t* tab, EncPut* P) {
    int bits = 0;
    {
        int t = dc - last_dc, t2 = t;
        if (t < 0) { t = -t; t2--; }
        const int n = enc_nbits(t);
        const uint32_t e = tab[n];
        bits += (int)(e & 0xff) + n;
        if (EMIT) P->put(((e >> 8) << n) | ((uint32_t)t2 & ((1u << n) - 1)), (int)(e & 0xff) + n);
    }
    int run = 0;
    const uint32_t zrl = tab[256 + 0xF0], eob = tab[256];
#pragma unroll
    for (int k = 1; k < 64; k++) {
        int t = (short)(cw[k >> 1] >> ((k & 1) * 16));
        if (t == 0) { run++; continue; }
        while (run > 15) {
            bits += (int)(zrl & 0xff);
            if (EMIT) P->put(zrl >> 8, (int)(zrl & 0xff));
            run -= 16;
        }
        int t2 = t;
        if (t < 0) { t = -t; t2--; }
        const int n = enc_nbits(t);
        const uint32_t e = tab[256 + (run << 4) + n];
        bits += (int)(e & 0xff) + n;
        if (EMIT) P->put(((e >> 8) << n) | ((uint32_t)t2 & ((1u << n) - 1)), (int)(e & 0xff) + n);
        run = 0;
    }
    if (run > 0) {
        bits += (int)(eob & 0xff);
        if (EMIT) P->put(eob >> 8, (int)(eob & 0xff));
    }
    return bits;
}

// result[4 * job] = bytes of the entropy-coded segment + EOI (what the file needs after its headers, whether or not it
// fitted), [4 * job + 1] = 1 when it did not fit out_cap, [4 * job + 2] = where the workgroup put a copy of the segment in
// `compact` (0xffffffff: it did not fit there).  The compact area is what the host fetches: the segments of a whole batch
// back to back (in the order the workgroups finish), one copy instead of one per image.  *cursor = how far it is filled.
// NT threads take NT blocks per pass.  NT = 256: whatever the blocks hold, a pass fits the window.  NT = 1024 (frames of more
// than 256 blocks): a pass takes the blocks from the front whose code still fits -- all 1024 unless they average more than
// 447 bits, which photographs do not come near -- and the rest come again in the next pass.
template <int NT>
__global__ __launch_bounds__(NT) void k_jpeg_enc_huff(const EncJob* __restrict__ jobs, const EncTables* __restrict__ tabs, uint32_t* __restrict__ result,
                                                      uint32_t* __restrict__ cursor, uint8_t* __restrict__ compact, uint32_t compact_cap) {
    __shared__ uint32_t s_win[ENC_WIN_WORDS];
    __shared__ int s_part[NT / 64], s_cnt[NT / 64], s_top[NT / 64];
    __shared__ uint32_t s_carry;
    __shared__ uint32_t s_huff[1024];
    const EncJob& J = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += NT) s_huff[i] = tabs->huff[i >> 8][i & 255];
    for (int i = tid; i < ENC_WIN_WORDS; i += NT) s_win[i] = 0;
    __syncthreads();
    int carry = 0;                      // bits of an unfinished byte at the top of s_win[0]
    long long out_pos = 0;              // bytes of the segment so far (counted even when they no longer fit)
    int taken = 0;
    for (int b0 = 0; b0 < J.nblocks; b0 += taken) {
        const int b = b0 + tid;
        const bool live = b < J.nblocks;
        int bits = 0, dc = 0, last_dc = 0;
        uint32_t cw[32];
        int tb = 0;
        if (live) {
            const int mcu = b / J.bpm, j = b - mcu * J.bpm;
            const uint4* blk = (const uint4*)(J.coef + (size_t)b * 64);
#pragma unroll
            for (int v = 0; v < 8; v++) {
                const uint4 q = blk[v];
                cw[4 * v] = q.x; cw[4 * v + 1] = q.y; cw[4 * v + 2] = q.z; cw[4 * v + 3] = q.w;
            }
            dc = enc_dc_of(J, mcu, j);
            if (J.bpm == 1) last_dc = mcu > 0 ? enc_dc_of(J, mcu - 1, 0) : 0;
            else if (j >= 4) { last_dc = mcu > 0 ? enc_dc_of(J, mcu - 1, j) : 0; tb = 512; }
            else last_dc = j > 0 ? enc_dc_of(J, mcu, j - 1) : (mcu > 0 ? enc_dc_of(J, mcu - 1, 3) : 0);
            bits = enc_code_block<false>(cw, last_dc, dc, &s_huff[tb], nullptr);
        }
        int total;
        const int incl = enc_block_scan<NT>(bits, s_part, &total);
        bool mine = live;
        taken = min(NT, J.nblocks - b0);
        if (NT > 256) {
            // the blocks whose code still fits the window form a prefix (the offsets grow): how many, and where their bits end
            mine = live && carry + incl <= ENC_WIN_WORDS * 32 - 64;
            const unsigned long long bal = __ballot(mine);
            const int cnt = __popcll(bal), top = __shfl(incl, cnt > 0 ? cnt - 1 : 0);
            if ((tid & 63) == 0) { s_cnt[tid >> 6] = cnt; s_top[tid >> 6] = cnt > 0 ? top : 0; }
            __syncthreads();
            taken = 0; total = 0;
#pragma unroll
            for (int k = 0; k < NT / 64; k++) { taken += s_cnt[k]; total = max(total, s_top[k]); }
        }
        if (mine) {
            EncPut P;
            P.start(s_win, carry + incl - bits);
            enc_code_block<true>(cw, last_dc, dc, &s_huff[tb], &P);
            P.finish();
        }
        __syncthreads();
        int nbits = carry + total;
        const bool last = b0 + taken >= J.nblocks;
        if (last && (nbits & 7)) {                              // flush_bits: ones up to the byte boundary
            if (tid == 0) {
                const int lb = nbits & 7;
                atomicOr(&s_win[nbits >> 5], ((1u << (8 - lb)) - 1) << (24 - ((nbits >> 3) & 3) * 8));
            }
            nbits = (nbits + 7) & ~7;
            __syncthreads();
        }
        const int nbytes = nbits >> 3;
        // whole bytes leave with a 00 stuffed behind every FF: a word-aligned run of bytes per thread
        const int per = ((nbytes + NT - 1) / NT + 3) & ~3;
        const int s = min(nbytes, tid * per), e = min(nbytes, s + per);
        int ff = 0;
        for (int i = s; i < e; i++) ff += ((s_win[i >> 2] >> (24 - (i & 3) * 8)) & 0xff) == 0xff;
        int ff_total;
        const int ff_incl = enc_block_scan<NT>(ff, s_part, &ff_total);
        {
            long long at = out_pos + s + (ff_incl - ff);
            for (int i = s; i < e; i++) {
                const uint32_t v = (s_win[i >> 2] >> (24 - (i & 3) * 8)) & 0xff;
                if (at < J.out_cap) J.out[at] = (uint8_t)v;
                at++;
                if (v == 0xff) { if (at < J.out_cap) J.out[at] = 0; at++; }
            }
        }
        out_pos += nbytes + ff_total;
        // the unfinished byte moves to the top of a cleared window
        carry = nbits & 7;
        if (tid == 0) s_carry = carry ? ((s_win[nbytes >> 2] >> (24 - (nbytes & 3) * 8)) & 0xff) << 24 : 0u;
        __syncthreads();
        const int used = (nbits + 31) >> 5;
        for (int i = tid; i < used; i += NT) s_win[i] = 0;
        __syncthreads();
        if (tid == 0) s_win[0] = s_carry;
        __syncthreads();
    }
    const uint32_t len = (uint32_t)(out_pos + 2);
    const bool fits = out_pos + 2 <= J.out_cap;
    if (tid == 0) {
        if (fits) { J.out[out_pos] = 0xff; J.out[out_pos + 1] = 0xd9; }
        uint32_t at = 0xffffffffu;
        if (fits) {
            at = atomicAdd(cursor, (len + 15u) & ~15u);
            if (at > compact_cap || len > compact_cap - at) at = 0xffffffffu;
        }
        result[4 * blockIdx.x] = len;
        result[4 * blockIdx.x + 1] = fits ? 0u : 1u;
        result[4 * blockIdx.x + 2] = at;
        s_carry = at;
    }
    __threadfence_block();
    __syncthreads();                                                // (also: every lane's bytes of the segment are written)
    const uint32_t at = s_carry;
    if (at != 0xffffffffu) {
        __threadfence();
        const uint4* src = (const uint4*)J.out;                     // both 16-byte aligned
        uint4* dst = (uint4*)(compact + at);
        for (uint32_t i = tid; i < (len + 15u) / 16u; i += NT) dst[i] = src[i];
    }
}

// ---------------------------------------------------------------- large frames: many workgroups per image
// One workgroup walking a whole 1080p frame takes 1.6 ms (48 passes of 1024 blocks).  A large frame is cut instead into
// segments of 256 blocks, one workgroup each, in two launches:
//   k_jpeg_enc_pack   sizes its blocks, publishes the segment's bit count, adds up the counts of the segments before it (they
//                     run at the same time: a segment's count does not depend on anything), and ORs its code into the
//                     image's UNSTUFFED stream at that bit offset -- through the LDS window, whole words to memory, only the
//                     two words it may share with its neighbours as atomics;
//   k_jpeg_enc_stuff  one workgroup per 16 KB of that stream: counts its FF bytes, publishes, adds up the chunks before it,
//                     and writes its bytes with the 00s behind their FFs where they belong; the chunk that holds the last
//                     byte adds EOI and the verdict.
// Workgroups take a ticket when they start (never blockIdx): the ones a workgroup waits for are then always running.
struct EncSeg { int job, local; };

// A record of these chains is ONE word: the value with bit 31 = "it is there" -- one relaxed atomic store to publish it, one
// relaxed atomic load (per look) to get it, no fence on either side: nothing else is handed over with it.  (As a value word
// and a flag word, released and acquired, a reader made two trips to memory with a cache invalidate between them and the
// writer a write-back before its flag: three such exchanges a thumbnail.)
__device__ __forceinline__ void enc_publish(uint32_t* rec, uint32_t value) {
    __hip_atomic_store(rec, (value & 0x7fffffffu) | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t enc_await(const uint32_t* rec) {           // the word (bit 31 clear: it never came)
    uint32_t v = 0;
    for (int spin = 0; spin < (1 << 22); spin++) {
        v = __hip_atomic_load(rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v & 0x80000000u) break;
        __builtin_amdgcn_s_sleep(8);
    }
    return v;
}

__device__ __forceinline__ uint32_t enc_sum_before(const uint32_t* __restrict__ rec, int upto, int* s_part, uint32_t* verdict) {
    // sum of rec[2 * p] over p < upto, each read once it is there (enc_await: the same bounded, relaxed poll as the
    // decoder's chain); all threads of the block get the sum.  A wait that runs out says so in the image's verdict word
    // (bit 1): what is written from a wrong offset is not a file, and the host must not hand it on as one.
    uint32_t acc = 0;
    for (int p = threadIdx.x; p < upto; p += 256) {
        const uint32_t v = enc_await(&rec[2 * p]);
        if (!(v & 0x80000000u)) atomicOr(verdict, 2u);
        acc += v & 0x7fffffffu;
    }
    int total;
    (void)enc_block_scan<256>((int)acc, s_part, &total);
    return (uint32_t)total;
}

__global__ __launch_bounds__(256) void k_jpeg_enc_pack(const EncJob* __restrict__ jobs, const EncSeg* __restrict__ map, const EncTables* __restrict__ tabs,
                                                       uint32_t* __restrict__ result, uint32_t* __restrict__ ticket) {
    __shared__ uint32_t s_win[ENC_WIN_WORDS];
    __shared__ int s_part[4];
    __shared__ uint32_t s_ticket;
    __shared__ uint32_t s_huff[1024];
    const int tid = threadIdx.x;
    if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
    for (int i = tid; i < 1024; i += 256) s_huff[i] = tabs->huff[i >> 8][i & 255];
    for (int i = tid; i < ENC_WIN_WORDS; i += 256) s_win[i] = 0;
    __syncthreads();
    const EncSeg me = map[s_ticket];
    const EncJob& J = jobs[me.job];
    const int sg = me.local, b = sg * 256 + tid;
    const bool live = b < J.nblocks;
    int bits = 0, dc = 0, last_dc = 0, tb = 0;
    uint32_t cw[32];
    if (live) {
        const int mcu = b / J.bpm, j = b - mcu * J.bpm;
        const uint4* blk = (const uint4*)(J.coef + (size_t)b * 64);
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const uint4 q = blk[v];
            cw[4 * v] = q.x; cw[4 * v + 1] = q.y; cw[4 * v + 2] = q.z; cw[4 * v + 3] = q.w;
        }
        dc = enc_dc_of(J, mcu, j);
        if (J.bpm == 1) last_dc = mcu > 0 ? enc_dc_of(J, mcu - 1, 0) : 0;
        else if (j >= 4) { last_dc = mcu > 0 ? enc_dc_of(J, mcu - 1, j) : 0; tb = 512; }
        else last_dc = j > 0 ? enc_dc_of(J, mcu, j - 1) : (mcu > 0 ? enc_dc_of(J, mcu - 1, 3) : 0);
        bits = enc_code_block<false>(cw, last_dc, dc, &s_huff[tb], nullptr);
    }
    int total;
    const int incl = enc_block_scan<256>(bits, s_part, &total);
    if (tid == 0) {
        enc_publish(&J.seg_rec[2 * sg], (uint32_t)total);
    }
    __syncthreads();                                                // (s_part is free again)
    const uint32_t base = enc_sum_before(J.seg_rec, sg, s_part, &result[4 * me.job + 1]);    // bits of the image in front of this segment
    const int lead = (int)(base & 31u);
    if (live) {
        EncPut P;
        P.start(s_win, lead + incl - bits);
        enc_code_block<true>(cw, last_dc, dc, &s_huff[tb], &P);
        P.finish();
    }
    __syncthreads();
    int nbits = lead + total;
    if (sg == J.nseg - 1) {                                         // the image's last segment: flush_bits' ones, and the stream's length
        const uint32_t all = base + (uint32_t)total;
        if (tid == 0) {
            if (all & 7u) atomicOr(&s_win[nbits >> 5], ((1u << (8 - (all & 7u))) - 1) << (24 - ((nbits >> 3) & 3) * 8));
            J.ubytes[0] = (all + 7u) >> 3;
        }
        nbits = (nbits + 7) & ~7;
        __syncthreads();
    }
    const int nwords = (nbits + 31) >> 5;
    uint32_t* U = J.ustream + (base >> 5);
    for (int i = tid; i < nwords; i += 256) {
        const uint32_t v = s_win[i];
        if (i == 0 || i == nwords - 1) { if (v) atomicOr(&U[i], v); }
        else U[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_jpeg_enc_stuff(const EncJob* __restrict__ jobs, const EncSeg* __restrict__ map, uint32_t* __restrict__ result,
                                                        uint32_t* __restrict__ ticket) {
    __shared__ int s_part[4];
    __shared__ uint32_t s_ticket;
    const int tid = threadIdx.x;
    if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    const EncSeg me = map[s_ticket];
    const EncJob& J = jobs[me.job];
    const int ck = me.local;
    const uint32_t nbytes = J.ubytes[0];
    const uint32_t c0 = (uint32_t)ck * 16384u;
    // a chunk past the stream's end (the grid is sized for the worst case) still publishes: nothing in it
    const uint32_t s = min(nbytes, c0 + (uint32_t)tid * 64u), e = min(nbytes, s + 64u);
    const uint32_t* U = J.ustream;
    int ff = 0;
    for (uint32_t i = s; i < e; i++) ff += ((U[i >> 2] >> (24 - (i & 3) * 8)) & 0xff) == 0xff;
    int total;
    const int incl = enc_block_scan<256>(ff, s_part, &total);
    if (tid == 0) {
        enc_publish(&J.chunk_rec[2 * ck], (uint32_t)total);
    }
    __syncthreads();
    if (c0 >= nbytes) return;                                       // (uniform)
    const uint32_t before = enc_sum_before(J.chunk_rec, ck, s_part, &result[4 * me.job + 1]);
    long long at = (long long)s + before + (incl - ff);
    for (uint32_t i = s; i < e; i++) {
        const uint32_t v = (U[i >> 2] >> (24 - (i & 3) * 8)) & 0xff;
        if (at < J.out_cap) J.out[at] = (uint8_t)v;
        at++;
        if (v == 0xff) { if (at < J.out_cap) J.out[at] = 0; at++; }
    }
    if (e == nbytes && s < e) {                                     // the lane that wrote the stream's last byte
        const long long len = at + 2;
        const bool fits = len <= J.out_cap;
        if (fits) { J.out[at] = 0xff; J.out[at + 1] = 0xd9; }
        const int r = me.job;
        result[4 * r] = (uint32_t)len;
        if (!fits) atomicOr(&result[4 * r + 1], 1u);
        result[4 * r + 2] = 0xffffffffu;                            // fetched from its own region, not from the compact area
    }
}

// ---------------------------------------------------------------- thumbnails: a few workgroups per image (round 5)
// k_jpeg_enc_huff<1024> gives a 224 x 168 thumbnail's 924 blocks ONE workgroup: sixteen waves on one compute unit, four to a
// SIMD, walk their blocks twice (35 us), and a request's answer waits for exactly that.  Here the image is cut into hseg
// segments of hper <= 256 blocks, a workgroup (four waves, one per SIMD) each, in ONE launch -- k_jpeg_enc_pack and
// k_jpeg_enc_stuff folded together, because a thumbnail's whole stream is a few KB and a second launch costs more than it:
//   size the blocks, publish the segment's bit count, add up the counts before it (they run at the same time);
//   emit the code into the LDS window at that bit offset within its first byte;
//   the byte a segment shares with the next one belongs to the NEXT one (it holds that byte's last bit): a segment publishes its
//   unfinished last byte and ORs its predecessor's into its first;
//   count the FF bytes among the bytes it owns, publish, add up the counts before it, write its bytes with the 00s in place;
//   the last segment pads with 1-bits and appends EOI; whoever finishes LAST (a counter) copies the file into the compact area.
// Workgroups take a ticket when they start, so the ones a workgroup waits for are always running.
__global__ __launch_bounds__(256) void k_jpeg_enc_huff_seg(const EncJob* __restrict__ jobs, const EncSeg* __restrict__ map, const EncTables* __restrict__ tabs,
                                                           uint32_t* __restrict__ result, uint32_t* __restrict__ cursor, uint8_t* __restrict__ compact,
                                                           uint32_t compact_cap, uint32_t* __restrict__ ticket) {
    __shared__ uint32_t s_win[ENC_WIN_WORDS];
    __shared__ uint32_t s_huff[1024];
    __shared__ int s_part[4];
    __shared__ uint32_t s_word;
    const int tid = threadIdx.x;
    if (tid == 0) s_word = atomicAdd(ticket, 1u);
    for (int i = tid; i < 1024; i += 256) s_huff[i] = tabs->huff[i >> 8][i & 255];
    for (int i = tid; i < ENC_WIN_WORDS; i += 256) s_win[i] = 0;
    __syncthreads();
    const EncSeg me = map[s_word];
    const EncJob& J = jobs[me.job];
    const int sg = me.local, nseg = J.hseg;
    const int b0 = sg * J.hper, b = b0 + tid;
    const bool live = tid < J.hper && b < J.nblocks;
    uint32_t* recA = J.hrec;
    uint32_t* recC = J.hrec + 2 * nseg;
    uint32_t* recB = J.hrec + 4 * nseg;
    uint32_t* verdict = &result[4 * me.job + 1];
    int bits = 0, dc = 0, last_dc = 0, tb = 0;
    uint32_t cw[32];
    if (live) {
        const int mcu = b / J.bpm, j = b - mcu * J.bpm;
        const uint4* blk = (const uint4*)(J.coef + (size_t)b * 64);
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const uint4 q = blk[v];
            cw[4 * v] = q.x; cw[4 * v + 1] = q.y; cw[4 * v + 2] = q.z; cw[4 * v + 3] = q.w;
        }
        dc = enc_dc_of(J, mcu, j);
        if (J.bpm == 1) last_dc = mcu > 0 ? enc_dc_of(J, mcu - 1, 0) : 0;
        else if (j >= 4) { last_dc = mcu > 0 ? enc_dc_of(J, mcu - 1, j) : 0; tb = 512; }
        else last_dc = j > 0 ? enc_dc_of(J, mcu, j - 1) : (mcu > 0 ? enc_dc_of(J, mcu - 1, 3) : 0);
        bits = enc_code_block<false>(cw, last_dc, dc, &s_huff[tb], nullptr);
    }
    int total;
    const int incl = enc_block_scan<256>(bits, s_part, &total);
    if (tid == 0) {
        enc_publish(&recA[2 * sg], (uint32_t)total);
    }
    __syncthreads();                                                // (s_part is free again)
    const uint32_t base = enc_sum_before(recA, sg, s_part, verdict);     // bits of the image in front of this segment
    const int lead = (int)(base & 7u);
    if (live) {
        EncPut P;
        P.start(s_win, lead + incl - bits);
        enc_code_block<true>(cw, last_dc, dc, &s_huff[tb], &P);
        P.finish();
    }
    __syncthreads();
    const bool last = sg == nseg - 1;
    int nbits = lead + total;
    int nown = nbits >> 3;                                          // whole bytes of the window: byte 0 is the file's byte base >> 3
    const int rem = nbits & 7;
    if (tid == 0) {
        if (last) {
            if (rem) { atomicOr(&s_win[nown >> 2], ((1u << (8 - rem)) - 1) << (24 - (nown & 3) * 8)); }     // flush_bits: ones up to the byte boundary
        } else {
            const uint32_t part = rem ? (s_win[nown >> 2] >> (24 - (nown & 3) * 8)) & 0xffu : 0u;          // the byte the next segment finishes
            enc_publish(&recB[2 * sg], part);
        }
        if (lead) {                                                 // the top bits of this segment's first byte are its predecessor's
            const uint32_t v = enc_await(&recB[2 * (sg - 1)]);
            if (!(v & 0x80000000u)) atomicOr(verdict, 2u);
            atomicOr(&s_win[0], (v & 0xffu) << 24);
        }
    }
    if (last && rem) nown++;
    __syncthreads();
    // the bytes this segment owns leave with a 00 behind every FF: a word-aligned run of bytes per thread
    const int per = ((nown + 255) / 256 + 3) & ~3;
    const int s = min(nown, tid * per), e = min(nown, s + per);
    int ff = 0;
    for (int i = s; i < e; i++) ff += ((s_win[i >> 2] >> (24 - (i & 3) * 8)) & 0xff) == 0xff;
    int ff_total;
    const int ff_incl = enc_block_scan<256>(ff, s_part, &ff_total);
    if (tid == 0) {
        enc_publish(&recC[2 * sg], (uint32_t)ff_total);
    }
    __syncthreads();
    const uint32_t ff_before = enc_sum_before(recC, sg, s_part, verdict);
    {
        long long at = (long long)(base >> 3) + ff_before + s + (ff_incl - ff);
        for (int i = s; i < e; i++) {
            const uint32_t v = (s_win[i >> 2] >> (24 - (i & 3) * 8)) & 0xff;
            if (at < J.out_cap) J.out[at] = (uint8_t)v;
            at++;
            if (v == 0xff) { if (at < J.out_cap) J.out[at] = 0; at++; }
        }
    }
    if (last && tid == 0) {
        const long long out_pos = (long long)(base >> 3) + ff_before + nown + ff_total;
        const bool fits = out_pos + 2 <= J.out_cap;
        if (fits) { J.out[out_pos] = 0xff; J.out[out_pos + 1] = 0xd9; }
        result[4 * me.job] = (uint32_t)(out_pos + 2);
        if (!fits) atomicOr(verdict, 1u);
    }
    // whoever finishes last has every byte of the file behind it: it puts the copy the host fetches into the compact area
    __threadfence();
    __syncthreads();
    if (tid == 0) s_word = __hip_atomic_fetch_add(&result[4 * me.job + 3], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_word != (uint32_t)(nseg - 1)) return;                     // (uniform)
    __threadfence();
    const uint32_t len = __hip_atomic_load(&result[4 * me.job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    if (tid == 0) {
        uint32_t at = 0xffffffffu;
        if (ok) {
            at = atomicAdd(cursor, (len + 15u) & ~15u);
            if (at > compact_cap || len > compact_cap - at) at = 0xffffffffu;
        }
        result[4 * me.job + 2] = at;
        s_word = at;
    }
    __syncthreads();
    const uint32_t at = s_word;
    if (at != 0xffffffffu) {
        const uint4* src = (const uint4*)J.out;                     // both 16-byte aligned
        uint4* dst = (uint4*)(compact + at);
        for (uint32_t i = tid; i < (len + 15u) / 16u; i += 256) dst[i] = src[i];
    }
}

// ---------------------------------------------------------------- host
const EncTables& enc_static_tables(EncTables* scratch, int quality) {
    static uint32_t huff[4][256];
    static std::once_flag once;
    std::call_once(once, [] {
        const uint8_t* bits[4] = {JSTD_BITS_DC_LUMA, JSTD_BITS_AC_LUMA, JSTD_BITS_DC_CHROMA, JSTD_BITS_AC_CHROMA};
        const uint8_t* vals[4] = {JSTD_VALS_DC, JSTD_VALS_AC_LUMA, JSTD_VALS_DC, JSTD_VALS_AC_CHROMA};
        for (int t = 0; t < 4; t++) {                          // jchuff.c jpeg_make_c_derived_tbl: canonical codes by length
            std::memset(huff[t], 0, sizeof(huff[t]));
            uint32_t code = 0;
            int p = 0;
            for (int l = 1; l <= 16; l++) {
                for (int i = 0; i < bits[t][l - 1]; i++, p++) huff[t][vals[t][p]] = (code++ << 8) | (uint32_t)l;
                code <<= 1;
            }
        }
    });
    std::memcpy(scratch->huff, huff, sizeof(huff));
    // jcparam.c jpeg_quality_scaling + jpeg_add_quant_table(force_baseline = TRUE); OpenCV clamps the quality to 0..100
    int q = quality <= 0 ? 1 : quality > 100 ? 100 : quality;
    q = q < 50 ? 5000 / q : 200 - q * 2;
    for (int t = 0; t < 2; t++)
        for (int i = 0; i < 64; i++) {
            long v = ((long)(t ? JSTD_Q_CHROMA : JSTD_Q_LUMA)[i] * q + 50L) / 100L;
            v = v <= 0 ? 1 : v > 255 ? 255 : v;
            scratch->q[t][JSTD_ZIGZAG[i]] = (uint16_t)v;
        }
    return *scratch;
}

void enc_put(std::vector<uint8_t>& f, int b) { f.push_back((uint8_t)b); }
void enc_put2(std::vector<uint8_t>& f, int v) { enc_put(f, v >> 8); enc_put(f, v & 255); }

// everything of the file in front of the entropy-coded segment, in jcmarker.c's order
std::vector<uint8_t> enc_headers(int w, int h, int nc, const EncTables& T) {
    std::vector<uint8_t> f;
    f.reserve(640);
    enc_put2(f, 0xFFD8);
    enc_put2(f, 0xFFE0); enc_put2(f, 16);
    for (const char ch : {'J', 'F', 'I', 'F', '\0'}) enc_put(f, ch);
    enc_put(f, 1); enc_put(f, 1); enc_put(f, 0); enc_put2(f, 1); enc_put2(f, 1); enc_put(f, 0); enc_put(f, 0);
    const int ntab = nc == 3 ? 2 : 1;
    for (int t = 0; t < ntab; t++) {
        enc_put2(f, 0xFFDB); enc_put2(f, 67); enc_put(f, t);
        for (int i = 0; i < 64; i++) enc_put(f, T.q[t][JSTD_ZIGZAG[i]]);
    }
    enc_put2(f, 0xFFC0); enc_put2(f, 8 + 3 * nc); enc_put(f, 8); enc_put2(f, h); enc_put2(f, w); enc_put(f, nc);
    for (int k = 0; k < nc; k++) { enc_put(f, k + 1); enc_put(f, k == 0 && nc == 3 ? 0x22 : 0x11); enc_put(f, k ? 1 : 0); }
    const uint8_t* bits[4] = {JSTD_BITS_DC_LUMA, JSTD_BITS_AC_LUMA, JSTD_BITS_DC_CHROMA, JSTD_BITS_AC_CHROMA};
    const uint8_t* vals[4] = {JSTD_VALS_DC, JSTD_VALS_AC_LUMA, JSTD_VALS_DC, JSTD_VALS_AC_CHROMA};
    for (int t = 0; t < 2 * ntab; t++) {
        int n = 0;
        for (int i = 0; i < 16; i++) n += bits[t][i];
        enc_put2(f, 0xFFC4); enc_put2(f, 2 + 1 + 16 + n); enc_put(f, ((t & 1) << 4) | (t >> 1));
        for (int i = 0; i < 16; i++) enc_put(f, bits[t][i]);
        for (int i = 0; i < n; i++) enc_put(f, vals[t][i]);
    }
    enc_put2(f, 0xFFDA); enc_put2(f, 6 + 2 * nc); enc_put(f, nc);
    for (int k = 0; k < nc; k++) { enc_put(f, k + 1); enc_put(f, k ? 0x11 : 0x00); }
    enc_put(f, 0); enc_put(f, 63); enc_put(f, 0);
    return f;
}

struct EncGeom { int nc, mcuw, mcuh, bpm, nblocks; };
bool enc_geom(int w, int h, int c, EncGeom* g) {
    if (w <= 0 || h <= 0 || w > 65500 || h > 65500 || (c != 1 && c != 3 && c != 4)) return false;    // JPEG_MAX_DIMENSION
    g->nc = c == 1 ? 1 : 3;
    const int m = c == 1 ? 8 : 16;
    g->mcuw = (w + m - 1) / m; g->mcuh = (h + m - 1) / m;
    g->bpm = c == 1 ? 1 : 6;
    const long long nb = (long long)g->mcuw * g->mcuh * g->bpm;
    if (nb > (1 << 24)) return false;
    g->nblocks = (int)nb;
    return true;
}

constexpr int ENC_MAX_BATCH = 256;

constexpr int ENC_BIG_BLOCKS = 2048;                            // frames of more block slots than this take the many-workgroup path

// An encode between the call that enqueued it (kernels and the copy of the segments into pinned memory) and the call that
// fetches the files: what the second half needs of the first.
struct EncState {
    std::vector<EncJob> jobs;
    std::vector<int> owner;                                     // job -> index into the caller's arrays
    std::vector<std::vector<uint8_t>> heads;
    std::vector<int> early;                                     // per image: IMP_OK, or what was wrong with it before anything ran
    int count = 0;
    void *coef = nullptr, *out = nullptr, *res = nullptr, *side = nullptr, *aux = nullptr, *hrec = nullptr;
    void *pin = nullptr, *token = nullptr, *mark = nullptr;
    size_t res_bytes = 0, compact_cap = 0;
    bool armed = false;                                         // something is in flight
    void drop() { dev_free(coef); dev_free(out); dev_free(res); dev_free(side); dev_free(aux); dev_free(hrec); coef = out = res = side = aux = hrec = nullptr; }
};

int encode_begin(const impgpu_image* const* images, int count, int quality, EncState& E) {
    hipStream_t s = env_stream();
    EncTables T;
    enc_static_tables(&T, quality);
    static const bool one_wg = ab_env("IMPGPU_JPEG_ENC_ONE_WG") != nullptr;       // A/B: every frame through k_jpeg_enc_huff
    std::vector<EncJob>& jobs = E.jobs;                         // small frames first, then the large ones
    std::vector<int>& owner = E.owner;
    std::vector<std::vector<uint8_t>>& heads = E.heads;
    E.count = count;
    E.early.assign((size_t)count, IMP_OK);
    // Thumbnails of 257 .. 2048 block slots: a few workgroups each (k_jpeg_enc_huff_seg) when the call is a request's or a broker
    // batch's -- up to sixteen frames -- and ONE workgroup of 1024 threads each when it is a queue's worth: measured, same box,
    // the lone call 72 -> 69 us at 224 x 168 and 86 -> 71 at 224 x 224 (1176 blocks: two passes of the one workgroup), the
    // broker's four lanes 16.95 -> 17.4 k requests/s at 16 workers and 23.7 -> 24.6 k at 32; but 64 frames per call 447 -> 504 us
    // and the eight-thread stream 46.8 -> 44.6 k.  IMPGPU_JPEG_ENC_SEG=0 | 1 forces either (A/B, read per call).
    const char* segenv = std::getenv("IMPGPU_JPEG_ENC_SEG");
    const bool seg_on = !one_wg && (segenv ? segenv[0] != '0' : count <= 16);
    for (int pass = 0; pass < 3; pass++)                        // 0: frames of up to 256 block slots (or all small ones), 1: up to 2048 in segments, 2: the large ones
        for (int i = 0; i < count; i++) {
            const impgpu_image* im = images[i];
            EncGeom g;
            if (pass == 0) {
                E.early[(size_t)i] = (!im || !enc_geom(im->w, im->h, im->c, &g)) ? IMP_ERROR_INVALID_ARGS : IMP_OK;
                if (E.early[(size_t)i] == IMP_OK && (size_t)g.nblocks * ENC_BLOCK_BYTES + 16 > 0x7fffffffu) E.early[(size_t)i] = IMP_ERROR_INVALID_ARGS;
            }
            if (E.early[(size_t)i] != IMP_OK) continue;
            (void)enc_geom(im->w, im->h, im->c, &g);
            const bool big = !one_wg && g.nblocks > ENC_BIG_BLOCKS && (uint64_t)g.nblocks * 1658u < (1ull << 32);
            const bool mid = !big && seg_on && g.nblocks > 256;
            if ((big ? 2 : mid ? 1 : 0) != pass) continue;
            EncJob J{};
            J.src = im->d; J.w = im->w; J.h = im->h; J.c = im->c; J.step = im->step;
            J.mcuw = g.mcuw; J.mcuh = g.mcuh; J.bpm = g.bpm; J.nblocks = g.nblocks;
            J.lbw = (im->w + 7) / 8; J.lbh = (im->h + 7) / 8; J.chh = (im->h + 1) / 2;
            J.out_cap = (int)((size_t)g.nblocks * ENC_BLOCK_BYTES + 16);
            if (big) {
                J.nseg = (g.nblocks + 255) / 256;
                J.nchunk = (int)(((size_t)g.nblocks * 1658 / 8 + 8 + 16383) / 16384);
            }
            if (mid) {
                J.hseg = (g.nblocks + 255) / 256;
                J.hper = (g.nblocks + J.hseg - 1) / J.hseg;     // (even shares: no segment is a handful of blocks)
            }
            jobs.push_back(J);
            owner.push_back(i);
            heads.push_back(enc_headers(im->w, im->h, g.nc, T));
        }
    const int nj = (int)jobs.size();
    if (!nj) return IMP_OK;
    int nsmall = 0, ntiny = 0;
    while (nsmall < nj && jobs[nsmall].nseg == 0) nsmall++;
    while (ntiny < nsmall && jobs[ntiny].hseg == 0) ntiny++;
    // device memory: coefficient blocks | segments | for the large frames: unstuffed streams, records, tickets (one area, zeroed)
    std::vector<EncMap> map;
    std::vector<EncSeg> pack_map, stuff_map, hseg_map;
    size_t hrec_words = 4;                                       // [0] = k_jpeg_enc_huff_seg's ticket, then 6 words per segment
    std::vector<size_t> o_hrec((size_t)jobs.size(), 0);
    size_t coef_bytes = 0, out_bytes = 0, aux_bytes = 16;        // aux: [0] [1] = the two tickets
    std::vector<size_t> o_coef((size_t)nj), o_out((size_t)nj), o_u((size_t)nj), o_seg((size_t)nj), o_chk((size_t)nj), o_ub((size_t)nj);
    for (int k = 0; k < nj; k++) {
        const EncJob& J = jobs[k];
        o_coef[k] = coef_bytes; coef_bytes += (size_t)J.nblocks * 128;
        o_out[k] = out_bytes; out_bytes += ((size_t)J.out_cap + 255) & ~size_t(255);
        for (int b = 0; b < J.nblocks; b += ENC_BLOCKS_PER_WG) map.push_back(EncMap{k, b});
        if (J.hseg) {
            o_hrec[(size_t)k] = hrec_words; hrec_words += (size_t)J.hseg * 6;
            for (int g = 0; g < J.hseg; g++) hseg_map.push_back(EncSeg{k, g});
        }
        if (J.nseg) {
            o_ub[k] = aux_bytes; aux_bytes += 16;
            o_seg[k] = aux_bytes; aux_bytes += (size_t)J.nseg * 8;
            o_chk[k] = aux_bytes; aux_bytes += (size_t)J.nchunk * 8;
            aux_bytes = (aux_bytes + 255) & ~size_t(255);
            o_u[k] = aux_bytes; aux_bytes += ((size_t)J.nchunk * 16384 + 64 + 255) & ~size_t(255);
            for (int g = 0; g < J.nseg; g++) pack_map.push_back(EncSeg{k, g});
            for (int g = 0; g < J.nchunk; g++) stuff_map.push_back(EncSeg{k, g});
        }
    }
    // what the host will fetch in its one copy: 32 bytes per block slot (a photograph at quality 90 needs about 13) and the
    // results in front; a batch that needs more than that costs a second copy and wait for the segments that did not fit.
    // (Large frames are fetched from their own regions once their length is known: they are worth a copy of their own.)
    size_t compact_cap = 0;
    for (int k = 0; k < nsmall; k++) compact_cap += std::min((size_t)jobs[k].out_cap, (size_t)jobs[k].nblocks * 32 + 256);
    compact_cap = (compact_cap + 255) & ~size_t(255);
    if (compact_cap > 0xfffffff0u) compact_cap = 0xfffffff0u & ~size_t(255);
    const size_t res_bytes = (((size_t)nj * 4 + 1) * 4 + 255) & ~size_t(255);
    E.res_bytes = res_bytes; E.compact_cap = compact_cap;
    void *&coef = E.coef, *&out = E.out, *&res = E.res, *&side = E.side, *&aux = E.aux;      // res = results | compact area
    if (!hseg_map.empty()) if (int rc = dev_alloc(hrec_words * 4, &E.hrec)) return rc;
    if (int rc = dev_alloc(coef_bytes, &coef)) { E.drop(); return rc; }
    if (int rc = dev_alloc(out_bytes, &out)) { E.drop(); return rc; }
    if (int rc = dev_alloc(res_bytes + compact_cap, &res)) { E.drop(); return rc; }
    if (nsmall < nj) if (int rc = dev_alloc(aux_bytes, &aux)) { E.drop(); return rc; }
    for (int k = 0; k < nj; k++) {
        EncJob& J = jobs[k];
        J.coef = (short*)((uint8_t*)coef + o_coef[k]);
        J.out = (uint8_t*)out + o_out[k];
        if (J.hseg) J.hrec = (uint32_t*)E.hrec + o_hrec[(size_t)k];
        if (J.nseg) {
            J.ustream = (uint32_t*)((uint8_t*)aux + o_u[k]);
            J.seg_rec = (uint32_t*)((uint8_t*)aux + o_seg[k]);
            J.chunk_rec = (uint32_t*)((uint8_t*)aux + o_chk[k]);
            J.ubytes = (uint32_t*)((uint8_t*)aux + o_ub[k]);
        }
    }
    // side blob: tables | jobs | map | segment map | chunk map
    auto up16 = [](size_t v) { return (v + 15) & ~size_t(15); };
    const size_t o_jobs = up16(sizeof(EncTables)), o_map = o_jobs + up16(jobs.size() * sizeof(EncJob)),
                 o_pack = o_map + up16(map.size() * sizeof(EncMap)), o_stuff = o_pack + up16(pack_map.size() * sizeof(EncSeg)),
                 o_hseg = o_stuff + up16(stuff_map.size() * sizeof(EncSeg));
    std::vector<uint8_t> blob(o_hseg + hseg_map.size() * sizeof(EncSeg) + 16);
    std::memcpy(blob.data(), &T, sizeof(T));
    std::memcpy(blob.data() + o_jobs, jobs.data(), jobs.size() * sizeof(EncJob));
    std::memcpy(blob.data() + o_map, map.data(), map.size() * sizeof(EncMap));
    if (!pack_map.empty()) std::memcpy(blob.data() + o_pack, pack_map.data(), pack_map.size() * sizeof(EncSeg));
    if (!stuff_map.empty()) std::memcpy(blob.data() + o_stuff, stuff_map.data(), stuff_map.size() * sizeof(EncSeg));
    if (!hseg_map.empty()) std::memcpy(blob.data() + o_hseg, hseg_map.data(), hseg_map.size() * sizeof(EncSeg));
    if (int rc = upload_small(blob.data(), blob.size(), &side, s)) { E.drop(); return rc; }
    const uint8_t* sd = (const uint8_t*)side;
    const EncJob* djobs = (const EncJob*)(sd + o_jobs);
    uint32_t* cursor = (uint32_t*)res + (size_t)nj * 4;
    hipError_t e = hipSuccess;                                  // (the verdict words and the compact area's cursor are cleared by k_jpeg_enc_blocks)
    if (aux) e = hipMemsetAsync(aux, 0, aux_bytes, s);
    hipLaunchKernelGGL(k_jpeg_enc_blocks, dim3((unsigned)map.size()), dim3(256), 0, s, djobs, (const EncMap*)(sd + o_map), (const EncTables*)sd,
                       (uint32_t*)res, (int)(res_bytes / 4), (uint32_t*)E.hrec, E.hrec ? (int)hrec_words : 0);
    if (ntiny) {                                                // (jobs [0, ntiny): one workgroup each)
        bool wide = false;                                      // any frame of more than 256 block slots (IMPGPU_JPEG_ENC_SEG=0): 1024 per pass
        for (int k = 0; k < ntiny; k++) wide = wide || jobs[k].nblocks > 256;
        if (wide) hipLaunchKernelGGL(k_jpeg_enc_huff<1024>, dim3((unsigned)ntiny), dim3(1024), 0, s, djobs, (const EncTables*)sd, (uint32_t*)res, cursor,
                                     (uint8_t*)res + res_bytes, (uint32_t)compact_cap);
        else hipLaunchKernelGGL(k_jpeg_enc_huff<256>, dim3((unsigned)ntiny), dim3(256), 0, s, djobs, (const EncTables*)sd, (uint32_t*)res, cursor,
                                (uint8_t*)res + res_bytes, (uint32_t)compact_cap);
    }
    if (!hseg_map.empty())
        hipLaunchKernelGGL(k_jpeg_enc_huff_seg, dim3((unsigned)hseg_map.size()), dim3(256), 0, s, djobs, (const EncSeg*)(sd + o_hseg), (const EncTables*)sd,
                           (uint32_t*)res, cursor, (uint8_t*)res + res_bytes, (uint32_t)compact_cap, (uint32_t*)E.hrec);
    if (nsmall < nj) {
        hipLaunchKernelGGL(k_jpeg_enc_pack, dim3((unsigned)pack_map.size()), dim3(256), 0, s, djobs, (const EncSeg*)(sd + o_pack), (const EncTables*)sd, (uint32_t*)res, (uint32_t*)aux);
        hipLaunchKernelGGL(k_jpeg_enc_stuff, dim3((unsigned)stuff_map.size()), dim3(256), 0, s, djobs, (const EncSeg*)(sd + o_stuff), (uint32_t*)res, (uint32_t*)aux + 1);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess)
        if (const int rc = stage_begin(res_bytes + compact_cap, &E.pin, &E.token)) { (void)hipStreamSynchronize(s); E.drop(); return rc; }    // (its own text: no pinned buffer free)
    if (e == hipSuccess) e = hipMemcpyAsync(E.pin, res, res_bytes + compact_cap, hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) { set_error("jpeg encode", e); (void)hipStreamSynchronize(s); E.drop(); return IMP_ERROR_DEVICE; }
    stage_hold(E.token, true);                                  // (the answers are read out of it by encode_finish, possibly calls later)
    if (int rc = lane_mark(&E.mark)) { (void)hipStreamSynchronize(s); stage_hold(E.token, false); E.drop(); return rc; }
    E.armed = true;
    return IMP_OK;
}

// The second half: waits for THAT encode (not for what the thread enqueued behind it), hands the files out.
int encode_finish(EncState& E, unsigned char* const* outs, const size_t* caps, size_t* lens, int* codes) {
    hipStream_t s = env_stream();
    for (int i = 0; i < E.count; i++) { lens[i] = 0; codes[i] = E.early[(size_t)i] != IMP_OK ? E.early[(size_t)i] : outs[i] ? IMP_OK : IMP_ERROR_INVALID_ARGS; }
    if (!E.armed) return IMP_OK;
    E.armed = false;
    const std::vector<EncJob>& jobs = E.jobs;
    const std::vector<int>& owner = E.owner;
    const std::vector<std::vector<uint8_t>>& heads = E.heads;
    const int nj = (int)jobs.size();
    void* mark = E.mark;
    E.mark = nullptr;
    if (int rc = lane_wait_mark(mark)) { (void)hipStreamSynchronize(s); stage_hold(E.token, false); E.drop(); return rc; }
    const uint32_t* seg = (const uint32_t*)E.pin;
    const uint8_t* compact = (const uint8_t*)E.pin + E.res_bytes;
    size_t more = 0;
    std::vector<size_t> at2((size_t)nj, 0);
    for (int k = 0; k < nj; k++) {
        const int i = owner[k];
        if (codes[i] != IMP_OK) continue;
        lens[i] = heads[k].size() + seg[4 * k];
        if (seg[4 * k + 1]) {
            codes[i] = IMP_ERROR_DEVICE;
            set_error_text((seg[4 * k + 1] & 2u) ? "jpeg encode: a wait between workgroups ran out" : "jpeg encode: a segment outgrew its bound");
            continue;
        }
        if (lens[i] > caps[i]) { codes[i] = IMP_ERROR_MALLOC_FAILED; continue; }       // lens[i] says what it takes
        if (seg[4 * k + 2] == 0xffffffffu) { at2[k] = more; more += (seg[4 * k] + 63) & ~size_t(63); }
    }
    void *pin2 = nullptr, *token2 = nullptr;
    hipError_t e = hipSuccess;
    bool straight = false;                                      // no pinned buffer free (two encodes begun): the large frames' segments go straight to the caller's memory
    if (more) {
        straight = stage_begin(more, &pin2, &token2) != IMP_OK;
        for (int k = 0; k < nj && e == hipSuccess; k++)
            if (codes[owner[k]] == IMP_OK && seg[4 * k + 2] == 0xffffffffu)
                e = straight ? hipMemcpy(outs[owner[k]] + heads[k].size(), jobs[k].out, seg[4 * k], hipMemcpyDeviceToHost)
                             : hipMemcpyAsync((uint8_t*)pin2 + at2[k], jobs[k].out, seg[4 * k], hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) { set_error("jpeg encode download", e); (void)hipStreamSynchronize(s); stage_hold(E.token, false); E.drop(); return IMP_ERROR_DEVICE; }
    }
    E.drop();                                                   // stream-ordered: after the copies
    if (more && !straight) {
        void* m2 = nullptr;
        int rc = lane_mark(&m2);
        if (!rc) rc = lane_wait_mark(m2);
        if (rc) { stage_hold(E.token, false); return rc; }
    }
    for (int k = 0; k < nj; k++) {
        const int i = owner[k];
        if (codes[i] != IMP_OK) continue;
        std::memcpy(outs[i], heads[k].data(), heads[k].size());
        if (seg[4 * k + 2] == 0xffffffffu && straight) continue;
        const uint8_t* from = seg[4 * k + 2] == 0xffffffffu ? (const uint8_t*)pin2 + at2[k] : compact + seg[4 * k + 2];
        std::memcpy(outs[i] + heads[k].size(), from, seg[4 * k]);
    }
    stage_hold(E.token, false);
    return IMP_OK;
}

int encode_group(const impgpu_image* const* images, int count, int quality, unsigned char* const* outs, const size_t* caps,
                 size_t* lens, int* codes) {
    EncState E;
    if (int rc = encode_begin(images, count, quality, E)) return rc;
    return encode_finish(E, outs, caps, lens, codes);
}

}  // namespace

}  // namespace imp

using namespace imp;

static thread_local char t_enc_thread_tag;                     // its address names the calling thread
struct impgpu_jpeg_encode { EncState E; const void* owner = nullptr; };

extern "C" {

size_t impgpu_jpeg_encode_bound(int width, int height, int channels) {
    EncGeom g;
    if (!enc_geom(width, height, channels, &g)) return 0;
    return 1024 + (size_t)g.nblocks * ENC_BLOCK_BYTES;
}

int impgpu_batch_encode_jpeg(const impgpu_image* const* images, int count, int quality, unsigned char* const* outs,
                             const size_t* capacities, size_t* lengths, int* codes) {
    if (count < 0 || (count && (!images || !outs || !capacities || !lengths || !codes))) return IMP_ERROR_INVALID_ARGS;
    if (!env_ready()) { set_error("impgpu_env_start has not been called", hipErrorNotInitialized); return IMP_ERROR_DEVICE; }
    TraceRange tr("IMP_STEP_ENCODE");                           // bridge.c:679-710
    IMP_FAULT_POINT(IMP_STEP_ENCODE);
    for (int at = 0; at < count; at += ENC_MAX_BATCH) {
        const int n = count - at < ENC_MAX_BATCH ? count - at : ENC_MAX_BATCH;
        if (int rc = encode_group(images + at, n, quality, outs + at, capacities + at, lengths + at, codes + at)) return rc;
    }
    return IMP_OK;
}

int impgpu_batch_encode_jpeg_begin(const impgpu_image* const* images, int count, int quality, impgpu_jpeg_encode** encode) {
    if (!encode) return IMP_ERROR_INVALID_ARGS;
    *encode = nullptr;
    if (count <= 0 || count > ENC_MAX_BATCH || !images) return IMP_ERROR_INVALID_ARGS;
    if (!env_ready()) { set_error("impgpu_env_start has not been called", hipErrorNotInitialized); return IMP_ERROR_DEVICE; }
    TraceRange tr("IMP_STEP_ENCODE");
    IMP_FAULT_POINT(IMP_STEP_ENCODE);
    impgpu_jpeg_encode* h = new impgpu_jpeg_encode();
    h->owner = &t_enc_thread_tag;
    if (int rc = encode_begin(images, count, quality, h->E)) { delete h; return rc; }
    *encode = h;
    return IMP_OK;
}

int impgpu_batch_encode_jpeg_finish(impgpu_jpeg_encode** encode, unsigned char* const* outs, const size_t* capacities, size_t* lengths, int* codes) {
    if (!encode || !*encode || !outs || !capacities || !lengths || !codes) return IMP_ERROR_INVALID_ARGS;
    impgpu_jpeg_encode* h = *encode;
    // (the staging buffer, the mark and the pool blocks belong to the beginning thread's lane)
    if (h->owner != &t_enc_thread_tag) { set_error_text("impgpu_batch_encode_jpeg_finish from another thread than _begin"); return IMP_ERROR_INVALID_ARGS; }
    *encode = nullptr;
    const int rc = encode_finish(h->E, outs, capacities, lengths, codes);
    delete h;
    return rc;
}

int impgpu_image_encode_jpeg(const impgpu_image* image, int quality, unsigned char* out, size_t capacity, size_t* length) {
    if (!image || !out || !length) return IMP_ERROR_INVALID_ARGS;
    int code = IMP_OK;
    if (int rc = impgpu_batch_encode_jpeg(&image, 1, quality, &out, &capacity, length, &code)) return rc;
    return code;
}

}  // extern "C"
