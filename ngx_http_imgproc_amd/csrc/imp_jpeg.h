// imp_jpeg.h -- the JPEG front of the pixel path: what bridge.c:545-552 (cvDecodeImage -> libjpeg) does on the host,
// moved in front of the operator chain on the device.  Shared by imp_jpeg.cpp (marker parser, table builder, scan
// preparation, the host entropy decoder kept as an A/B path) and imp_jpeg.hip (entropy decoder + IDCT / upsampling /
// colour kernels).  Not installed; the ABI is impgpu_image_decode_jpeg in include/impgpu.h.
#pragma once
#include "imp_internal.h"

namespace imp {

#ifndef JPEG_LOOKBITS_N
#define JPEG_LOOKBITS_N 10                   // (A/B builds: -DJPEG_LOOKBITS_N=9)
#endif
constexpr int JPEG_SUB_ENTRIES = 256;        // second-level entries per table (the Annex K tables need 128 or fewer)
constexpr int JPEG_LOOKBITS = JPEG_LOOKBITS_N;   // codes up to this length resolve with one LDS lookup
constexpr int JPEG_CHUNK_WORDS = 32;         // a decoder lane owns 1024 bits of the unstuffed stream -- or 512 / 256 for a launch too
constexpr int JPEG_CHUNK_BYTES_MAX = 512;    // the longest chunk jpeg_prepare_scan cuts
constexpr int JPEG_CHUNK_BYTES = JPEG_CHUNK_WORDS * 4;   // small to fill the device (JpegScan::chunk_bytes, chosen by jpeg_chunk_bytes_for)

struct JpegHuffSpec {                        // a DHT table as the file gives it
    bool present = false;
    uint8_t bits[17] = {};                   // bits[l] = number of codes of length l
    uint8_t vals[256] = {};
    int nvals = 0;
};

struct JpegComp {
    int id, h, v, tq, td, ta;
    int bw, bh;                              // blocks per row / column of the MCU-padded plane
    int dsw, dsh;                            // libjpeg's downsampled_width / _height: the real samples
};

struct JpegHeader {
    int width = 0, height = 0, ncomp = 0;
    int hs = 1, vs = 1;                      // luma sampling factors (chroma is 1x1)
    int mcux = 0, mcuy = 0, bpm = 0;         // MCUs per row / column, blocks per MCU
    JpegComp comp[3] = {};
    uint16_t qt[4][64] = {};                 // natural order
    bool qt_present[4] = {};
    JpegHuffSpec dc[4], ac[4];
    int restart_interval = 0;
    bool ycc = true;                         // three components mean YCbCr (else R,G,B stored as such)
    size_t scan_begin = 0;                   // first entropy-coded byte
    int why = 0;                             // JPEG_WHY_*: the reason for an IMP_ERROR_UNSUPPORTED (impgpu_jpeg_counters, impgpu_jpeg_classify)
};
// why the device does not take a file (round 5: what the cvDecodeImage fallback of bridge.c:545-552 is paid for)
enum { JPEG_WHY_NONE = 0, JPEG_WHY_PROGRESSIVE = 1 /* SOF2 */, JPEG_WHY_PROCESS = 2 /* arithmetic, lossless, hierarchical */,
       JPEG_WHY_12BIT = 3, JPEG_WHY_COMPONENTS = 4 /* CMYK / YCCK, two components */, JPEG_WHY_SCANS = 5 /* not one interleaved scan */,
       JPEG_WHY_SAMPLING = 6 /* factors other than 1x1 / 2x1 / 1x2 / 2x2 luma over 1x1 chroma */, JPEG_WHY_OTHER = 7 /* DNL height, not a JPEG, ... */,
       JPEG_WHY_COUNT = 8 };

// Marker segments up to the first SOS.  IMP_OK, IMP_ERROR_UNSUPPORTED (a JPEG this path does not take: progressive,
// arithmetic, 12-bit, CMYK, several scans, sampling other than 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 -- the caller decodes on the
// host as before) or IMP_ERROR_DECODE_FAILED (malformed).
int jpeg_parse(const uint8_t* blob, size_t size, JpegHeader* H);

// One Huffman table as the kernels read it.
struct JpegHuffDev {
    uint16_t lut[1 << JPEG_LOOKBITS];        // jpeg_lut_entry() of the code a JPEG_LOOKBITS-bit peek starts with; length 0 = a longer code:
                                             // 0x8000 | nb << 12 | (off / 2) << 5 = look in sub[off + the next nb bits], or 0 = walk the limits
    uint16_t sub[JPEG_SUB_ENTRIES];          // second level: jpeg_lut_entry() of the codes longer than JPEG_LOOKBITS (0 = no such code)
    uint32_t limit[18];                      // limit[l]: 16-bit left-aligned peeks below it start with a code of length <= l
    int32_t offs[18];                        // symbol index = offs[l] + (peek16 >> (16 - l))
    uint8_t vals[256];
};
int jpeg_build_table(const JpegHuffSpec& spec, bool is_dc, JpegHuffDev* out);

// Everything the kernels need to know about one file (a kernel argument by value).
struct JpegFrame {
    int width, height, ncomp, hs, vs, mcux, mcuy, bpm, ycc;
    int bw[3], bh[3], dsw[3], dsh[3];
    unsigned coef_off[3];                    // first coefficient of the component's plane, in shorts
    int dctab[3], actab[3];                  // which of the two DC / two AC device tables the component uses
    int slots_per_seg;                       // restart_interval * bpm * 64 coefficient slots (whole scan when no DRI)
    unsigned total_slots;                    // mcux * mcuy * bpm * 64
    unsigned nchunks, nsegs;
    unsigned chunk_bits;                     // 1024, 512 or 256: what a lane of the entropy kernel owns
    unsigned overlap_bits;                   // how far in front of its chunk a walk of k_jpeg_select starts (jpeg_overlap_bits_for)
    unsigned wsplit;                         // lanes of k_jpeg_write per chunk: 1, or 2 (the second enters at the chunk's middle, JpegSpan::mid)
};

// The entropy-coded segment made ready for the device: FF00 unstuffed, restart intervals cut at their RSTn markers, every
// interval starting on a chunk boundary, padded with 1-bits, one extra all-ones chunk at the end.
struct JpegScan {
    std::vector<uint32_t> seg_first_chunk;   // per interval
    std::vector<uint32_t> seg_bits;          // per interval: payload length in bits (8 * bytes)
    size_t nchunks = 0;                      // chunks holding payload (the trailing guard chunk is not counted)
    size_t chunk_bytes = JPEG_CHUNK_BYTES;   // IN: 256, 128, 64 or 32
};
// How a launch's files are cut: 128-byte chunks fill the device when there are many of them; a small launch (a lone request,
// up to 4 MB of entropy-coded data)
// is a chain of per-chunk walks that nothing else overlaps with, so its counting and writing walks -- which touch every
// chunk once -- finish sooner on shorter chunks, while the resynchronising rounds take the same time either way (a round's
// length and the number of rounds trade against each other).  launch_bytes = entropy-coded bytes of the whole launch.
// IMPGPU_JPEG_CHUNK_WORDS = 8 | 16 | 32 overrides (A/B).
size_t jpeg_chunk_bytes_for(size_t file_bytes, size_t launch_bytes, bool busy = false);
// How far in front of its chunk a synchronising walk starts: long enough to hold a block end or two of THIS file (its
// entropy-coded bytes over its blocks), so that one of the walks has fallen into step with the true decoder by the chunk's
// first bit.  IMPGPU_JPEG_OVERLAP (bits) overrides (A/B).
unsigned jpeg_overlap_bits_for(unsigned chunk_bits, size_t scan_bytes, size_t total_blocks);
// worst-case bytes jpeg_prepare_scan writes for `scan_bytes` of entropy-coded data and `nsegs` intervals
size_t jpeg_scan_capacity(size_t scan_bytes, size_t nsegs);
int jpeg_prepare_scan(const uint8_t* blob, size_t size, const JpegHeader& H, uint8_t* out, size_t cap, JpegScan* scan);

// Host entropy decoder (A/B path, IMPGPU_JPEG_HUFF=host): fills the MCU-padded coefficient planes, natural order.
int jpeg_host_entropy(const uint8_t* blob, size_t size, const JpegHeader& H, int16_t* coef, const JpegFrame& F);

// What the kernels need to know about the file, and which of the file's DC / AC tables fill the two device slots of each
// kind (-1 = unused).  IMP_ERROR_UNSUPPORTED for three distinct tables of a kind.
int jpeg_frame_setup(const JpegHeader& H, JpegFrame* F, int dc_ids[2], int ac_ids[2]);
int jpeg_build_tables(const JpegHeader& H, const int dc_ids[2], const int ac_ids[2], JpegHuffDev tabs[4]);   // [0..1] DC, [2..3] AC
// per chunk its interval, then seg_first_chunk[], then seg_bits[]: the entropy kernel's side input
void jpeg_scan_meta(const JpegScan& scan, std::vector<uint32_t>* meta);
// the device's entropy stage run lane by lane on the host (CPU tests / diagnostics; imp_jpeg_core.h)
int jpeg_emulate_entropy(const uint8_t* blob, size_t size, const JpegHeader& H, const JpegFrame& F, const int dc_ids[2],
                         const int ac_ids[2], int16_t* coef, unsigned* status, int* rounds);

// ---- imp_jpeg.hip
// status word the entropy kernels leave behind: 0 = every interval decoded to exactly its MCUs
constexpr unsigned JPEG_ST_BAD_CODE = 1u, JPEG_ST_BAD_COUNT = 2u, JPEG_ST_CHAIN_TIMEOUT = 4u, JPEG_ST_OVERRUN = 8u;
// a workgroup of k_jpeg_select took "every candidate leads to the same candidate" for "the true state leads there too", and its
// predecessor's final word said otherwise (the true state was none of the candidates and had not fallen into step with them)
constexpr unsigned JPEG_ST_CHAIN_GUESS = 16u;
// One file of a launch.  All kernels take a table of these plus a map from workgroup number to (job, workgroup within
// the job), so any number of files -- a request, an album, a queue's worth of requests -- costs the same few launches.
struct JpegJob {
    JpegFrame F;
    const uint32_t* words;                   // the prepared scan
    const uint32_t* chunk_seg;               // per chunk: its interval
    const uint32_t* seg_first_chunk;
    const uint32_t* seg_bits;
    const JpegHuffDev* tables;               // [0..1] DC, [2..3] AC
    const uint16_t* qt;                      // [3][64] natural order: the components' quantisation tables
    int16_t* coef;                           // the MCU-padded coefficient planes (every block written whole by k_jpeg_write)
    uint32_t* header;                        // 4 words, zeroed: [1] status, [2] repair walks, [3] chunks reached by a chase
    uint32_t* records;                       // JPEG_CTL_REC words per workgroup of k_jpeg_select, zeroed: the chain
    uint64_t *cand_in, *cand_out;            // [bpm][nchunks], k_jpeg_walks -> k_jpeg_select: the state of walk k of a chunk at its first bit / behind its last
    uint32_t* cand_n;                        // ... and the coefficient slots passed between the two
    uint8_t* cand_nib;                       // [bpm][nchunks], k_jpeg_mend -> k_jpeg_select: candidate k of the chunk before leads into walk (nib & 15) of this one
                                             // (15 = none); | 16 = found by a repair walk; 32 | k2 = the same state as candidate k2: its answer
    uint64_t* rep_out;                       // ... the repair walk's exit state and slot count
    uint32_t* rep_n;
    uint64_t *cand_mid, *rep_mid;            // [bpm][nchunks]: a walk's / a repair walk's state at the chunk's middle (JpegSpan::mid) ...
    uint32_t *cand_nmid, *rep_nmid;          // ... and the slots it passed up to there
    struct JpegHuffTabs* tabs;               // the tables as the decoder lanes read them (written by k_jpeg_walks)
    uint32_t* ext_idx;                       // [bpm][nchunks]: the record of a candidate whose repair walk joined nothing (valid where cand_nib == 15 | 16)
    uint32_t* ext;                           // records of JPEG_EXT_WORDS words: [0] chunk, [1] candidate, [2] steps, [3] joined candidate of the last step's chunk
                                             // (14 = the chain ended at an interval's end, 15 = not joined), then per step {entry state (2), slots};
                                             // the last two words: the state behind the last step when not joined
    uint32_t* ext_count;                     // records handed out (zeroed with the control area)
    uint32_t ext_cap;
    uint64_t* chunk_entry;                   // per chunk, k_jpeg_select -> k_jpeg_write: the decoder state at its first symbol
    uint32_t* chunk_n;                       // ... and the coefficient slots its symbols pass
    uint64_t* chunk_mid;                     // ... the state at its middle (a dead state: not known -- one lane decodes the whole chunk)
    uint32_t* chunk_nmid;                    // ... and the slots passed up to there
    // k_jpeg_write and k_jpeg_dcfix work in UNITS: F.wsplit lanes per chunk, unit u = chunk u / wsplit, half u % wsplit
    uint32_t* chunk_slot0;                   // per unit, k_jpeg_write -> k_jpeg_dcfix: the slot of its first symbol
    int* chunk_dc;                           // ... and [4]: its DC differences summed per component, its DC symbols
    int* wg_dc;                              // [8] per workgroup of k_jpeg_write: the sums over its chunks [0..2]; its clock at start and end [4], [5], at its walk's [6], [7]
    int16_t* dcadd;                          // per block, in scan order, k_jpeg_dcfix -> k_jpeg_pixels: what its DC term in the planes lacks
                                             // (null: the planes hold absolute DC terms -- the host's entropy stage)
    uint8_t* dst;                            // the frame
    int dstep;
    uint32_t* verdict;                       // where k_jpeg_dcfix leaves header[0..3] for the host (pinned memory; null: the host copies them)
};
// a workgroup's record in the chain of k_jpeg_select (words):
//   [1] 1 = its map is in [2]; 2 = its final word is in [3..5]
//   [2] the composed map of its chunks: which candidate of its last chunk follows from each candidate of its predecessor's
//   [3] the final word: 0..5 = that candidate of its last chunk is the true exit state; 15 = the state in [4..5] is;
//       13 = the next chunk is step [5] of record [4]
//   [6] coefficient slots passed by its chunks      [20..31] its clock at the phase boundaries (IMPGPU_JPEG_TRACE=2)
#ifndef JPEG_EXT_STEPS_N
#define JPEG_EXT_STEPS_N 6     // chunks a repair walk that joined nothing decodes on through, speculatively, in k_jpeg_mend (A/B with -DJPEG_EXT_STEPS_N=4 / 2, 64 files: k_jpeg_mend 165 -> 156 / 154 us, k_jpeg_select level: the kernel is its repair walks, not these)
#endif
constexpr int JPEG_EXT_STEPS = JPEG_EXT_STEPS_N, JPEG_EXT_WORDS = 4 + 3 * JPEG_EXT_STEPS + 2;
constexpr int JPEG_CTL_REC = 32;
#ifndef JPEG_SYNC_BLOCK_N
#define JPEG_SYNC_BLOCK_N 256  // (A/B: -DJPEG_SYNC_BLOCK_N=512 -- fewer, larger workgroups in k_jpeg_select's chain)
#endif
constexpr int JPEG_SYNC_BLOCK = JPEG_SYNC_BLOCK_N;         // lanes per workgroup of k_jpeg_select: (chunk, block of the MCU) pairs
#ifndef JPEG_HUFF_BLOCK_N
#define JPEG_HUFF_BLOCK_N 256
#endif
constexpr int JPEG_HUFF_BLOCK = JPEG_HUFF_BLOCK_N;         // chunks (= lanes) per workgroup of k_jpeg_write / k_jpeg_dcfix
constexpr int JPEG_TILE_W = 256, JPEG_TILE_H = 64;   // pixels a workgroup of the pixel kernel produces
struct JpegMapEntry { uint32_t job, local; };
inline unsigned jpeg_sync_chunks_per_block(int bpm) { return (unsigned)JPEG_SYNC_BLOCK / (unsigned)bpm; }   // 256, 85, 64, 42
inline unsigned jpeg_sync_blocks(unsigned nchunks, int bpm) { const unsigned c = jpeg_sync_chunks_per_block(bpm); return (nchunks + c - 1) / c; }
inline unsigned jpeg_entropy_blocks(unsigned nunits) { return (nunits + JPEG_HUFF_BLOCK - 1) / JPEG_HUFF_BLOCK; }   // (units: nchunks * wsplit)
// `ticket` = one zeroed word per launch; both maps in job-major order (a job's workgroups in increasing order):
// sync_map for k_jpeg_select (jpeg_sync_blocks per job), chunk_map for k_jpeg_write and k_jpeg_dcfix (jpeg_entropy_blocks)
// `marks`: null, or five events recorded behind k_jpeg_walks, _mend, _select, _write, _dcfix (impgpu_jpeg_profile)
int launch_jpeg_entropy(const JpegJob* jobs, const JpegMapEntry* sync_map, unsigned sync_blocks, const JpegMapEntry* chunk_map,
                        unsigned chunk_blocks, uint32_t* ticket, hipStream_t s, hipEvent_t* marks = nullptr, bool small = false);
// `small`: walks, mend and select as ONE launch (k_jpeg_entropy_small) -- for launches that do not fill the device anyway
// dequantise + ISLOW IDCT + fancy upsampling + YCbCr->BGR, coefficient planes -> frames; all jobs of one sampling class
int launch_jpeg_pixels(int hs, int vs, int ncomp, const JpegJob* jobs, const JpegMapEntry* tile_map, unsigned total_tiles, hipStream_t s);

}  // namespace imp
