// imp_jpeg.h -- the JPEG front of the pixel path: what bridge.c:545-552 (cvDecodeImage -> libjpeg) does on the host,
// moved in front of the operator chain on the device.  Shared by imp_jpeg.cpp (marker parser, table builder, scan
// preparation, the host entropy decoder kept as an A/B path) and imp_jpeg.hip (entropy decoder + IDCT / upsampling /
// colour kernels).  Not installed; the ABI is impgpu_image_decode_jpeg in include/impgpu.h.
#pragma once
#include "imp_internal.h"

namespace imp {

constexpr int JPEG_LOOKBITS = 10;            // codes up to this length resolve with one LDS lookup
constexpr int JPEG_CHUNK_WORDS = 32;         // a decoder lane owns 1024 bits of the unstuffed stream -- or 512 / 256 for a launch too
constexpr int JPEG_CHUNK_BYTES = JPEG_CHUNK_WORDS * 4;   // small to fill the device (JpegScan::chunk_bytes, chosen by jpeg_chunk_bytes_for)
constexpr int JPEG_HUFF_BLOCK = 256;         // chunks (= lanes) per workgroup of the entropy kernel

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
};

// Marker segments up to the first SOS.  IMP_OK, IMP_ERROR_UNSUPPORTED (a JPEG this path does not take: progressive,
// arithmetic, 12-bit, CMYK, several scans, sampling other than 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 -- the caller decodes on the
// host as before) or IMP_ERROR_DECODE_FAILED (malformed).
int jpeg_parse(const uint8_t* blob, size_t size, JpegHeader* H);

// One Huffman table as the kernels read it.
struct JpegHuffDev {
    uint16_t lut[1 << JPEG_LOOKBITS];        // jpeg_lut_entry() of the code a JPEG_LOOKBITS-bit peek starts with; length 0 = a longer code
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
    unsigned overlap_bits;                   // how far in front of its chunk a walk of k_jpeg_sync starts (jpeg_overlap_bits_for)
};

// The entropy-coded segment made ready for the device: FF00 unstuffed, restart intervals cut at their RSTn markers, every
// interval starting on a chunk boundary, padded with 1-bits, one extra all-ones chunk at the end.
struct JpegScan {
    std::vector<uint32_t> seg_first_chunk;   // per interval
    std::vector<uint32_t> seg_bits;          // per interval: payload length in bits (8 * bytes)
    size_t nchunks = 0;                      // chunks holding payload (the trailing guard chunk is not counted)
    size_t chunk_bytes = JPEG_CHUNK_BYTES;   // IN: 128, 64 or 32
};
// How a launch's files are cut: 128-byte chunks fill the device when there are many of them; a small launch (a lone request,
// up to 4 MB of entropy-coded data)
// is a chain of per-chunk walks that nothing else overlaps with, so its counting and writing walks -- which touch every
// chunk once -- finish sooner on shorter chunks, while the resynchronising rounds take the same time either way (a round's
// length and the number of rounds trade against each other).  launch_bytes = entropy-coded bytes of the whole launch.
// IMPGPU_JPEG_CHUNK_WORDS = 8 | 16 | 32 overrides (A/B).
size_t jpeg_chunk_bytes_for(size_t file_bytes, size_t launch_bytes);
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
// status word the entropy kernel leaves behind: 0 = every interval decoded to exactly its MCUs
constexpr unsigned JPEG_ST_BAD_CODE = 1u, JPEG_ST_BAD_COUNT = 2u, JPEG_ST_CHAIN_TIMEOUT = 4u, JPEG_ST_OVERRUN = 8u;
// One file of a launch.  Both kernels take a table of these plus a map from workgroup number to (job, workgroup within
// the job), so any number of files -- a request, an album, a queue's worth of requests -- costs the same two launches.
struct JpegJob {
    JpegFrame F;
    const uint32_t* words;                   // the prepared scan
    const uint32_t* chunk_seg;               // per chunk: its interval
    const uint32_t* seg_first_chunk;
    const uint32_t* seg_bits;
    const JpegHuffDev* tables;               // [0..1] DC, [2..3] AC
    const uint16_t* qt;                      // [3][64] natural order: the components' quantisation tables
    int16_t* coef;                           // the MCU-padded coefficient planes (zeroed before the entropy launch)
    uint32_t* header;                        // 4 words, zeroed: [1] status, [2] / [3] most rounds a workgroup took before / after the hand-over
    uint32_t* records;                       // JPEG_CTL_REC words per workgroup of the job, zeroed: the chain
    uint8_t* dst;                            // the frame
    int dstep;
};
constexpr int JPEG_CTL_REC = 24;             // [0..1] the tentative exit state, one 64-bit word kept up to date by the last lane (0 = nothing yet), [3..5] final exit
                                             // state (flag, lo, hi), [6..10] totals (flag, n, dc0..2), [12..22] the workgroup's clock at its phase boundaries, low words
                                             // of wall_clock64() (IMPGPU_JPEG_TRACE=2 prints them)
constexpr int JPEG_TILE_W = 256, JPEG_TILE_H = 64;   // pixels a workgroup of the pixel kernel produces
struct JpegMapEntry { uint32_t job, local; };
inline unsigned jpeg_entropy_blocks(unsigned nchunks) { return (nchunks + JPEG_HUFF_BLOCK - 1) / JPEG_HUFF_BLOCK; }
// `ticket` = one zeroed word per launch; block_map in job-major order (a job's workgroups in increasing order)
int launch_jpeg_entropy(const JpegJob* jobs, const JpegMapEntry* block_map, unsigned total_blocks, uint32_t* ticket, hipStream_t s);
// dequantise + ISLOW IDCT + fancy upsampling + YCbCr->BGR, coefficient planes -> frames; all jobs of one sampling class
int launch_jpeg_pixels(int hs, int vs, int ncomp, const JpegJob* jobs, const JpegMapEntry* tile_map, unsigned total_tiles, hipStream_t s);

}  // namespace imp
