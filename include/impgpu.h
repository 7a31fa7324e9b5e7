/*
 * impgpu.h -- C ABI of libimpgpu.so: the MI355X (gfx950) pixel-transform path that
 * replaces the operator segment of IMP's RunJob (reference bridge.c:574-656) and the
 * OpenCV/filters.c calls below it.  Plain C types only; no torch, no C++ in signatures.
 *
 * Each entry point cites the reference interface it stands in for.  Argument strings
 * are the reference's GET-parameter values verbatim (docs/03 - Usage.md); return codes
 * are the reference's IMP_* codes (required.h:28-41).  All pixel work runs in HIP
 * kernels on the device selected by impgpu_env_start(); there is no CPU fallback --
 * every device-touching call returns IMP_ERROR_DEVICE when HIP is unavailable.
 *
 * Images are 8-bit interleaved B,G,R[,A] (required.h:66-69), top-left origin, rows
 * padded to 4 bytes exactly like cvCreateImage (widthStep = (w*channels + 3) & ~3).
 */
#ifndef IMPGPU_H
#define IMPGPU_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes: required.h:28-41 ---- */
#define IMP_OK                      0
#define IMP_ERROR_UNSUPPORTED       1
#define IMP_ERROR_MALLOC_FAILED     2
#define IMP_ERROR_DECODE_FAILED     3
#define IMP_ERROR_ENCODE_FAILED     4
#define IMP_ERROR_INVALID_ARGS      50
#define IMP_ERROR_UPSCALE           51
#define IMP_ERROR_NO_SUCH_FILTER    52
#define IMP_ERROR_NO_SUCH_WATERMARK 53
#define IMP_ERROR_TOO_BIG_TARGET    54
#define IMP_ERROR_TOO_MUCH_FILTERS  55
#define IMP_ERROR_FEATURE_DISABLED  56
/* not in the reference: a HIP runtime call failed or no device/env (maps to HTTP 500 like any other code) */
#define IMP_ERROR_DEVICE            90

/* ---- step codes: required.h:46-54 ---- */
#define IMP_STEP_START     0
#define IMP_STEP_VALIDATE  1
#define IMP_STEP_DECODE    2
#define IMP_STEP_CROP      3
#define IMP_STEP_RESIZE    4
#define IMP_STEP_FILTERING 5
#define IMP_STEP_WATERMARK 6
#define IMP_STEP_INFO      7
#define IMP_STEP_ENCODE    8

/* ---- interpolation ids: OpenCV 2.4 CV_INTER_* as passed to cvResize at bridge.c:190-191 ---- */
#define IMP_INTER_NN       0
#define IMP_INTER_LINEAR   1
#define IMP_INTER_CUBIC    2
#define IMP_INTER_AREA     3
#define IMP_INTER_LANCZOS4 4

/* Device-resident stand-in for the IplImage* the reference passes between operators
 * (required.h:129-134 Frame.Image). Opaque; query with the accessors. */
typedef struct impgpu_image impgpu_image;

/* The fields of Config (required.h:108-118) that the operators read, plus Position
 * (required.h:86-91).  watermark replaces RecoverInfo (required.h:99-106): the overlay's
 * pixels live in HBM, uploaded once per worker by impgpu_prepare_watermark. */
typedef struct {
    unsigned int  max_target_w;         /* MaxTargetDimensions->W, 0 = unlimited (module.c:172-175 default 2000) */
    unsigned int  max_target_h;         /* MaxTargetDimensions->H */
    int           max_filters_count;    /* MaxFiltersCount (module.c:181 default 5) */
    int           allow_experiments;    /* AllowExperiments */
    int           watermark_opacity;    /* WatermarkOpacity 1..100 (module.c:144-148) */
    char          watermark_gravity_x;  /* Position.GravityX: 'l' 'c' 'r' */
    char          watermark_gravity_y;  /* Position.GravityY: 't' 'c' 'b' */
    int           watermark_offset_x;   /* Position.OffsetX */
    int           watermark_offset_y;   /* Position.OffsetY */
    impgpu_image* watermark;            /* WatermarkInfo; NULL = no watermark configured */
} impgpu_config;

/* What RunJob has parsed out of the query string by bridge.c:372, i.e. the inputs of
 * the operator segment bridge.c:574-656. */
typedef struct {
    const char*        crop;            /* value of crop=, NULL if absent */
    const char*        gravity;         /* value of gravity=, NULL if absent */
    const char*        resize;          /* value of resize=, NULL if absent */
    int                simple;          /* bridge.c:594: GIF output forces nearest-neighbour */
    const char* const* filters;         /* "name=args" strings in query order (bridge.c:365) */
    int                filter_count;
    int                need_flatten;    /* encoder lacks alpha (bridge.c:642-648) */
} impgpu_job;

/* ---- lifecycle: OnEnvStart / OnEnvDestroy (bridge.h:1-2, bridge.c:10-16; called from
 *      module.c:100-107 once per worker process, after fork) ---- */
/* device < 0: take $IMPGPU_DEVICE, else $LOCAL_RANK, else 0.
 * impgpu_env_start also registers impgpu_env_destroy with atexit() (once per process): a worker that leaves through
 * exit() with a live env gives its streams, events, pinned buffers and pool blocks back BEFORE the HIP runtime's own exit
 * handlers run, not during them.  That exit-time teardown waits at most $IMPGPU_EXIT_WAIT_MS (default 2000) for the
 * device to go idle and otherwise leaves everything to the driver; a fork()ed copy of a process that started the env
 * does nothing at exit.  impgpu_env_destroy is idempotent. */
int         impgpu_env_start(int device);
void        impgpu_env_destroy(void);
int         impgpu_env_device(void);            /* -1 when no env */
/* NUMA: the node the env's device hangs off (its PCI function's numa_node; -1 = unknown or a one-node
 * host), and a call that binds the CALLING thread to that node's CPUs -- a worker's staging copies then run next to
 * the GPU's PCIe root, and pinned memory the thread allocates afterwards is that node's (first touch).  An nginx worker
 * calls it once after impgpu_env_start; IMPGPU_NUMA_BIND=1 does it for every thread at its first call into the library.
 * IMP_ERROR_UNSUPPORTED when the node is unknown or none of its CPUs is open to this process. */
int         impgpu_env_numa_node(void);
int         impgpu_env_bind_thread(void);
const char* impgpu_last_error(void);            /* text of the last HIP failure on this thread */
int         impgpu_sync(void);                  /* wait for the env stream */
void*       impgpu_env_stream(void);            /* the env's hipStream_t */

/* Test hook (failure detection): the nth entry (1 = the next) into the given IMP_STEP_* behaves as if its first
 * HIP call had failed -- IMP_ERROR_DEVICE, impgpu_last_error() says "injected fault", the failing step is reported like
 * any other -- so the path a lost device takes can be tested without losing one.  step < 0 disarms.  Armed by this call
 * only, never by the environment. */
int         impgpu_fault_arm(int step, long nth);

/* ---- frames: the decode -> operators hand-over (bridge.c:547-552, advancedio.c:310-318)
 *      and the operators -> encode hand-over (bridge.c:703-704, advancedio.c:65-101) ---- */
int   impgpu_image_upload(const unsigned char* data, int width, int height, int channels,
                          int step, impgpu_image** out);   /* pinned staging + hipMemcpyAsync */
int   impgpu_image_create(int width, int height, int channels, impgpu_image** out);
/* Pinned-host variants for callers that decode into / encode from page-locked memory
 * (impgpu_host_alloc): no staging pass, fully asynchronous on the env stream.  The host
 * buffer must stay untouched until impgpu_sync() (upload) / is valid after impgpu_sync() (download).
 * `step` may be any host row pitch >= width * channels: a pitch other than the frame's own (4-byte padded rows) still
 * crosses the link as one linear copy and is re-pitched on the device; on download the bytes between the caller's rows
 * are then written as zeros. */
void* impgpu_host_alloc(size_t bytes);
void  impgpu_host_free(void* ptr);
int   impgpu_image_upload_pinned(const unsigned char* data, int width, int height, int channels,
                                 int step, impgpu_image** out);
int   impgpu_image_download_pinned(const impgpu_image* image, unsigned char* data, int step);
/* The FreeImage side of the hand-overs (advancedio.c): bitmaps there are bottom-up with 4-byte aligned rows.
 * upload_fi32 = LoadSingle's copy loop (advancedio.c:310-318): 32-bit bits -> 4-channel top-down frame.
 * download_fi = IplToFI32 / IplToFI24 (advancedio.c:65-101): flip, and B,G,R,A with A = 255 for 3-channel frames
 * (bpp 32) or B,G,R with alpha dropped (bpp 24), written into the encoder's bitmap; the repack runs on the device. */
int   impgpu_image_upload_fi32(const unsigned char* bits, int width, int height, int pitch, impgpu_image** out);
int   impgpu_image_download_fi(const impgpu_image* image, int bpp, unsigned char* bits, int pitch);   /* syncs */
/* The decode itself moved in front of the operators: IplImage* image = cvDecodeImage(&rawencoded, -1)   bridge.c:545-552
 * for a JPEG blob (SIG_JPG, bridge.c:376-378), i.e. libjpeg with its default parameters (ISLOW IDCT, fancy upsampling)
 * and OpenCV's R/B swap.  The compressed bytes cross the link instead of the pixels; Huffman decoding, dequantisation,
 * IDCT, chroma upsampling and YCbCr -> B,G,R run on the device and the frame (3 channels, or 1 for a gray file, exactly
 * what cvDecodeImage(.., -1) returns) is bit-identical to libjpeg-turbo's.  Takes 8-bit baseline / extended-sequential
 * Huffman files with one interleaved scan and 4:4:4, 4:2:2, 4:4:0, 4:2:0 or gray sampling, with or without restart
 * intervals.  IMP_ERROR_UNSUPPORTED = a JPEG outside that set (progressive, CMYK, ...): decode it with cvDecodeImage as
 * before and impgpu_image_upload the pixels.  IMP_ERROR_DECODE_FAILED = damaged data; libjpeg would warn and deliver
 * what it could, so the same fallback applies.  Waits for the device's verdict on the entropy-coded data before it
 * returns (the only wait on the request path besides the download).  A launch with less than 40 KB of entropy-coded
 * data keeps its Huffman decoding on the calling thread (6 ns per byte against a fixed few hundred microseconds of
 * device rounds); IMPGPU_JPEG_HUFF=device|host forces one.  The pixels do not depend on it. */
int   impgpu_image_decode_jpeg(const unsigned char* blob, size_t size, impgpu_image** out);
/* The same for `count` files at once -- the frames of a request queue (BASELINE configs[4]) or whatever else arrives
 * together: ONE entropy launch and one pixel launch per chroma sampling for all of them, one wait for all verdicts, so the
 * device is filled by a single caller.  codes[i] / images[i] are what impgpu_image_decode_jpeg would return for file i
 * (images[i] = NULL unless codes[i] == IMP_OK); the return value is IMP_OK unless the call itself failed (no env, a HIP
 * error), in which case no image is returned. */
int   impgpu_batch_decode_jpeg(const unsigned char* const* blobs, const size_t* sizes, int count,
                               impgpu_image** images, int* codes);
/* The same in two halves, for a caller that has more to do than wait (round 4): _begin parses, copies the scans into pinned
 * memory and enqueues the whole decode -- it does NOT wait; _finish sleeps until THAT batch is done (not whatever the thread
 * enqueued after it), reads the verdicts and fills images[] / codes[] as impgpu_batch_decode_jpeg would.  Between the two the
 * thread may begin the next batch or enqueue anything else; the blobs must stay readable until _finish (a file the device
 * defers is read again).  Both halves belong to ONE thread: _finish from another thread returns IMP_ERROR_INVALID_ARGS and
 * leaves the batch with its owner.  A thread has FOUR slots for decodes in flight, shared by the batches it has begun and
 * by its ordinary decode calls (one each; a one-call batch of 32 or more files takes two when two are free): with four
 * batches begun and not finished, any further decode on that thread returns IMP_ERROR_INVALID_ARGS.  count <= 256.
 * The batch runs on a second stream of the calling thread, so work the thread enqueues between the two calls (the resize and
 * the answers of the batch before) overlaps it.  (impgpu_batch_decode_jpeg itself already prepares the second half of a large
 * batch while the device works on the first; profiles/r05_jpeg_stream_native.txt holds both forms side by side.) */
typedef struct impgpu_jpeg_batch impgpu_jpeg_batch;
int   impgpu_batch_decode_jpeg_begin(const unsigned char* const* blobs, const size_t* sizes, int count, impgpu_jpeg_batch** batch);
int   impgpu_batch_decode_jpeg_finish(impgpu_jpeg_batch** batch, impgpu_image** images, int* codes);
/* Files whose entropy-coded segment has been taken out of its byte stuffing by the CALLER (round 5: a worker process does it
 * while it copies its request into the broker's shared memory -- impgpu_jpeg_unstuff in libimpgpu_client.so -- so the pass over
 * the compressed bytes is made by sixteen sleeping workers instead of the four threads that drive the device).
 *   head, head_size   the file from SOI up to and including its SOS header (what jpeg headers parse); a file WITH a restart
 *                     interval cannot be handed over this way (IMP_ERROR_INVALID_ARGS for that file);
 *   scan, scan_size   the entropy-coded bytes that followed, FF 00 -> FF, fill bytes dropped, up to (not including) the marker
 *                     that ends them; behind them the caller has written IMPGPU_JPEG_SCAN_TAIL bytes of 0xFF (not counted);
 *                     scan = NULL: `head` is a whole file as it came (the two kinds may share a batch);
 *   registered        != 0: `scan` lies in memory made known with impgpu_host_register -- the bytes go to the device from
 *                     where they are (no pass over them on the calling thread at all); they must not change until the call
 *                     returns.
 * Pixels, codes and refusals are those of impgpu_batch_decode_jpeg on the original files. */
#define IMPGPU_JPEG_SCAN_TAIL 1024
typedef struct {
    const unsigned char* head; size_t head_size;
    const unsigned char* scan; size_t scan_size;
    int registered;
} impgpu_jpeg_prepared;
int   impgpu_batch_decode_jpeg_prepared(const impgpu_jpeg_prepared* files, int count, impgpu_image** images, int* codes);
/* ... and in two halves like impgpu_batch_decode_jpeg_begin, but on the calling thread's OWN stream: what the thread enqueues
 * afterwards runs behind this decode, what it enqueued before runs in front of it (a broker lane begins the next batch's
 * decode while the answers of the batch before are still being written: the device never waits for the host's share).
 * The array is copied; head / scan bytes must stay readable until impgpu_batch_decode_jpeg_finish.  count <= 256. */
int   impgpu_batch_decode_jpeg_prepared_begin(const impgpu_jpeg_prepared* files, int count, impgpu_jpeg_batch** batch);
/* The frames of a batch begun on the calling thread's own stream (impgpu_batch_decode_jpeg_prepared_begin), AHEAD of their verdicts:
 * images[i] = the frame file i is being decoded into -- geometry final, pixels there once the batch's kernels have run, which is
 * before anything the thread enqueues on it afterwards runs -- or NULL where there is none yet (refused at its header, or a file
 * whose Huffman stage the host keeps: those come out of _finish as before).  Ownership passes to the caller; _finish then
 * returns NULL for that file and its CODE: not IMP_OK = the frame does not hold the file's pixels (release it, or whatever was
 * made of it, and fall back like for any refused file).  So a request's operators and its answer's encode can be enqueued
 * BEHIND the decode and the thread waits once, at the end, instead of once for the verdicts and once for the answer: a lone
 * 640 x 480 -> thumbnail -> JPEG request 0.29 -> see DESIGN 4b. */
int   impgpu_batch_decode_jpeg_pending(impgpu_jpeg_batch* batch, impgpu_image** images);
/* Page-locks `bytes` at `p` (hipHostRegister) so that copies out of it need no staging; impgpu_host_unregister before the
 * memory goes away.  Needs impgpu_env_start. */
int   impgpu_host_register(void* p, size_t bytes);
int   impgpu_host_unregister(void* p);
/* Where a decode call's time goes (per-stage timing): impgpu_jpeg_profile(1) makes every later decode call leave
 * its stages with the calling thread, impgpu_jpeg_stage_times reads the last call's, in microseconds:
 * [0] marker segments, [1] FF00 unstuffing into pinned memory, [2] tables + job table, [3] enqueue, [4] wait for the verdicts
 * (host clock); [5] upload, [6] k_jpeg_walks, [7] k_jpeg_mend, [8] k_jpeg_select, [9] k_jpeg_write, [10] k_jpeg_dcfix,
 * [11] verdict copy + k_jpeg_pixels (events on the stream; the call then also waits for the pixels); [12] files decoded.
 * Returns the previous setting / IMP_OK. */
int   impgpu_jpeg_profile(int on);
/* Process-wide counts since the library was loaded: [0] files whose entropy stage ran on the device, [1] of those, refused
 * by its verdict (the caller's cvDecodeImage fallback took them), [2] of those, because a wait between workgroups ran out --
 * must stay 0 on a healthy box, however many processes share the device -- [3] files kept on the calling thread because
 * their blocks are too long for the device scheme (more than 200 bits each).
 * Round 5, what the cvDecodeImage fallback (bridge.c:545-552) is paid for -- files a decode call refused at their header, by
 * reason: [5] progressive (SOF2), [6] arithmetic / lossless / hierarchical, [7] 12-bit samples, [8] CMYK / YCCK or another
 * component count, [9] not one interleaved scan, [10] sampling factors outside 1x1 / 2x1 / 1x2 / 2x2 over 1x1, [11] anything
 * else (a DNL height, not a JPEG at all), [12] headers damaged before the first scan.  ([4] is unused.)
 * impgpu_jpeg_classify gives the same verdict for one file without decoding it (host, no device): 0 = the device takes it,
 * 1..7 = the reasons in the order above, 8 = damaged; tools/corpus_probe.py adds them up over a directory. */
int   impgpu_jpeg_counters(unsigned long long* counters, int n);
int   impgpu_jpeg_classify(const unsigned char* blob, size_t size);
int   impgpu_jpeg_stage_times(double* microseconds, int n);
/* The other end of the request: CvMat* encoded = cvEncodeImage(".jpg", image, basicCoderopt)          bridge.c:704
 * with basicCoderopt = {CV_IMWRITE_JPEG_QUALITY, quality} (bridge.c:474-486; OpenCV clamps the value to 0..100), for the
 * frame the operators left in HBM: colour conversion, chroma downsampling, forward DCT, quantisation and Huffman coding run
 * on the device and the compressed file crosses the link instead of the pixels.  OpenCV 2.4.9's JpegEncoder leaves
 * everything else at libjpeg's defaults -- baseline, YCbCr 4:2:0 (one gray component for a 1-channel frame), the Annex K
 * tables, ISLOW DCT, JFIF 1.01 -- and the file is the same, byte for byte, as libjpeg-turbo's.  A 4-channel frame loses
 * its alpha exactly as in cvEncodeImage (RunJob flattens it first, bridge.c:642-656); an album handle encodes frame 0,
 * like bridge.c:703.  *length receives the file's size; IMP_ERROR_MALLOC_FAILED = capacity is smaller than that (nothing
 * is written; impgpu_jpeg_encode_bound(w, h, c) is always enough).  Waits for the device. */
int   impgpu_image_encode_jpeg(const impgpu_image* image, int quality, unsigned char* out, size_t capacity, size_t* length);
/* `count` frames (the thumbnails of a request queue) in two launches and two waits; codes[i] / lengths[i] are what
 * impgpu_image_encode_jpeg would give for frame i. */
int   impgpu_batch_encode_jpeg(const impgpu_image* const* images, int count, int quality, unsigned char* const* outs,
                               const size_t* capacities, size_t* lengths, int* codes);
/* The same in two halves (round 5, for a caller that serves more than one batch: a broker lane): _begin enqueues everything --
 * kernels and the copy of the files' segments into pinned memory -- and does NOT wait; _finish sleeps until THAT encode is
 * done (not what the thread enqueued behind it, e.g. the decode of the next batch), and fills outs[] / lengths[] / codes[] as
 * impgpu_batch_encode_jpeg would.  The frames must stay alive until _finish.  Both halves belong to ONE thread.  The answers
 * of an encode wait in one of the thread's TWO pinned staging buffers until they are fetched: with two encodes begun, anything
 * else of that thread that stages through pinned memory (a third encode, a decode, an upload) returns IMP_ERROR_INVALID_ARGS
 * until one is finished -- a lane keeps one in flight.  count <= 256. */
typedef struct impgpu_jpeg_encode impgpu_jpeg_encode;
int   impgpu_batch_encode_jpeg_begin(const impgpu_image* const* images, int count, int quality, impgpu_jpeg_encode** encode);
int   impgpu_batch_encode_jpeg_finish(impgpu_jpeg_encode** encode, unsigned char* const* outs, const size_t* capacities,
                                      size_t* lengths, int* codes);
size_t impgpu_jpeg_encode_bound(int width, int height, int channels);
/* the SOF header alone (host, no device): the size checks the module makes before decoding */
int   impgpu_jpeg_info(const unsigned char* blob, size_t size, int* width, int* height, int* channels);
/* Diagnostics (host, no device): the quantised coefficients of the file's components, MCU-padded planes one after the
 * other, blocks in raster order, 64 shorts each in row-major order.  how = 0: a plain sequential entropy decoder;
 * how = 1: the device's chunk-parallel scheme executed lane by lane on the host (same code as the kernel's lanes).
 * info[0] = shorts written, [1] = the scheme's status word, [2] = chunks whose true entry state was not among their
 * walks' (how = 1), [3..5] = first short of each component's plane, [6..8] / [9..11] = blocks per row / column.
 * impgpu_jpeg_sync_stats: more about the calling thread's last how = 1 call -- stats[0] = chunks, [1] = the same misses,
 * [2] = repair walks (speculative ones included), [3] = chunks reached by an explicit state ("chase"), [4] = chunk bits,
 * [5] = overlap bits, [6] = walks per chunk. */
int   impgpu_jpeg_coefficients(const unsigned char* blob, size_t size, int how, short* out, size_t capacity, int* info);
void  impgpu_jpeg_sync_stats(int stats[8]);
/* ---- PNG in (round 4, a bounded experiment: DESIGN.md "PNG decode") ----
 * cvDecodeImage(&rawencoded, -1) (bridge.c:545-552) for a PNG blob (SIG_PNG, bridge.c:376-378), i.e. OpenCV 2.4's PngDecoder
 * over libpng with the "unchanged" flag: 8-bit gray -> 1 channel, RGB -> BGR, RGBA -> BGRA, tRNS not expanded.  The zlib
 * stream is inflated on the HOST (csrc/imp_inflate.cpp) straight into pinned memory; the filtered scanlines cross the link and the five
 * scanline filters (PNG specification 9.2) are undone on the device.  Takes bit depth 8, colour types 0 / 2 / 6, not
 * interlaced, width <= 4096, height <= 16384; IMP_ERROR_UNSUPPORTED = any other PNG (or not a PNG): decode it with
 * cvDecodeImage as before.  IMP_ERROR_DECODE_FAILED = damaged (a chunk's CRC, the deflate stream, too little data, a
 * filter type above 4; the stream's Adler-32 trailer is not checked).  Does not wait for the device.
 * impgpu_png_stage_times: the calling thread's last decode, host clock, microseconds: [0] header, [1] chunk walk + CRC +
 * inflate, [2] filter-type check + enqueue (upload, kernel); [3] = bytes of filtered scanlines. */
int   impgpu_image_decode_png(const unsigned char* blob, size_t size, impgpu_image** out);
int   impgpu_png_info(const unsigned char* blob, size_t size, int* width, int* height, int* channels);
int   impgpu_png_stage_times(double* microseconds, int n);
/* Diagnostics (host, no device): the filtered scanlines of the file -- height rows of (1 filter byte + width * channels bytes),
 * as the device receives them -- after the chunk walk, the CRC checks and the inflate.  *length = the bytes the image needs
 * (set whenever the header was readable); IMP_ERROR_MALLOC_FAILED when capacity is smaller. */
int   impgpu_png_scanlines(const unsigned char* blob, size_t size, unsigned char* out, size_t capacity, size_t* length);
int   impgpu_image_wrap(void* device_ptr, int width, int height, int channels, int step,
                        impgpu_image** out);               /* borrow memory already in HBM */
int   impgpu_image_clone(const impgpu_image* src, impgpu_image** out);
int   impgpu_image_download(const impgpu_image* image, unsigned char* data, int step); /* syncs */
/* the same for every frame of an album (bridge.c:680-710 reads them all): all copies are enqueued, then ONE wait */
int   impgpu_batch_download(const impgpu_image* const* images, int count, unsigned char* const* datas, const int* steps);
/* An ALBUM: the frames of one animation (Album, required.h:56-66; filled by LoadGIF, advancedio.c:184-289, or by the
 * one-frame decoders at bridge.c:554-572).  All frames have the canvas' geometry, so they live in ONE device block and
 * the handle stands for all of them: impgpu_run_ops on an album handle runs every operator of the request as ONE launch
 * over all frames (the reference loops `for fid < album.Count` around each operator, bridge.c:577-655), and so does
 * every single operator below.  Info() and ASCII() read frame 0, as bridge.c:283-300 / 669 do; impgpu_image_download*
 * and the batch entry points take single images.  steps may be NULL (= width * channels, tightly packed rows). */
int   impgpu_album_upload(const unsigned char* const* datas, int count, int width, int height, int channels,
                          const int* steps, impgpu_image** out);
int   impgpu_album_download(const impgpu_image* album, unsigned char* const* datas, const int* steps);   /* one wait */
int   impgpu_album_count(const impgpu_image* image);        /* 1 for a single image */
int   impgpu_image_width(const impgpu_image* image);
int   impgpu_image_height(const impgpu_image* image);
int   impgpu_image_channels(const impgpu_image* image);
int   impgpu_image_step(const impgpu_image* image);
void* impgpu_image_device_ptr(const impgpu_image* image);
void  impgpu_image_release(impgpu_image** image);          /* cvReleaseImage */

/* ---- operators.  `pointer` operators may replace *pointer and release the old image,
 *      exactly as the reference's IplImage** operators do. ---- */

/* int Crop(IplImage** pointer, char* args, char* gravity)            bridge.h:4, bridge.c:18-141 */
int impgpu_crop(impgpu_image** pointer, const char* args, const char* gravity);
/* int Resize(IplImage** pointer, char* args, Config* config, int simple)  bridge.h:5, bridge.c:143-197 */
int impgpu_resize(impgpu_image** pointer, const char* args, const impgpu_config* config, int simple);
/* cvResize(image, resized, filter)                                    bridge.c:189-191.
 * The reference only ever passes NN / CUBIC (enlarging) / AREA (shrinking); this entry
 * point also takes LINEAR and LANCZOS4 and CUBIC-on-downscale with OpenCV 2.4.9 semantics. */
int impgpu_cv_resize(impgpu_image** pointer, int width, int height, int interpolation);
/* int Filter(IplImage** pointer, char* request, int allowExperiments) filters.h:1, filters.c:43-70 */
int impgpu_filter(impgpu_image** pointer, const char* request, int allow_experiments);
/* int PrepareWatermark(Config* cfg, ngx_pool_t* pool)                 bridge.h:6, bridge.c:199-237.
 * File read + decode stay on the host (cvDecodeImage); this takes the decoded pixels
 * (what bridge.c:221-234 parks in RecoverInfo) and uploads them to HBM. */
int impgpu_prepare_watermark(impgpu_config* config, const unsigned char* pixels, int width,
                             int height, int channels, int step);
/* int Watermark(IplImage* image, Config* config)                      bridge.h:7, bridge.c:239-281 */
int impgpu_watermark(impgpu_image* image, const impgpu_config* config);
/* void BlendWithPaper(IplImage* source)                               filters.h:30, filters.c:666-687 */
int impgpu_blend_with_paper(impgpu_image* image);
/* float CalcPerceivedBrightness(IplImage* image)                      filters.h:34, filters.c:707-729 */
int impgpu_calc_perceived_brightness(const impgpu_image* image, float* brightness);
/* Memory ASCII(IplImage* input, char* args, ngx_pool_t* pool)         filters.h:18, filters.c:486-522.
 * out must hold (width+1)*height-1 bytes; like the reference it leaves the image in HSV. */
int impgpu_ascii(impgpu_image* image, const char* args, unsigned char* out, long capacity, long* length);
/* cvCvtColor(image, colored, CV_GRAY2BGR)                             bridge.c:613-618 */
int impgpu_gray2bgr(impgpu_image** pointer);
/* void RGB2HSV(IplImage*) / void HSV2RGB(IplImage*)                   helpers.h:17-18, helpers.c:70-176 */
int impgpu_rgb2hsv(impgpu_image* image);
int impgpu_hsv2rgb(impgpu_image* image);

/* The operator segment of RunJob, bridge.c:574-656: crop -> resize -> [gray->BGR] ->
 * filters in order -> watermark -> flatten.  *step receives the IMP_STEP_* that was
 * running when a non-zero code was returned (JobResult.Step, required.h:78-84).
 * Crop is folded into the next operator's source view, consecutive pointwise filters
 * run as one kernel. */
int impgpu_run_ops(impgpu_image** pointer, const impgpu_job* job, const impgpu_config* config, int* step);

/* ---- argument grammar only (host, no device): what Crop / Resize decide before they
 *      touch pixels.  Used by the module to answer 400/405/413 without a GPU round trip. ---- */
int impgpu_crop_geometry(int width, int height, const char* args, const char* gravity,
                         int* x, int* y, int* w, int* h);                      /* bridge.c:18-128 */
int impgpu_resize_geometry(int width, int height, const char* args, const impgpu_config* config,
                           int simple, int* w, int* h, int* interpolation);   /* bridge.c:143-190 */
int impgpu_filter_check(const char* request, int allow_experiments);           /* filters.c:43-70 + per-filter arg checks */
int impgpu_check_destructive(const char* request);                             /* filters.c:32-40 */

/* ---- request front end (host, no device): the part of RunJob above the operators.
 *      URI unescape + GET grammar of bridge.c:304-372 ('?' split, '&' tokens, prefix-matched keys,
 *      last crop/gravity/resize/quality/format/page wins, filters appended up to
 *      config->max_filters_count) and the encoder choice of bridge.c:413-466 that decides
 *      job.simple (GIF, bridge.c:594) and job.need_flatten (bridge.c:642-648).
 *      `extension` is req->exten, used when format= is absent (bridge.c:413-416).
 *      On error *out is still a valid object to free. The job's strings live inside it. ---- */
typedef struct impgpu_request impgpu_request;
int               impgpu_parse_request(const char* uri, const char* extension, const impgpu_config* config,
                                       impgpu_request** out);
const impgpu_job* impgpu_request_job(const impgpu_request* request);
const char*       impgpu_request_quality(const impgpu_request* request);    /* value of quality= or NULL */
const char*       impgpu_request_format(const impgpu_request* request);     /* value of format= or NULL */
/* the page LoadGIF gets (DecodeRequest.Page, bridge.c:563): the page= value; when absent, 0 for every encoder that
 * takes one frame and -1 (= all pages) only for GIF output and the json exit (bridge.c:324, :433-435, :448-450) */
int               impgpu_request_page(const impgpu_request* request);
int               impgpu_request_mime(const impgpu_request* request);       /* IMP_MIME_* of required.h:57-62 */
int               impgpu_request_destructive(const impgpu_request* request); /* CheckDestructive over the filters */
void              impgpu_request_free(impgpu_request** request);

/* ---- GIF album hand-over: LoadGIF (advancedio.c:103-262).  FreeImage keeps decoding the pages on the host; the
 *      per-pixel compositing loop of advancedio.c:204-247 -- the frame placed at FrameLeft/FrameTop on the first
 *      page's canvas, the transparent index, the `master` index canvas carried from page to page when the request
 *      is destructive (advancedio.c:195-240), the RGBQUAD palette lookup into a 4-channel frame (:242-246) -- runs
 *      on the device.  A page is what FreeImage hands that loop: 8-bit indices in scanline order (bottom-up,
 *      `pitch` bytes per row; FreeImage_ConvertTo8Bits first when the page is not 8-bit, :181-185).
 *      frames[i] = page i (all `count` of them) when page == -1.  With page >= 0 the walk is always destructive, a
 *      page past the last one means page 0 (advancedio.c:111-116), the walk stops at that page and frames[0] is
 *      that page alone (advancedio.c:256-273).  page < -1 returns IMP_ERROR_INVALID_ARGS (undefined in the reference). ---- */
#define IMP_GIF_DISPOSAL_UNSPECIFIED 0
#define IMP_GIF_DISPOSAL_LEAVE       1
#define IMP_GIF_DISPOSAL_BACKGROUND  2
#define IMP_GIF_DISPOSAL_PREVIOUS    3
typedef struct impgpu_gif_page {
    const unsigned char* indices;   /* FreeImage_GetBits of the 8-bit page (host memory) */
    int width, height, pitch;       /* FreeImage_GetWidth / GetHeight / GetPitch */
    int left, top;                  /* FIMD_ANIMATION FrameLeft / FrameTop (advancedio.c:159-174) */
    int dispose;                    /* FIMD_ANIMATION DisposalMethod (advancedio.c:146-157) -> Frame.Dispose */
    int transparency_key;           /* FreeImage_GetTransparentIndex, -1 = none (advancedio.c:176) */
    const unsigned char* palette;   /* FreeImage_GetPalette: 256 RGBQUAD = B,G,R,reserved bytes (advancedio.c:178) */
} impgpu_gif_page;
int impgpu_gif_compose(const impgpu_gif_page* pages, int count, int destructive, int page, impgpu_image** frames);
/* the same frames as ONE album handle (see impgpu_album_upload): what RunJob's operator segment then runs on */
int impgpu_gif_compose_album(const impgpu_gif_page* pages, int count, int destructive, int page, impgpu_image** album);

/* ---- batch entry points (benchmark / multi-frame albums: bridge.c:578,591,608,632).
 *      `count` frames of identical geometry, frame i at base + i*frame_stride bytes,
 *      already resident in HBM.  stream = hipStream_t to launch on (NULL = env stream).
 *      Asynchronous: returns after enqueue. ---- */
int impgpu_batch_cv_resize(const void* src, long long src_frame_stride, int src_width, int src_height, int src_step,
                           void* dst, long long dst_frame_stride, int dst_width, int dst_height, int dst_step,
                           int channels, int count, int interpolation, void* stream);
/* The per-frame Resize() loop of bridge.c:588-604 over frames that all DIFFER in size (BASELINE configs[4], frames already
 * in HBM): each item is cvResize'd with the interpolation Resize() picks for it -- NN when `simple`, CUBIC when either
 * side grows, AREA otherwise (bridge.c:188-192) -- but frames that share a kernel ride in one launch (a descriptor per
 * frame), so a run of thumbnails costs a handful of launches instead of one per request.  Same bytes as calling
 * impgpu_batch_cv_resize once per item.  Nothing is launched if any item is malformed (IMP_ERROR_INVALID_ARGS). */
typedef struct impgpu_resize_item {
    const void* src; int src_width, src_height, src_step;
    void*       dst; int dst_width, dst_height, dst_step;
} impgpu_resize_item;
int impgpu_batch_resize_mixed(const impgpu_resize_item* items, int count, int channels, int simple, void* stream);
/* cfg3 chain on a batch: resize (AREA/CUBIC by the reference's rule) -> rotate -> watermark.
 * rotate in {0, 90, 180, 270}; config->watermark may be NULL. dst geometry must match. */
int impgpu_batch_resize_rotate_watermark(const void* src, long long src_frame_stride, int src_width, int src_height, int src_step,
                                         void* dst, long long dst_frame_stride, int dst_step,
                                         int resize_width, int resize_height, int rotate,
                                         const impgpu_config* config, int channels, int count, void* stream);

/* A run of pointwise filter-* requests (everything but flip / rotate / blur) applied in place to every frame of a
 * batch with ONE fused launch: the per-frame filter loop of bridge.c:606-627 for albums (animated GIFs).  Returns
 * IMP_ERROR_UNSUPPORTED if a request is not pointwise (the caller then goes frame by frame), otherwise the code the
 * first failing Filter() would return. */
int impgpu_batch_filters(void* frames, long long frame_stride, int width, int height, int channels, int step, int count,
                         const char* const* filters, int filter_count, int allow_experiments, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IMPGPU_H */
