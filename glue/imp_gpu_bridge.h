/*
 * imp_gpu_bridge.h -- the nginx-side glue between ngx_http_imgproc's RunJob and libimpgpu.so.
 *
 * Compiled INTO the reference module (it uses the reference's own types: Album, Frame, Config, Memory from required.h),
 * next to bridge.c; glue/apply_glue.sh makes the edits in the reference's files that call these functions:
 *   bridge.c:10-16    OnEnvStart / OnEnvDestroy bodies      -> ImpGpuEnvStart / ImpGpuEnvDestroy
 *   bridge.c:545      before cvDecodeImage                  -> ImpGpuDecode     (a JPEG / PNG is decoded on the device; else as before)
 *   bridge.c:574-656  crop / resize / filter / watermark / flatten loops over the album -> ImpGpuOperators
 *   bridge.c:661      Info()                                -> ImpGpuInfo      (brightness reduced on the device)
 *   bridge.c:669-670  ASCII()                               -> ImpGpuASCII
 *   bridge.c:681      before an encoder that reads IplImages -> ImpGpuDownload  (frames back into IplImages)
 *   bridge.c:703-709  cvEncodeImage(".jpg")                 -> ImpGpuEncodeJpeg (the file is written on the device)
 *   bridge.c:714      finalize:                             -> ImpGpuRelease
 *   required.h:117    Config gains `void* WatermarkDevice`  (per-worker handle of the uploaded overlay)
 *   required.h:139    Album gains `void* Device`            (the frames FiLoadFrames put into HBM / SaveSingle fetches from it)
 * and in advancedio.c (round 4: the FreeImage side hands frames over without an IplImage in between):
 *   advancedio.c:187-248  LoadGIF's per-pixel compositing loop  -> ImpGpuGifPage (collect) + ImpGpuGifCompose at :260
 *   advancedio.c:295-318  LoadSingle's flip-and-copy loop       -> ImpGpuLoadSingle
 *   advancedio.c:428-429  SaveSingle's IplToFI32 / IplToFI24    -> ImpGpuFetchFi into the bitmap FreeImage encodes
 * Needs nginx, OpenCV 2.4 and FreeImage headers exactly like the files around it, so it is not BUILT in this repository;
 * tests/test_glue.py compiles it (and the patched bridge.c) with -fsyntax-only against declaration-only stand-ins for
 * those headers (tests/c/decls/, a compile check of this glue and nothing else), and the C ABI underneath it is exercised
 * from C by tests/c/runjob_harness.c.
 */
#ifndef IMP_GPU_BRIDGE_H
#define IMP_GPU_BRIDGE_H

#include <impgpu.h>
#include <impgpu_broker.h>

/* The device-resident counterpart of the request's Album: ONE handle for all of its frames (every frame of an album has
 * the canvas' geometry, advancedio.c:187, so they share a block and each operator is one launch for the whole animation;
 * impgpu_album_upload in impgpu.h).
 *
 * BROKER MODE (round 5; $IMPGPU_BROKER names the segment of an `impgpu_broker`, include/impgpu_broker.h): the worker never
 * touches HIP.  ImpGpuDecode and ImpGpuOperators only NOTE the file and the job; the request's exit -- ImpGpuEncodeJpeg,
 * ImpGpuDownload, ImpGpuInfo, ImpGpuASCII -- knows what kind of answer is wanted and makes the ONE round trip that
 * carries {file, job, config} to the process that owns the GPU, where it rides a launch with whatever other workers
 * have queued.  A failure comes back with the answer and names its own step (JobResult.Step through `Step`).  Requests
 * whose frames come from FreeImage (IMP_FEATURE_ADVANCED_IO: GIF albums, LoadSingle) still need a device in the worker:
 * their first occurrence starts an in-process env (ImpGpuEnvStart's other branch), everything else goes to the broker. */
typedef struct {
    impgpu_image* Handle;
    /* ---- broker mode only (zero otherwise) ---- */
    const u_char* Blob;                 /* the file ImpGpuDecode was shown: sent as it is */
    size_t        BlobSize;
    int           Deferred;             /* the operators have been noted, nothing has run yet */
    impgpu_job    Job;                  /* (its strings live in the request pool, like RunJob's own) */
    impgpu_config Cfg;
    int           WatermarkId;
    int*          Step;                 /* &answer->Step */
    Album*        Source;               /* the request's album (frames decoded on the host go as pixels) */
} ImpGpuAlbum;

/* once per worker process, after fork (module.c:100-107).  worker = ngx_worker (nginx >= 1.9.1) or ngx_process_slot:
 * worker i drives GPU i mod #GPUs. */
#if defined(nginx_version) && nginx_version >= 1009001
#define IMP_GPU_WORKER_INDEX ((int)ngx_worker)
#else
#define IMP_GPU_WORKER_INDEX ((int)ngx_process_slot)
#endif
void   ImpGpuEnvStart(int worker);
void   ImpGpuEnvDestroy(void);

/* bridge.c:545-552 for a JPEG or PNG blob: returns 1 when the file was decoded on the device (album gets its one frame with
 * Image = NULL, gpu holds the device frame), 0 when the caller must decode on the host as before (neither format, a file the
 * device decoders do not take, or a damaged one), and -IMP_ERROR_* when the DEVICE failed: the request then fails at
 * its DECODE step (HTTP 500) instead of paying for a host decode whose upload would fail the same way. */
int    ImpGpuDecode(u_char* blob, size_t size, Album* album, ImpGpuAlbum* gpu, ngx_pool_t* pool);

/* Steps 3-7 of RunJob for all frames of the album at once: upload (unless ImpGpuDecode put the frame there), then crop ->
 * resize -> [gray->BGR] -> filters -> watermark -> flatten in the reference's fixed order, each ONE launch over the album.  `lacksAlpha` = the chosen encoder cannot store alpha (bridge.c:643-647).
 * Returns the IMP_* code and leaves the failing IMP_STEP_* in *step (JobResult.Step). */
int    ImpGpuOperators(Album* album, ImpGpuAlbum* gpu, ngx_pool_t* pool, char* crop, char* gravity, char* resize, int simple,
                       char** filters, int filterCount, int lacksAlpha, Config* config, int* step);
u_char* ImpGpuInfo(ImpGpuAlbum* gpu, Album* album, ngx_pool_t* pool, int* code); /* bridge.c:283-300; NULL + *code on a device error */
Memory ImpGpuASCII(ImpGpuAlbum* gpu, char* args, ngx_pool_t* pool);              /* filters.c:488-522 */
int    ImpGpuDownload(ImpGpuAlbum* gpu, Album* album, ngx_pool_t* pool);         /* results -> fresh IplImages, one wait */
/* bridge.c:703-709 when the answer is a JPEG from the basic encoder: the file is written on the device, nothing is downloaded */
int    ImpGpuEncodeJpeg(ImpGpuAlbum* gpu, int quality, ngx_pool_t* pool, u_char** bytes, size_t* length);
void   ImpGpuRelease(ImpGpuAlbum* gpu);

/* ---- the FreeImage side (advancedio.c, IMP_FEATURE_ADVANCED_IO) ---- */
/* the pages of a GIF as LoadGIF walks them (advancedio.c:128-259): indices and palette are copied out of the locked page */
typedef struct {
    impgpu_gif_page* Pages;
    int Count;
} ImpGpuGif;
int    ImpGpuGifPage(ImpGpuGif* gif, ngx_pool_t* pool, int frameid, int framecount, const unsigned char* bits, int w, int h, int pitch,
                     int left, int top, int dispose, int key, const void* palette, int canvasW, int canvasH);
int    ImpGpuGifCompose(ImpGpuGif* gif, int isdestructive, int page, Album* result);     /* advancedio.c:204-247 on the device */
int    ImpGpuLoadSingle(Album* result, ngx_pool_t* pool, const unsigned char* bits, int w, int h, int pitch);   /* advancedio.c:295-318 */
int    ImpGpuFrameWidth(void* device);
int    ImpGpuFrameHeight(void* device);
int    ImpGpuFetchFi(void* device, int bpp, unsigned char* bits, int pitch);              /* advancedio.c:65-101 */

#endif
