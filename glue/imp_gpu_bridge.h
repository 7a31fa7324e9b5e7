/*
 * imp_gpu_bridge.h -- the nginx-side glue between ngx_http_imgproc's RunJob and libimpgpu.so.
 *
 * Compiled INTO the reference module (it uses the reference's own types: Album, Frame, Config, Memory from required.h),
 * next to bridge.c; glue/apply_glue.sh makes the edits in the reference's files that call these functions:
 *   bridge.c:10-16    OnEnvStart / OnEnvDestroy bodies      -> ImpGpuEnvStart / ImpGpuEnvDestroy
 *   bridge.c:574-656  crop / resize / filter / watermark / flatten loops over the album -> ImpGpuOperators
 *   bridge.c:661      Info()                                -> ImpGpuInfo      (brightness reduced on the device)
 *   bridge.c:669-670  ASCII()                               -> ImpGpuASCII
 *   bridge.c:681      before either encoder runs            -> ImpGpuDownload  (frames back into IplImages)
 *   bridge.c:714      finalize:                             -> ImpGpuRelease
 *   required.h:117    Config gains `void* WatermarkDevice`  (per-worker handle of the uploaded overlay)
 * Needs nginx, OpenCV 2.4 and FreeImage headers exactly like the files around it, so it is not built in this repository;
 * the C ABI underneath it is exercised from C by tests/c/runjob_harness.c.
 */
#ifndef IMP_GPU_BRIDGE_H
#define IMP_GPU_BRIDGE_H

#include <impgpu.h>

typedef struct {
    impgpu_image** Frames;   /* device-resident counterparts of Album.Frames[i].Image, ngx_palloc'ed in req->pool */
    int            Count;
} ImpGpuAlbum;

/* once per worker process, after fork (module.c:100-107).  worker = ngx_worker: worker i drives GPU i mod #GPUs. */
void   ImpGpuEnvStart(int worker);
void   ImpGpuEnvDestroy(void);

/* Steps 3-7 of RunJob for every frame of the album: upload, then crop -> resize -> [gray->BGR] -> filters -> watermark ->
 * flatten in the reference's fixed order.  `lacksAlpha` = the chosen encoder cannot store alpha (bridge.c:643-647).
 * Returns the IMP_* code and leaves the failing IMP_STEP_* in *step (JobResult.Step). */
int    ImpGpuOperators(Album* album, ImpGpuAlbum* gpu, ngx_pool_t* pool, char* crop, char* gravity, char* resize, int simple,
                       char** filters, int filterCount, int lacksAlpha, Config* config, int* step);
u_char* ImpGpuInfo(ImpGpuAlbum* gpu, Album* album, ngx_pool_t* pool);            /* bridge.c:283-300 */
Memory ImpGpuASCII(ImpGpuAlbum* gpu, char* args, ngx_pool_t* pool);              /* filters.c:488-522 */
int    ImpGpuDownload(ImpGpuAlbum* gpu, Album* album);                           /* results -> fresh IplImages */
void   ImpGpuRelease(ImpGpuAlbum* gpu);

#endif
