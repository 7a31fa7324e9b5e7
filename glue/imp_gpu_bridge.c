/*
 * imp_gpu_bridge.c -- see imp_gpu_bridge.h.  C (gnu99) like the module it is compiled into.
 */
#include "required.h"
#include "helpers.h"
#include "imp_gpu_bridge.h"

void ImpGpuEnvStart(int worker) {
    /* HIP is initialised here, in the worker, never in the master before fork.  A failure is not fatal for nginx:
     * every later call then answers IMP_ERROR_DEVICE, which BodyFilter maps to 500 (module.c:305). */
    if (impgpu_env_start(worker) != IMP_OK) {
        fprintf(stderr, "imp::no usable GPU for worker %d: %s\n", worker, impgpu_last_error());
        return;
    }
    /* the worker runs next to its GPU: staging copies and pinned buffers on the device's NUMA node (a no-op on a one-node
     * host or when the node's CPUs are not open to this process) */
    (void)impgpu_env_bind_thread();
}

void ImpGpuEnvDestroy(void) {
    impgpu_env_destroy();
}

/* The Config fields the operators read (required.h:108-118) in the ABI's plain struct.  The overlay is uploaded by the
 * first request that needs it in this worker: PrepareWatermark ran in the master (module.c:159) and parked the decoded
 * pixels in the conf pool, which every forked worker inherits; the device handle is per worker AND per location --
 * each location's Config is its own object (OnConfigMerge, module.c:130-190) -- so it lives in that Config. */
static int FillConfig(Config* config, impgpu_config* g) {
    memset(g, 0, sizeof(*g));
    g->max_target_w      = config->MaxTargetDimensions->W;
    g->max_target_h      = config->MaxTargetDimensions->H;
    g->max_filters_count = (int)config->MaxFiltersCount;
    g->allow_experiments = (int)config->AllowExperiments;
    if (config->WatermarkInfo) {
        if (!config->WatermarkDevice) {
            RecoverInfo* inf = config->WatermarkInfo;
            impgpu_config once;
            memset(&once, 0, sizeof(once));
            int rc = impgpu_prepare_watermark(&once, inf->Pointer, inf->Size.width, inf->Size.height, inf->Channels, inf->Step);
            if (rc) {
                return rc;
            }
            config->WatermarkDevice = once.watermark;
        }
        g->watermark           = (impgpu_image*)config->WatermarkDevice;
        g->watermark_opacity   = (int)config->WatermarkOpacity;
        g->watermark_gravity_x = config->WatermarkPosition->GravityX;
        g->watermark_gravity_y = config->WatermarkPosition->GravityY;
        g->watermark_offset_x  = config->WatermarkPosition->OffsetX;
        g->watermark_offset_y  = config->WatermarkPosition->OffsetY;
    }
    return IMP_OK;
}

int ImpGpuDecode(u_char* blob, size_t size, Album* album, ImpGpuAlbum* gpu, ngx_pool_t* pool) {
    impgpu_image* frame = NULL;
    /* SIG_JPG (bridge.c:8).  Anything the device decoder does not take -- progressive, CMYK, damaged files: a non-zero
     * code -- goes to cvDecodeImage exactly as before, so no request changes its answer */
    if (size < 3 || blob[0] != 0xFF || blob[1] != 0xD8 || blob[2] != 0xFF) {
        return 0;
    }
    if (impgpu_image_decode_jpeg(blob, size, &frame) != IMP_OK) {
        return 0;
    }
    album->Frames = ngx_palloc(pool, sizeof(Frame));
    if (!album->Frames) {
        impgpu_image_release(&frame);
        return 0;
    }
    gpu->Handle = frame;
    /* the host never sees the decoded pixels: Image stays NULL until ImpGpuDownload creates the encoder's input
     * (cvReleaseImage at bridge.c:719 accepts a NULL image) */
    album->Count = 1;
    album->Frames[0].Image = NULL;
    album->Frames[0].Time = album->Frames[0].Dispose = album->Frames[0].TransparencyKey = 0;
    return 1;
}

int ImpGpuOperators(Album* album, ImpGpuAlbum* gpu, ngx_pool_t* pool, char* crop, char* gravity, char* resize, int simple,
                    char** filters, int filterCount, int lacksAlpha, Config* config, int* step) {
    impgpu_config gcfg;
    impgpu_job job;
    int fid;

    *step = IMP_STEP_WATERMARK;
    int rc = FillConfig(config, &gcfg);
    if (rc) {
        return rc;
    }

    job.crop         = crop;
    job.gravity      = gravity;
    job.resize       = resize;
    job.simple       = simple;
    job.filters      = (const char* const*)filters;
    job.filter_count = filterCount;
    job.need_flatten = lacksAlpha;      /* applied only to 4-channel frames, like bridge.c:642-656 */

    if (!gpu->Handle) {             /* (a frame ImpGpuDecode put on the device is already there) */
        const unsigned char** rows = ngx_palloc(pool, album->Count * sizeof(unsigned char*));
        int* steps = ngx_palloc(pool, album->Count * sizeof(int));
        IplImage* first = album->Frames[0].Image;
        *step = IMP_STEP_DECODE;
        if (!rows || !steps) {
            return IMP_ERROR_MALLOC_FAILED;
        }
        for (fid = 0; fid < album->Count; fid++) {
            IplImage* image = album->Frames[fid].Image;
            if (image->width != first->width || image->height != first->height || image->nChannels != first->nChannels) {
                return IMP_ERROR_INVALID_ARGS;      /* not an Album LoadGIF / the decoders can produce */
            }
            rows[fid]  = (const unsigned char*)image->imageData;
            steps[fid] = image->widthStep;
        }
        rc = impgpu_album_upload(rows, album->Count, first->width, first->height, first->nChannels, steps, &gpu->Handle);
        if (rc) {
            return rc;
        }
    }
    /* nothing is waited for here: the upload, every operator (one launch each for the whole album) and the next request's
     * upload overlap on the worker's stream */
    return impgpu_run_ops(&gpu->Handle, &job, &gcfg, step);
}

u_char* ImpGpuInfo(ImpGpuAlbum* gpu, Album* album, ngx_pool_t* pool, int* code) {
    float brightness = 0;
    u_char* json = ngx_palloc(pool, 256 * sizeof(u_char));
    *code = json ? impgpu_calc_perceived_brightness(gpu->Handle, &brightness) : IMP_ERROR_MALLOC_FAILED;
    if (*code) {                    /* a lost device must not read as "brightness 0", HTTP 200 */
        return NULL;
    }
    sprintf(
        (char*)json,
        "{"
            "\"width\":%d,"
            "\"height\":%d,"
            "\"brightness\":%d,"
            "\"count\":%d"
        "}",
        impgpu_image_width(gpu->Handle),
        impgpu_image_height(gpu->Handle),
        (int)round(brightness * 100),
        album->Count
    );
    return json;
}

Memory ImpGpuASCII(ImpGpuAlbum* gpu, char* args, ngx_pool_t* pool) {
    Memory result;
    impgpu_image* image = gpu->Handle;      /* frame 0 of an album, like bridge.c:669 */
    long buflen = (long)(impgpu_image_width(image) + 1) * impgpu_image_height(image) - 1;
    result.Buffer = ngx_palloc(pool, buflen > 0 ? buflen : 1);
    result.Length = 0;
    result.Error  = result.Buffer ? impgpu_ascii(image, args, result.Buffer, buflen, &result.Length) : IMP_ERROR_MALLOC_FAILED;
    return result;
}

int ImpGpuDownload(ImpGpuAlbum* gpu, Album* album, ngx_pool_t* pool) {
    int fid, rc = IMP_OK;
    int count = impgpu_album_count(gpu->Handle);
    IplImage** fresh = ngx_pcalloc(pool, count * sizeof(IplImage*));
    unsigned char** rows = ngx_pcalloc(pool, count * sizeof(unsigned char*));
    int* steps = ngx_pcalloc(pool, count * sizeof(int));
    if (!fresh || !rows || !steps || count != album->Count) {
        return IMP_ERROR_MALLOC_FAILED;
    }
    for (fid = 0; fid < count; fid++) {
        /* same header rules as every cvCreateImage in bridge.c: 8-bit, rows padded to 4 bytes -- the layout the
         * device frames already have, so cvEncodeImage / IplToFI32 / IplToFI24 read them unchanged */
        fresh[fid] = cvCreateImage(cvSize(impgpu_image_width(gpu->Handle), impgpu_image_height(gpu->Handle)), IPL_DEPTH_8U,
                                   impgpu_image_channels(gpu->Handle));
        if (!fresh[fid] || !fresh[fid]->imageData) {
            rc = IMP_ERROR_MALLOC_FAILED;
            break;
        }
        rows[fid]  = (unsigned char*)fresh[fid]->imageData;
        steps[fid] = fresh[fid]->widthStep;
    }
    /* all frames of the album in one transfer, one wait */
    if (!rc) {
        rc = impgpu_album_download(gpu->Handle, rows, steps);
    }
    for (fid = 0; fid < count; fid++) {
        if (rc) {
            if (fresh[fid]) {
                cvReleaseImage(&fresh[fid]);
            }
        } else {
            IplImage* old = album->Frames[fid].Image;
            cvReleaseImage(&old);
            album->Frames[fid].Image = fresh[fid];
        }
    }
    return rc;
}

int ImpGpuEncodeJpeg(ImpGpuAlbum* gpu, int quality, ngx_pool_t* pool, u_char** bytes, size_t* length) {
    /* cvEncodeImage(".jpg", album.Frames[0].Image, basicCoderopt) at bridge.c:703-709 for the frame in HBM: the same file,
     * and the compressed bytes are all that crosses the link.  The buffer is sized for the worst case and lives in the
     * request pool like the reference's own copy of the encoder's output. */
    size_t capacity = impgpu_jpeg_encode_bound(impgpu_image_width(gpu->Handle), impgpu_image_height(gpu->Handle),
                                               impgpu_image_channels(gpu->Handle));
    u_char* output = capacity ? ngx_palloc(pool, capacity) : NULL;
    if (!output) {
        return IMP_ERROR_MALLOC_FAILED;
    }
    int rc = impgpu_image_encode_jpeg(gpu->Handle, quality, output, capacity, length);
    if (rc) {
        return rc;
    }
    *bytes = output;
    return IMP_OK;
}

void ImpGpuRelease(ImpGpuAlbum* gpu) {
    impgpu_image_release(&gpu->Handle);
}
