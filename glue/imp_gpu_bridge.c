/*
 * imp_gpu_bridge.c -- see imp_gpu_bridge.h.  C (gnu99) like the module it is compiled into.
 */
#include "required.h"
#include "helpers.h"
#include "imp_gpu_bridge.h"
#include <string.h>

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
    /* SIG_JPG / SIG_PNG (bridge.c:8).  Anything the device decoders do not take -- progressive or CMYK JPEGs, 16-bit,
     * palette or interlaced PNGs, damaged files: a non-zero code -- goes to cvDecodeImage exactly as before, so no request
     * changes its answer */
    static const u_char png[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    int isJpeg = size >= 3 && blob[0] == 0xFF && blob[1] == 0xD8 && blob[2] == 0xFF;
    int isPng  = size >= 8 && memcmp(blob, png, 8) == 0;
    if (!isJpeg && !isPng) {
        return 0;
    }
    {
        int rc = isJpeg ? impgpu_image_decode_jpeg(blob, size, &frame) : impgpu_image_decode_png(blob, size, &frame);
        if (rc == IMP_ERROR_UNSUPPORTED || rc == IMP_ERROR_DECODE_FAILED) {
            return 0;               /* not this decoder's file: cvDecodeImage, as before */
        }
        if (rc != IMP_OK) {
            return -rc;             /* the device (or its memory) failed: the request fails here */
        }
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
    /* a realistic first size (half a byte per sample and some headroom; the worst case is 432 bytes per 8x8 block: 21 MB
     * for a 1080p answer, held until the request ends, where the reference allocates the encoded length, bridge.c:705-706);
     * when the file is larger the call says by how much in *length, and one retry has exactly that */
    size_t capacity = (size_t)impgpu_image_width(gpu->Handle) * impgpu_image_height(gpu->Handle) / 2 + 65536;
    u_char* output = ngx_palloc(pool, capacity);
    int rc;
    if (!output) {
        return IMP_ERROR_MALLOC_FAILED;
    }
    rc = impgpu_image_encode_jpeg(gpu->Handle, quality, output, capacity, length);
    if (rc == IMP_ERROR_MALLOC_FAILED && *length > capacity) {
        ngx_pfree(pool, output);
        capacity = *length;
        output = ngx_palloc(pool, capacity);
        if (!output) {
            return IMP_ERROR_MALLOC_FAILED;
        }
        rc = impgpu_image_encode_jpeg(gpu->Handle, quality, output, capacity, length);
    }
    if (rc) {
        return rc;
    }
    *bytes = output;
    return IMP_OK;
}

void ImpGpuRelease(ImpGpuAlbum* gpu) {
    impgpu_image_release(&gpu->Handle);
}

#ifdef IMP_FEATURE_ADVANCED_IO
/* ---- the FreeImage side.  LoadGIF (advancedio.c:103-274) keeps its walk over the pages -- metadata, the 8-bit conversion,
 * the lock / unlock -- and hands every page's indices and palette to ImpGpuGifPage instead of compositing it pixel by
 * pixel (advancedio.c:187-248); ImpGpuGifCompose then runs that loop for all pages on the device. */
int ImpGpuGifPage(ImpGpuGif* gif, ngx_pool_t* pool, int frameid, int framecount, const unsigned char* bits, int w, int h, int pitch,
                  int left, int top, int dispose, int key, const void* palette, int canvasW, int canvasH) {
    impgpu_gif_page* p;
    unsigned char* copy;
    (void)canvasW; (void)canvasH;                   /* (the canvas is the first page's size: impgpu_gif_compose takes it from there) */
    if (!gif->Pages) {
        gif->Pages = ngx_pcalloc(pool, framecount * sizeof(impgpu_gif_page));
        gif->Count = 0;
        if (!gif->Pages) {
            return IMP_ERROR_MALLOC_FAILED;
        }
    }
    if (frameid != gif->Count || frameid >= framecount) {
        return IMP_ERROR_INVALID_ARGS;
    }
    /* the page is unlocked (or unloaded) before the next one is read: keep what the compositing needs */
    copy = ngx_palloc(pool, (size_t)h * pitch + 1024);
    if (!copy) {
        return IMP_ERROR_MALLOC_FAILED;
    }
    memcpy(copy, bits, (size_t)h * pitch);
    memcpy(copy + (size_t)h * pitch, palette, 1024);                /* 256 RGBQUADs */
    p = &gif->Pages[gif->Count++];
    p->indices = copy;
    p->width = w; p->height = h; p->pitch = pitch;
    p->left = left; p->top = top;
    p->dispose = dispose;
    p->transparency_key = key;
    p->palette = copy + (size_t)h * pitch;
    return IMP_OK;
}

int ImpGpuGifCompose(ImpGpuGif* gif, int isdestructive, int page, Album* result) {
    impgpu_image* frames = NULL;
    int rc;
    if (!gif->Pages || gif->Count < 1) {
        return IMP_ERROR_DECODE_FAILED;
    }
    rc = impgpu_gif_compose_album(gif->Pages, gif->Count, isdestructive, page, &frames);
    if (rc == IMP_OK) {
        result->Device = frames;                    /* RunJob takes it over (gpu.Handle) right after FiLoadFrames */
    }
    return rc;
}

/* LoadSingle (advancedio.c:276-321): the 32-bit bitmap goes to the device as it is, bottom-up; the flip into a top-down
 * 4-channel frame (advancedio.c:310-318) happens there */
int ImpGpuLoadSingle(Album* result, ngx_pool_t* pool, const unsigned char* bits, int w, int h, int pitch) {
    impgpu_image* frame = NULL;
    int rc = impgpu_image_upload_fi32(bits, w, h, pitch, &frame);
    if (rc) {
        return rc;
    }
    result->Frames = ngx_palloc(pool, sizeof(Frame));
    if (!result->Frames) {
        impgpu_image_release(&frame);
        return IMP_ERROR_MALLOC_FAILED;
    }
    result->Count = 1;
    result->Frames[0].Image = NULL;
    result->Frames[0].Time = result->Frames[0].TransparencyKey = result->Frames[0].Dispose = 0;
    result->Device = frame;
    return IMP_OK;
}

int ImpGpuFrameWidth(void* device)  { return impgpu_image_width((impgpu_image*)device); }
int ImpGpuFrameHeight(void* device) { return impgpu_image_height((impgpu_image*)device); }

/* SaveSingle (advancedio.c:427-446): IplToFI32 / IplToFI24 (advancedio.c:65-101) -- the flip and the 32 / 24-bit repack -- run
 * on the device and land in the bitmap FreeImage is about to encode */
int ImpGpuFetchFi(void* device, int bpp, unsigned char* bits, int pitch) {
    return impgpu_image_download_fi((impgpu_image*)device, bpp, bits, pitch);
}
#endif
