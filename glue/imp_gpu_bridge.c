/*
 * imp_gpu_bridge.c -- see imp_gpu_bridge.h.  C (gnu99) like the module it is compiled into.
 */
#include "required.h"
#include "helpers.h"
#include "imp_gpu_bridge.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- broker mode: include/impgpu_broker.h.  g_broker_name != NULL: this worker is a client of `impgpu_broker`. ---- */
static const char*    g_broker_name;
static impgpu_client* g_client;
static int            g_worker = 0;
static int            g_own_env = 0;
static struct { Config* Cfg; int Id; } g_broker_marks[16];   /* per location: the id impgpu_client_prepare_watermark gave */
static int            g_broker_nmarks;

static int BrokerMode(void) { return g_broker_name != NULL; }

static int OwnEnv(void) {
    /* HIP is initialised here, in the worker, never in the master before fork.  A failure is not fatal for nginx:
     * every later call then answers IMP_ERROR_DEVICE, which BodyFilter maps to 500 (module.c:305). */
    if (g_own_env) {
        return IMP_OK;
    }
    if (impgpu_env_start(g_worker) != IMP_OK) {
        fprintf(stderr, "imp::no usable GPU for worker %d: %s\n", g_worker, impgpu_last_error());
        return IMP_ERROR_DEVICE;
    }
    /* the worker runs next to its GPU: staging copies and pinned buffers on the device's NUMA node (a no-op on a one-node
     * host or when the node's CPUs are not open to this process) */
    (void)impgpu_env_bind_thread();
    g_own_env = 1;
    return IMP_OK;
}

void ImpGpuEnvStart(int worker) {
    const char* name = getenv("IMPGPU_BROKER");
    g_worker = worker;
    if (name && *name) {
        /* one broker per GPU, one segment each: "/impgpu-broker-%d" with $IMPGPU_BROKER_COUNT brokers spreads the workers
         * round-robin like impgpu_env_start(worker) spreads them over the devices */
        static char chosen[128];
        const char* n = getenv("IMPGPU_BROKER_COUNT");
        int brokers = n ? atoi(n) : 1;
        snprintf(chosen, sizeof(chosen), name, brokers > 0 ? worker % brokers : 0);
        g_broker_name = chosen;
        /* (the broker may come up after the workers: a failed attach is tried again by the first request) */
        if (impgpu_client_attach(g_broker_name, &g_client) != IMP_OK) {
            fprintf(stderr, "imp::worker %d: %s\n", worker, impgpu_client_last_error());
        }
        return;
    }
    (void)OwnEnv();
}

void ImpGpuEnvDestroy(void) {
    impgpu_client_detach(&g_client);
    impgpu_env_destroy();
}

static int BrokerClient(void) {
    if (!g_client && impgpu_client_attach(g_broker_name, &g_client) != IMP_OK) {
        return IMP_ERROR_DEVICE;
    }
    return IMP_OK;
}

/* The one round trip of a request in broker mode.  A file the device decoders do not take (IMPB_NOT_TAKEN: progressive,
 * CMYK, damaged ...) is decoded here by cvDecodeImage exactly as bridge.c:545-552 does and goes again as pixels, so no
 * request changes its answer.  Returns the request's IMP_* code; a failure's step goes to *gpu->Step. */
static int BrokerRun(ImpGpuAlbum* gpu, Album* album, int outKind, int quality, const char* asciiArgs, impgpu_client_answer* a) {
    impgpu_client_request r;
    IplImage* decoded = NULL;
    int rc = BrokerClient();
    if (rc) {
        return rc;
    }
    memset(&r, 0, sizeof(r));
    r.job = gpu->Deferred ? &gpu->Job : NULL;
    r.config = &gpu->Cfg;
    r.watermark_id = gpu->WatermarkId;
    r.out_kind = outKind;
    r.quality = quality;
    r.ascii_args = asciiArgs;
    if (gpu->Blob) {
        r.in_kind = IMPB_IN_FILE;
        r.input = gpu->Blob;
        r.input_bytes = gpu->BlobSize;
        rc = impgpu_client_run(g_client, &r, a);
        if (rc == IMP_OK && a->code == IMPB_NOT_TAKEN) {
            CvMat rawencoded = cvMat(1, (int)gpu->BlobSize, CV_8UC1, (void*)gpu->Blob);
            decoded = cvDecodeImage(&rawencoded, -1);
            if (!decoded) {
                if (gpu->Step) {
                    *gpu->Step = IMP_STEP_DECODE;
                }
                return IMP_ERROR_DECODE_FAILED;
            }
        } else {
            goto answered;
        }
    } else if (album->Count != 1 || !album->Frames[0].Image) {
        return IMP_ERROR_UNSUPPORTED;       /* (albums of several frames come from FreeImage: the in-process path) */
    }
    {
        IplImage* image = decoded ? decoded : album->Frames[0].Image;
        r.in_kind = IMPB_IN_FRAME;
        r.input = (const unsigned char*)image->imageData;
        r.input_bytes = (size_t)image->widthStep * image->height;
        r.width = image->width; r.height = image->height; r.channels = image->nChannels; r.step = image->widthStep;
        rc = impgpu_client_run(g_client, &r, a);
        if (decoded) {
            cvReleaseImage(&decoded);
        }
    }
answered:
    if (rc) {
        fprintf(stderr, "imp::broker: %s\n", impgpu_client_last_error());
        return rc;                          /* IMP_ERROR_DEVICE: the request fails like a lost device (HTTP 500) */
    }
    if (a->code) {
        if (gpu->Step) {
            *gpu->Step = a->step;
        }
        return a->code > 0 ? a->code : IMP_ERROR_DECODE_FAILED;
    }
    return IMP_OK;
}

/* The Config fields the operators read (required.h:108-118) in the ABI's plain struct.  The overlay is uploaded by the
 * first request that needs it in this worker: PrepareWatermark ran in the master (module.c:159) and parked the decoded
 * pixels in the conf pool, which every forked worker inherits; the device handle is per worker AND per location --
 * each location's Config is its own object (OnConfigMerge, module.c:130-190) -- so it lives in that Config. */
static int FillConfig(Config* config, impgpu_config* g, int* brokerMark) {
    memset(g, 0, sizeof(*g));
    if (brokerMark) {
        *brokerMark = 0;
    }
    g->max_target_w      = config->MaxTargetDimensions->W;
    g->max_target_h      = config->MaxTargetDimensions->H;
    g->max_filters_count = (int)config->MaxFiltersCount;
    g->allow_experiments = (int)config->AllowExperiments;
    if (config->WatermarkInfo && brokerMark) {
        /* broker mode: the overlay's pixels are registered with the broker once per worker and location (the client
         * registers them again by itself when the broker has been replaced) */
        int i, id = 0;
        for (i = 0; i < g_broker_nmarks && !id; i++) {
            if (g_broker_marks[i].Cfg == config) {
                id = g_broker_marks[i].Id;
            }
        }
        if (!id) {
            RecoverInfo* inf = config->WatermarkInfo;
            int rc = BrokerClient();
            if (!rc && g_broker_nmarks >= 16) {
                rc = IMP_ERROR_NO_SUCH_WATERMARK;
            }
            if (!rc) {
                rc = impgpu_client_prepare_watermark(g_client, inf->Pointer, inf->Size.width, inf->Size.height, inf->Channels, inf->Step, &id);
            }
            if (rc) {
                return rc;
            }
            g_broker_marks[g_broker_nmarks].Cfg = config;
            g_broker_marks[g_broker_nmarks++].Id = id;
        }
        *brokerMark = id;
        g->watermark_opacity   = (int)config->WatermarkOpacity;
        g->watermark_gravity_x = config->WatermarkPosition->GravityX;
        g->watermark_gravity_y = config->WatermarkPosition->GravityY;
        g->watermark_offset_x  = config->WatermarkPosition->OffsetX;
        g->watermark_offset_y  = config->WatermarkPosition->OffsetY;
    } else if (config->WatermarkInfo) {
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
    if (BrokerMode()) {
        /* noted, not sent: the request's exit makes the one round trip (imp_gpu_bridge.h) */
        album->Frames = ngx_palloc(pool, sizeof(Frame));
        if (!album->Frames) {
            return 0;
        }
        gpu->Blob = blob;
        gpu->BlobSize = size;
        album->Count = 1;
        album->Frames[0].Image = NULL;
        album->Frames[0].Time = album->Frames[0].Dispose = album->Frames[0].TransparencyKey = 0;
        return 1;
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
    if (BrokerMode() && !gpu->Handle && !album->Device) {
        /* noted: crop / resize / filters / watermark / flatten run in the broker, in the reference's order, when the exit
         * asks for the answer (a wrong argument then fails the request THERE, with the step the broker names) */
        int rcb = FillConfig(config, &gpu->Cfg, &gpu->WatermarkId);
        if (rcb) {
            return rcb;
        }
        gpu->Job.crop         = crop;
        gpu->Job.gravity      = gravity;
        gpu->Job.resize       = resize;
        gpu->Job.simple       = simple;
        gpu->Job.filters      = (const char* const*)filters;
        gpu->Job.filter_count = filterCount;
        gpu->Job.need_flatten = lacksAlpha;
        gpu->Deferred = 1;
        gpu->Step = step;
        gpu->Source = album;
        *step = IMP_STEP_INFO;
        return IMP_OK;
    }
    if (BrokerMode() && OwnEnv() != IMP_OK) {       /* frames from FreeImage: this worker needs a device of its own after all */
        *step = IMP_STEP_DECODE;
        return IMP_ERROR_DEVICE;
    }
    int rc = FillConfig(config, &gcfg, NULL);
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
    int width, height;
    if (json && BrokerMode() && !gpu->Handle) {
        impgpu_client_answer a;
        *code = BrokerRun(gpu, album, IMPB_OUT_INFO, 0, NULL, &a);
        brightness = a.brightness; width = a.width; height = a.height;
    } else {
        *code = json ? impgpu_calc_perceived_brightness(gpu->Handle, &brightness) : IMP_ERROR_MALLOC_FAILED;
        width = impgpu_image_width(gpu->Handle); height = impgpu_image_height(gpu->Handle);
    }
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
        width,
        height,
        (int)round(brightness * 100),
        album->Count
    );
    return json;
}

Memory ImpGpuASCII(ImpGpuAlbum* gpu, char* args, ngx_pool_t* pool) {
    Memory result;
    impgpu_image* image = gpu->Handle;      /* frame 0 of an album, like bridge.c:669 */
    if (BrokerMode() && !gpu->Handle) {
        impgpu_client_answer a;
        result.Buffer = NULL;
        result.Length = 0;
        result.Error = gpu->Source ? BrokerRun(gpu, gpu->Source, IMPB_OUT_ASCII, 0, args, &a) : IMP_ERROR_INVALID_ARGS;
        if (!result.Error) {
            result.Buffer = ngx_palloc(pool, a.bytes ? a.bytes : 1);
            if (!result.Buffer) {
                result.Error = IMP_ERROR_MALLOC_FAILED;
            } else {
                memcpy(result.Buffer, a.data, a.bytes);     /* (the slot is the next request's) */
                result.Length = (long)a.bytes;
            }
        }
        return result;
    }
    long buflen = (long)(impgpu_image_width(image) + 1) * impgpu_image_height(image) - 1;
    result.Buffer = ngx_palloc(pool, buflen > 0 ? buflen : 1);
    result.Length = 0;
    result.Error  = result.Buffer ? impgpu_ascii(image, args, result.Buffer, buflen, &result.Length) : IMP_ERROR_MALLOC_FAILED;
    return result;
}

int ImpGpuDownload(ImpGpuAlbum* gpu, Album* album, ngx_pool_t* pool) {
    int fid, rc = IMP_OK;
    if (BrokerMode() && !gpu->Handle) {
        /* the answer's pixels come back in the worker's slot: into a fresh IplImage with the header rules of every
         * cvCreateImage in bridge.c, for the host encoder that asked */
        impgpu_client_answer a;
        IplImage* fresh;
        int y;
        rc = BrokerRun(gpu, album, IMPB_OUT_FRAME, 0, NULL, &a);
        if (rc) {
            return rc;
        }
        fresh = cvCreateImage(cvSize(a.width, a.height), IPL_DEPTH_8U, a.channels);
        if (!fresh || !fresh->imageData) {
            return IMP_ERROR_MALLOC_FAILED;
        }
        for (y = 0; y < a.height; y++) {
            memcpy(fresh->imageData + (size_t)y * fresh->widthStep, a.data + (size_t)y * a.row_step, (size_t)a.width * a.channels);
        }
        {
            IplImage* old = album->Frames[0].Image;
            cvReleaseImage(&old);
            album->Frames[0].Image = fresh;
        }
        return IMP_OK;
    }
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
    if (BrokerMode() && !gpu->Handle) {
        impgpu_client_answer a;
        int rcb = gpu->Source ? BrokerRun(gpu, gpu->Source, IMPB_OUT_JPEG, quality, NULL, &a) : IMP_ERROR_INVALID_ARGS;
        u_char* copy;
        if (rcb) {
            return rcb;
        }
        copy = ngx_palloc(pool, a.bytes ? a.bytes : 1);     /* (the reference copies the encoder's output too, bridge.c:705-706) */
        if (!copy) {
            return IMP_ERROR_MALLOC_FAILED;
        }
        memcpy(copy, a.data, a.bytes);
        *bytes = copy;
        *length = a.bytes;
        return IMP_OK;
    }
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
    if (BrokerMode() && OwnEnv() != IMP_OK) {       /* FreeImage's frames need a device in this worker (imp_gpu_bridge.h) */
        return IMP_ERROR_DEVICE;
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
    int rc = (BrokerMode() && OwnEnv() != IMP_OK) ? IMP_ERROR_DEVICE : impgpu_image_upload_fi32(bits, w, h, pitch, &frame);
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
