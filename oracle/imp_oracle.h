/*
 * imp_oracle.h -- CPU restatement of the IMP pixel-transform path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library (libimpgpu.so, the
 * ngx_http_imgproc_amd package) may include, link or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker, never as the thing measured for the headline or shipped.
 *
 * PARITY UNPINNED.  The reference (tommiv/ngx_http_imgproc) has no tests, no
 * golden vectors and no fixtures, and cannot be compiled in this image: every
 * source file includes <ngx_*.h>, <opencv/cv.h> and <FreeImage.h>
 * (required.h:12-19), none of which exist here.  This file therefore restates
 * the reference's own C (bridge.c / filters.c / helpers.c, cited per function)
 * and, for the two calls whose arithmetic lives in the absent third-party
 * dependency OpenCV 2.4.9 (cvResize bridge.c:191, cvSmooth filters.c:204), the
 * published algorithm of OpenCV 2.4.9 modules/imgproc/src/{imgwarp,smooth,
 * filter}.cpp as built for x86-64 (SSE2 paths on).  The only reference-derived
 * known answers are the seven Crop geometry results recorded in SURVEY.md
 * section 8(c); tests/test_oracle_kat.py checks them.
 */
#ifndef IMP_ORACLE_H
#define IMP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* required.h:28-41 */
#define ORC_OK                      0
#define ORC_ERROR_UNSUPPORTED       1
#define ORC_ERROR_MALLOC_FAILED     2
#define ORC_ERROR_INVALID_ARGS      50
#define ORC_ERROR_UPSCALE           51
#define ORC_ERROR_NO_SUCH_FILTER    52
#define ORC_ERROR_NO_SUCH_WATERMARK 53
#define ORC_ERROR_TOO_BIG_TARGET    54
#define ORC_ERROR_TOO_MUCH_FILTERS  55
#define ORC_ERROR_FEATURE_DISABLED  56

/* required.h:46-54 */
#define ORC_STEP_CROP      3
#define ORC_STEP_RESIZE    4
#define ORC_STEP_FILTERING 5
#define ORC_STEP_WATERMARK 6
#define ORC_STEP_INFO      7

/* OpenCV 2.4 CV_INTER_* */
#define ORC_INTER_NN       0
#define ORC_INTER_LINEAR   1
#define ORC_INTER_CUBIC    2
#define ORC_INTER_AREA     3
#define ORC_INTER_LANCZOS4 4

/* 8-bit interleaved image, top-left origin: the subset of IplImage the path uses
 * (required.h:129-134; helpers.h:1-4). step = cvCreateImage's 4-byte aligned row. */
typedef struct {
    unsigned char* data;
    int width, height, channels, step;
} orc_image;

orc_image* orc_image_create(int width, int height, int channels);
orc_image* orc_image_from(const unsigned char* data, int width, int height, int channels, int step);
orc_image* orc_image_clone(const orc_image* src);
void       orc_image_free(orc_image* img);
unsigned char* orc_image_data(orc_image* img);
int orc_image_width(const orc_image* img);
int orc_image_height(const orc_image* img);
int orc_image_channels(const orc_image* img);
int orc_image_step(const orc_image* img);

/* bridge.c:18-141 */
int orc_crop_geometry(int col, int row, const char* args, const char* gravity,
                      int* x, int* y, int* w, int* h);
int orc_crop(orc_image** pointer, const char* args, const char* gravity);

/* bridge.c:143-197 */
int orc_resize_geometry(int col, int row, const char* args, unsigned max_w, unsigned max_h,
                        int simple, int* w, int* h, int* interpolation);
int orc_resize(orc_image** pointer, const char* args, unsigned max_w, unsigned max_h, int simple);

/* OpenCV 2.4.9 cvResize for 8-bit 1/3/4-channel images. simd=1 follows the x86-64
 * SSE2 build (float vertical pass for cubic), simd=0 the scalar templates. */
int  orc_cv_resize(const orc_image* src, orc_image* dst, int interpolation);
void orc_set_cv_simd(int simd);

/* OpenCV 2.4.9 cvSmooth(CV_GAUSSIAN, 0, 0, sigma, 0) in place. */
int orc_cv_smooth_gaussian(orc_image* img, double sigma);
int orc_gaussian_ksize(double sigma);

/* filters.c:43-70 and the callbacks it dispatches */
int orc_filter(orc_image** pointer, const char* request, int allow_experiments);

/* helpers.c:70-176 */
void orc_rgb2hsv(orc_image* img);
void orc_hsv2rgb(orc_image* img);

/* bridge.c:239-281 + filters.c:619-662. overlay is the RecoverInfo image. */
int orc_watermark(orc_image* img, const orc_image* overlay, char gravity_x, char gravity_y,
                  int offset_x, int offset_y, int opacity);
/* filters.c:666-687 */
void orc_blend_with_paper(orc_image* img);
/* filters.c:707-729 */
float orc_calc_perceived_brightness(const orc_image* img);
/* filters.c:486-522; out must hold (w+1)*h-1 bytes; returns length. Mutates img (RGB2HSV). */
long orc_ascii(orc_image* img, const char* args, unsigned char* out);
/* bridge.c:613-618 */
int orc_gray2bgr(orc_image** pointer);

/* bridge.c:574-656: crop -> resize -> [gray->BGR] -> filters -> watermark -> flatten. */
typedef struct {
    const char* crop;        /* NULL = absent */
    const char* gravity;
    const char* resize;
    int simple;              /* bridge.c:594 */
    const char* const* filters;
    int filter_count;
    int allow_experiments;
    unsigned max_w, max_h;
    const orc_image* overlay; /* NULL = no watermark configured */
    char gravity_x, gravity_y;
    int offset_x, offset_y, opacity;
    int flatten;             /* encoder lacks alpha (bridge.c:642-648) */
} orc_chain;
int orc_run_chain(orc_image** pointer, const orc_chain* chain, int* step);

/* advancedio.c:65-101 IplToFI32/24 and advancedio.c:310-318 LoadSingle's copy */
int orc_ipl_to_fi(const orc_image* img, int bpp, unsigned char* out, int pitch);
orc_image* orc_fi32_to_ipl(const unsigned char* bits, int width, int height, int pitch);

/* advancedio.c:103-262 LoadGIF, the compositing loop of :204-247.  A page is what FreeImage hands that loop:
 * 8-bit palette indices in FreeImage scanline order (bottom-up, `pitch` bytes per row), the FrameLeft / FrameTop /
 * DisposalMethod tags, the transparent index (-1 = none) and the 256-entry RGBQUAD palette (B,G,R,reserved).
 * frames[i] receives page i composed on the first page's canvas (4 channels); with page >= 0 the walk stops at
 * that page and only frames[0] (= that page) is returned, like advancedio.c:254-262. */
typedef struct {
    const unsigned char* indices;
    int width, height, pitch;
    int left, top;
    int dispose;             /* 0 unspecified, 1 leave, 2 background, 3 previous (FreeImage GIF_DISPOSAL_*) */
    int transparency_key;
    const unsigned char* palette;
} orc_gif_page;
int orc_gif_compose(const orc_gif_page* pages, int count, int destructive, int page, orc_image** frames);

/* bridge.c:304-372 request parsing + bridge.c:413-466 encoder choice. Strings point into `buffer`. */
#define ORC_MAX_FILTERS 64
typedef struct {
    char* buffer;
    char *crop, *gravity, *resize, *quality, *format;
    int page;
    char* filters[ORC_MAX_FILTERS];
    int filter_count;
    int mime;            /* required.h:57-62: -1 jpg, -2 png, -3 json, -4 FreeImage, -5 text */
    int simple;          /* encoder is GIF */
    int need_flatten;    /* encoder cannot store alpha */
} orc_request;
int  orc_parse_request(const char* uri, const char* exten, int max_filters, orc_request** out);
void orc_request_free(orc_request* r);

#ifdef __cplusplus
}
#endif
#endif
