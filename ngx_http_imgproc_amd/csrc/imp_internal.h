// imp_internal.h -- shared declarations of libimpgpu.so (not installed; the public ABI is include/impgpu.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/impgpu.h"

namespace imp {

// ---------------------------------------------------------------- images
}  // namespace imp

struct impgpu_image {
    uint8_t* d = nullptr;   // device pointer
    int w = 0, h = 0, c = 0, step = 0;
    size_t cap = 0;         // bytes owned (0 for borrowed views)
    bool owned = false;
    // an album (the frames of one animation, bridge.c:554-572): `frames` frames of this geometry in ONE block, frame i at
    // d + i * fstride, so that every operator of the request is one launch for the whole album.  1 / 0 for a single image.
    int frames = 1;
    size_t fstride = 0;
};

namespace imp {

// A read-only window into device memory: how Crop is folded into the next operator.
struct View {
    const uint8_t* d;
    int w, h, c, step;
};
inline View view_of(const impgpu_image* im) { return View{im->d, im->w, im->h, im->c, im->step}; }
inline View view_sub(const View& v, int x, int y, int w, int h) {
    return View{v.d + (size_t)y * v.step + (size_t)x * v.c, w, h, v.c, v.step};
}
inline int aligned_step(int w, int c) { return (w * c + 3) & ~3; }
// what the kernels' 32-bit pixel / byte indexing can address (see image_new)
inline bool frame_fits(int w, int h, int c) {
    const long long step = ((long long)w * c + 3) & ~3LL;
    return (long long)w * h <= (1LL << 30) && step * h <= 0xffffffffLL && step <= 0x7fffffffLL;
}
// A caller-described frame (batch entry points, impgpu_image_wrap): positive size, a pitch that holds a row, and everything
// the kernels compute in 32 bits -- pixel count, row offsets, byte offsets inside a frame -- in range.  64-bit arithmetic:
// `step < w * c` in int wraps for the sizes this is meant to refuse.
inline bool view_fits(long long w, long long h, int c, long long step) {
    return w > 0 && h > 0 && w <= 0x7fffffffLL && h <= 0x7fffffffLL && step >= w * c && step <= 0x7fffffffLL &&
           w * h <= (1LL << 30) && step * h <= 0xffffffffLL;
}

// ---------------------------------------------------------------- A/B switches
// The environment switches of the measurement sessions (tools/switch_matrix.sh; DESIGN.md lists them with what each one
// measured) select kernel variants that LOST their comparison or tuning values other than the chosen ones.  They are compiled
// in only with -DIMPGPU_AB_SWITCHES (IMPGPU_EXTRA_FLAGS=-DIMPGPU_AB_SWITCHES python ngx_http_imgproc_amd/build.py): the
// shipped library reads none of them, and the variants only they reach are not in its binary.
#ifdef IMPGPU_AB_SWITCHES
inline const char* ab_env(const char* name) { return std::getenv(name); }
#else
constexpr const char* ab_env(const char*) { return nullptr; }
#endif
inline int ab_env_int(const char* name, int otherwise) {
    const char* s = ab_env(name);
    return s ? std::atoi(s) : otherwise;
}

// ---------------------------------------------------------------- runtime (imp_runtime.hip)
void set_error(const char* what, hipError_t e);
void set_error_text(const char* what);             // impgpu_last_error() text for failures that are not HIP errors
bool env_ready();
hipStream_t env_stream();
int  dev_alloc(size_t bytes, void** out);          // stream-ordered pool; IMP_* code
void dev_free(void* p);
int  image_new(int w, int h, int c, impgpu_image** out);
int  image_new_album(int w, int h, int c, int frames, impgpu_image** out);
void image_delete(impgpu_image* im);
// Copy a small host blob (tables, taps) into pool memory through the pinned ring, ordered on `s`.
int  upload_small(const void* host, size_t bytes, void** dev, hipStream_t s);
int  upload_to(void* dev, const void* host, size_t bytes, hipStream_t s);   // the same copy into memory the caller holds
// Larger host-built blobs: fill the pinned buffer stage_begin returns, then stage_upload copies it to `dev` on the lane stream.
int  stage_begin(size_t bytes, void** host, void** token);
int  stage_upload(void* token, void* dev, size_t bytes);
void stage_hold(void* token, bool held);                   // a download enqueued into the buffer is read by a later call: keep it out of circulation until then
size_t stage_capacity(void* token);                       // bytes the buffer behind `token` really holds (>= what stage_begin was asked for)
int  stage_upload_part(void* token, size_t offset, void* dev, size_t bytes, bool last);   // pieces of the same buffer; `last` fences it
// Pool memory and caller-supplied ("foreign") streams -- the batch entry points.  The pool recycles blocks in the order
// of the lane's own stream; these keep that sound without a host wait:
void dev_free_on(void* p, hipStream_t s);                  // free once the work enqueued on `s` so far is done
int  dev_alloc_on(size_t bytes, void** out, hipStream_t s); // dev_alloc + make `s` wait for the lane stream's tail
int  stream_join(hipStream_t s);                           // `s` waits for everything the lane stream holds now
int  stream_join_back(hipStream_t s);                      // the lane stream waits for everything `s` holds now
bool lane_stream_idle();
uint32_t* lane_mailbox();                                  // pinned words of the calling thread's lane: [0, 1024) for any caller, then 1024 per JPEG group in flight
int  lane_mark(void** mark);                               // a point of the lane's stream ...
int  lane_mark_on(hipStream_t s, void** mark);             // ... or of another stream of the thread (nullptr: the lane's)
hipStream_t lane_side_stream();                            // the lane's second stream (made on first use; nullptr if it cannot be)
int  lane_wait_mark(void* mark);                           // ... to sleep until (nullptr: the whole stream)
int  lane_wait();                                          // wait for the lane's stream (sleeping, unless IMPGPU_SYNC=spin)
bool on_lane_stream(hipStream_t s);
// Per-lane caches (resize tables): owned by the lane, dropped with it (impgpu_env_destroy), no lock.
struct LaneCache { virtual ~LaneCache() {} };
constexpr int LANE_CACHE_SLOTS = 2;
LaneCache** lane_cache_slot(int which);
// rocTX ranges named after the reference's step codes (required.h:46-54) around each stage of a request, so a
// `rocprofv3 --marker-trace` timeline reads like JobResult.Step.  Resolved with dlopen at impgpu_env_start: no link-time
// dependency, nothing emitted unless a profiler's rocTX library is already in the process or IMPGPU_ROCTX=1.
void trace_push(const char* name);
void trace_pop();
struct TraceRange {
    explicit TraceRange(const char* name) { trace_push(name); }
    ~TraceRange() { trace_pop(); }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};
// Fault injection: impgpu_fault_arm(step, n) makes the n-th entry into that IMP_STEP_* behave as if its first
// HIP call had failed: IMP_ERROR_DEVICE, impgpu_last_error() says so, the failing step is reported like any other -- the
// path a lost device takes, testable without losing one.  Never armed from the environment.
bool fault_hit(int step);
#define IMP_FAULT_POINT(step)                                     \
    do {                                                          \
        if (imp::fault_hit(step)) return IMP_ERROR_DEVICE;        \
    } while (0)
#define IMP_HIP(call)                                             \
    do {                                                          \
        hipError_t _e = (call);                                   \
        if (_e != hipSuccess) { imp::set_error(#call, _e); return IMP_ERROR_DEVICE; } \
    } while (0)

// ---------------------------------------------------------------- AlphaBlendOver on one BGRA pixel pair (filters.c:633-659)
// Shared by k_blend_over (imp_pixel.hip) and the fused resize + rotate + watermark kernel (imp_resize.hip): the float
// sequence of the reference, one operation per rounding (`alpha` = 1 - opacity, filters.c:620).
#ifdef __HIPCC__          // device code: the host-only builds of the grammar files (sanitizer tests) skip it
__device__ __forceinline__ uint32_t blend_over_bgra(uint32_t d, uint32_t s, float alpha) {
    const int dB = d & 0xff, dG = (d >> 8) & 0xff, dR = (d >> 16) & 0xff;
    const float dA = (float)((double)(d >> 24) / 255.0);
    const int sB = s & 0xff, sG = (s >> 8) & 0xff, sR = (s >> 16) & 0xff;
    float sA = (float)((double)(s >> 24) / 255.0);
    sA = (float)fmax((double)__fsub_rn(sA, alpha), 0.0);
    const float inv = __fsub_rn(1.f, sA);
    const float tA = __fadd_rn(sA, __fmul_rn(dA, inv));
    int tB = 0, tG = 0, tR = 0;
    if (tA != 0.f) {
        tB = (int)__fdiv_rn(__fadd_rn(__fmul_rn((float)sB, sA), __fmul_rn(__fmul_rn((float)dB, dA), inv)), tA);
        tG = (int)__fdiv_rn(__fadd_rn(__fmul_rn((float)sG, sA), __fmul_rn(__fmul_rn((float)dG, dA), inv)), tA);
        tR = (int)__fdiv_rn(__fadd_rn(__fmul_rn((float)sR, sA), __fmul_rn(__fmul_rn((float)dR, dA), inv)), tA);
    }
    const float a255 = __fmul_rn(tA, 255.f);
    const int tAi = (a255 > -2147483904.f && a255 < 2147483648.f) ? (int)a255 : (int)0x80000000;
    return (uint32_t)(tB & 0xff) | ((uint32_t)(tG & 0xff) << 8) | ((uint32_t)(tR & 0xff) << 16) | ((uint32_t)(tAi & 0xff) << 24);
}
// The same on a 3-channel destination pixel (packed B | G << 8 | R << 16): its alpha is 1 (filters.c:636) and none is written.
__device__ __forceinline__ uint32_t blend_over_bgr(uint32_t d, uint32_t s, float alpha) {
    const int dB = d & 0xff, dG = (d >> 8) & 0xff, dR = (d >> 16) & 0xff;
    const float dA = 1.f;
    const int sB = s & 0xff, sG = (s >> 8) & 0xff, sR = (s >> 16) & 0xff;
    float sA = (float)((double)(s >> 24) / 255.0);
    sA = (float)fmax((double)__fsub_rn(sA, alpha), 0.0);
    const float inv = __fsub_rn(1.f, sA);
    const float tA = __fadd_rn(sA, __fmul_rn(dA, inv));
    int tB = 0, tG = 0, tR = 0;
    if (tA != 0.f) {
        tB = (int)__fdiv_rn(__fadd_rn(__fmul_rn((float)sB, sA), __fmul_rn(__fmul_rn((float)dB, dA), inv)), tA);
        tG = (int)__fdiv_rn(__fadd_rn(__fmul_rn((float)sG, sA), __fmul_rn(__fmul_rn((float)dG, dA), inv)), tA);
        tR = (int)__fdiv_rn(__fadd_rn(__fmul_rn((float)sR, sA), __fmul_rn(__fmul_rn((float)dR, dA), inv)), tA);
    }
    return (uint32_t)(tB & 0xff) | ((uint32_t)(tG & 0xff) << 8) | ((uint32_t)(tR & 0xff) << 16);
}
#endif

// ---------------------------------------------------------------- host grammar (imp_args.cpp)
int crop_geometry(int col, int row, const char* args, const char* gravity, int* x, int* y, int* w, int* h);
int resize_geometry(int col, int row, const char* args, unsigned max_w, unsigned max_h, int simple,
                    int* w, int* h, int* interp);

// One stage of the fused pointwise program (imp_pixel.hip runs them per pixel, in order,
// keeping the reference's per-stage 8-bit truncation).
enum StageKind : int {
    ST_LUT4 = 0,      // per-channel 256-entry tables, channels 0..3 (identity rows where unused)
    ST_RGB2HSV,       // helpers.c:70-107
    ST_HSV2RGB,       // helpers.c:109-176
    ST_GRADMAP,       // filters.c:264-276: 768-byte RGB table indexed by ((R+G+B)/3)*3
    ST_VIGNETTE,      // filters.c:312-317 on HSV value; params cx, cy, maxrad, power
    ST_RAINBOW,       // filters.c:368-398; param sat
    ST_SCANLINE,      // filters.c:434-451; params freq, width, s_byte, v_byte
};
struct Stage {
    int kind;
    int lut_off;      // byte offset into the program's table blob (LUT4: 1024 B, GRADMAP: 768 B)
    int i0, i1, i2, i3;
    float f0, f1;
};
struct PixelProgram {
    std::vector<Stage> stages;
    std::vector<uint8_t> tables;
    void clear() { stages.clear(); tables.clear(); }
    bool empty() const { return stages.empty(); }
};

// What one filter-* request turns into.
enum FilterClass : int { FC_POINTWISE = 0, FC_FLIP, FC_ROTATE, FC_BLUR, FC_NOOP };
struct FilterPlan {
    int cls = FC_NOOP;
    int flip_mode = 0;       // cvFlip mode: 0 vertical, 1 horizontal, -1 both
    int rotate = 0;          // 90 / 180 / 270
    float sigma = 0;         // blur
    // pointwise stages are appended to the caller's PixelProgram
};
// Parses "name=args" (filters.c:43-70 + the callback's own checks). For pointwise filters
// appends stages for an image of `channels` channels and w x h pixels to `prog`.
int filter_plan(const char* request, int allow_experiments, int channels, int w, int h,
                FilterPlan* plan, PixelProgram* prog);
int check_destructive(const char* request);

// ---------------------------------------------------------------- coefficient tables (imp_tables.cpp)
struct TapAxis {            // LINEAR / CUBIC / LANCZOS4: one axis of cv::resize's generic branch
    int ksize;
    std::vector<int> ofs;       // dsize entries: source index of the tap-window centre
    std::vector<short> coef;    // dsize * ksize fixed-point (11-bit) weights
};
void build_tap_axis(int ssize, int dsize, double scale, int interp, bool is_x, TapAxis* out);
struct AreaAxis {           // general INTER_AREA: per destination index a run of source indices
    std::vector<int> start;     // first source index of the run
    std::vector<int> count;     // run length
    std::vector<int> aoff;      // offset of the run's first weight in alpha
    std::vector<float> alpha;
    int max_count = 0;
};
void build_area_axis(int ssize, int dsize, double scale, AreaAxis* out);
int  area_max_count(int ssize, int dsize, double scale);          // AreaAxis::max_count alone
int  gaussian_ksize(double sigma);
void gaussian_kernel_fixed(int n, double sigma, std::vector<int>* ik);

// ---------------------------------------------------------------- launchers
struct Frames {             // `count` frames of one geometry
    const uint8_t* src; long long src_stride; View v;   // v.d == src (frame 0)
    uint8_t* dst; long long dst_stride; int dw, dh, dstep;
    int count;
};
// imp_resize.hip
int launch_cv_resize(const Frames& f, int interp, hipStream_t s);
// `count` frames of different geometry, each Resize()d by the reference's rule (BASELINE configs[4], device-resident)
struct MixFrame { const uint8_t* src; int sw, sh, sstep; uint8_t* dst; int dw, dh, dstep; };
int launch_resize_mixed(const MixFrame* frames, int count, int channels, int simple, hipStream_t s);
// exact-2x AREA + rotate 90/270 of BGRA in one pass; IMP_ERROR_UNSUPPORTED when the geometry does not qualify.
// With a 4-channel overlay the Watermark step (bridge.c:629-640) rides on the store phase of the same kernel.
struct OverlayArgs { const uint8_t* ov; int ostep, rx, ry, maxcol, maxrow; float alpha; };
int launch_area2x2_rotate(const Frames& f, int amount, const OverlayArgs* overlay, hipStream_t s);
// general INTER_AREA of BGRA frames + rotate 0/90/180/270 + watermark in one pass (f.dw x f.dh = the resized geometry,
// f.dst = the final frames); IMP_ERROR_UNSUPPORTED when the geometry takes another resize kernel.
int launch_area_rotate(const Frames& f, int amount, const OverlayArgs* overlay, hipStream_t s);
// imp_geom.hip
int launch_copy(const Frames& f, hipStream_t s);                       // crop copy / clone (dw,dh = v.w,v.h)
int launch_flip(const Frames& f, int mode, hipStream_t s);             // cvFlip
int launch_rotate(const Frames& f, int amount, hipStream_t s);         // 90 / 270 (dw,dh = v.h,v.w), 180
int launch_gray2bgr(const Frames& f, hipStream_t s);
int launch_pack_fi(const View& v, int bpp, uint8_t* dst, int dpitch, hipStream_t s);   // IplToFI32/24: flip + repack
struct GifPageDev { long long idx_off, pal_off; int w, h, pitch, left, top, dispose, key, pad; };   // offsets into one device blob
int launch_gif_compose(const uint8_t* blob, const GifPageDev* pages, uint8_t* const* outs, int npages, int cw, int ch,
                       int ostep, int destructive, int only, hipStream_t s);                 // LoadGIF's compositing loop
// imp_pixel.hip
int launch_pixel_program(uint8_t* d, long long stride, int w, int h, int c, int step, int count,
                         const PixelProgram& prog, hipStream_t s);
int launch_blend_over(uint8_t* d, long long stride, int w, int h, int c, int step, int count,
                      const impgpu_image* overlay, int rx, int ry, int maxcol, int maxrow,
                      float alpha, hipStream_t s);
int launch_blend_paper(uint8_t* d, long long stride, int w, int h, int step, int count, hipStream_t s);
int launch_brightness(const View& v, float* host_result, hipStream_t s);
int launch_ascii(uint8_t* d, int w, int h, int c, int step, const uint8_t* table, int tablelen,
                 float factor, uint8_t* dev_out, hipStream_t s);
// imp_blur.hip
int launch_gaussian(uint8_t* d, long long stride, int w, int h, int c, int step, int count,
                    double sigma, hipStream_t s);

// fused single-pass BGRA Gaussian, out of place; IMP_ERROR_UNSUPPORTED when it does not apply
int launch_gaussian_fused(const Frames& f, double sigma, hipStream_t s);

// Watermark placement (bridge.c:254-274 + cvSetImageROI clipping). Returns IMP_* code.
int watermark_rect(int basew, int baseh, int overw, int overh, const impgpu_config* cfg,
                   int* rx, int* ry, int* maxcol, int* maxrow);


// The dynamic-LDS limit of a kernel is a per-function, process-wide attribute: raised ONCE per kernel instantiation (a
// thread-safe static) to everything the device allows beside the kernel's static LDS -- not before every launch with that
// launch's own size, where two request threads with different sizes could interleave set(small) between another thread's
// set(large) and its launch, and every request paid a runtime call for a value that never changes (round 4's review).
#if defined(__HIPCC__)
template <auto Kernel>
inline hipError_t lds_limit_once() {
    static const hipError_t err = [] {
        int dev = 0, cap = 0;
        hipFuncAttributes fa;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
        if (e == hipSuccess) e = hipFuncGetAttributes(&fa, (const void*)Kernel);
        if (e != hipSuccess) return e;
        const int room = cap - (int)fa.sharedSizeBytes;
        return hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, room > 0 ? room : 0);
    }();
    return err;
}
#endif
}  // namespace imp
