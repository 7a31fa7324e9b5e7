// imp_runtime.hip -- per-worker environment: device, streams, HBM buffer pools, pinned staging,
// frame upload / download.  Stands where the reference's empty OnEnvStart / OnEnvDestroy
// (bridge.c:10-16) and its cvCreateImage / cvReleaseImage calls are.
//
// One env per process (one nginx worker = one process, module.c:100-107).  Inside it every
// calling thread gets its own LANE: a HIP stream, a size-bucketed pool of device buffers, a
// pinned staging buffer for frame upload / download and a pinned ring for small tables.  All
// operators of a request are enqueued on the lane's stream, so a buffer released by one operator
// can be handed to the next without a device sync (stream order is the only ordering needed), and
// requests driven from different threads overlap their H2D copies, kernels and D2H copies with no
// shared lock on the hot path.  A frame belongs to the lane (thread) that created it.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include "imp_internal.h"

namespace imp {

struct Lane {
    hipStream_t stream = nullptr;
    std::mutex mu;                              // guards the pool: frees may come from other threads
    std::multimap<size_t, void*> free_list;     // bucket size -> buffer
    std::map<void*, size_t> live;               // buffer -> bucket size
    uint8_t* stage = nullptr;                   // pinned staging for pageable upload / download
    size_t stage_cap = 0;
    hipEvent_t stage_done = nullptr;
    bool stage_busy = false;
    uint8_t* ring = nullptr;                    // pinned ring for per-launch tables (LUTs, Gaussian taps)
    size_t ring_cap = 0, ring_pos = 0;
};

struct Env {
    int device = -1;
    unsigned long long generation = 0;
    std::mutex mu;
    std::vector<Lane*> lanes;
};

static Env* g_env = nullptr;
static unsigned long long g_generation = 0;
static thread_local Lane* t_lane = nullptr;
static thread_local unsigned long long t_lane_gen = 0;
static thread_local std::string t_error;

void set_error(const char* what, hipError_t e) {
    t_error = std::string(what) + ": " + hipGetErrorString(e);
}
bool env_ready() { return g_env != nullptr; }

static int no_env() {
    t_error = "impgpu_env_start has not been called";
    return IMP_ERROR_DEVICE;
}

// The calling thread's lane, created on first use (HIP's current device is per thread too).
static Lane* lane() {
    Env* E = g_env;
    if (!E) return nullptr;
    if (t_lane && t_lane_gen == E->generation) return t_lane;
    if (hipSetDevice(E->device) != hipSuccess) return nullptr;
    Lane* L = new Lane();
    if (hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&L->stage_done, hipEventDisableTiming) != hipSuccess) {
        delete L;
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(E->mu);
        E->lanes.push_back(L);
    }
    t_lane = L;
    t_lane_gen = E->generation;
    return L;
}

hipStream_t env_stream() {
    Lane* L = lane();
    return L ? L->stream : nullptr;
}

static size_t bucket_of(size_t bytes) {
    size_t b = 4096;
    while (b < bytes) {
        // 1, 1.5, 2, 3, 4, 6 ... x 4 KiB: at most 33 % slack
        size_t half = b + b / 2;
        if (half >= bytes) return half;
        b <<= 1;
    }
    return b;
}

int dev_alloc(size_t bytes, void** out) {
    Lane* L = lane();
    if (!L) return no_env();
    const size_t b = bucket_of(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lk(L->mu);
        auto it = L->free_list.find(b);
        if (it != L->free_list.end()) {
            *out = it->second;
            L->free_list.erase(it);
            L->live[*out] = b;
            return IMP_OK;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, b);
    if (e != hipSuccess) {
        // drop this lane's cache and retry once
        {
            std::lock_guard<std::mutex> lk(L->mu);
            (void)hipStreamSynchronize(L->stream);
            for (auto& kv : L->free_list) (void)hipFree(kv.second);
            L->free_list.clear();
        }
        e = hipMalloc(&p, b);
        if (e != hipSuccess) { set_error("hipMalloc", e); return IMP_ERROR_MALLOC_FAILED; }
    }
    std::lock_guard<std::mutex> lk(L->mu);
    L->live[p] = b;
    *out = p;
    return IMP_OK;
}

static bool lane_take_back(Lane* L, void* p) {
    std::lock_guard<std::mutex> lk(L->mu);
    auto it = L->live.find(p);
    if (it == L->live.end()) return false;
    L->free_list.emplace(it->second, p);
    L->live.erase(it);
    return true;
}

void dev_free(void* p) {
    Env* E = g_env;
    if (!p || !E) return;
    Lane* mine = (t_lane && t_lane_gen == E->generation) ? t_lane : nullptr;
    if (mine && lane_take_back(mine, p)) return;
    // released from another thread (e.g. a garbage collector): hand it back to the lane that owns it
    std::vector<Lane*> lanes;
    {
        std::lock_guard<std::mutex> lk(E->mu);
        lanes = E->lanes;
    }
    for (Lane* L : lanes)
        if (L != mine && lane_take_back(L, p)) return;
}

int image_new(int w, int h, int c, impgpu_image** out) {
    if (w <= 0 || h <= 0 || (c != 1 && c != 3 && c != 4)) return IMP_ERROR_INVALID_ARGS;
    // the kernels index pixels and row bytes in 32 bits: a frame is at most 2^30 pixels and 4 GiB - 1 of rows.  Larger
    // requests (resize=2000000000,1,up with the size watchdog off) end like a failed cvCreateImage, not in a wrapped pitch.
    if (!frame_fits(w, h, c)) { t_error = "frame too large"; return IMP_ERROR_MALLOC_FAILED; }
    impgpu_image* im = new impgpu_image();
    im->w = w; im->h = h; im->c = c;
    im->step = aligned_step(w, c);
    im->cap = (size_t)im->step * h;
    void* p = nullptr;
    int rc = dev_alloc(im->cap + 16, &p);
    if (rc) { delete im; return rc; }
    im->d = (uint8_t*)p;
    im->owned = true;
    *out = im;
    return IMP_OK;
}

void image_delete(impgpu_image* im) {
    if (!im) return;
    if (im->owned) dev_free(im->d);
    delete im;
}

int upload_small(const void* host, size_t bytes, void** dev, hipStream_t s) {
    Lane* L = lane();
    if (!L) return no_env();
    void* p = nullptr;
    int rc = dev_alloc(bytes, &p);
    if (rc) return rc;
    const size_t need = (bytes + 63) & ~size_t(63);
    if (need > L->ring_cap) {       // first use, or a blob larger than the ring
        (void)hipDeviceSynchronize();
        if (L->ring) (void)hipHostFree(L->ring);
        L->ring = nullptr;
        L->ring_cap = 0;
        const size_t cap = need * 2 > (size_t(1) << 20) ? need * 2 : (size_t(1) << 20);
        hipError_t e = hipHostMalloc((void**)&L->ring, cap, hipHostMallocDefault);
        if (e != hipSuccess) { set_error("hipHostMalloc(ring)", e); dev_free(p); return IMP_ERROR_DEVICE; }
        L->ring_cap = cap;
        L->ring_pos = 0;
    }
    if (L->ring_pos + need > L->ring_cap) {   // wrap: everything that read the ring must be done
        (void)hipDeviceSynchronize();         // (device-wide: a batch call may have used a foreign stream)
        L->ring_pos = 0;
    }
    uint8_t* slot = L->ring + L->ring_pos;
    L->ring_pos += need;
    std::memcpy(slot, host, bytes);
    hipError_t e = hipMemcpyAsync(p, slot, bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) { set_error("hipMemcpyAsync(small)", e); dev_free(p); return IMP_ERROR_DEVICE; }
    *dev = p;
    return IMP_OK;
}

static int stage_reserve(Lane* L, size_t bytes) {
    if (L->stage_busy) {
        IMP_HIP(hipEventSynchronize(L->stage_done));
        L->stage_busy = false;
    }
    if (L->stage_cap >= bytes) return IMP_OK;
    if (L->stage) IMP_HIP(hipHostFree(L->stage));
    L->stage = nullptr;
    L->stage_cap = 0;
    const size_t cap = bucket_of(bytes);
    IMP_HIP(hipHostMalloc((void**)&L->stage, cap, hipHostMallocDefault));
    L->stage_cap = cap;
    return IMP_OK;
}

static void lane_destroy(Lane* L) {
    (void)hipStreamSynchronize(L->stream);
    for (auto& kv : L->free_list) (void)hipFree(kv.second);
    for (auto& kv : L->live) (void)hipFree(kv.first);
    if (L->stage) (void)hipHostFree(L->stage);
    if (L->ring) (void)hipHostFree(L->ring);
    (void)hipEventDestroy(L->stage_done);
    (void)hipStreamDestroy(L->stream);
    delete L;
}

}  // namespace imp

using namespace imp;

extern "C" {

int impgpu_env_start(int device) {
    if (g_env) return IMP_OK;
    if (device < 0) {
        const char* s = std::getenv("IMPGPU_DEVICE");
        if (!s) s = std::getenv("LOCAL_RANK");
        device = s ? std::atoi(s) : 0;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        if (e != hipSuccess) set_error("hipGetDeviceCount", e);
        else t_error = "no HIP device visible";
        return IMP_ERROR_DEVICE;
    }
    device %= n;    // round-robin of workers over the node's GPUs
    IMP_HIP(hipSetDevice(device));
    Env* E = new Env();
    E->device = device;
    E->generation = ++g_generation;
    g_env = E;
    if (!lane()) {  // the calling thread's lane: fails loudly here rather than at the first operator
        g_env = nullptr;
        delete E;
        t_error = "could not create a HIP stream";
        return IMP_ERROR_DEVICE;
    }
    return IMP_OK;
}

void impgpu_env_destroy(void) {
    Env* E = g_env;
    if (!E) return;
    (void)hipDeviceSynchronize();
    for (Lane* L : E->lanes) lane_destroy(L);
    g_env = nullptr;        // other threads' t_lane pointers are invalidated by the generation counter
    t_lane = nullptr;
    delete E;
}

int impgpu_env_device(void) { return g_env ? g_env->device : -1; }
const char* impgpu_last_error(void) { return t_error.c_str(); }
void* impgpu_env_stream(void) { return (void*)env_stream(); }

int impgpu_sync(void) {
    Lane* L = lane();
    if (!L) return no_env();
    IMP_HIP(hipStreamSynchronize(L->stream));
    return IMP_OK;
}

int impgpu_image_create(int width, int height, int channels, impgpu_image** out) {
    if (!out) return IMP_ERROR_INVALID_ARGS;
    return image_new(width, height, channels, out);
}

int impgpu_image_upload(const unsigned char* data, int width, int height, int channels, int step,
                        impgpu_image** out) {
    if (!data || !out || (long long)step < (long long)width * channels) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    impgpu_image* im = nullptr;
    int rc = image_new(width, height, channels, &im);
    if (rc) return rc;
    const size_t bytes = (size_t)im->step * height;
    rc = stage_reserve(L, bytes);
    if (rc) { image_delete(im); return rc; }
    // repack into the device row pitch (cvCreateImage alignment) inside pinned memory
    const size_t rowbytes = (size_t)width * channels;
    if ((size_t)step == (size_t)im->step) {
        std::memcpy(L->stage, data, bytes - (im->step - rowbytes));
    } else {
        for (int y = 0; y < height; y++)
            std::memcpy(L->stage + (size_t)y * im->step, data + (size_t)y * step, rowbytes);
    }
    hipError_t e = hipMemcpyAsync(im->d, L->stage, bytes, hipMemcpyHostToDevice, L->stream);
    if (e == hipSuccess) e = hipEventRecord(L->stage_done, L->stream);
    if (e != hipSuccess) { set_error("hipMemcpyAsync(upload)", e); image_delete(im); return IMP_ERROR_DEVICE; }
    L->stage_busy = true;
    *out = im;
    return IMP_OK;
}

void* impgpu_host_alloc(size_t bytes) {
    if (!lane()) { no_env(); return nullptr; }
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { set_error("hipHostMalloc", e); return nullptr; }
    return p;
}

void impgpu_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int impgpu_image_upload_pinned(const unsigned char* data, int width, int height, int channels, int step,
                               impgpu_image** out) {
    if (!data || !out || (long long)step < (long long)width * channels) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    impgpu_image* im = nullptr;
    int rc = image_new(width, height, channels, &im);
    if (rc) return rc;
    // straight from the caller's pinned frame: no staging pass.  hipMemcpy2DAsync repacks the row pitch.
    hipError_t e = hipMemcpy2DAsync(im->d, (size_t)im->step, data, (size_t)step, (size_t)width * channels, (size_t)height,
                                    hipMemcpyHostToDevice, L->stream);
    if (e != hipSuccess) { set_error("hipMemcpy2DAsync(upload_pinned)", e); image_delete(im); return IMP_ERROR_DEVICE; }
    *out = im;
    return IMP_OK;
}

int impgpu_image_download_pinned(const impgpu_image* im, unsigned char* data, int step) {
    if (!im || !data || step < im->w * im->c) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    IMP_HIP(hipMemcpy2DAsync(data, (size_t)step, im->d, (size_t)im->step, (size_t)im->w * im->c, (size_t)im->h,
                             hipMemcpyDeviceToHost, L->stream));
    return IMP_OK;
}

int impgpu_image_upload_fi32(const unsigned char* bits, int width, int height, int pitch, impgpu_image** out) {
    if (!bits || !out || pitch < width * 4) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    impgpu_image* im = nullptr;
    int rc = image_new(width, height, 4, &im);
    if (rc) return rc;
    const size_t bytes = (size_t)im->step * height;
    rc = stage_reserve(L, bytes);
    if (rc) { image_delete(im); return rc; }
    // LoadSingle (advancedio.c:310-318): FreeImage rows are bottom-up; the flip rides on the staging copy
    for (int y = 0; y < height; y++)
        std::memcpy(L->stage + (size_t)(height - 1 - y) * im->step, bits + (size_t)y * pitch, (size_t)width * 4);
    hipError_t e = hipMemcpyAsync(im->d, L->stage, bytes, hipMemcpyHostToDevice, L->stream);
    if (e == hipSuccess) e = hipEventRecord(L->stage_done, L->stream);
    if (e != hipSuccess) { set_error("hipMemcpyAsync(upload_fi32)", e); image_delete(im); return IMP_ERROR_DEVICE; }
    L->stage_busy = true;
    *out = im;
    return IMP_OK;
}

int impgpu_image_download_fi(const impgpu_image* im, int bpp, unsigned char* bits, int pitch) {
    if (!im || !bits || (bpp != 24 && bpp != 32) || im->c < 3 || pitch < im->w * (bpp / 8)) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    const int dpitch = (im->w * (bpp / 8) + 3) & ~3;            // FreeImage's own pitch rule
    const size_t bytes = (size_t)dpitch * im->h;
    void* tmp = nullptr;
    int rc = dev_alloc(bytes, &tmp);
    if (rc) return rc;
    rc = launch_pack_fi(view_of(im), bpp, (uint8_t*)tmp, dpitch, L->stream);
    if (!rc) rc = stage_reserve(L, bytes);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(L->stage, tmp, bytes, hipMemcpyDeviceToHost, L->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(L->stream);
        if (e != hipSuccess) { set_error("download_fi", e); rc = IMP_ERROR_DEVICE; }
    }
    dev_free(tmp);
    if (rc) return rc;
    const size_t rowbytes = (size_t)im->w * (bpp / 8);
    for (int y = 0; y < im->h; y++) std::memcpy(bits + (size_t)y * pitch, L->stage + (size_t)y * dpitch, rowbytes);
    return IMP_OK;
}

int impgpu_image_wrap(void* device_ptr, int width, int height, int channels, int step, impgpu_image** out) {
    if (!device_ptr || !out || width <= 0 || height <= 0 || (channels != 1 && channels != 3 && channels != 4) ||
        (long long)step < (long long)width * channels || !frame_fits(width, height, channels) ||
        (long long)step * height > 0xffffffffLL)
        return IMP_ERROR_INVALID_ARGS;
    impgpu_image* im = new impgpu_image();
    im->d = (uint8_t*)device_ptr;
    im->w = width; im->h = height; im->c = channels; im->step = step;
    im->cap = 0;
    im->owned = false;
    *out = im;
    return IMP_OK;
}

int impgpu_image_download(const impgpu_image* im, unsigned char* data, int step) {
    if (!im || !data || step < im->w * im->c) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    const size_t bytes = (size_t)im->step * im->h;
    int rc = stage_reserve(L, bytes);
    if (rc) return rc;
    IMP_HIP(hipMemcpyAsync(L->stage, im->d, bytes, hipMemcpyDeviceToHost, L->stream));
    IMP_HIP(hipStreamSynchronize(L->stream));
    const size_t rowbytes = (size_t)im->w * im->c;
    for (int y = 0; y < im->h; y++)
        std::memcpy(data + (size_t)y * step, L->stage + (size_t)y * im->step, rowbytes);
    return IMP_OK;
}

int impgpu_image_width(const impgpu_image* im) { return im ? im->w : 0; }
int impgpu_image_height(const impgpu_image* im) { return im ? im->h : 0; }
int impgpu_image_channels(const impgpu_image* im) { return im ? im->c : 0; }
int impgpu_image_step(const impgpu_image* im) { return im ? im->step : 0; }
void* impgpu_image_device_ptr(const impgpu_image* im) { return im ? im->d : nullptr; }

void impgpu_image_release(impgpu_image** im) {
    if (!im || !*im) return;
    image_delete(*im);
    *im = nullptr;
}

}  // extern "C"
