// imp_runtime.hip -- per-worker environment: device, stream, HBM buffer pool, pinned staging,
// frame upload / download.  Stands where the reference's empty OnEnvStart / OnEnvDestroy
// (bridge.c:10-16) and its cvCreateImage / cvReleaseImage calls are.
//
// One env per process (one nginx worker = one process = one stream, module.c:100-107).
// Device buffers come from a size-bucketed free list; because every operator of a request
// is enqueued on the same stream, a buffer released by one operator can be handed to the
// next without a device sync (stream order is the only ordering needed).
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include "imp_internal.h"

namespace imp {

struct Env {
    int device = -1;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::multimap<size_t, void*> free_list;     // bucket size -> buffer
    std::map<void*, size_t> live;               // buffer -> bucket size
    size_t pooled_bytes = 0;
    // pinned staging for upload / download
    uint8_t* stage = nullptr;
    size_t stage_cap = 0;
    hipEvent_t stage_done = nullptr;
    bool stage_busy = false;
    // pinned ring for the small per-launch tables (LUTs, Gaussian taps): async H2D copies need a
    // source that outlives the call
    uint8_t* ring = nullptr;
    size_t ring_cap = 0, ring_pos = 0;
};

static Env* g_env = nullptr;
static thread_local std::string t_error;

void set_error(const char* what, hipError_t e) {
    t_error = std::string(what) + ": " + hipGetErrorString(e);
}
bool env_ready() { return g_env != nullptr; }
hipStream_t env_stream() { return g_env ? g_env->stream : nullptr; }

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
    if (!g_env) { t_error = "impgpu_env_start has not been called"; return IMP_ERROR_DEVICE; }
    size_t b = bucket_of(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lk(g_env->mu);
        auto it = g_env->free_list.find(b);
        if (it != g_env->free_list.end()) {
            *out = it->second;
            g_env->free_list.erase(it);
            g_env->live[*out] = b;
            return IMP_OK;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, b);
    if (e != hipSuccess) {
        // drop the cache and retry once
        {
            std::lock_guard<std::mutex> lk(g_env->mu);
            (void)hipStreamSynchronize(g_env->stream);
            for (auto& kv : g_env->free_list) (void)hipFree(kv.second);
            g_env->free_list.clear();
        }
        e = hipMalloc(&p, b);
        if (e != hipSuccess) { set_error("hipMalloc", e); return IMP_ERROR_MALLOC_FAILED; }
    }
    std::lock_guard<std::mutex> lk(g_env->mu);
    g_env->live[p] = b;
    g_env->pooled_bytes += b;
    *out = p;
    return IMP_OK;
}

void dev_free(void* p) {
    if (!p || !g_env) return;
    std::lock_guard<std::mutex> lk(g_env->mu);
    auto it = g_env->live.find(p);
    if (it == g_env->live.end()) return;
    g_env->free_list.emplace(it->second, p);
    g_env->live.erase(it);
}

int image_new(int w, int h, int c, impgpu_image** out) {
    if (w <= 0 || h <= 0 || (c != 1 && c != 3 && c != 4)) return IMP_ERROR_INVALID_ARGS;
    impgpu_image* im = new impgpu_image();
    im->w = w; im->h = h; im->c = c;
    im->step = aligned_step(w, c);
    im->cap = (size_t)im->step * h;
    void* p = nullptr;
    int rc = dev_alloc(im->cap + 16, &p);   // +16: kernels may read one 16-byte vector that ends past the last pixel
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
    if (!g_env) { t_error = "impgpu_env_start has not been called"; return IMP_ERROR_DEVICE; }
    void* p = nullptr;
    int rc = dev_alloc(bytes, &p);
    if (rc) return rc;
    uint8_t* slot;
    {
        std::lock_guard<std::mutex> lk(g_env->mu);
        Env* E = g_env;
        const size_t need = (bytes + 63) & ~size_t(63);
        if (need > E->ring_cap) {       // first use, or a blob larger than the ring
            (void)hipDeviceSynchronize();
            if (E->ring) (void)hipHostFree(E->ring);
            E->ring = nullptr;
            E->ring_cap = 0;
            size_t cap = need * 2 > (size_t(4) << 20) ? need * 2 : (size_t(4) << 20);
            hipError_t e = hipHostMalloc((void**)&E->ring, cap, hipHostMallocDefault);
            if (e != hipSuccess) { set_error("hipHostMalloc(ring)", e); dev_free(p); return IMP_ERROR_DEVICE; }
            E->ring_cap = cap;
            E->ring_pos = 0;
        }
        if (E->ring_pos + need > E->ring_cap) {   // wrap: everything that read the ring must be done
            (void)hipDeviceSynchronize();
            E->ring_pos = 0;
        }
        slot = E->ring + E->ring_pos;
        E->ring_pos += need;
    }
    std::memcpy(slot, host, bytes);
    hipError_t e = hipMemcpyAsync(p, slot, bytes, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) { set_error("hipMemcpyAsync(small)", e); dev_free(p); return IMP_ERROR_DEVICE; }
    *dev = p;
    return IMP_OK;
}

static int stage_reserve(size_t bytes) {
    Env* E = g_env;
    if (E->stage_busy) {
        IMP_HIP(hipEventSynchronize(E->stage_done));
        E->stage_busy = false;
    }
    if (E->stage_cap >= bytes) return IMP_OK;
    if (E->stage) IMP_HIP(hipHostFree(E->stage));
    E->stage = nullptr;
    E->stage_cap = 0;
    size_t cap = bucket_of(bytes);
    IMP_HIP(hipHostMalloc((void**)&E->stage, cap, hipHostMallocDefault));
    E->stage_cap = cap;
    return IMP_OK;
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
    e = hipStreamCreateWithFlags(&E->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { set_error("hipStreamCreate", e); delete E; return IMP_ERROR_DEVICE; }
    e = hipEventCreateWithFlags(&E->stage_done, hipEventDisableTiming);
    if (e != hipSuccess) { set_error("hipEventCreate", e); (void)hipStreamDestroy(E->stream); delete E; return IMP_ERROR_DEVICE; }
    g_env = E;
    return IMP_OK;
}

void impgpu_env_destroy(void) {
    Env* E = g_env;
    if (!E) return;
    (void)hipStreamSynchronize(E->stream);
    for (auto& kv : E->free_list) (void)hipFree(kv.second);
    for (auto& kv : E->live) (void)hipFree(kv.first);
    if (E->stage) (void)hipHostFree(E->stage);
    if (E->ring) (void)hipHostFree(E->ring);
    (void)hipEventDestroy(E->stage_done);
    (void)hipStreamDestroy(E->stream);
    g_env = nullptr;
    delete E;
}

int impgpu_env_device(void) { return g_env ? g_env->device : -1; }
const char* impgpu_last_error(void) { return t_error.c_str(); }
void* impgpu_env_stream(void) { return g_env ? (void*)g_env->stream : nullptr; }

int impgpu_sync(void) {
    if (!g_env) { t_error = "impgpu_env_start has not been called"; return IMP_ERROR_DEVICE; }
    IMP_HIP(hipStreamSynchronize(g_env->stream));
    return IMP_OK;
}

int impgpu_image_create(int width, int height, int channels, impgpu_image** out) {
    if (!out) return IMP_ERROR_INVALID_ARGS;
    return image_new(width, height, channels, out);
}

int impgpu_image_upload(const unsigned char* data, int width, int height, int channels, int step,
                        impgpu_image** out) {
    if (!data || !out || step < width * channels) return IMP_ERROR_INVALID_ARGS;
    if (!g_env) { t_error = "impgpu_env_start has not been called"; return IMP_ERROR_DEVICE; }
    impgpu_image* im = nullptr;
    int rc = image_new(width, height, channels, &im);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_env->mu);   // staging buffer is shared
    size_t bytes = (size_t)im->step * height;
    rc = stage_reserve(bytes);
    if (rc) { image_delete(im); return rc; }
    // repack into the device row pitch (cvCreateImage alignment) inside pinned memory
    size_t rowbytes = (size_t)width * channels;
    if ((size_t)step == (size_t)im->step) {
        std::memcpy(g_env->stage, data, bytes - (im->step - rowbytes));
    } else {
        for (int y = 0; y < height; y++)
            std::memcpy(g_env->stage + (size_t)y * im->step, data + (size_t)y * step, rowbytes);
    }
    hipError_t e = hipMemcpyAsync(im->d, g_env->stage, bytes, hipMemcpyHostToDevice, g_env->stream);
    if (e == hipSuccess) e = hipEventRecord(g_env->stage_done, g_env->stream);
    if (e != hipSuccess) { set_error("hipMemcpyAsync(upload)", e); image_delete(im); return IMP_ERROR_DEVICE; }
    g_env->stage_busy = true;
    *out = im;
    return IMP_OK;
}

int impgpu_image_wrap(void* device_ptr, int width, int height, int channels, int step, impgpu_image** out) {
    if (!device_ptr || !out || width <= 0 || height <= 0 || (channels != 1 && channels != 3 && channels != 4) ||
        step < width * channels)
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
    if (!g_env) { t_error = "impgpu_env_start has not been called"; return IMP_ERROR_DEVICE; }
    std::lock_guard<std::mutex> lk(g_env->mu);
    size_t bytes = (size_t)im->step * im->h;
    int rc = stage_reserve(bytes);
    if (rc) return rc;
    IMP_HIP(hipMemcpyAsync(g_env->stage, im->d, bytes, hipMemcpyDeviceToHost, g_env->stream));
    IMP_HIP(hipStreamSynchronize(g_env->stream));
    size_t rowbytes = (size_t)im->w * im->c;
    for (int y = 0; y < im->h; y++)
        std::memcpy(data + (size_t)y * step, g_env->stage + (size_t)y * im->step, rowbytes);
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
