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
//
// Nothing on the per-request path waits for the device except the call that hands pixels back to the
// host: pool reuse is ordered by the lane's stream; memory that a caller-supplied ("foreign") stream
// still reads is parked behind an event and taken back when the event has fired (dev_free_on); the
// pinned ring for small tables is cut into segments, each fenced by its own event, so a wrap waits
// only for copies issued a whole ring ago; frame staging alternates between two pinned buffers, so
// the host copy of request N+1 overlaps the H2D DMA of request N.
#include <dlfcn.h>
#include <sched.h>
#include <time.h>
#include <unistd.h>
#include <cstdlib>
#include <atomic>
#include <cctype>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <vector>
#include "imp_internal.h"

namespace imp {

constexpr int    RING_SEGS = 8;
constexpr size_t RING_SEG_BYTES = size_t(128) << 10;   // blobs above this take a one-off pinned buffer
constexpr int    N_STAGE = 2;
constexpr size_t MAILBOX_BYTES = 5 * 4096;              // pinned words a kernel's verdict is copied into (behind the ring): 1024 for everyone, then 1024 per JPEG group in flight (imp_jpeg_api.cpp)

struct Staging {
    uint8_t* p = nullptr;
    size_t cap = 0;
    hipEvent_t done = nullptr;
    bool busy = false;
    bool held = false;                  // a download that has been enqueued into it is read later (stage_hold): not handed out meanwhile
    int small_streak = 0;               // reservations in a row that needed less than an eighth of the buffer
};

struct Parked {             // a pool block waiting for an event of a foreign stream
    hipEvent_t ev;
    void* dev;
};

struct Lane {
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;                 // a second stream of the lane, made on first use (lane_side_stream)
    std::mutex mu;                              // guards the pool: frees may come from other threads
    std::multimap<size_t, void*> free_list;     // bucket size -> buffer
    size_t free_bytes = 0;                      // sum of the free list (trimmed above pool_cap() when the lane is idle)
    std::map<void*, size_t> live;               // buffer -> bucket size
    Staging stage[N_STAGE];                     // pinned staging for pageable upload / download, used alternately
    int stage_next = 0;
    uint8_t* ring = nullptr;                    // pinned ring for per-launch tables (LUTs, Gaussian taps, resize tables)
    int seg = 0;                                // segment being filled
    size_t seg_pos = 0;
    hipEvent_t seg_done[RING_SEGS] = {};        // recorded on the lane stream when a segment is left
    bool seg_busy[RING_SEGS] = {};
    hipEvent_t join_ev = nullptr;               // lane stream -> foreign stream ordering
    hipEvent_t sync_ev = nullptr;               // blocking wait for the lane stream (lane_wait)
    std::vector<hipEvent_t> ev_pool;
    std::vector<hipEvent_t> mark_pool;          // blocking-sync events for lane_mark / lane_wait_mark
    std::deque<Parked> parked;
    LaneCache* caches[LANE_CACHE_SLOTS] = {};
};

struct Env {
    int device = -1;
    int numa_node = -1;                         // of the device's PCI function (-1: unknown / a single-node host)
    std::vector<int> node_cpus;                 // that node's CPUs, already cut down to what this process may run on
    unsigned long long generation = 0;
    std::mutex mu;
    std::vector<Lane*> lanes;
    std::vector<Lane*> idle;                    // lanes whose thread has exited: the next new thread takes one over
};

static void lane_destroy(Lane* L);
static std::atomic<Env*> g_env{nullptr};
static std::mutex g_env_mu;                         // env start / destroy against threads that end meanwhile (LaneReturn)
static unsigned long long g_generation = 0;
static thread_local Lane* t_lane = nullptr;
static thread_local unsigned long long t_lane_gen = 0;

// A thread that ends hands its lane (stream, pool, staging, caches) back instead of leaking it: a server that churns
// threads would otherwise collect a stream and tens of MB of pinned memory per dead thread until impgpu_env_destroy.
struct LaneReturn {
    ~LaneReturn() {
        // under the env lock: impgpu_env_destroy cannot free the env between the look and the push, and the thread forgets
        // the lane it hands back (a later thread_local destructor calling into the library gets a fresh one)
        std::lock_guard<std::mutex> lk(g_env_mu);
        Env* E = g_env.load();
        if (E && t_lane && t_lane_gen == E->generation) {
            std::lock_guard<std::mutex> lk2(E->mu);
            E->idle.push_back(t_lane);
        }
        t_lane = nullptr;
        t_lane_gen = 0;
    }
};
static thread_local LaneReturn t_lane_return;
static thread_local std::string t_error;

// ---- rocTX (optional) and fault injection
static void (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static std::atomic<int> g_fault_step{-1};
static long g_fault_countdown = 0;
static std::mutex g_fault_mu;

static void trace_init() {
    g_roctx_push = nullptr;
    g_roctx_pop = nullptr;
    const bool want = std::getenv("IMPGPU_ROCTX") != nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);            // a profiler already brought it in
        if (!h && want) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        auto push = (void (*)(const char*))dlsym(h, "roctxRangePushA");
        auto pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) { g_roctx_push = push; g_roctx_pop = pop; return; }
    }
}
void trace_push(const char* name) { if (g_roctx_push) g_roctx_push(name); }
void trace_pop() { if (g_roctx_pop) (void)g_roctx_pop(); }

// Armed only by an explicit call (impgpu_fault_arm): a stray environment variable inherited by an nginx worker must not be
// able to fail requests.  step < 0 disarms.
static void fault_arm(int step, long nth) {
    std::lock_guard<std::mutex> lk(g_fault_mu);
    g_fault_step = -1;
    g_fault_countdown = 0;
    if (step < IMP_STEP_START || step > IMP_STEP_ENCODE) return;
    g_fault_countdown = nth < 1 ? 1 : nth;
    g_fault_step = step;
}

void set_error(const char* what, hipError_t e);
bool fault_hit(int step) {
    if (g_fault_step.load(std::memory_order_relaxed) < 0) return false;   // the only cost when not armed
    std::lock_guard<std::mutex> lk(g_fault_mu);
    if (step != g_fault_step || g_fault_countdown <= 0) return false;
    if (--g_fault_countdown > 0) return false;
    g_fault_step = -1;
    set_error("injected fault (IMPGPU_FAULT)", hipErrorUnknown);
    return true;
}

void set_error(const char* what, hipError_t e) {
    t_error = std::string(what) + ": " + hipGetErrorString(e);
}
void set_error_text(const char* what) { t_error = what; }
bool env_ready() { return g_env != nullptr; }

static int no_env() {
    t_error = "impgpu_env_start has not been called";
    return IMP_ERROR_DEVICE;
}

// ---- NUMA ("one host thread per GPU, NUMA-local"): which node the device hangs off, and that node's CPUs
static void numa_probe(Env* E) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, E->device) != hipSuccess) return;
    for (char* c = bus; *c; c++) *c = (char)std::tolower((unsigned char)*c);
    char path[160];
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = std::fopen(path, "r");
    if (!f) return;
    int node = -1;
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
    if (node < 0) return;
    std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = std::fopen(path, "r");
    if (!f) return;
    char list[4096] = {0};
    const bool got = std::fgets(list, sizeof list, f) != nullptr;
    std::fclose(f);
    if (!got) return;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    for (char* tok = std::strtok(list, ",\n"); tok; tok = std::strtok(nullptr, ",\n")) {      // "0-15,32-47"
        int lo = 0, hi = 0;
        const int n = std::sscanf(tok, "%d-%d", &lo, &hi);
        if (n < 1) continue;
        if (n == 1) hi = lo;
        for (int c = lo; c <= hi && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET(c, &allowed)) E->node_cpus.push_back(c);
    }
    E->numa_node = node;
}

// Bind the calling thread to the CPUs of the device's node: its staging copies then run next to the PCIe root the GPU
// hangs off, and the pinned memory it allocates afterwards (first touch) is that node's.
static int numa_bind_thread(Env* E) {
    if (!E || E->numa_node < 0 || E->node_cpus.empty()) return IMP_ERROR_UNSUPPORTED;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : E->node_cpus) CPU_SET(c, &set);
    return sched_setaffinity(0, sizeof set, &set) == 0 ? IMP_OK : IMP_ERROR_UNSUPPORTED;
}

// The calling thread's lane, created on first use (HIP's current device is per thread too).
static Lane* lane() {
    Env* E = g_env;
    if (!E) return nullptr;
    if (t_lane && t_lane_gen == E->generation) return t_lane;
    if (hipSetDevice(E->device) != hipSuccess) return nullptr;
    (void)&t_lane_return;                       // instantiate this thread's returner
    {
        std::lock_guard<std::mutex> lk(E->mu);
        if (!E->idle.empty()) {                 // everything the previous owner enqueued is ordered on the lane's stream
            t_lane = E->idle.back();
            E->idle.pop_back();
            t_lane_gen = E->generation;
            return t_lane;
        }
    }
    static const bool bind = [] { const char* s = std::getenv("IMPGPU_NUMA_BIND"); return s && *s == '1'; }();
    if (bind) (void)numa_bind_thread(E);        // before the lane's pinned buffers are allocated (first touch)
    Lane* L = new Lane();
    // IMPGPU_LANE_CU_SPLIT=n (2, 4, 8; A/B builds only -- it LOST): the k-th lane of the process launches on a contiguous n-th of
    // the device's CU mask bits -- which the hardware deals round over the XCDs, so a part spans all eight
    // (tools/cu_place_probe.hip) -- instead of the whole device.  The idea: the broker's lanes run chains of small kernels side
    // by side, and unmasked the workgroups of concurrent small kernels share compute units while others idle (4 streams x 64
    // workgroups: 179 CUs used, 63 shared by up to four workgroups; masked: 256 used, none shared).  Measured with four lanes,
    // requests/s at 1 / 8 / 16 / 32 workers: unmasked 2.26 / 10.3 / 14.2 / 20.0 k, quarters 1.98 / 9.3 / 12.8 / 17.5 k, halves
    // 2.25 / 10.2 / 13.6 / 19.5 k -- the chains are slowed by the device's total vector work, not by where it lands.
    static const int split = [] { const char* s = ab_env("IMPGPU_LANE_CU_SPLIT"); const int v = s ? std::atoi(s) : 0; return (v == 2 || v == 4 || v == 8) ? v : 0; }();
    bool made = false;
    if (split) {
        static std::atomic<int> next_part{0};
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, E->device) == hipSuccess && prop.multiProcessorCount >= 8 * split && prop.multiProcessorCount <= 1024) {
            const int ncu = prop.multiProcessorCount, per = ncu / split, part = next_part.fetch_add(1) % split;
            uint32_t mask[32] = {0};
            for (int i = part * per; i < (part + 1) * per; i++) mask[i / 32] |= 1u << (i % 32);
            made = hipExtStreamCreateWithCUMask(&L->stream, (uint32_t)((ncu + 31) / 32), mask) == hipSuccess;
            if (!made) { L->stream = nullptr; (void)hipGetLastError(); }
        }
    }
    bool ok = (made || hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking) == hipSuccess) &&
              hipEventCreateWithFlags(&L->join_ev, hipEventDisableTiming) == hipSuccess &&
              hipHostMalloc((void**)&L->ring, RING_SEGS * RING_SEG_BYTES + MAILBOX_BYTES, hipHostMallocDefault) == hipSuccess &&
              hipEventCreateWithFlags(&L->sync_ev, hipEventDisableTiming | hipEventBlockingSync) == hipSuccess;
    for (int i = 0; ok && i < N_STAGE; i++) ok = hipEventCreateWithFlags(&L->stage[i].done, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < RING_SEGS; i++) ok = hipEventCreateWithFlags(&L->seg_done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        lane_destroy(L);
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

static bool lane_take_back(Lane* L, void* p);

// Take back whatever the parked events have released (front to back: a stream's events fire in order, and a few
// unfinished entries of another stream in front only delay the ones behind them until the next call).
static void reap(Lane* L, bool wait) {
    while (!L->parked.empty()) {
        Parked& k = L->parked.front();
        if (wait) (void)hipEventSynchronize(k.ev);
        else if (hipEventQuery(k.ev) != hipSuccess) break;
        lane_take_back(L, k.dev);
        L->ev_pool.push_back(k.ev);
        L->parked.pop_front();
    }
}

static void park(Lane* L, void* dev, hipStream_t s) {
    hipEvent_t ev = nullptr;
    if (!L->ev_pool.empty()) { ev = L->ev_pool.back(); L->ev_pool.pop_back(); }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    if (!ev || hipEventRecord(ev, s) != hipSuccess) {       // cannot fence it: wait now rather than hand it out early
        (void)hipStreamSynchronize(s);
        if (ev) L->ev_pool.push_back(ev);
        lane_take_back(L, dev);
        return;
    }
    L->parked.push_back(Parked{ev, dev});
}

int dev_alloc(size_t bytes, void** out) {
    Lane* L = lane();
    if (!L) return no_env();
    if (!L->parked.empty()) reap(L, false);
    const size_t b = bucket_of(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lk(L->mu);
        auto it = L->free_list.find(b);
        if (it != L->free_list.end()) {
            *out = it->second;
            L->free_list.erase(it);
            L->free_bytes -= b;
            L->live[*out] = b;
            return IMP_OK;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, b);
    if (e != hipSuccess) {
        // drop this lane's cache and retry once
        reap(L, true);
        {
            std::lock_guard<std::mutex> lk(L->mu);
            (void)hipStreamSynchronize(L->stream);
            for (auto& kv : L->free_list) (void)hipFree(kv.second);
            L->free_list.clear();
            L->free_bytes = 0;
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
    L->free_bytes += it->second;
    L->live.erase(it);
    return true;
}

// Bytes a lane may keep on its free list.  Without a bound a worker that once saw a burst of large frames holds their
// buckets for ever, and N workers x lanes share one GPU.  IMPGPU_POOL_CAP_MB (default 8192 -- 288 GB of HBM and a
// hipFree / hipMalloc pair that costs more than a 1080p request: the default only catches a pool that has run away,
// a 256-file JPEG batch legitimately recycles 1.2 GB of coefficient planes per call; 0 = never trim).
static size_t pool_cap() {
    static const size_t cap = [] {
        const char* s = std::getenv("IMPGPU_POOL_CAP_MB");
        return (size_t)(s ? std::atoll(s) : 8192) << 20;
    }();
    return cap;
}

// Called when the lane's stream has just been waited for (nothing enqueued can still read a free block): give the largest
// free blocks back to the driver until the list is under the cap.  hipFree is a device-wide wait, which is why this is not
// done on the enqueue path and only past the cap.
static void pool_trim(Lane* L) {
    const size_t cap = pool_cap();
    if (!cap || L->free_bytes <= cap) return;
    std::lock_guard<std::mutex> lk(L->mu);
    while (L->free_bytes > cap / 2 && !L->free_list.empty()) {
        auto it = std::prev(L->free_list.end());
        (void)hipFree(it->second);
        L->free_bytes -= it->first;
        L->free_list.erase(it);
    }
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
        if (L != mine && lane_take_back(L, p)) {
            // A frame belongs to the lane (thread) that created it: the block goes back to its owner's free list with no
            // ordering against work this thread may still have enqueued on it.  Supported only for frames that are idle
            // (a garbage collector's finaliser); -DIMPGPU_DEBUG builds stop here so that a misuse is found in testing.
#ifdef IMPGPU_DEBUG
            std::fprintf(stderr, "impgpu: frame memory %p released from a thread that does not own it\n", p);
            std::abort();
#endif
            return;
        }
}

// Free `p` once everything enqueued on `s` so far is done.  On the lane's own stream that is plain stream order (the
// next user of the block is enqueued behind it); a foreign stream's work is fenced with an event, no host wait.
void dev_free_on(void* p, hipStream_t s) {
    Lane* L = lane();
    if (!p || !L) return;
    if (s == L->stream) { dev_free(p); return; }
    bool mine;
    {
        std::lock_guard<std::mutex> lk(L->mu);
        mine = L->live.count(p) != 0;
    }
    if (mine) { park(L, p, s); return; }        // stays in `live` until the event has fired
    (void)hipStreamSynchronize(s);              // another lane's block: its owner cannot see our event
    dev_free(p);
}

// Make the foreign stream `s` wait for what the lane's stream holds right now: pool blocks are recycled in lane-stream
// order, so a block about to be used on `s` may still be read by an earlier lane-stream kernel.
int stream_join(hipStream_t s) {
    Lane* L = lane();
    if (!L) return no_env();
    if (s == L->stream) return IMP_OK;
    IMP_HIP(hipEventRecord(L->join_ev, L->stream));
    IMP_HIP(hipStreamWaitEvent(s, L->join_ev, 0));
    return IMP_OK;
}

// The other direction: the lane's stream waits for what `s` holds now (before a block `s` was reading is recycled).
int stream_join_back(hipStream_t s) {
    Lane* L = lane();
    if (!L) return no_env();
    if (s == L->stream) return IMP_OK;
    IMP_HIP(hipEventRecord(L->join_ev, s));
    IMP_HIP(hipStreamWaitEvent(L->stream, L->join_ev, 0));
    return IMP_OK;
}

bool lane_stream_idle() {
    Lane* L = lane();
    return L && hipStreamQuery(L->stream) == hipSuccess;
}

int dev_alloc_on(size_t bytes, void** out, hipStream_t s) {
    if (int rc = dev_alloc(bytes, out)) return rc;
    if (int rc = stream_join(s)) { dev_free(*out); *out = nullptr; return rc; }
    return IMP_OK;
}

bool on_lane_stream(hipStream_t s) {
    Lane* L = lane();
    return L && s == L->stream;
}

// A few pinned words per lane for results a kernel leaves behind (a D2H copy into pageable memory would make the runtime
// wait inside the copy call, and hold its locks while it does).
uint32_t* lane_mailbox() {
    Lane* L = lane();
    return L ? (uint32_t*)(L->ring + RING_SEGS * RING_SEG_BYTES) : nullptr;
}

// Wait for the lane's stream.  hipStreamSynchronize spins on a host core for the whole wait; with more waiting threads than
// cores (a request stream, nginx workers sharing a box) that starves the threads that have host work to do, so the default
// is an event created with hipEventBlockingSync: the thread sleeps until the interrupt.  IMPGPU_SYNC=spin takes the other one.
int lane_wait() {
    Lane* L = lane();
    if (!L) return no_env();
    static const bool spin = [] { const char* s = std::getenv("IMPGPU_SYNC"); return s && !std::strcmp(s, "spin"); }();
    if (spin) { IMP_HIP(hipStreamSynchronize(L->stream)); }
    else {
        IMP_HIP(hipEventRecord(L->sync_ev, L->stream));
        IMP_HIP(hipEventSynchronize(L->sync_ev));
    }
    if (L->parked.empty()) pool_trim(L);        // (blocks parked behind a foreign stream's event are not on the free list)
    return IMP_OK;
}

// A point of the lane's stream to come back to: lane_wait_mark sleeps until everything enqueued BEFORE the mark is done --
// not what the thread enqueued after it (a JPEG group's verdicts are read while the next group is already on the device).
// A second stream for work that should overlap what the thread enqueues next on its lane stream (a JPEG batch begun ahead of
// the answers of the batch before).  Pool memory used on it follows the foreign-stream rules: stream_join before the first
// use, dev_free_on afterwards.
hipStream_t lane_side_stream() {
    Lane* L = lane();
    if (!L) return nullptr;
    if (!L->side && hipStreamCreateWithFlags(&L->side, hipStreamNonBlocking) != hipSuccess) { L->side = nullptr; (void)hipGetLastError(); }
    return L->side;
}

int lane_mark(void** mark) { return lane_mark_on(nullptr, mark); }

int lane_mark_on(hipStream_t s, void** mark) {
    Lane* L = lane();
    if (!L) return no_env();
    if (!s) s = L->stream;
    hipEvent_t ev = nullptr;
    if (!L->mark_pool.empty()) { ev = L->mark_pool.back(); L->mark_pool.pop_back(); }
    else IMP_HIP(hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming));
    const hipError_t e = hipEventRecord(ev, s);
    if (e != hipSuccess) { L->mark_pool.push_back(ev); set_error("hipEventRecord(mark)", e); return IMP_ERROR_DEVICE; }
    *mark = (void*)ev;
    return IMP_OK;
}

int lane_wait_mark(void* mark) {
    Lane* L = lane();
    if (!L) return no_env();
    if (!mark) return lane_wait();
    hipEvent_t ev = (hipEvent_t)mark;
    static const bool spin = [] { const char* s = std::getenv("IMPGPU_SYNC"); return s && !std::strcmp(s, "spin"); }();
    hipError_t e = hipSuccess;
    if (spin) { while ((e = hipEventQuery(ev)) == hipErrorNotReady) {} }
    else e = hipEventSynchronize(ev);
    L->mark_pool.push_back(ev);
    if (e != hipSuccess) { set_error("hipEventSynchronize(mark)", e); return IMP_ERROR_DEVICE; }
    return IMP_OK;
}

LaneCache** lane_cache_slot(int which) {
    Lane* L = lane();
    return (L && which >= 0 && which < LANE_CACHE_SLOTS) ? &L->caches[which] : nullptr;
}

int image_new(int w, int h, int c, impgpu_image** out) { return image_new_album(w, h, c, 1, out); }

int image_new_album(int w, int h, int c, int frames, impgpu_image** out) {
    if (w <= 0 || h <= 0 || frames <= 0 || frames > 65535 || (c != 1 && c != 3 && c != 4)) return IMP_ERROR_INVALID_ARGS;
    // the kernels index pixels and row bytes in 32 bits: a frame is at most 2^30 pixels and 4 GiB - 1 of rows.  Larger
    // requests (resize=2000000000,1,up with the size watchdog off) end like a failed cvCreateImage, not in a wrapped pitch.
    if (!frame_fits(w, h, c)) { t_error = "frame too large"; return IMP_ERROR_MALLOC_FAILED; }
    impgpu_image* im = new impgpu_image();
    im->w = w; im->h = h; im->c = c;
    im->step = aligned_step(w, c);
    im->cap = (size_t)im->step * h;
    if (frames > 1) {                                       // frames start on 256-byte boundaries of one block
        im->frames = frames;
        im->fstride = (im->cap + 255) & ~size_t(255);
        im->cap = im->fstride * frames;
    }
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

// The next staging buffer, free to overwrite and at least `bytes` large.  Two buffers alternate: while the DMA of
// one request still reads buffer A the host fills buffer B for the next.
static int stage_reserve(Lane* L, size_t bytes, Staging** out) {
    Staging* S = &L->stage[L->stage_next];
    L->stage_next = (L->stage_next + 1) % N_STAGE;
    if (S->held) {                                             // (an answer begun and not fetched yet lies in it: the other one)
        S = &L->stage[L->stage_next];
        L->stage_next = (L->stage_next + 1) % N_STAGE;
        if (S->held) { t_error = "both staging buffers of the thread hold answers that have not been fetched"; return IMP_ERROR_INVALID_ARGS; }
    }
    if (S->busy) {
        IMP_HIP(hipEventSynchronize(S->done));
        S->busy = false;
    }
    // A buffer that one outsized request made larger than IMPGPU_STAGE_CAP_MB (default 512: a 256-file JPEG batch stages
    // 112 MB) goes back when it is next taken for something a quarter of that cap would hold: a worker does not keep an
    // unswappable gigabyte for life because of one request.  (Here and not after a wait: the caller still reads a download
    // out of the buffer then.)
    static const size_t trim_cap = [] {
        const char* s = std::getenv("IMPGPU_STAGE_CAP_MB");
        return (size_t)(s ? std::atoll(s) : 512) << 20;
    }();
    bool outsized = trim_cap && S->cap > trim_cap && bytes <= trim_cap / 4;
    // ... and below that cap: a buffer of more than 64 MB (one PNG of 4096 x 16384 stages 268 MB for a file that may weigh
    // 260 KB) that the last sixteen requests each used less than an eighth of goes back too -- N workers x lanes x two
    // buffers of unswappable memory is the worst case an operator has to budget for
    S->small_streak = (S->cap > (size_t(64) << 20) && bytes <= S->cap / 8) ? S->small_streak + 1 : 0;
    if (S->small_streak >= 16) outsized = true;
    if (S->cap < bytes || outsized) {
        S->small_streak = 0;
        if (S->p) IMP_HIP(hipHostFree(S->p));
        S->p = nullptr;
        S->cap = 0;
        // (hipHostFree above waits for the device: start at a 1080p frame so the common sizes never regrow)
        const size_t cap = bucket_of(bytes < (size_t(8) << 20) ? (size_t(8) << 20) : bytes);
        IMP_HIP(hipHostMalloc((void**)&S->p, cap, hipHostMallocDefault));
        S->cap = cap;
    }
    *out = S;
    return IMP_OK;
}

// Small host blob -> pool memory, visible to work enqueued on `s` afterwards.  The copy itself always rides the lane's
// stream (so ONE event per ring segment fences every copy out of it); a foreign `s` is made to wait for it.
int upload_small(const void* host, size_t bytes, void** dev, hipStream_t s) {
    void* p = nullptr;
    int rc = dev_alloc(bytes, &p);
    if (rc) return rc;
    rc = upload_to(p, host, bytes, s);
    if (rc) { dev_free(p); return rc; }
    *dev = p;
    return IMP_OK;
}

// The copy of upload_small into memory the caller already holds (pool memory of this lane, or a part of it).
int upload_to(void* p, const void* host, size_t bytes, hipStream_t s) {
    Lane* L = lane();
    if (!L) return no_env();
    int rc;
    const size_t need = (bytes + 63) & ~size_t(63);
    hipError_t e;
    if (need > RING_SEG_BYTES) {            // GIF albums and the like: through the frame staging buffers
        Staging* S = nullptr;
        rc = stage_reserve(L, bytes, &S);
        if (rc) return rc;
        std::memcpy(S->p, host, bytes);
        e = hipMemcpyAsync(p, S->p, bytes, hipMemcpyHostToDevice, L->stream);
        if (e == hipSuccess) e = hipEventRecord(S->done, L->stream);
        if (e != hipSuccess) { set_error("hipMemcpyAsync(upload)", e); (void)hipStreamSynchronize(L->stream); return IMP_ERROR_DEVICE; }
        S->busy = true;
    } else {
        if (L->seg_pos + need > RING_SEG_BYTES) {             // leave this segment: fence its copies, enter the next one
            if (hipEventRecord(L->seg_done[L->seg], L->stream) == hipSuccess) L->seg_busy[L->seg] = true;
            else (void)hipStreamSynchronize(L->stream);
            L->seg = (L->seg + 1) % RING_SEGS;
            L->seg_pos = 0;
            if (L->seg_busy[L->seg]) {                        // copies issued a whole ring ago: long done in practice
                (void)hipEventSynchronize(L->seg_done[L->seg]);
                L->seg_busy[L->seg] = false;
            }
        }
        uint8_t* slot = L->ring + (size_t)L->seg * RING_SEG_BYTES + L->seg_pos;
        L->seg_pos += need;
        std::memcpy(slot, host, bytes);
        e = hipMemcpyAsync(p, slot, bytes, hipMemcpyHostToDevice, L->stream);
        if (e != hipSuccess) { set_error("hipMemcpyAsync(small)", e); return IMP_ERROR_DEVICE; }
    }
    if (s != L->stream) {
        rc = stream_join(s);
        if (rc) return rc;
    }
    return IMP_OK;
}

// A host-built blob too large for the ring (a JPEG's entropy-coded segment, its coefficient planes): the caller fills the
// pinned buffer stage_begin hands out, stage_upload enqueues the copy on the lane's stream and fences the buffer.
int stage_begin(size_t bytes, void** host, void** token) {
    Lane* L = lane();
    if (!L) return no_env();
    Staging* S = nullptr;
    if (int rc = stage_reserve(L, bytes ? bytes : 1, &S)) return rc;
    *host = S->p;
    *token = S;
    return IMP_OK;
}

size_t stage_capacity(void* token) { return token ? ((Staging*)token)->cap : 0; }
void stage_hold(void* token, bool held) { if (token) ((Staging*)token)->held = held; }

int stage_upload(void* token, void* dev, size_t bytes) {
    Lane* L = lane();
    if (!L) return no_env();
    Staging* S = (Staging*)token;
    if (!S || !bytes) return IMP_OK;                 // nothing to send: the buffer is free again at once
    hipError_t e = hipMemcpyAsync(dev, S->p, bytes, hipMemcpyHostToDevice, L->stream);
    if (e == hipSuccess) e = hipEventRecord(S->done, L->stream);
    if (e != hipSuccess) { set_error("hipMemcpyAsync(stage_upload)", e); (void)hipStreamSynchronize(L->stream); return IMP_ERROR_DEVICE; }
    S->busy = true;
    return IMP_OK;
}

// The same in pieces: part of the pinned buffer (from `offset`) to `dev`, on the lane stream; the buffer is fenced when `last`
// (bytes may be 0 then: a decode that gives up after some pieces still has to fence them).
int stage_upload_part(void* token, size_t offset, void* dev, size_t bytes, bool last) {
    Lane* L = lane();
    if (!L) return no_env();
    Staging* S = (Staging*)token;
    if (!S) return IMP_OK;
    hipError_t e = hipSuccess;
    if (bytes) e = hipMemcpyAsync(dev, S->p + offset, bytes, hipMemcpyHostToDevice, L->stream);
    if (e == hipSuccess && last) { e = hipEventRecord(S->done, L->stream); if (e == hipSuccess) S->busy = true; }
    if (e != hipSuccess) { set_error("hipMemcpyAsync(stage_upload_part)", e); (void)hipStreamSynchronize(L->stream); return IMP_ERROR_DEVICE; }
    return IMP_OK;
}

static void lane_destroy(Lane* L) {
    if (L->stream) (void)hipStreamSynchronize(L->stream);
    reap(L, true);
    for (LaneCache*& c : L->caches) { delete c; c = nullptr; }       // host side only: their device blocks are in `live`
    for (auto& kv : L->free_list) (void)hipFree(kv.second);
    for (auto& kv : L->live) (void)hipFree(kv.first);
    for (Staging& S : L->stage) {
        if (S.p) (void)hipHostFree(S.p);
        if (S.done) (void)hipEventDestroy(S.done);
    }
    if (L->ring) (void)hipHostFree(L->ring);
    for (hipEvent_t ev : L->seg_done) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : L->ev_pool) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : L->mark_pool) (void)hipEventDestroy(ev);
    if (L->join_ev) (void)hipEventDestroy(L->join_ev);
    if (L->sync_ev) (void)hipEventDestroy(L->sync_ev);
    if (L->side) { (void)hipStreamSynchronize(L->side); (void)hipStreamDestroy(L->side); }
    if (L->stream) (void)hipStreamDestroy(L->stream);
    delete L;
}

// rows of many frames, each to its own destination and pitch, in one launch (impgpu_batch_download)
struct GatherItem {
    const uint8_t* src;
    uint8_t* dst;
    int rows, rowbytes, sstep, dstep;
};

__global__ __launch_bounds__(256) void k_gather_rows(const GatherItem* __restrict__ items, int rows_per_block) {
    const GatherItem it = items[blockIdx.x];
    const int y0 = blockIdx.y * rows_per_block, y1 = min(it.rows, y0 + rows_per_block);
    const int lane = threadIdx.x & 63;
    const bool words = !(((uintptr_t)it.src | (uintptr_t)it.dst | (uintptr_t)it.sstep | (uintptr_t)it.dstep) & 3);
    for (int y = y0 + (int)(threadIdx.x >> 6); y < y1; y += 4) {             // a wave per row
        const uint8_t* s = it.src + (size_t)y * it.sstep;
        uint8_t* d = it.dst + (size_t)y * it.dstep;
        int done = 0;
        if (words) {
            const int nw = it.rowbytes >> 2;
            for (int i = lane; i < nw; i += 64) ((uint32_t*)d)[i] = ((const uint32_t*)s)[i];
            done = nw << 2;
        }
        for (int i = done + lane; i < it.rowbytes; i += 64) d[i] = s[i];
    }
}

// the device's view of a host address, when the device can write there (hipHostMalloc / hipHostRegister memory)
static bool host_device_ptr(const void* p, void** dev) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
    if (dev) *dev = a.devicePointer;
    return true;
}

}  // namespace imp

using namespace imp;

extern "C" {

// A process that ends with a live env (a worker told to quit between requests, a script that never calls
// impgpu_env_destroy) used to leave the lanes' streams, blocking-sync events, pinned rings and pool blocks alive into the
// HIP runtime's own exit handlers -- and, when another library shares the runtime (torch), into theirs.  impgpu_env_start
// registers this once: handlers run in reverse order of registration, so it runs BEFORE the handlers of everything that
// was initialised before it (the HIP runtime: hipGetDeviceCount below comes first) and after those of whatever came later.
// The wait is bounded: a device that no longer answers must not keep a worker from exiting -- then nothing is touched and
// the env is left to the kernel (IMPGPU_EXIT_WAIT_MS, default 2000).  A forked child of a process that started the env
// (same memory, another pid) must not talk to the parent's device context at all.
static pid_t g_env_pid = 0;

static void env_atexit() {
    if (g_env_pid != getpid()) return;
    {
        std::lock_guard<std::mutex> lk(g_env_mu);
        Env* E = g_env.load();
        if (!E) return;
        const char* s = std::getenv("IMPGPU_EXIT_WAIT_MS");
        const long budget_ms = s ? std::atol(s) : 2000;
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        bool idle = false;
        for (;;) {
            idle = true;
            {
                std::lock_guard<std::mutex> lk2(E->mu);
                for (Lane* L : E->lanes) {
                    if (L->stream && hipStreamQuery(L->stream) == hipErrorNotReady) idle = false;
                    if (L->side && hipStreamQuery(L->side) == hipErrorNotReady) idle = false;
                }
            }
            (void)hipGetLastError();
            if (idle) break;
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000 + (t1.tv_nsec - t0.tv_nsec) / 1000000 > budget_ms) break;
            timespec nap{0, 200000};
            nanosleep(&nap, nullptr);
        }
        if (!idle) {                            // leak on purpose: no further HIP call from this library
            g_env.store(nullptr);
            std::fprintf(stderr, "impgpu: device still busy at exit after %ld ms; env left to the driver\n", budget_ms);
            return;
        }
    }
    impgpu_env_destroy();
}

int impgpu_env_start(int device) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    if (g_env.load()) return IMP_OK;
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
    trace_init();
    Env* E = new Env();
    E->device = device;
    E->generation = ++g_generation;
    numa_probe(E);
    g_env = E;
    g_env_pid = getpid();
    static std::once_flag exit_hook;
    std::call_once(exit_hook, [] { (void)std::atexit(env_atexit); });
    if (!lane()) {  // the calling thread's lane: fails loudly here rather than at the first operator
        g_env = nullptr;
        delete E;
        t_error = "could not create a HIP stream";
        return IMP_ERROR_DEVICE;
    }
    return IMP_OK;
}

void impgpu_env_destroy(void) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    Env* E = g_env.exchange(nullptr);   // from here on no thread finds the env (other threads' t_lane pointers are
    if (!E) return;                     // invalidated by the generation counter); a thread that ends meanwhile waits on the lock
    (void)hipDeviceSynchronize();
    for (Lane* L : E->lanes) lane_destroy(L);
    t_lane = nullptr;
    t_lane_gen = 0;
    delete E;
}

int impgpu_env_numa_node(void) {
    Env* E = g_env.load();
    return E ? E->numa_node : -1;
}

int impgpu_env_bind_thread(void) {
    Env* E = g_env.load();
    if (!E) return no_env();
    return numa_bind_thread(E);
}

int impgpu_fault_arm(int step, long nth) {
    fault_arm(step, nth);
    return IMP_OK;
}

int impgpu_env_device(void) {
    Env* E = g_env.load();
    return E ? E->device : -1;
}
const char* impgpu_last_error(void) { return t_error.c_str(); }
void* impgpu_env_stream(void) { return (void*)env_stream(); }

int impgpu_sync(void) { return lane_wait(); }

int impgpu_image_create(int width, int height, int channels, impgpu_image** out) {
    if (!out) return IMP_ERROR_INVALID_ARGS;
    return image_new(width, height, channels, out);
}

int impgpu_image_upload(const unsigned char* data, int width, int height, int channels, int step,
                        impgpu_image** out) {
    if (!data || !out || (long long)step < (long long)width * channels) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    TraceRange tr("IMP_STEP_DECODE");                       // the decoded frame's hand-over (bridge.c:541-572)
    IMP_FAULT_POINT(IMP_STEP_DECODE);
    impgpu_image* im = nullptr;
    int rc = image_new(width, height, channels, &im);
    if (rc) return rc;
    const size_t bytes = (size_t)im->step * height;
    Staging* S = nullptr;
    rc = stage_reserve(L, bytes, &S);
    if (rc) { image_delete(im); return rc; }
    // repack into the device row pitch (cvCreateImage alignment) inside pinned memory
    const size_t rowbytes = (size_t)width * channels;
    if ((size_t)step == (size_t)im->step) {
        std::memcpy(S->p, data, bytes - (im->step - rowbytes));
    } else {
        for (int y = 0; y < height; y++)
            std::memcpy(S->p + (size_t)y * im->step, data + (size_t)y * step, rowbytes);
    }
    hipError_t e = hipMemcpyAsync(im->d, S->p, bytes, hipMemcpyHostToDevice, L->stream);
    if (e == hipSuccess) e = hipEventRecord(S->done, L->stream);
    if (e != hipSuccess) { set_error("hipMemcpyAsync(upload)", e); (void)hipStreamSynchronize(L->stream); image_delete(im); return IMP_ERROR_DEVICE; }
    S->busy = true;
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
    // straight from the caller's pinned frame: no staging pass.  Equal pitches (every BGRA frame, every frame whose rows are
    // padded to 4 bytes the way cvCreateImage and FreeImage pad them) are one linear DMA.  A different host pitch (tightly
    // packed BGR rows) crosses the link as ONE linear DMA too, into a pool block, and a device copy re-pitches it:
    // hipMemcpy2DAsync moved such frames at 0.8 GB/s (130 requests/s on the mixed-size stream).
    if (step == im->step) {
        const hipError_t e = hipMemcpyAsync(im->d, data, (size_t)step * height, hipMemcpyHostToDevice, L->stream);
        if (e != hipSuccess) { set_error("hipMemcpyAsync(upload_pinned)", e); image_delete(im); return IMP_ERROR_DEVICE; }
    } else {
        void* tmp = nullptr;
        const size_t bytes = (size_t)step * (height - 1) + (size_t)width * channels;
        rc = dev_alloc(bytes, &tmp);
        if (rc) { image_delete(im); return rc; }
        const hipError_t e = hipMemcpyAsync(tmp, data, bytes, hipMemcpyHostToDevice, L->stream);
        if (e != hipSuccess) { set_error("hipMemcpyAsync(upload_pinned)", e); dev_free(tmp); image_delete(im); return IMP_ERROR_DEVICE; }
        Frames f{};
        f.src = (const uint8_t*)tmp; f.v = View{(const uint8_t*)tmp, width, height, channels, step};
        f.dst = im->d; f.dw = width; f.dh = height; f.dstep = im->step; f.count = 1;
        rc = launch_copy(f, L->stream);
        dev_free(tmp);                                         // (the pool hands it out again in lane-stream order)
        if (rc) { image_delete(im); return rc; }
    }
    *out = im;
    return IMP_OK;
}

int impgpu_image_download_pinned(const impgpu_image* im, unsigned char* data, int step) {
    if (!im || !data || step < im->w * im->c) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    if (step == im->step) {
        IMP_HIP(hipMemcpyAsync(data, im->d, (size_t)step * im->h, hipMemcpyDeviceToHost, L->stream));
        return IMP_OK;
    }
    void* tmp = nullptr;                                       // re-pitch on the device, then one linear DMA
    const size_t bytes = (size_t)step * (im->h - 1) + (size_t)im->w * im->c;
    if (int rc = dev_alloc(bytes, &tmp)) return rc;
    Frames f{};
    f.src = im->d; f.v = View{im->d, im->w, im->h, im->c, im->step};
    f.dst = (uint8_t*)tmp; f.dw = im->w; f.dh = im->h; f.dstep = step; f.count = 1;
    // (the bytes between the rows of the caller's buffer arrive as zeros, not as whatever the pool block held before)
    if (hipMemsetAsync(tmp, 0, bytes, L->stream) != hipSuccess) { set_error("hipMemsetAsync(download_pinned)", hipGetLastError()); dev_free(tmp); return IMP_ERROR_DEVICE; }
    int rc = launch_copy(f, L->stream);
    if (!rc) {
        const hipError_t e = hipMemcpyAsync(data, tmp, bytes, hipMemcpyDeviceToHost, L->stream);
        if (e != hipSuccess) { set_error("hipMemcpyAsync(download_pinned)", e); rc = IMP_ERROR_DEVICE; }
    }
    dev_free(tmp);
    return rc;
}

int impgpu_image_upload_fi32(const unsigned char* bits, int width, int height, int pitch, impgpu_image** out) {
    if (!bits || !out || pitch < width * 4) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    impgpu_image* im = nullptr;
    int rc = image_new(width, height, 4, &im);
    if (rc) return rc;
    const size_t bytes = (size_t)im->step * height;
    Staging* S = nullptr;
    rc = stage_reserve(L, bytes, &S);
    if (rc) { image_delete(im); return rc; }
    // LoadSingle (advancedio.c:310-318): FreeImage rows are bottom-up; the flip rides on the staging copy
    for (int y = 0; y < height; y++)
        std::memcpy(S->p + (size_t)(height - 1 - y) * im->step, bits + (size_t)y * pitch, (size_t)width * 4);
    hipError_t e = hipMemcpyAsync(im->d, S->p, bytes, hipMemcpyHostToDevice, L->stream);
    if (e == hipSuccess) e = hipEventRecord(S->done, L->stream);
    if (e != hipSuccess) { set_error("hipMemcpyAsync(upload_fi32)", e); (void)hipStreamSynchronize(L->stream); image_delete(im); return IMP_ERROR_DEVICE; }
    S->busy = true;
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
    Staging* S = nullptr;
    if (!rc) rc = stage_reserve(L, bytes, &S);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(S->p, tmp, bytes, hipMemcpyDeviceToHost, L->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(L->stream);
        if (e != hipSuccess) { set_error("download_fi", e); rc = IMP_ERROR_DEVICE; }
    }
    dev_free(tmp);
    if (rc) return rc;
    const size_t rowbytes = (size_t)im->w * (bpp / 8);
    for (int y = 0; y < im->h; y++) std::memcpy(bits + (size_t)y * pitch, S->p + (size_t)y * dpitch, rowbytes);
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
    TraceRange tr("IMP_STEP_ENCODE");                       // the encoder's hand-over (bridge.c:680-710)
    IMP_FAULT_POINT(IMP_STEP_ENCODE);
    const size_t bytes = (size_t)im->step * im->h;
    Staging* S = nullptr;
    int rc = stage_reserve(L, bytes, &S);
    if (rc) return rc;
    IMP_HIP(hipMemcpyAsync(S->p, im->d, bytes, hipMemcpyDeviceToHost, L->stream));
    if (int rcw = lane_wait()) return rcw;
    const size_t rowbytes = (size_t)im->w * im->c;
    for (int y = 0; y < im->h; y++)
        std::memcpy(data + (size_t)y * step, S->p + (size_t)y * im->step, rowbytes);
    return IMP_OK;
}

int impgpu_batch_download(const impgpu_image* const* images, int count, unsigned char* const* datas, const int* steps) {
    if (count < 0 || (count && (!images || !datas || !steps))) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    TraceRange tr("IMP_STEP_ENCODE");
    IMP_FAULT_POINT(IMP_STEP_ENCODE);
    size_t total = 0;
    int most_rows = 0;
    for (int i = 0; i < count; i++) {
        if (!images[i] || !datas[i] || steps[i] < images[i]->w * images[i]->c) return IMP_ERROR_INVALID_ARGS;
        total += ((size_t)images[i]->step * images[i]->h + 63) & ~size_t(63);
        most_rows = std::max(most_rows, images[i]->h);
    }
    if (!total) return IMP_OK;
    // One gather launch instead of a copy per frame (64 thumbnails were 64 hipMemcpyAsync calls: 1.3 ms of a request
    // thread's time).  Destinations the device can write -- pinned memory, impgpu_host_alloc -- get their rows straight
    // from the kernel, in the caller's layout: no DMA call, no staging pass.  Ordinary memory gets the frames gathered
    // into one block, ONE copy into pinned staging, and the rows copied out on the host.
    std::vector<GatherItem> items((size_t)count);
    bool direct = true;
    for (int i = 0; i < count && direct; i++) {
        void* dev = nullptr;
        direct = host_device_ptr(datas[i], &dev) &&
                 host_device_ptr(datas[i] + (size_t)(images[i]->h - 1) * steps[i] + (size_t)images[i]->w * images[i]->c - 1, nullptr);
        items[(size_t)i].dst = (uint8_t*)dev;
        items[(size_t)i].dstep = steps[i];
    }
    void* block = nullptr;
    Staging* S = nullptr;
    if (!direct) {
        if (int rc = stage_reserve(L, total, &S)) return rc;
        if (int rc = dev_alloc(total, &block)) return rc;
    }
    size_t at = 0;
    for (int i = 0; i < count; i++) {
        const impgpu_image* im = images[i];
        GatherItem& it = items[(size_t)i];
        it.src = im->d; it.rows = im->h; it.rowbytes = im->w * im->c; it.sstep = im->step;
        if (!direct) { it.dst = (uint8_t*)block + at; it.dstep = im->step; }
        at += ((size_t)im->step * im->h + 63) & ~size_t(63);
    }
    void* dev_items = nullptr;
    int rc = upload_small(items.data(), items.size() * sizeof(GatherItem), &dev_items, L->stream);
    if (!rc) {
        const int rpb = 16;
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)count, (unsigned)((most_rows + rpb - 1) / rpb)), dim3(256), 0, L->stream, (const GatherItem*)dev_items, rpb);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("k_gather_rows", e); rc = IMP_ERROR_DEVICE; }
        dev_free(dev_items);
    }
    if (!rc && !direct) {
        const hipError_t e = hipMemcpyAsync(S->p, block, total, hipMemcpyDeviceToHost, L->stream);
        if (e != hipSuccess) { set_error("hipMemcpyAsync(batch_download)", e); rc = IMP_ERROR_DEVICE; }
    }
    if (block) dev_free(block);
    if (rc) { (void)lane_wait(); return rc; }
    if (int rcw = lane_wait()) return rcw;
    if (direct) return IMP_OK;
    at = 0;
    for (int i = 0; i < count; i++) {
        const impgpu_image* im = images[i];
        const size_t rowbytes = (size_t)im->w * im->c;
        for (int y = 0; y < im->h; y++) std::memcpy(datas[i] + (size_t)y * steps[i], S->p + at + (size_t)y * im->step, rowbytes);
        at += ((size_t)im->step * im->h + 63) & ~size_t(63);
    }
    return IMP_OK;
}

int impgpu_album_upload(const unsigned char* const* datas, int count, int width, int height, int channels,
                        const int* steps, impgpu_image** out) {
    if (!datas || !out || count <= 0) return IMP_ERROR_INVALID_ARGS;
    for (int i = 0; i < count; i++)
        if (!datas[i] || (steps && (long long)steps[i] < (long long)width * channels)) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    TraceRange tr("IMP_STEP_DECODE");
    IMP_FAULT_POINT(IMP_STEP_DECODE);
    impgpu_image* im = nullptr;
    if (int rc = image_new_album(width, height, channels, count, &im)) return rc;
    const size_t fstride = count > 1 ? im->fstride : (size_t)im->step * height;
    const size_t rowbytes = (size_t)width * channels;
    // the whole album goes through one staging buffer and one copy when it fits the ring, else frame by frame
    const size_t total = fstride * count;
    const int per = total <= (size_t(64) << 20) ? count : 1;
    for (int at = 0; at < count; at += per) {
        Staging* S = nullptr;
        int rc = stage_reserve(L, fstride * per, &S);
        if (rc) { (void)hipStreamSynchronize(L->stream); image_delete(im); return rc; }
        for (int k = 0; k < per; k++) {
            const unsigned char* src = datas[at + k];
            const size_t sstep = steps ? (size_t)steps[at + k] : rowbytes;
            for (int y = 0; y < height; y++) std::memcpy(S->p + k * fstride + (size_t)y * im->step, src + (size_t)y * sstep, rowbytes);
        }
        const size_t bytes = fstride * (per - 1) + (size_t)im->step * height;
        hipError_t e = hipMemcpyAsync(im->d + (size_t)at * fstride, S->p, bytes, hipMemcpyHostToDevice, L->stream);
        if (e == hipSuccess) e = hipEventRecord(S->done, L->stream);
        if (e != hipSuccess) { set_error("hipMemcpyAsync(album upload)", e); (void)hipStreamSynchronize(L->stream); image_delete(im); return IMP_ERROR_DEVICE; }
        S->busy = true;
    }
    *out = im;
    return IMP_OK;
}

int impgpu_album_download(const impgpu_image* im, unsigned char* const* datas, const int* steps) {
    if (!im || !datas) return IMP_ERROR_INVALID_ARGS;
    const size_t rowbytes = (size_t)im->w * im->c;
    for (int i = 0; i < im->frames; i++)
        if (!datas[i] || (steps && (size_t)steps[i] < rowbytes)) return IMP_ERROR_INVALID_ARGS;
    Lane* L = lane();
    if (!L) return no_env();
    TraceRange tr("IMP_STEP_ENCODE");
    IMP_FAULT_POINT(IMP_STEP_ENCODE);
    const size_t frame = (size_t)im->step * im->h;
    const size_t fstride = im->frames > 1 ? im->fstride : frame;
    // one copy and ONE wait for the whole album when it fits the staging ring, else frame by frame
    const int per = fstride * im->frames <= (size_t(64) << 20) ? im->frames : 1;
    for (int at = 0; at < im->frames; at += per) {
        const size_t bytes = fstride * (per - 1) + frame;
        Staging* S = nullptr;
        if (int rc = stage_reserve(L, bytes, &S)) return rc;
        IMP_HIP(hipMemcpyAsync(S->p, im->d + (size_t)at * fstride, bytes, hipMemcpyDeviceToHost, L->stream));
        if (int rc = lane_wait()) return rc;
        for (int k = 0; k < per; k++) {
            const size_t dstep = steps ? (size_t)steps[at + k] : rowbytes;
            for (int y = 0; y < im->h; y++)
                std::memcpy(datas[at + k] + (size_t)y * dstep, S->p + k * fstride + (size_t)y * im->step, rowbytes);
        }
    }
    return IMP_OK;
}

int impgpu_album_count(const impgpu_image* im) { return im ? im->frames : 0; }
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
