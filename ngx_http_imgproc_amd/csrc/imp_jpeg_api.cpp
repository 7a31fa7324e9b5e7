// imp_jpeg_api.cpp -- impgpu_image_decode_jpeg / impgpu_batch_decode_jpeg: the reference's cvDecodeImage(&rawencoded, -1)
// for JPEG blobs (bridge.c:545-552) with everything but marker parsing and FF00 unstuffing on the device (imp_jpeg.h).
// One file or many, the device sees the same thing: a table of jobs, one entropy launch, one pixel launch per sampling
// class, one wait for all verdicts.
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "imp_jpeg_core.h"

using namespace imp;

namespace {

// Where the entropy stage of a launch runs.  The device's five launches cost about a quarter of a millisecond however small
// the file is; the calling thread's decoder takes 6 ns per byte.  One request at a time (tools/jpeg_tiny_probe.py,
// profiles/r04_jpeg_small_files.txt, frame complete, ms: device / calling thread / libjpeg-turbo on one core) -- 64x64 0.17 /
// 0.06 / 0.04, 160x120 0.21 / 0.09 / 0.09, 320x240 (22 KB) 0.24 / 0.17 / 0.23, 480x360 (50 KB) 0.26 / 0.31 / 0.45, 640x480
// (88 KB) 0.26 / 0.54 / 0.82, 1080p 0.37 / 3.4 / 5.4 -- they cross near 40 KB of entropy-coded data (round 3's stage, with its
// rounds: near 100 KB), so a launch smaller than that keeps its Huffman decoding on the thread that is about to sleep in the
// wait anyway (dequantisation, IDCT, upsampling and colour stay on the device either way), and everything larger -- every
// batch -- is the device's.  IMPGPU_JPEG_HUFF = device | host forces one (read per call: a getenv is nothing next to a decode).
bool entropy_on_device(size_t launch_bytes) {
    const char* s = std::getenv("IMPGPU_JPEG_HUFF");
    if (s && !std::strcmp(s, "host")) return false;
    if (s && !std::strcmp(s, "device")) return true;
    return launch_bytes >= (size_t(40) << 10);
}

// impgpu_jpeg_profile(1): every decode call leaves its stages' durations with the calling thread (impgpu_jpeg_stage_times) --
// the host's from its clock, the device's from events recorded between the kernels
std::atomic<int> g_profile{0};
// process-wide: files whose entropy stage ran on the device, of those refused by its verdict, of those with a chain wait
// that ran out (JPEG_ST_CHAIN_TIMEOUT: the file is then decoded by the caller's fallback -- a box that does this silently
// looks healthy and is not), files kept on the calling thread because their blocks are too long
std::atomic<unsigned long long> g_count[4 + JPEG_WHY_COUNT + 1];    // [4 + why]: files refused at their header, by reason; [4 + JPEG_WHY_COUNT]: damaged headers
thread_local double t_stage[16];

// IMPGPU_JPEG_TRACE=1: one line per call on stderr with the host's share of it, in microseconds
struct Stopwatch {
    bool on;
    std::chrono::steady_clock::time_point t0;
    double marks[8] = {};
    int n = 0;
    Stopwatch() : on(std::getenv("IMPGPU_JPEG_TRACE") != nullptr || g_profile.load(std::memory_order_relaxed)), t0(std::chrono::steady_clock::now()) {}
    void mark() {
        if (!on || n >= 8) return;
        const auto t = std::chrono::steady_clock::now();
        marks[n++] = std::chrono::duration<double, std::micro>(t - t0).count();
        t0 = t;
    }
};

constexpr int MAX_BATCH = 256;                      // verdicts of one launch fit the lane's pinned mailbox
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Prep {                                       // one file on its way to the device
    int code = IMP_OK;
    JpegHeader H;
    JpegFrame F;
    int dc_ids[2] = {-1, -1}, ac_ids[2] = {-1, -1};
    JpegScan scan;
    size_t nsegs = 0;
    size_t words_off = 0, words_cap = 0;            // bytes, in the staging buffer and in the device copy alike
    size_t coef_off = 0;                            // bytes in the coefficient area
    size_t side_tables = 0, side_qt = 0, side_meta = 0;   // byte offsets in the side blob
    size_t ctl_header = 0, ctl_records = 0, ctl_ext = 0;   // word offsets in the control area
    size_t work_off = 0;                            // bytes in the per-chunk work area (entry states, slot counts, DC sums)
    impgpu_image* im = nullptr;
    size_t scan_len = 0;                            // entropy-coded bytes: as the file has them, or unstuffed by the caller (prepared)
    const impgpu_jpeg_prepared* pre = nullptr;      // the caller has unstuffed the scan (impgpu_batch_decode_jpeg_prepared)
    bool given = false;                             // the frame has been handed to the caller ahead of its verdict (impgpu_batch_decode_jpeg_pending)
    bool direct = false;                            // ... into registered memory: its words go to the device from there
    size_t direct_off = 0;                          // bytes behind the staged part of the words' block
};

// A file whose blocks average more bits than this keeps its Huffman stage on the calling thread even inside a launch that is
// the device's: blocks that long seldom end inside a walk's overlap, so their chunks are reached one after the other by an
// explicit state (tools/jpeg_sync_probe.py: 4.5 % of the chunks at 210 bits per block, a quality-98 photograph; 25-70 % for
// white noise), and a 4 MB file of noise would hold a workgroup chain for most of a second where the host needs 25 ms.
constexpr size_t DENSE_BITS_PER_BLOCK = 200;
constexpr int CODE_DEFERRED = -0x7fff;              // internal: "decode this one in the host-entropy group"

// A group of files between the call that enqueued its decode and the call that reads its verdicts.  Everything the device
// works on (stream words, planes, tables, work areas) and the pinned words its verdicts arrive in belong to the group until
// group_finish; the caller's blobs must stay readable until then (a file deferred to the host-entropy group is read again).
constexpr int GROUP_SLOTS = 4;                      // groups in flight per calling thread: one slot of the lane's mailbox each
struct Group {
    std::vector<Prep> P;
    const unsigned char* const* blobs = nullptr;
    const size_t* sizes = nullptr;
    int count = 0;
    bool on_device = false, profile = false;
    bool one_block = false;                         // d_side and d_ctl lie inside d_words' block (one upload for all three)
    void *d_words = nullptr, *d_coef = nullptr, *d_side = nullptr, *d_ctl = nullptr, *d_work = nullptr;
    void* mark = nullptr;                           // the point of the group's stream where its last kernel was enqueued
    hipStream_t stream = nullptr;                   // the lane's stream, or its side stream (a batch begun ahead: its decode overlaps what the thread enqueues next)
    uint32_t* mailbox = nullptr;
    int slot = -1;
    size_t live = 0, ctl_total = 0;
    hipEvent_t ev[8] = {};
    Stopwatch sw;
    const void* owner = nullptr;                    // the thread that began the group (its slot, its lane): only it may finish it
    const impgpu_jpeg_prepared* prep = nullptr;     // per file, or nullptr: every blob is a whole file
};
thread_local char t_thread_tag;                     // its address names the calling thread
thread_local unsigned t_slots_busy = 0;             // bit k: mailbox slot k holds a group that has not been finished
std::atomic<int> g_groups_in_flight{0};             // over all threads: groups begun and not finished

// A prepared file as the file it came from (head, the scan with its 00s behind the FFs again, EOI): what the host's entropy
// decoder reads -- the A/B path and the files a device launch defers (dense blocks), never the common case.
std::vector<uint8_t> restuffed(const impgpu_jpeg_prepared& f) {
    std::vector<uint8_t> out;
    out.reserve(f.head_size + f.scan_size + f.scan_size / 64 + 16);
    out.insert(out.end(), f.head, f.head + f.head_size);
    for (size_t i = 0; i < f.scan_size; i++) {
        out.push_back(f.scan[i]);
        if (f.scan[i] == 0xFF) out.push_back(0);
    }
    out.push_back(0xFF);
    out.push_back(0xD9);
    return out;
}

void group_release(Group& G) {
    for (int i = 0; i < 8; i++) if (G.ev[i]) { (void)hipEventDestroy(G.ev[i]); G.ev[i] = nullptr; }
    void* blocks[5] = {G.d_words, G.d_coef, G.one_block ? nullptr : G.d_side, G.one_block ? nullptr : G.d_ctl, G.d_work};
    for (void* b : blocks)
        if (b) { if (G.stream && !on_lane_stream(G.stream)) dev_free_on(b, G.stream); else dev_free(b); }
    G.d_words = G.d_coef = G.d_side = G.d_ctl = G.d_work = nullptr;
    if (G.slot >= 0) { t_slots_busy &= ~(1u << G.slot); g_groups_in_flight.fetch_sub(1, std::memory_order_relaxed); }
    G.slot = -1;
}

// ---- a few helper threads for the host's per-file work of a BATCH (never started by single-file calls)
class HostPool {
public:
    // (never destroyed: its threads sleep on the condition variable for life, and destroying a condition variable that has
    // waiters blocks -- a static instance made every process that had run a batch hang in its exit handlers)
    static HostPool& get() { static HostPool* p = new HostPool(); return *p; }
    int helpers() const { return (int)threads_.size(); }
    // runs fn(items[k]) for every k, the caller taking part; returns when all are done
    template <class Fn>
    void run(const std::vector<int>& items, Fn& fn) {
        std::atomic<size_t> next{0}, done{0};
        const size_t n = items.size();
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1, std::memory_order_relaxed);
                if (k >= n) break;
                fn(items[k]);
                done.fetch_add(1, std::memory_order_release);
            }
        };
        std::function<void()> job = work;
        {
            std::lock_guard<std::mutex> lk(mu_);
            const int want = (int)std::min<size_t>(threads_.size(), n > 1 ? n - 1 : 0);
            for (int i = 0; i < want; i++) queue_.push_back(&job);
        }
        cv_.notify_all();
        work();
        // helpers that took the job but found nothing left have touched nothing of ours; those in the middle of an item are waited for
        while (done.load(std::memory_order_acquire) < n) std::this_thread::yield();
        std::unique_lock<std::mutex> lk(mu_);
        for (auto it = queue_.begin(); it != queue_.end();) it = (*it == &job) ? queue_.erase(it) : it + 1;
        idle_.wait(lk, [&] { return running_ == 0 || !uses(&job); });
    }
private:
    HostPool() {
        const char* s = std::getenv("IMPGPU_HOST_THREADS");
        unsigned hw = std::thread::hardware_concurrency();
        int n = s ? std::atoi(s) : (int)std::min(3u, hw / 8);              // helpers beside the caller
        if (n < 0) n = 0;
        if (n > 15) n = 15;
        for (int i = 0; i < n; i++) threads_.emplace_back([this] { loop(); });
        for (auto& t : threads_) t.detach();
    }
    bool uses(std::function<void()>* j) const { for (auto* c : current_) if (c == j) return true; return false; }
    void loop() {
        for (;;) {
            std::function<void()>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !queue_.empty(); });
                job = queue_.front();
                queue_.pop_front();
                current_.push_back(job);
                running_++;
            }
            (*job)();
            {
                std::lock_guard<std::mutex> lk(mu_);
                running_--;
                for (auto it = current_.begin(); it != current_.end(); ++it) if (*it == job) { current_.erase(it); break; }
            }
            idle_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, idle_;
    std::deque<std::function<void()>*> queue_;
    std::vector<std::function<void()>*> current_;
    std::vector<std::thread> threads_;
    int running_ = 0;
};

template <class Fn>
void host_parallel(const std::vector<int>& items, size_t bytes, Fn& fn) {
    if (items.size() >= 4 && bytes >= (size_t(256) << 10) && HostPool::get().helpers() > 0) HostPool::get().run(items, fn);
    else for (int i : items) fn(i);
}

// Everything up to the last enqueue: headers, the unstuffing copy (or the host's entropy decoding), job tables, uploads,
// the entropy and pixel kernels, the verdicts' copy.  Does NOT wait.  A non-zero return means nothing is in flight and
// nothing is held (codes[] of the caller are then filled by the caller from G.P where they are set, else with the return).
int group_begin(Group& G, const unsigned char* const* blobs, const size_t* sizes, int count, int force_host, bool side = false,
                const impgpu_jpeg_prepared* prep = nullptr) {
    Stopwatch& sw = G.sw;
    hipStream_t s = side ? lane_side_stream() : nullptr;
    if (!s) s = env_stream();
    if (!s) return IMP_ERROR_DEVICE;
    G.stream = s;
    G.blobs = blobs; G.sizes = sizes; G.count = count; G.prep = prep;
    G.P.assign((size_t)count, Prep());
    std::vector<Prep>& P = G.P;
    for (int k = 0; k < GROUP_SLOTS && G.slot < 0; k++)
        if (!(t_slots_busy & (1u << k))) G.slot = k;
    if (G.slot < 0) { set_error_text("too many JPEG batches begun and not finished on this thread"); return IMP_ERROR_INVALID_ARGS; }
    t_slots_busy |= 1u << G.slot;
    G.owner = &t_thread_tag;
    g_groups_in_flight.fetch_add(1, std::memory_order_relaxed);
    // ---- headers, geometry, the sizes of everything
    size_t words_total = 0, coef_total = 0, direct_total = 0;
    for (int i = 0; i < count; i++) {
        Prep& p = P[(size_t)i];
        if (!blobs[i]) { p.code = IMP_ERROR_INVALID_ARGS; continue; }
        if (prep && prep[i].scan) p.pre = &prep[i];
        p.code = jpeg_parse(blobs[i], sizes[i], &p.H);
        // (a prepared file: its head ends where its scan began, and nothing cuts the scan into intervals)
        if (!p.code && p.pre && (p.H.scan_begin != sizes[i] || p.H.restart_interval || !p.pre->scan_size)) {
            set_error_text("prepared JPEG: the head does not end at its scan, or the file has a restart interval");
            p.code = IMP_ERROR_INVALID_ARGS;
            continue;
        }
        if (!p.code) p.scan_len = p.pre ? p.pre->scan_size : sizes[i] - p.H.scan_begin;
        if (!p.code && !frame_fits(p.H.width, p.H.height, p.H.ncomp)) { p.code = IMP_ERROR_UNSUPPORTED; p.H.why = JPEG_WHY_OTHER; }
        if (p.code == IMP_ERROR_UNSUPPORTED) g_count[4 + (p.H.why > 0 && p.H.why < JPEG_WHY_COUNT ? p.H.why : JPEG_WHY_OTHER)].fetch_add(1, std::memory_order_relaxed);
        else if (p.code == IMP_ERROR_DECODE_FAILED) g_count[4 + JPEG_WHY_COUNT].fetch_add(1, std::memory_order_relaxed);
        if (!p.code) p.code = jpeg_frame_setup(p.H, &p.F, p.dc_ids, p.ac_ids);
        if (p.code) continue;
        const size_t total_mcus = (size_t)p.H.mcux * p.H.mcuy;
        p.nsegs = p.H.restart_interval ? (total_mcus + p.H.restart_interval - 1) / p.H.restart_interval : 1;
        // Refused before anything is sized by what the header claims: an interval needs a data byte and its marker, a block a DC
        // code and an end-of-block code -- a few hundred bytes that announce 30000 x 30000 pixels, or a restart interval of one
        // MCU on a frame of 2^30, would otherwise have pinned and device memory allocated by the gigabyte before the decode fails.
        const size_t scan_bytes = p.scan_len;
        if (p.nsegs > scan_bytes / 3 + 1 || (size_t)p.F.total_slots / 64 > 4 * scan_bytes + 64) { p.code = IMP_ERROR_DECODE_FAILED; continue; }
        p.words_cap = align_up(jpeg_scan_capacity(scan_bytes, p.nsegs), JPEG_CHUNK_BYTES_MAX);
        p.direct = p.pre && p.pre->registered && !force_host;
        if (p.direct) { p.direct_off = direct_total; direct_total += p.words_cap; }
        else { p.words_off = words_total; words_total += p.words_cap; }
        p.coef_off = coef_total;
        coef_total += align_up((size_t)p.F.total_slots * sizeof(int16_t), 256);
    }
    size_t launch_bytes = 0;
    for (int i = 0; i < count; i++)
        if (!P[(size_t)i].code) launch_bytes += P[(size_t)i].scan_len;
    const bool on_device = G.on_device = !force_host && entropy_on_device(launch_bytes);
    if (!on_device && direct_total) {                               // (a launch of under 40 KB: its files are staged after all)
        for (Prep& p : P) if (!p.code && p.direct) { p.direct = false; p.words_off = words_total; words_total += p.words_cap; }
        direct_total = 0;
    }
    if (on_device && !std::getenv("IMPGPU_JPEG_HUFF"))
        for (int i = 0; i < count; i++) {
            Prep& p = P[(size_t)i];
            if (!p.code && p.scan_len * 8 > DENSE_BITS_PER_BLOCK * ((size_t)p.F.total_slots / 64)) {
                p.code = CODE_DEFERRED;
                g_count[3].fetch_add(1, std::memory_order_relaxed);
            }
        }
    sw.mark();                                                      // [0] headers
    // ---- the compressed bytes: FF00 unstuffing while they are copied into pinned memory -- the only pass the host makes
    // over them (device entropy stage), or the whole entropy decoding into pinned coefficient planes (A/B path)
    void* host = nullptr;
    void* token = nullptr;
    int rc = stage_begin(on_device ? words_total : coef_total, &host, &token);
    if (rc) { group_release(G); return rc; }
    size_t& live = G.live;
    // The unstuffing copies of a batch are independent of each other (a file each, disjoint parts of the pinned buffer): from four
    // files and 256 KB on they are shared out to a few helper threads (host_pool) -- 31 us per 437 KB file on the calling thread
    // were 0.2 ms of a broker lane's 0.9-ms cycle at seven files a batch, and 2.0 of a one-thread stream's 5.8 ms per 64 files.
    // A lone request never gets here with more than one file, so an nginx worker that links the library never starts a thread.
    if (on_device) {
        std::vector<int> todo;
        for (int i = 0; i < count; i++) if (!P[(size_t)i].code) todo.push_back(i);
        const bool busy = g_groups_in_flight.load(std::memory_order_relaxed) > 1;    // (this group is counted already)
        auto prepare = [&](int i) {
            Prep& p = P[(size_t)i];
            p.scan.chunk_bytes = jpeg_chunk_bytes_for(p.scan_len, launch_bytes, busy);
            if (!p.pre) { p.code = jpeg_prepare_scan(blobs[i], sizes[i], p.H, (uint8_t*)host + p.words_off, p.words_cap, &p.scan); return; }
            // the caller's unstuffed bytes are one interval: what jpeg_prepare_scan would have left -- the bytes, 1-bits up to
            // the chunk boundary, one all-ones guard chunk (a registered scan brings them: IMPGPU_JPEG_SCAN_TAIL)
            const size_t CBY = p.scan.chunk_bytes, n = p.pre->scan_size, padded = (n + CBY - 1) / CBY * CBY;
            p.scan.seg_first_chunk.assign(1, 0u);
            p.scan.seg_bits.assign(1, (uint32_t)(n * 8));
            p.scan.nchunks = padded / CBY;
            if ((uint64_t)padded * 8 >= (1ull << 32)) { p.code = IMP_ERROR_UNSUPPORTED; return; }
            if (p.direct) return;
            uint8_t* out = (uint8_t*)host + p.words_off;
            std::memcpy(out, p.pre->scan, n);
            std::memset(out + n, 0xFF, padded - n + JPEG_CHUNK_BYTES);
        };
        host_parallel(todo, launch_bytes, prepare);
    }
    for (int i = 0; i < count; i++) {
        Prep& p = P[(size_t)i];
        if (p.code) continue;
        if (on_device) {
            if (!p.code) {
                p.F.nchunks = (unsigned)p.scan.nchunks;
                p.F.nsegs = (unsigned)p.scan.seg_first_chunk.size();
                p.F.chunk_bits = (unsigned)p.scan.chunk_bytes * 8;
                p.F.overlap_bits = jpeg_overlap_bits_for(p.F.chunk_bits, p.scan_len, (size_t)p.F.total_slots / 64);
                // k_jpeg_write decodes a chunk with two lanes, the second entering at the chunk's middle in the state the chunk's
                // true walk passed there: its chain is half as long and the synchronising phases are none the longer.  Where the
                // launch is a latency chain, that is (a lone 640 x 480: k_jpeg_write 65 -> 56 us, 4K 99 -> 89); a launch that fills
                // the device gains nothing from shorter chains and loses to the second round of workgroups its LDS forces
                // (64 files, 28 MB: 349 -> 415 us), so from 4 MB on -- where the chunks grow to 256 bytes -- a chunk keeps its
                // one lane.  IMPGPU_JPEG_SPLIT=0 | 1 forces either (A/B, read per call).
                { const char* sp = std::getenv("IMPGPU_JPEG_SPLIT"); p.F.wsplit = sp ? (sp[0] == '0' ? 1u : 2u) : launch_bytes > (size_t(4) << 20) ? 1u : 2u; }
            }
        } else {
            int16_t* planes = (int16_t*)((uint8_t*)host + p.coef_off);
            std::memset(planes, 0, (size_t)p.F.total_slots * sizeof(int16_t));
            if (p.pre) {
                const std::vector<uint8_t> file = restuffed(*p.pre);
                p.code = jpeg_host_entropy(file.data(), file.size(), p.H, planes, p.F);
            } else p.code = jpeg_host_entropy(blobs[i], sizes[i], p.H, planes, p.F);
        }
        if (!p.code) p.code = image_new(p.H.width, p.H.height, p.H.ncomp, &p.im);
        if (!p.code) live++;
    }
    sw.mark();                                                      // [1] unstuffing copy / host entropy decoding
    void *&d_words = G.d_words, *&d_coef = G.d_coef, *&d_side = G.d_side, *&d_ctl = G.d_ctl, *&d_work = G.d_work;
    bool& profile = G.profile;
    profile = g_profile.load(std::memory_order_relaxed) != 0;
    hipEvent_t* ev = G.ev;
    size_t njobs = 0;
    size_t& ctl_total = G.ctl_total;
    uint32_t* mailbox = G.mailbox = lane_mailbox() ? lane_mailbox() + 1024 * (1 + G.slot) : nullptr;      // (slot 0 of the mailbox is the other callers': brightness, the encoder)
    if (live == 0) { (void)stage_upload(token, nullptr, 0); return IMP_OK; }
    if (!mailbox) { rc = IMP_ERROR_DEVICE; goto fail; }
    {
        // ---- the side blob: per job its tables, quantisers and interval arrays, then the job table and the workgroup maps
        size_t side = 0, ctl_words = 4;                             // control: [0] ticket, then the jobs' headers, then their records
        size_t total_blocks = 0, sync_blocks = 0, work = 0;
        size_t tiles[5] = {0, 0, 0, 0, 0};                          // per sampling class
        auto klass = [](const JpegFrame& F) { return F.ncomp == 1 ? 0 : F.hs == 1 ? (F.vs == 1 ? 1 : 3) : (F.vs == 1 ? 2 : 4); };
        for (Prep& p : P) {
            if (p.code) continue;
            p.ctl_header = ctl_words;
            ctl_words += 4;
        }
        for (Prep& p : P) {
            if (p.code) continue;
            njobs++;
            if (on_device) {
                p.side_tables = side;
                side += align_up(4 * sizeof(JpegHuffDev), 64);
                p.side_meta = side;
                side += align_up((p.scan.nchunks + 2 * p.scan.seg_first_chunk.size()) * sizeof(uint32_t), 64);
                p.ctl_records = ctl_words;
                ctl_words += (size_t)jpeg_sync_blocks(p.F.nchunks, p.F.bpm) * JPEG_CTL_REC;
                p.ctl_ext = ctl_words;
                ctl_words += 4;
                sync_blocks += jpeg_sync_blocks(p.F.nchunks, p.F.bpm);
                total_blocks += jpeg_entropy_blocks(p.F.nchunks * p.F.wsplit);
                // per chunk: entry and middle state (8 bytes each), slots, slots up to the middle; per UNIT of k_jpeg_write
                // (wsplit per chunk): first slot, DC sums [4]; per workgroup of k_jpeg_write: DC sums [4];
                // per chunk and block of the MCU: what its walk and its repair walk found (64 bytes)
                p.work_off = work;
                work += align_up((size_t)p.F.nchunks * (24 + 20 * (size_t)p.F.wsplit + 64 * (size_t)p.F.bpm) + 8 + (size_t)jpeg_entropy_blocks(p.F.nchunks * p.F.wsplit) * 32 + sizeof(JpegHuffTabs) + 64 +
                                 ((size_t)p.F.nchunks / 8 + 16) * JPEG_EXT_WORDS * 4 + align_up((size_t)p.F.total_slots / 64 * 2, 64) + 64, 256);
            }
            p.side_qt = side;
            side += align_up(3 * 64 * sizeof(uint16_t), 64);
            tiles[klass(p.F)] += (size_t)((p.F.width + JPEG_TILE_W - 1) / JPEG_TILE_W) * ((p.F.height + JPEG_TILE_H - 1) / JPEG_TILE_H);
        }
        const size_t side_jobs = side;
        side += align_up(njobs * sizeof(JpegJob), 64);
        const size_t side_blocks = side;
        side += align_up(total_blocks * sizeof(JpegMapEntry), 64);
        const size_t side_sync = side;
        side += align_up(sync_blocks * sizeof(JpegMapEntry), 64);
        size_t side_tiles[5];
        for (int k = 0; k < 5; k++) { side_tiles[k] = side; side += align_up(tiles[k] * sizeof(JpegMapEntry), 64); }
        // The scan words, the side blob and the (zeroed) control area cross the link as ONE copy when the pinned buffer the
        // words were unstuffed into has room behind them (it is at least 8 MB): three commands on the stream -- two copies
        // and a fill -- were 30-40 us of a lone file's 220 (tools/jpeg_stage_probe.py), the bytes themselves 3.
        const size_t off_side = align_up(words_total, 256), off_ctl = off_side + align_up(side, 256), in_total = off_ctl + ctl_words * sizeof(uint32_t);
        const bool one = G.one_block = on_device && stage_capacity(token) >= in_total;
        // (the scans a caller unstuffed into registered memory lie behind what is staged: each arrives by a copy of its own)
        const size_t direct_base = align_up(one ? in_total : words_total, 256);
        if (one) {
            rc = dev_alloc(direct_base + direct_total, &d_words);
            if (!rc) { d_side = (uint8_t*)d_words + off_side; d_ctl = (uint8_t*)d_words + off_ctl; }
        } else {
            rc = dev_alloc(side, &d_side);
            if (!rc && on_device) rc = dev_alloc(direct_base + direct_total, &d_words);
            if (!rc && on_device) rc = dev_alloc(ctl_words * sizeof(uint32_t), &d_ctl);
        }
        if (!rc) rc = dev_alloc(coef_total, &d_coef);
        if (!rc && on_device) rc = dev_alloc(work, &d_work);
        ctl_total = ctl_words;
        if (!rc) rc = stream_join(s);                               // (side stream: the pool recycles in lane-stream order; the frames were allocated there too)
        if (rc) goto fail;
        std::vector<uint8_t> blob_mem(one ? 0 : side);
        struct { uint8_t* p; uint8_t* data() const { return p; } } blob{one ? (uint8_t*)host + off_side : blob_mem.data()};
        if (one) { std::memset((uint8_t*)host + words_total, 0, off_side - words_total); std::memset((uint8_t*)host + off_side, 0, in_total - off_side); }
        uint32_t* verdict_dev = nullptr;                            // the device's view of the group's pinned verdict words
        if (on_device && hipHostGetDevicePointer((void**)&verdict_dev, mailbox, 0) != hipSuccess) { verdict_dev = nullptr; (void)hipGetLastError(); }
        JpegJob* jobs = (JpegJob*)(blob.data() + side_jobs);
        JpegMapEntry* bmap = (JpegMapEntry*)(blob.data() + side_blocks);
        JpegMapEntry* smap = (JpegMapEntry*)(blob.data() + side_sync);
        JpegMapEntry* tmap[5];
        for (int k = 0; k < 5; k++) tmap[k] = (JpegMapEntry*)(blob.data() + side_tiles[k]);
        size_t j = 0, nb = 0, ns = 0, nt[5] = {0, 0, 0, 0, 0};
        std::vector<uint32_t> meta;
        for (Prep& p : P) {
            if (p.code) continue;
            JpegJob& J = jobs[j];
            std::memset(&J, 0, sizeof J);
            J.F = p.F;
            if (on_device) {
                rc = jpeg_build_tables(p.H, p.dc_ids, p.ac_ids, (JpegHuffDev*)(blob.data() + p.side_tables));
                if (rc) { p.code = rc; rc = IMP_OK; }               // (cannot happen after jpeg_parse; keeps the job inert)
                jpeg_scan_meta(p.scan, &meta);
                std::memcpy(blob.data() + p.side_meta, meta.data(), meta.size() * sizeof(uint32_t));
                J.words = (const uint32_t*)((uint8_t*)d_words + (p.direct ? direct_base + p.direct_off : p.words_off));
                J.chunk_seg = (const uint32_t*)((uint8_t*)d_side + p.side_meta);
                J.seg_first_chunk = J.chunk_seg + p.F.nchunks;
                J.seg_bits = J.seg_first_chunk + p.F.nsegs;
                J.tables = (const JpegHuffDev*)((uint8_t*)d_side + p.side_tables);
                J.header = (uint32_t*)d_ctl + p.ctl_header;
                J.records = (uint32_t*)d_ctl + p.ctl_records;
                uint8_t* wk = (uint8_t*)d_work + p.work_off;
                const size_t NC = p.F.nchunks, NU = NC * p.F.wsplit, NB = NC * (size_t)p.F.bpm;
                J.chunk_entry = (uint64_t*)wk;
                J.chunk_mid = (uint64_t*)(wk + NC * 8);
                J.chunk_n = (uint32_t*)(wk + NC * 16);
                J.chunk_nmid = (uint32_t*)(wk + NC * 20);
                J.chunk_slot0 = (uint32_t*)(wk + NC * 24);
                J.chunk_dc = (int*)(wk + NC * 24 + NU * 4);
                J.wg_dc = (int*)(wk + NC * 24 + NU * 20);
                uint8_t* cd = wk + align_up(NC * 24 + NU * 20, 8) + (size_t)jpeg_entropy_blocks((unsigned)NU) * 32;
                J.cand_in = (uint64_t*)cd;
                J.cand_out = (uint64_t*)(cd + NB * 8);
                J.rep_out = (uint64_t*)(cd + NB * 16);
                J.cand_mid = (uint64_t*)(cd + NB * 24);
                J.rep_mid = (uint64_t*)(cd + NB * 32);
                J.cand_n = (uint32_t*)(cd + NB * 40);
                J.rep_n = (uint32_t*)(cd + NB * 44);
                J.cand_nmid = (uint32_t*)(cd + NB * 48);
                J.rep_nmid = (uint32_t*)(cd + NB * 52);
                J.cand_nib = (uint8_t*)(cd + NB * 56);                                          // (a byte each; the four-byte arrays go on at 60)
                J.ext_idx = (uint32_t*)(cd + NB * 60);
                J.tabs = (JpegHuffTabs*)(cd + NB * 64);
                J.ext = (uint32_t*)((uint8_t*)J.tabs + align_up(sizeof(JpegHuffTabs), 64));
                J.ext_cap = (uint32_t)(p.F.nchunks / 8 + 16);
                J.ext_count = (uint32_t*)d_ctl + p.ctl_ext;
                J.dcadd = (int16_t*)((uint8_t*)J.ext + align_up((size_t)J.ext_cap * JPEG_EXT_WORDS * 4, 64));
                for (unsigned b = 0; b < jpeg_entropy_blocks(p.F.nchunks * p.F.wsplit); b++) bmap[nb++] = JpegMapEntry{(uint32_t)j, b};
                for (unsigned b = 0; b < jpeg_sync_blocks(p.F.nchunks, p.F.bpm); b++) smap[ns++] = JpegMapEntry{(uint32_t)j, b};
            }
            uint16_t* qt3 = (uint16_t*)(blob.data() + p.side_qt);
            for (int c = 0; c < p.H.ncomp; c++) std::memcpy(qt3 + 64 * c, p.H.qt[p.H.comp[c].tq], 64 * sizeof(uint16_t));
            J.qt = (const uint16_t*)((uint8_t*)d_side + p.side_qt);
            J.coef = (int16_t*)((uint8_t*)d_coef + p.coef_off);
            J.dst = p.im->d;
            J.dstep = p.im->step;
            J.verdict = (on_device && verdict_dev) ? verdict_dev + 4 * j : nullptr;
            const int k = klass(p.F);
            const uint32_t ntiles = (uint32_t)(((p.F.width + JPEG_TILE_W - 1) / JPEG_TILE_W) * ((p.F.height + JPEG_TILE_H - 1) / JPEG_TILE_H));
            for (uint32_t tl = 0; tl < ntiles; tl++) tmap[k][nt[k]++] = JpegMapEntry{(uint32_t)j, tl};
            j++;
        }
        sw.mark();                                                  // [2] tables + job table
        if (profile && on_device) {
            for (int i = 0; i < 8; i++) if (hipEventCreate(&ev[i]) != hipSuccess) { ev[i] = nullptr; profile = false; }
            if (profile) (void)hipEventRecord(ev[0], s);
        }
        rc = stage_upload(token, on_device ? d_words : d_coef, one ? in_total : on_device ? words_total : coef_total);
        token = nullptr;
        if (!rc && !one) rc = upload_to(d_side, blob.data(), side, s);      // (both copies ride the lane's stream; `s` is made to wait for them here)
        if (!rc && one) rc = stream_join(s);
        for (const Prep& p : P) {
            if (rc || p.code || !p.direct) continue;
            const hipError_t e = hipMemcpyAsync((uint8_t*)d_words + direct_base + p.direct_off, p.pre->scan, align_up(p.pre->scan_size, JPEG_CHUNK_BYTES_MAX) + JPEG_CHUNK_BYTES_MAX,
                                                hipMemcpyHostToDevice, s);
            if (e != hipSuccess) { set_error("hipMemcpyAsync(jpeg scan)", e); rc = IMP_ERROR_DEVICE; }
        }
        if (rc) goto fail;
        if (on_device) {
            // (the coefficient planes are not cleared: k_jpeg_write stores whole blocks)
            if (!one && hipMemsetAsync(d_ctl, 0, ctl_words * sizeof(uint32_t), s) != hipSuccess) {
                set_error("hipMemsetAsync(jpeg)", hipGetLastError());
                rc = IMP_ERROR_DEVICE;
                goto fail;
            }
            if (profile) (void)hipEventRecord(ev[1], s);
            // a launch too small to fill the device with any of its kernels runs walks, mend and select as ONE launch
            // (k_jpeg_entropy_small; measured, one file at a time, fused / five kernels: 640 x 480 0.221 / 0.235 ms, 720p equal,
            // 1080p 0.323 / 0.313, 4K 0.546 / 0.510 -- from a few hundred KB on every phase is bound by what it computes, not by
            // its launch, and a workgroup that waits inside a kernel holds slots a kernel boundary would have given to others).
            // IMPGPU_JPEG_FUSED=0 keeps the five kernels, =1 fuses up to 4 MB (A/B, read per call).
            const char* fz = std::getenv("IMPGPU_JPEG_FUSED");
            const size_t fuse_up_to = fz && fz[0] == '1' ? (size_t(4) << 20) : (size_t(192) << 10);
            const bool small = launch_bytes <= fuse_up_to && !(fz && fz[0] == '0');
            rc = launch_jpeg_entropy((const JpegJob*)((uint8_t*)d_side + side_jobs), (const JpegMapEntry*)((uint8_t*)d_side + side_sync), (unsigned)sync_blocks,
                                     (const JpegMapEntry*)((uint8_t*)d_side + side_blocks), (unsigned)total_blocks, (uint32_t*)d_ctl, s, profile ? ev + 2 : nullptr, small);
            if (rc) goto fail;
            // the kernel's verdicts (did every interval decode to exactly its MCUs?) are read before a frame is handed on:
            // k_jpeg_dcfix has written them into the pinned words; copied only where the device cannot address those
            if (!verdict_dev) {
                const hipError_t e = hipMemcpyAsync(mailbox, (uint32_t*)d_ctl + 4, njobs * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
                if (e != hipSuccess) { set_error("hipMemcpyAsync(jpeg verdicts)", e); rc = IMP_ERROR_DEVICE; goto fail; }
            }
        }
        static const int KH[5] = {1, 1, 2, 1, 2}, KV[5] = {1, 1, 1, 2, 2}, KN[5] = {1, 3, 3, 3, 3};
        for (int k = 0; k < 5 && !rc; k++)
            rc = launch_jpeg_pixels(KH[k], KV[k], KN[k], (const JpegJob*)((uint8_t*)d_side + side_jobs),
                                    (const JpegMapEntry*)((uint8_t*)d_side + side_tiles[k]), (unsigned)tiles[k], s);
        if (rc) goto fail;
        if (profile && on_device) (void)hipEventRecord(ev[7], s);
    }
    sw.mark();                                                      // [3] enqueue
    rc = lane_mark_on(s, &G.mark);
    if (rc) goto fail;
    return IMP_OK;
fail:
    if (token) (void)stage_upload(token, nullptr, 0);
    (void)hipStreamSynchronize(s);
    (void)lane_wait();                                              // nothing of this call may still be running when its buffers go back
    for (Prep& p : P) if (p.im) { image_delete(p.im); p.im = nullptr; }
    group_release(G);
    return rc;
}

// The other half: waits for the group's last kernel (not for whatever the thread enqueued after it), reads the verdicts,
// hands the frames out, gives the device memory back.
int group_finish(Group& G, impgpu_image** images, int* codes) {
    std::vector<Prep>& P = G.P;
    Stopwatch& sw = G.sw;
    const bool on_device = G.on_device;
    const int count = G.count;
    const size_t live = G.live, ctl_total = G.ctl_total;
    void *d_ctl = G.d_ctl, *d_work = G.d_work;
    uint32_t* mailbox = G.mailbox;
    hipEvent_t* ev = G.ev;
    const bool profile = G.profile;
    int rc = IMP_OK;
    if (live == 0) goto done;
    rc = lane_wait_mark(G.mark);
    G.mark = nullptr;
    if (rc) goto fail;
    if (on_device) {
        if (const char* tr = std::getenv("IMPGPU_JPEG_TRACE"); tr && !std::strcmp(tr, "2")) {
            // the workgroups' clocks at their phase boundaries (the kernels' stamp()), microseconds since the launch's
            // first workgroup started: wg: start | candidates loaded | twins | maps | scan + look-back | done (incl. the check of a guess), then
            // the rounds of picking and the chunks the workgroup chased
            std::vector<uint32_t> ctl(ctl_total);
            if (hipMemcpy(ctl.data(), d_ctl, ctl_total * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess) {
                uint32_t t0 = 0;
                bool have = false;
                for (const Prep& p : P)
                    if (!p.code)
                        for (unsigned b = 0; b < jpeg_sync_blocks(p.F.nchunks, p.F.bpm); b++) {
                            const uint32_t v = ctl[p.ctl_records + (size_t)b * JPEG_CTL_REC + 20];
                            if (!have || (int32_t)(v - t0) < 0) { t0 = v; have = true; }
                        }
                for (const Prep& p : P) {                          // k_jpeg_write: every workgroup's start and end, and its walk's
                    if (p.code) continue;
                    const unsigned nb2 = jpeg_entropy_blocks(p.F.nchunks * p.F.wsplit);
                    std::vector<int> wg((size_t)nb2 * 8);
                    if (hipMemcpy(wg.data(), (uint8_t*)d_work + p.work_off + (size_t)p.F.nchunks * (24 + 20 * (size_t)p.F.wsplit), wg.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) continue;
                    for (unsigned b = 0; b < nb2; b++)
                        std::fprintf(stderr, "ww %dx%d %u/%u: %.1f %.1f walk %.1f %.1f\n", p.H.width, p.H.height, b, nb2, (double)(int32_t)((uint32_t)wg[8 * b + 4] - t0) / 100.0,
                                     (double)(int32_t)((uint32_t)wg[8 * b + 5] - t0) / 100.0, (double)(int32_t)((uint32_t)wg[8 * b + 6] - t0) / 100.0, (double)(int32_t)((uint32_t)wg[8 * b + 7] - t0) / 100.0);
                }
                for (const Prep& p : P) {
                    if (p.code) continue;
                    const unsigned nb = jpeg_sync_blocks(p.F.nchunks, p.F.bpm);
                    for (unsigned b = 0; b < nb; b++) {
                        const uint32_t* r = &ctl[p.ctl_records + (size_t)b * JPEG_CTL_REC + 20];
                        std::fprintf(stderr, "wg %dx%d %u/%u:", p.H.width, p.H.height, b, nb);
                        for (int k = 0; k <= 5; k++) std::fprintf(stderr, " %.1f", r[k] ? (double)(int32_t)(r[k] - t0) / 100.0 : -1.0);
                        std::fprintf(stderr, " rounds %u chased %u", r[9], r[10]);
                        if (r[6]) std::fprintf(stderr, " | fused: start %.1f walks %.1f mend %.1f", (double)(int32_t)(r[6] - t0) / 100.0, (double)(int32_t)(r[7] - t0) / 100.0, (double)(int32_t)(r[8] - t0) / 100.0);
                        std::fprintf(stderr, "\n");
                    }
                }
            }
        }
        if (const char* tr = std::getenv("IMPGPU_JPEG_TRACE"); tr && !std::strcmp(tr, "3")) {
            // every chunk's entry state and slot count as k_jpeg_select left them ("ce <chunk> <state> <slots>": the host model prints the same)
            for (const Prep& p : P) {
                if (p.code) continue;
                std::vector<uint64_t> ent(p.F.nchunks);
                std::vector<uint32_t> cn(p.F.nchunks);
                const uint8_t* wk = (const uint8_t*)d_work + p.work_off;
                if (hipMemcpy(ent.data(), wk, ent.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) continue;
                if (hipMemcpy(cn.data(), wk + (size_t)p.F.nchunks * 16, cn.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) continue;
                for (size_t g = 0; g < ent.size(); g++) std::fprintf(stderr, "ce %zu %016llx %u\n", g, (unsigned long long)ent[g], cn[g]);
            }
        }
        size_t j = 0;
        for (Prep& p : P) {
            if (p.code) continue;
            const uint32_t status = mailbox[4 * j + 1];
            if (sw.on && std::getenv("IMPGPU_JPEG_TRACE")) std::fprintf(stderr, "jpeg %dx%d: %u chunks of %u bits (overlap %u), %u repair walks, %u chunks chased, %u walks in k_jpeg_select, status %u\n", p.H.width, p.H.height, p.F.nchunks, p.F.chunk_bits, p.F.overlap_bits, mailbox[4 * j + 2], mailbox[4 * j + 3], mailbox[4 * j + 0], status);
            g_count[0].fetch_add(1, std::memory_order_relaxed);
            if (status) g_count[1].fetch_add(1, std::memory_order_relaxed);
            if (status & JPEG_ST_CHAIN_TIMEOUT) g_count[2].fetch_add(1, std::memory_order_relaxed);
            if (status) {
                char text[96];
                std::snprintf(text, sizeof text, "jpeg entropy stage refused the scan (status 0x%x)", status);
                set_error_text(text);
                p.code = IMP_ERROR_DECODE_FAILED;
            }
            j++;
        }
    }
    sw.mark();                                                      // [4] wait for the verdicts
    if (g_profile.load(std::memory_order_relaxed)) {
        for (int i = 0; i < 16; i++) t_stage[i] = 0;
        for (int i = 0; i < 5; i++) t_stage[i] = sw.marks[i];
        if (profile && on_device && hipEventSynchronize(ev[7]) == hipSuccess)
            for (int i = 0; i < 7; i++) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) == hipSuccess) t_stage[5 + i] = 1e3 * ms;
            }
        t_stage[12] = (double)live;
    }
    if (sw.on && std::getenv("IMPGPU_JPEG_TRACE"))
        std::fprintf(stderr, "jpeg x%d (%zu live): headers %.0f %s %.0f jobs %.0f enqueue %.0f wait %.0f us\n", count, live, sw.marks[0],
                     on_device ? "unstuff" : "host-entropy", sw.marks[1], sw.marks[2], sw.marks[3], sw.marks[4]);
done:
    for (int i = 0; i < count; i++) {
        Prep& p = P[(size_t)i];
        if (p.given) { images[i] = nullptr; codes[i] = p.code; continue; }      // (the caller has the frame: the code says whether it holds the file's pixels)
        if (p.code && p.im) { image_delete(p.im); p.im = nullptr; }
        images[i] = p.im;
        codes[i] = p.code;
    }
    group_release(G);
    return IMP_OK;
fail:
    if (G.stream) (void)hipStreamSynchronize(G.stream);
    (void)lane_wait();                                              // nothing of this group may still be running when its buffers go back
    for (int i = 0; i < count; i++) {
        if (P[(size_t)i].im && !P[(size_t)i].given) image_delete(P[(size_t)i].im);
        images[i] = nullptr;
        codes[i] = P[(size_t)i].code ? P[(size_t)i].code : rc;
    }
    group_release(G);
    return rc;
}

// files a device group deferred (dense blocks: CODE_DEFERRED) are decoded in a host-entropy group of their own
int group_deferred(const unsigned char* const* blobs, const size_t* sizes, int count, impgpu_image** images, int* codes, const impgpu_jpeg_prepared* prep = nullptr) {
    std::vector<int> late;
    for (int i = 0; i < count; i++) if (codes[i] == CODE_DEFERRED) late.push_back(i);
    if (late.empty()) return IMP_OK;
    std::vector<const unsigned char*> b2(late.size());
    std::vector<size_t> s2(late.size());
    std::vector<impgpu_image*> i2(late.size(), nullptr);
    std::vector<int> c2(late.size(), IMP_OK);
    std::vector<impgpu_jpeg_prepared> p2;
    for (size_t k = 0; k < late.size(); k++) { b2[k] = blobs[late[k]]; s2[k] = sizes[late[k]]; if (prep) p2.push_back(prep[late[k]]); }
    Group H;
    int rc = group_begin(H, b2.data(), s2.data(), (int)late.size(), 1, false, prep ? p2.data() : nullptr);
    if (!rc) rc = group_finish(H, i2.data(), c2.data());
    else for (size_t k = 0; k < late.size(); k++) c2[k] = H.P[k].code ? H.P[k].code : rc;
    for (size_t k = 0; k < late.size(); k++) { images[late[k]] = rc ? nullptr : i2[k]; codes[late[k]] = rc ? rc : c2[k]; }
    if (rc) for (int i = 0; i < count; i++) if (images[i]) { impgpu_image_release(&images[i]); codes[i] = rc; }
    return rc;
}

// One group, begun and finished in one go -- except that a large group is cut in two and the second half is PREPARED (headers,
// unstuffing copy, job table: 2.3 of a 64-file call's 5.8 ms) while the device already works on the first: one calling thread
// went 11 -> 15 k requests/s with that.  When other threads keep the device busy anyway (four or more groups in flight) the
// group stays whole: two launches of half the size are the less efficient way to fill a device that is already full.
int decode_group(const unsigned char* const* blobs, const size_t* sizes, int count, impgpu_image** images, int* codes) {
    const bool whole = std::getenv("IMPGPU_JPEG_WHOLE") != nullptr;      // measurements of ONE launch per batch (tools/jpeg_prof_r04.sh, bench.py's stage profile); read per call
    // (the two halves take a mailbox slot each: with batches begun and not finished on this thread -- impgpu_batch_decode_jpeg_begin --
    // holding three of the four, the group stays whole instead of failing for want of a second slot)
    const int free_slots = GROUP_SLOTS - __builtin_popcount(t_slots_busy & ((1u << GROUP_SLOTS) - 1));
    const int first = !whole && count >= 32 && free_slots >= 2 && g_groups_in_flight.load(std::memory_order_relaxed) < 4 ? count / 2 : count;
    Group A, B;
    int rc = group_begin(A, blobs, sizes, first, 0);
    if (rc) {
        for (int i = 0; i < count; i++) { images[i] = nullptr; codes[i] = i < first && A.P.size() > (size_t)i && A.P[(size_t)i].code ? A.P[(size_t)i].code : rc; }
        return rc;
    }
    int rcb = IMP_OK;
    // Both halves on the lane's stream (the second half on the lane's side stream was measured on one box, two repetitions: the
    // same at 1, 2 and 8 threads, 7 % slower at 4 -- so it is not used here; impgpu_batch_decode_jpeg_begin does use it).
    if (first < count) rcb = group_begin(B, blobs + first, sizes + first, count - first, 0);
    rc = group_finish(A, images, codes);
    if (first < count) {
        if (!rcb) rcb = group_finish(B, images + first, codes + first);
        else for (int i = first; i < count; i++) { images[i] = nullptr; codes[i] = rcb; }
        if (!rc) rc = rcb;
    }
    if (rc) {
        for (int i = 0; i < count; i++) if (images[i]) { impgpu_image_release(&images[i]); codes[i] = rc; }
        return rc;
    }
    return group_deferred(blobs, sizes, count, images, codes);
}

}  // namespace

struct impgpu_jpeg_batch {
    Group G;
    std::vector<impgpu_jpeg_prepared> files;        // impgpu_batch_decode_jpeg_prepared_begin: its own copy of the caller's array
    std::vector<const unsigned char*> blobs;
    std::vector<size_t> sizes;
};

extern "C" {

int impgpu_batch_decode_jpeg(const unsigned char* const* blobs, const size_t* sizes, int count, impgpu_image** images, int* codes) {
    if (count < 0 || (count && (!blobs || !sizes || !images || !codes))) return IMP_ERROR_INVALID_ARGS;
    if (!env_ready()) { set_error_text("impgpu_env_start has not been called"); return IMP_ERROR_DEVICE; }
    TraceRange tr("IMP_STEP_DECODE");
    IMP_FAULT_POINT(IMP_STEP_DECODE);
    for (int at = 0; at < count; at += MAX_BATCH) {
        const int n = count - at < MAX_BATCH ? count - at : MAX_BATCH;
        if (int rc = decode_group(blobs + at, sizes + at, n, images + at, codes + at)) {
            for (int i = 0; i < at; i++) impgpu_image_release(&images[i]);
            for (int i = at + n; i < count; i++) { images[i] = nullptr; codes[i] = rc; }
            return rc;
        }
    }
    return IMP_OK;
}

int impgpu_batch_decode_jpeg_begin(const unsigned char* const* blobs, const size_t* sizes, int count, impgpu_jpeg_batch** batch) {
    if (!batch) return IMP_ERROR_INVALID_ARGS;
    *batch = nullptr;
    if (count <= 0 || count > MAX_BATCH || !blobs || !sizes) return IMP_ERROR_INVALID_ARGS;
    if (!env_ready()) { set_error_text("impgpu_env_start has not been called"); return IMP_ERROR_DEVICE; }
    TraceRange tr("IMP_STEP_DECODE");
    IMP_FAULT_POINT(IMP_STEP_DECODE);
    impgpu_jpeg_batch* b = new impgpu_jpeg_batch();
    // on the lane's side stream: what the thread enqueues before _finish overlaps it (IMPGPU_JPEG_AHEAD_STREAM=lane: behind it on the lane's own)
    const char* where = std::getenv("IMPGPU_JPEG_AHEAD_STREAM");
    const int rc = group_begin(b->G, blobs, sizes, count, 0, !(where && !std::strcmp(where, "lane")));
    if (rc) { delete b; return rc; }
    *batch = b;
    return IMP_OK;
}

int impgpu_batch_decode_jpeg_finish(impgpu_jpeg_batch** batch, impgpu_image** images, int* codes) {
    if (!batch || !*batch || !images || !codes) return IMP_ERROR_INVALID_ARGS;
    impgpu_jpeg_batch* b = *batch;
    // the slot, the mark event and the pool blocks belong to the beginning thread's lane: another thread would clear its own
    // slot bits and hand the event to its own lane.  Refused; the batch stays valid for its owner.
    if (b->G.owner != &t_thread_tag) { set_error_text("impgpu_batch_decode_jpeg_finish from another thread than _begin"); return IMP_ERROR_INVALID_ARGS; }
    *batch = nullptr;
    const unsigned char* const* blobs = b->G.blobs;
    const size_t* sizes = b->G.sizes;
    const int count = b->G.count;
    int rc = group_finish(b->G, images, codes);
    if (!rc) rc = group_deferred(blobs, sizes, count, images, codes, b->G.prep);
    delete b;
    return rc;
}

int impgpu_batch_decode_jpeg_pending(impgpu_jpeg_batch* batch, impgpu_image** images) {
    if (!batch || !images) return IMP_ERROR_INVALID_ARGS;
    Group& G = batch->G;
    if (G.owner != &t_thread_tag) { set_error_text("impgpu_batch_decode_jpeg_pending from another thread than _begin"); return IMP_ERROR_INVALID_ARGS; }
    // (only a batch whose kernels are on the thread's OWN stream: what the caller enqueues on the frames runs behind them)
    if (G.stream && !on_lane_stream(G.stream)) { set_error_text("impgpu_batch_decode_jpeg_pending: the batch runs on the thread's second stream"); return IMP_ERROR_INVALID_ARGS; }
    for (int i = 0; i < G.count; i++) {
        Prep& p = G.P[(size_t)i];
        images[i] = nullptr;
        if (p.code || !p.im || p.given) continue;               // refused at its header, deferred to the host's Huffman stage, or handed out already
        images[i] = p.im;
        p.given = true;
    }
    return IMP_OK;
}

int impgpu_batch_decode_jpeg_prepared_begin(const impgpu_jpeg_prepared* files, int count, impgpu_jpeg_batch** batch) {
    if (!batch) return IMP_ERROR_INVALID_ARGS;
    *batch = nullptr;
    if (count <= 0 || count > MAX_BATCH || !files) return IMP_ERROR_INVALID_ARGS;
    if (!env_ready()) { set_error_text("impgpu_env_start has not been called"); return IMP_ERROR_DEVICE; }
    TraceRange tr("IMP_STEP_DECODE");
    IMP_FAULT_POINT(IMP_STEP_DECODE);
    impgpu_jpeg_batch* b = new impgpu_jpeg_batch();
    b->files.assign(files, files + count);
    b->blobs.resize((size_t)count);
    b->sizes.resize((size_t)count);
    for (int i = 0; i < count; i++) { b->blobs[(size_t)i] = files[i].head; b->sizes[(size_t)i] = files[i].head_size; }
    const int rc = group_begin(b->G, b->blobs.data(), b->sizes.data(), count, 0, false, b->files.data());
    if (rc) { delete b; return rc; }
    *batch = b;
    return IMP_OK;
}

int impgpu_batch_decode_jpeg_prepared(const impgpu_jpeg_prepared* files, int count, impgpu_image** images, int* codes) {
    if (count < 0 || (count && (!files || !images || !codes))) return IMP_ERROR_INVALID_ARGS;
    if (!env_ready()) { set_error_text("impgpu_env_start has not been called"); return IMP_ERROR_DEVICE; }
    TraceRange tr("IMP_STEP_DECODE");
    IMP_FAULT_POINT(IMP_STEP_DECODE);
    std::vector<const unsigned char*> blobs((size_t)count);
    std::vector<size_t> sizes((size_t)count);
    for (int i = 0; i < count; i++) { blobs[(size_t)i] = files[i].head; sizes[(size_t)i] = files[i].head_size; images[i] = nullptr; codes[i] = IMP_OK; }
    for (int at = 0; at < count; at += MAX_BATCH) {
        const int n = count - at < MAX_BATCH ? count - at : MAX_BATCH;
        Group A;
        int rc = group_begin(A, blobs.data() + at, sizes.data() + at, n, 0, false, files + at);
        if (rc) for (int i = 0; i < n; i++) codes[at + i] = A.P.size() > (size_t)i && A.P[(size_t)i].code ? A.P[(size_t)i].code : rc;
        else rc = group_finish(A, images + at, codes + at);
        if (!rc) rc = group_deferred(blobs.data() + at, sizes.data() + at, n, images + at, codes + at, files + at);
        if (rc) {
            for (int i = 0; i < at + n; i++) if (images[i]) impgpu_image_release(&images[i]);
            for (int i = at + n; i < count; i++) { images[i] = nullptr; codes[i] = rc; }
            return rc;
        }
    }
    return IMP_OK;
}

int impgpu_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return IMP_ERROR_INVALID_ARGS;
    if (!env_ready() || !env_stream()) { set_error_text("impgpu_env_start has not been called"); return IMP_ERROR_DEVICE; }
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { set_error("hipHostRegister", e); (void)hipGetLastError(); return IMP_ERROR_DEVICE; }
    return IMP_OK;
}

int impgpu_host_unregister(void* p) {
    if (!p) return IMP_ERROR_INVALID_ARGS;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { set_error("hipHostUnregister", e); (void)hipGetLastError(); return IMP_ERROR_DEVICE; }
    return IMP_OK;
}

int impgpu_jpeg_counters(unsigned long long* counters, int n) {
    if (!counters || n < 0) return IMP_ERROR_INVALID_ARGS;
    for (int i = 0; i < n; i++) counters[i] = i < 4 + JPEG_WHY_COUNT + 1 ? g_count[i].load(std::memory_order_relaxed) : 0ull;
    return IMP_OK;
}

int impgpu_jpeg_profile(int on) {
    return g_profile.exchange(on ? 1 : 0);
}

int impgpu_jpeg_stage_times(double* microseconds, int n) {
    if (!microseconds || n < 0) return IMP_ERROR_INVALID_ARGS;
    for (int i = 0; i < n; i++) microseconds[i] = i < 16 ? t_stage[i] : 0.0;
    return IMP_OK;
}

int impgpu_image_decode_jpeg(const unsigned char* blob, size_t size, impgpu_image** out) {
    if (!blob || !out) return IMP_ERROR_INVALID_ARGS;
    *out = nullptr;
    int code = IMP_OK;
    const int rc = impgpu_batch_decode_jpeg(&blob, &size, 1, out, &code);
    return rc ? rc : code;
}

}  // extern "C"
