// imp_broker.cpp -- `impgpu_broker`: the one process per GPU that owns the HIP context when IMP runs as N worker
// processes (docs/02 - Configuration.md:18 worker_processes; module.c:100-107 OnEnvStart per worker; module.c:298 /
// bridge.c:302 RunJob synchronous, one request at a time).  Protocol and rationale: include/impgpu_broker.h.
//
// Every broker thread is a lane of libimpgpu.so (its own stream, pools, pinned staging).  A thread takes ALL requests that
// are queued when it looks (up to --batch), so the number of files per launch follows the load by itself: one idle worker
// gets a batch of one (the latency of the in-process path plus two futex hops), 32 busy workers ride 16-32 to a launch --
//     files          impgpu_batch_decode_jpeg            cvDecodeImage, bridge.c:545-552
//     resize-only    impgpu_batch_resize_mixed           Resize(), bridge.c:588-604     (anything else: impgpu_run_ops per frame,
//                                                                                        bridge.c:574-656 in the reference's order)
//     JPEG answers   impgpu_batch_encode_jpeg            cvEncodeImage(".jpg"), bridge.c:704
//     pixel answers  impgpu_batch_download               for the host encoders (PNG, WebP, FreeImage formats)
// Nothing a worker writes into its slot is trusted further than a request is: the request record is copied out of shared
// memory once and validated (sizes against the slot, offsets against the text area, frame geometry against the bytes).
//
//   impgpu_broker [--name /impgpu-broker-0] [--device 0] [--slots 64] [--slot-mb 32] [--register-mb 8] [--pipeline 1] [--threads 2] [--batch 64]
//                 [--gather-us 0] [--supervise] [--ready-file PATH]
// --supervise: this process only forks and watches; the child is the broker.  A child that dies (a lost device, a bug) is
// replaced by a FRESH child -- fork() from a parent that never touched the GPU, no exec of a process that did.
#include <impgpu_broker.h>

#include <errno.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Options {
    std::string name = IMPB_DEFAULT_NAME;
    int device = 0;
    int slots = 64;
    long slot_mb = 32;
    long register_mb = 8;               // page-locked at the front of every slot (0: none)
    // --pipeline 1: a lane unpacks the next batch while the device writes the answers of the one before (Batch, below).  Measured
    // (one box, four lanes, requests/s at 8 / 16 / 32 workers): 11.5 / 15.6 / 21.1 k against 11.7 / 16.7 / 23.0 k one batch at a time.
    // The device idles less, and it does not matter: with N synchronous workers the rate is N / latency, and a request whose
    // lane also unpacks its successor and hands out its predecessor waits longer for its own answer.  It pays with a backlog
    // (a caller that keeps many requests in flight per connection); IMP's workers do not have one.
    bool pipeline = false;
    long split_kb = 0;
    // (Also measured and not kept: the frames taken ahead of their verdicts -- impgpu_batch_decode_jpeg_pending -- so that a batch's
    // operators and answers are enqueued behind its decode and the lane waits once.  From C, in process, a lone request gains 8-11 us
    // (tests/c/latency_harness.c); through the broker a lone worker's request is level (p50 0.36 ms both ways) and under load the
    // lanes lose: 7.5 / 10.7 / 14.9 / 17.8 k requests/s at 4 / 8 / 16 / 32 workers against 8.6 / 11.9 / 16.8 / 23.5 k.)
    int threads = 2;
    int batch = 64;
    int gather_us = 0;
    bool supervise = false;
    std::string ready_file;
};

struct Segment {
    uint8_t* base = nullptr;
    size_t bytes = 0;
    impb_header_fields* h = nullptr;
    impb_slot* slots = nullptr;
    uint8_t* data = nullptr;
    uint64_t slot_bytes = 0;
    uint64_t registered = 0;            // the first so many bytes of every slot's data area are page-locked (copies to the device start there)
    uint8_t* slot_data(int i) const { return data + (uint64_t)i * slot_bytes; }
};

std::atomic<bool> g_stop{false};
// where a batch's time goes (microseconds, summed over all batches; printed when the broker stops)
std::atomic<uint64_t> g_us_prepare{0}, g_us_decode{0}, g_us_ops{0}, g_us_answer{0}, g_us_idle{0};
void on_signal(int) { g_stop = true; }

long futex(volatile uint32_t* addr, int op, uint32_t val, const timespec* to) {
    return syscall(SYS_futex, addr, op, val, to, nullptr, 0);
}
bool pid_alive(uint32_t pid) { return pid != 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM); }
double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}
template <class T> T aload(volatile T* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
template <class T> void astore(volatile T* p, T v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

// ---- the segment: created, or adopted when one with the same geometry is already there (workers keep their mapping and
// their slots across a broker restart)
bool open_segment(const Options& o, Segment* S) {
    const uint64_t slot_bytes = (uint64_t)o.slot_mb << 20;
    const uint64_t slots_offset = sizeof(impb_header);
    const uint64_t data_offset = slots_offset + (uint64_t)o.slots * sizeof(impb_slot);
    const uint64_t total = data_offset + (uint64_t)o.slots * slot_bytes;
    bool fresh = false;
    int fd = shm_open(o.name.c_str(), O_RDWR, 0600);
    if (fd >= 0) {
        struct stat st;
        impb_header hdr;
        const bool same = fstat(fd, &st) == 0 && (uint64_t)st.st_size == total && pread(fd, &hdr, sizeof hdr, 0) == (ssize_t)sizeof hdr &&
                          hdr.f.magic == IMPB_MAGIC && hdr.f.version == IMPB_VERSION && hdr.f.nslots == (uint32_t)o.slots &&
                          hdr.f.slot_data_bytes == slot_bytes;
        if (same && pid_alive(hdr.f.broker_pid) && hdr.f.broker_pid != (uint32_t)getpid()) {
            std::fprintf(stderr, "impgpu_broker: %s is served by pid %u\n", o.name.c_str(), hdr.f.broker_pid);
            close(fd);
            return false;
        }
        if (!same) {                       // another layout: workers of the old one find a dead broker and re-open by name
            close(fd);
            shm_unlink(o.name.c_str());
            fd = -1;
        }
    }
    if (fd < 0) {
        fd = shm_open(o.name.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
        if (fd < 0) { std::perror("impgpu_broker: shm_open"); return false; }
        if (ftruncate(fd, (off_t)total) != 0) { std::perror("impgpu_broker: ftruncate"); close(fd); shm_unlink(o.name.c_str()); return false; }
        fresh = true;
    }
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { std::perror("impgpu_broker: mmap"); return false; }
    S->base = (uint8_t*)p; S->bytes = total;
    S->h = &((impb_header*)p)->f;
    S->slots = (impb_slot*)(S->base + slots_offset);
    S->data = S->base + data_offset;
    S->slot_bytes = slot_bytes;
    impb_header_fields* h = S->h;
    astore(&h->broker_pid, 0u);
    if (fresh) {
        h->magic = IMPB_MAGIC; h->version = IMPB_VERSION; h->nslots = (uint32_t)o.slots;
        h->slot_data_bytes = slot_bytes; h->slots_offset = slots_offset; h->data_offset = data_offset;
        h->epoch = 0;
    }
    __atomic_add_fetch(&h->epoch, 1u, __ATOMIC_SEQ_CST);
    h->device = (uint32_t)o.device;
    h->served = 0; h->batches = 0; h->sleepers = 0;
    // requests the previous broker died with: answered "device lost"; slots of dead workers: free
    for (int i = 0; i < o.slots; i++) {
        impb_slot_fields* s = &S->slots[i].f;
        const uint32_t st = aload(&s->state), owner = aload(&s->owner_pid);
        if (st == IMPB_FREE) continue;
        if (owner && !pid_alive(owner)) { astore(&s->owner_pid, 0u); astore(&s->state, (uint32_t)IMPB_FREE); continue; }
        if (st == IMPB_SUBMITTED || st == IMPB_TAKEN) {
            s->code = IMP_ERROR_DEVICE; s->step = IMP_STEP_START; s->out_bytes = 0; s->out_offset = 0;
            std::snprintf(s->error, sizeof s->error, "the broker was restarted");
            astore(&s->state, (uint32_t)IMPB_DONE);
            futex(&s->state, FUTEX_WAKE, 1, nullptr);
        }
    }
    return true;
}

// ---- one request, copied out of its slot
struct Req {
    int slot = -1;
    impb_slot_fields q;                 // private copy of the record (the strings included)
    const uint8_t* in = nullptr;        // the slot's data area (shared: read once by the decoder)
    impgpu_image* img = nullptr;
    impgpu_image* out = nullptr;        // resize-only path: the thumbnail
    int code = IMP_OK, step = IMP_STEP_START;
    bool done = false;                  // the answer is final (an error, NOT_TAKEN, a registration)
    std::string err;
    std::vector<const char*> filters;
    impgpu_job job{};
    impgpu_config cfg{};
    double t_taken = 0;
};

struct Watermarks {
    std::mutex mu;
    std::vector<impgpu_image*> imgs;    // id - 1 -> frame (lives as long as the broker: a location's overlay)
} g_marks;

bool text_ok(const impb_slot_fields&, int at) { return at == -1 || (at >= 0 && at < IMPB_TEXT_BYTES); }

void fail(Req& r, int code, int step, const char* what) {
    r.code = code; r.step = step; r.done = true;
    r.err = what ? what : "";
}

// validate + build job / config out of the private copy
void prepare(Req& r, const Segment& S) {
    impb_slot_fields& q = r.q;
    q.text[IMPB_TEXT_BYTES - 1] = 0;
    if (q.in_bytes > S.slot_bytes) return fail(r, IMP_ERROR_INVALID_ARGS, IMP_STEP_VALIDATE, "in_bytes past the slot");
    if (q.in_kind > IMPB_IN_WATERMARK || q.out_kind > IMPB_OUT_ASCII) return fail(r, IMP_ERROR_INVALID_ARGS, IMP_STEP_VALIDATE, "unknown kind");
    if (q.filter_count < 0 || q.filter_count > IMPB_MAX_FILTERS || !text_ok(q, q.crop_at) || !text_ok(q, q.gravity_at) || !text_ok(q, q.resize_at) || !text_ok(q, q.ascii_at))
        return fail(r, IMP_ERROR_INVALID_ARGS, IMP_STEP_VALIDATE, "bad text offsets");
    for (int i = 0; i < q.filter_count; i++) {
        if (q.filter_at[i] < 0 || q.filter_at[i] >= IMPB_TEXT_BYTES) return fail(r, IMP_ERROR_INVALID_ARGS, IMP_STEP_VALIDATE, "bad filter offset");
        r.filters.push_back(q.text + q.filter_at[i]);
    }
    if (q.in_scan_bytes) {              // a JPEG whose scan the worker has unstuffed (impgpu_jpeg_unstuff)
        if (q.in_kind != IMPB_IN_FILE || q.in_head_bytes < 4 || q.in_head_bytes > q.in_scan_at || (q.in_scan_at & 255) || q.in_scan_at > q.in_bytes ||
            q.in_scan_bytes > q.in_bytes - q.in_scan_at || q.in_bytes - q.in_scan_at - q.in_scan_bytes < IMPGPU_JPEG_SCAN_TAIL)
            return fail(r, IMP_ERROR_INVALID_ARGS, IMP_STEP_VALIDATE, "prepared file: offsets past its bytes");
    }
    if (q.in_kind != IMPB_IN_FILE) {
        const long long need = (long long)q.in_step * q.in_h;
        if (q.in_w <= 0 || q.in_h <= 0 || (q.in_c != 1 && q.in_c != 3 && q.in_c != 4) || (long long)q.in_step < (long long)q.in_w * q.in_c ||
            need <= 0 || (unsigned long long)need > q.in_bytes)
            return fail(r, IMP_ERROR_INVALID_ARGS, IMP_STEP_VALIDATE, "frame geometry does not match its bytes");
    }
    r.job.crop = q.crop_at >= 0 ? q.text + q.crop_at : nullptr;
    r.job.gravity = q.gravity_at >= 0 ? q.text + q.gravity_at : nullptr;
    r.job.resize = q.resize_at >= 0 ? q.text + q.resize_at : nullptr;
    r.job.simple = q.simple; r.job.need_flatten = q.need_flatten;
    r.job.filters = r.filters.data(); r.job.filter_count = q.filter_count;
    r.cfg.max_target_w = q.max_target_w; r.cfg.max_target_h = q.max_target_h;
    r.cfg.max_filters_count = q.max_filters_count; r.cfg.allow_experiments = q.allow_experiments;
    if (q.watermark_id) {
        std::lock_guard<std::mutex> lk(g_marks.mu);
        if (q.watermark_id < 0 || (size_t)q.watermark_id > g_marks.imgs.size())
            return fail(r, IMP_ERROR_NO_SUCH_WATERMARK, IMP_STEP_WATERMARK, "watermark id of another broker epoch");
        r.cfg.watermark = g_marks.imgs[(size_t)q.watermark_id - 1];
        r.cfg.watermark_opacity = q.watermark_opacity;
        r.cfg.watermark_gravity_x = q.watermark_gravity_x; r.cfg.watermark_gravity_y = q.watermark_gravity_y;
        r.cfg.watermark_offset_x = q.watermark_offset_x; r.cfg.watermark_offset_y = q.watermark_offset_y;
    }
}

bool is_jpeg(const uint8_t* p, uint64_t n) { return n >= 3 && p[0] == 0xFF && p[1] == 0xD8 && p[2] == 0xFF; }
bool is_png(const uint8_t* p, uint64_t n) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    return n >= 8 && !std::memcmp(p, sig, 8);
}

// Does the request's operator segment come down to one Resize() of a colour frame?  (Then it rides the mixed launch.)
bool resize_only(const Req& r) {
    if (!r.job.resize || r.job.crop || r.job.filter_count || r.cfg.watermark) return false;
    const int c = impgpu_image_channels(r.img);
    if (c == 1) return false;                                   // gray -> BGR first (bridge.c:613-618): run_ops knows
    if (c == 4 && r.job.need_flatten) return false;
    return impgpu_album_count(r.img) == 1;
}

// A batch on its way through a lane.  Its three steps -- begin (copy the records, enqueue the JPEG decode), middle (read the
// verdicts, operators, enqueue the answers' encode), end (fetch the files, wake the workers) -- are separate because the lane
// runs them INTERLEAVED with the next batch's: while the device writes batch k's answers the thread unpacks batch k + 1 and
// puts its decode behind them on the same stream, and while that decode runs it hands batch k's files out.  One lane, one
// stream, no second hardware queue (more than four queues in use slow every kernel on them: tools/contention_probe.sh).
struct Batch {
    std::vector<Req> reqs;
    std::vector<size_t> who;                    // requests whose JPEG rides the batch's decode
    std::vector<impgpu_jpeg_prepared> pre;
    impgpu_jpeg_batch* dec = nullptr;           // begun, not finished
    impgpu_jpeg_encode* enc = nullptr;          // the answers of ONE quality, begun and not fetched
    std::vector<size_t> enc_who;
    std::vector<const impgpu_image*> enc_im;
    bool live = false;
};

struct Worker {
    const Segment& S;
    const Options& O;
    int id;
    Batch slots_[2];
    // scratch reused from batch to batch
    std::vector<impgpu_image*> imgs;
    std::vector<int> codes;

    Worker(const Segment& s, const Options& o, int i) : S(s), O(o), id(i) {}

    // --split-kb K (A/B): a launch takes files of ONE size class -- up to K KB, or above -- so that a small file does not wait
    // for the decode of a large one it happens to share a launch with (a launch's kernels take what its longest file takes)
    int klass_ = -1;
    int take(std::vector<int>& mine, int start) {
        const int n = (int)S.h->nslots;
        int got = 0;
        if (mine.empty()) klass_ = -1;
        for (int k = 0; k < n && (int)mine.size() < O.batch; k++) {
            const int i = (start + k) % n;
            impb_slot_fields* s = &S.slots[i].f;
            if (aload(&s->state) != IMPB_SUBMITTED) continue;
            if (O.split_kb > 0) {
                const int kl = s->in_bytes > ((uint64_t)O.split_kb << 10) ? 1 : 0;
                if (klass_ >= 0 && kl != klass_) continue;
                klass_ = kl;
            }
            uint32_t expect = IMPB_SUBMITTED;
            if (__atomic_compare_exchange_n(&s->state, &expect, (uint32_t)IMPB_TAKEN, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) {
                mine.push_back(i);
                got++;
            }
        }
        return got;
    }

    void finish(Req& r) {
        impb_slot_fields* s = &S.slots[r.slot].f;
        s->code = r.code; s->step = r.step;
        std::snprintf(s->error, sizeof s->error, "%s", r.err.c_str());
        s->broker_us = (uint32_t)(now_us() - r.t_taken);
        impgpu_image_release(&r.img);
        impgpu_image_release(&r.out);
        __atomic_add_fetch(&S.h->served, (uint64_t)1, __ATOMIC_RELAXED);
        if (aload(&s->owner_pid) == 0) { astore(&s->state, (uint32_t)IMPB_FREE); return; }     // abandoned by a worker that timed out
        astore(&s->state, (uint32_t)IMPB_DONE);
        futex(&s->state, FUTEX_WAKE, 1, nullptr);
    }

    void begin(Batch& B, const std::vector<int>& mine) {
        std::vector<Req>& reqs = B.reqs;
        const size_t n = mine.size();
        const double t0 = now_us();
        B.live = true;
        reqs.clear();
        reqs.resize(n);
        for (size_t k = 0; k < n; k++) {
            Req& r = reqs[k];
            r.slot = mine[k];
            std::memcpy(&r.q, (const void*)&S.slots[r.slot].f, sizeof r.q);
            r.in = S.slot_data(r.slot);
            r.t_taken = t0;
            prepare(r, S);
            impb_slot_fields* s = &S.slots[r.slot].f;
            s->out_offset = 0; s->out_bytes = 0; s->out_w = s->out_h = s->out_c = s->out_step = 0; s->brightness = 0;
            s->batch_size = (int32_t)n;
        }
        __atomic_add_fetch(&S.h->batches, (uint64_t)1, __ATOMIC_RELAXED);
        const double t1 = now_us();
        g_us_prepare += (uint64_t)(t1 - t0);

        // ---- decode (bridge.c:541-572): all JPEG files of the batch in one call, enqueued here, waited for in middle()
        std::vector<size_t>& who = B.who;
        std::vector<impgpu_jpeg_prepared>& pre = B.pre;
        who.clear(); pre.clear();
        for (size_t k = 0; k < n; k++) {
            Req& r = reqs[k];
            if (r.done) continue;
            r.step = IMP_STEP_DECODE;
            if (r.q.in_kind != IMPB_IN_FILE || !is_jpeg(r.in, r.q.in_bytes)) continue;
            who.push_back(k);
            impgpu_jpeg_prepared f{r.in, (size_t)r.q.in_bytes, nullptr, 0, 0};
            if (r.q.in_scan_bytes) {
                // (its bytes go to the device from the slot when they lie in its page-locked part; the worker sleeps until DONE)
                f.head_size = (size_t)r.q.in_head_bytes;
                f.scan = r.in + r.q.in_scan_at;
                f.scan_size = (size_t)r.q.in_scan_bytes;
                f.registered = r.q.in_scan_at + r.q.in_scan_bytes + IMPGPU_JPEG_SCAN_TAIL <= S.registered;
            }
            pre.push_back(f);
        }
        B.dec = nullptr;
        if (!who.empty()) {
            const int rc = impgpu_batch_decode_jpeg_prepared_begin(pre.data(), (int)who.size(), &B.dec);
            if (rc != IMP_OK) {
                B.dec = nullptr;
                for (size_t j = 0; j < who.size(); j++) fail(reqs[who[j]], rc, IMP_STEP_DECODE, impgpu_last_error());
                who.clear();
            }
        }
        g_us_decode += (uint64_t)(now_us() - t1);
    }

    void middle(Batch& B) {
        std::vector<Req>& reqs = B.reqs;
        const std::vector<size_t>& who = B.who;
        const size_t n = reqs.size();
        const double t1 = now_us();
        if (B.dec) {
            imgs.assign(who.size(), nullptr);
            codes.assign(who.size(), IMP_OK);
            const int rc = impgpu_batch_decode_jpeg_finish(&B.dec, imgs.data(), codes.data());
            B.dec = nullptr;
            for (size_t j = 0; j < who.size(); j++) {
                Req& r = reqs[who[j]];
                const int c = rc != IMP_OK ? rc : codes[j];
                if (c == IMP_OK) { r.img = imgs[j]; continue; }
                if (c == IMP_ERROR_UNSUPPORTED || c == IMP_ERROR_DECODE_FAILED) fail(r, IMPB_NOT_TAKEN, IMP_STEP_DECODE, "not a file the device decodes");
                else fail(r, c, IMP_STEP_DECODE, impgpu_last_error());
            }
        }
        for (size_t k = 0; k < n; k++) {
            Req& r = reqs[k];
            if (r.done || r.img) continue;
            int rc = IMP_OK;
            if (r.q.in_kind == IMPB_IN_FILE) {
                if (is_png(r.in, r.q.in_bytes)) {
                    rc = impgpu_image_decode_png(r.in, (size_t)r.q.in_bytes, &r.img);
                    if (rc == IMP_ERROR_UNSUPPORTED || rc == IMP_ERROR_DECODE_FAILED) { fail(r, IMPB_NOT_TAKEN, IMP_STEP_DECODE, "not a file the device decodes"); continue; }
                } else { fail(r, IMPB_NOT_TAKEN, IMP_STEP_DECODE, "neither JPEG nor PNG"); continue; }
            } else {
                rc = impgpu_image_upload(r.in, r.q.in_w, r.q.in_h, r.q.in_c, r.q.in_step, &r.img);
            }
            if (rc != IMP_OK) { fail(r, rc, IMP_STEP_DECODE, impgpu_last_error()); continue; }
            if (r.q.in_kind == IMPB_IN_WATERMARK) {             // PrepareWatermark (bridge.c:199-237): kept, answered with its id
                std::lock_guard<std::mutex> lk(g_marks.mu);
                g_marks.imgs.push_back(r.img);
                r.img = nullptr;
                S.slots[r.slot].f.out_w = (int32_t)g_marks.imgs.size();
                r.code = IMP_OK; r.step = IMP_STEP_INFO; r.done = true;
            }
        }
        // a registered overlay must be complete before another lane's request reads it
        for (size_t k = 0; k < n; k++) if (reqs[k].q.in_kind == IMPB_IN_WATERMARK && reqs[k].code == IMP_OK) { (void)impgpu_sync(); break; }

        const double t2 = now_us();
        g_us_decode += (uint64_t)(t2 - t1);
        // ---- operators (bridge.c:574-656)
        std::map<int, std::vector<size_t>> mixed;                // channels * 2 + simple -> requests of one mixed launch
        for (size_t k = 0; k < n; k++) {
            Req& r = reqs[k];
            if (r.done) continue;
            if (r.cfg.max_filters_count > 0 && r.job.filter_count > r.cfg.max_filters_count) { fail(r, IMP_ERROR_TOO_MUCH_FILTERS, IMP_STEP_START, ""); continue; }
            if (n > 1 && resize_only(r)) {
                int ow = 0, oh = 0, ip = 0;
                const int sw = impgpu_image_width(r.img), sh = impgpu_image_height(r.img), c = impgpu_image_channels(r.img);
                const int rc = impgpu_resize_geometry(sw, sh, r.job.resize, &r.cfg, r.job.simple, &ow, &oh, &ip);
                if (rc != IMP_OK) { fail(r, rc, IMP_STEP_RESIZE, ""); continue; }
                if (ow != sw || oh != sh) {
                    const int rc2 = impgpu_image_create(ow, oh, c, &r.out);
                    if (rc2 != IMP_OK) { fail(r, rc2, IMP_STEP_RESIZE, impgpu_last_error()); continue; }
                    mixed[c * 2 + (r.job.simple ? 1 : 0)].push_back(k);
                    continue;
                }
            }
            int step = IMP_STEP_START;
            const int rc = impgpu_run_ops(&r.img, &r.job, &r.cfg, &step);
            if (rc != IMP_OK) fail(r, rc, step, rc == IMP_ERROR_DEVICE ? impgpu_last_error() : "");
        }
        for (auto& kv : mixed) {
            std::vector<impgpu_resize_item> items(kv.second.size());
            for (size_t j = 0; j < kv.second.size(); j++) {
                Req& r = reqs[kv.second[j]];
                items[j].src = impgpu_image_device_ptr(r.img); items[j].src_width = impgpu_image_width(r.img);
                items[j].src_height = impgpu_image_height(r.img); items[j].src_step = impgpu_image_step(r.img);
                items[j].dst = impgpu_image_device_ptr(r.out); items[j].dst_width = impgpu_image_width(r.out);
                items[j].dst_height = impgpu_image_height(r.out); items[j].dst_step = impgpu_image_step(r.out);
            }
            const int rc = impgpu_batch_resize_mixed(items.data(), (int)items.size(), kv.first / 2, kv.first & 1, nullptr);
            for (size_t j = 0; j < kv.second.size(); j++) {
                Req& r = reqs[kv.second[j]];
                if (rc != IMP_OK) { fail(r, rc, IMP_STEP_RESIZE, impgpu_last_error()); continue; }
                impgpu_image_release(&r.img);                    // (pool memory: recycled in stream order, behind the launch)
                r.img = r.out;
                r.out = nullptr;
            }
        }

        const double t3 = now_us();
        g_us_ops += (uint64_t)(t3 - t2);
        // ---- answers (bridge.c:659-710)
        std::map<int, std::vector<size_t>> by_quality;
        std::vector<size_t> raw;
        for (size_t k = 0; k < n; k++) {
            Req& r = reqs[k];
            if (r.done) continue;
            impb_slot_fields* s = &S.slots[r.slot].f;
            const uint64_t at = (r.q.in_bytes + 63) & ~uint64_t(63);
            s->out_w = impgpu_image_width(r.img); s->out_h = impgpu_image_height(r.img); s->out_c = impgpu_image_channels(r.img);
            s->out_offset = at;
            if (r.q.out_kind == IMPB_OUT_INFO) {
                r.step = IMP_STEP_INFO;
                float b = 0;
                const int rc = impgpu_calc_perceived_brightness(r.img, &b);
                if (rc != IMP_OK) { fail(r, rc, IMP_STEP_INFO, impgpu_last_error()); continue; }
                s->brightness = b;
                r.code = IMP_OK; r.done = true;
            } else if (r.q.out_kind == IMPB_OUT_ASCII) {            // the text exit (bridge.c:669-670): (width + 1) * height - 1 characters
                r.step = IMP_STEP_INFO;
                const long need = (long)(s->out_w + 1) * s->out_h - 1;
                if (at >= S.slot_bytes || (uint64_t)(need > 0 ? need : 1) > S.slot_bytes - at) { fail(r, IMP_ERROR_MALLOC_FAILED, IMP_STEP_INFO, "answer does not fit the slot"); continue; }
                long len = 0;
                const int rc = impgpu_ascii(r.img, r.q.ascii_at >= 0 ? r.q.text + r.q.ascii_at : "", S.slot_data(r.slot) + at, need, &len);
                if (rc != IMP_OK) { fail(r, rc, IMP_STEP_INFO, rc == IMP_ERROR_DEVICE ? impgpu_last_error() : ""); continue; }
                s->out_bytes = (uint64_t)len;
                r.code = IMP_OK; r.done = true;
            } else if (r.q.out_kind == IMPB_OUT_JPEG) {
                r.step = IMP_STEP_ENCODE;
                if (at >= S.slot_bytes || S.slot_bytes - at < impgpu_jpeg_encode_bound(s->out_w, s->out_h, s->out_c)) { fail(r, IMP_ERROR_MALLOC_FAILED, IMP_STEP_ENCODE, "answer does not fit the slot"); continue; }
                by_quality[r.q.quality].push_back(k);
            } else {
                r.step = IMP_STEP_ENCODE;
                const uint64_t need = (uint64_t)impgpu_image_step(r.img) * (uint64_t)s->out_h;
                if (at >= S.slot_bytes || S.slot_bytes - at < need) { fail(r, IMP_ERROR_MALLOC_FAILED, IMP_STEP_ENCODE, "answer does not fit the slot"); continue; }
                s->out_step = impgpu_image_step(r.img);
                s->out_bytes = need;
                raw.push_back(k);
            }
        }
        // the JPEG answers of the batch's most common quality are only ENQUEUED here (end() fetches them); what else the batch
        // wants -- other qualities, pixels for host encoders -- is answered on the spot, before that encode goes on the stream
        int async_quality = -1;
        size_t most = 0;
        for (auto& kv : by_quality) if (kv.second.size() > most) { most = kv.second.size(); async_quality = kv.first; }
        for (auto& kv : by_quality) {
            if (kv.first == async_quality) continue;
            const size_t m = kv.second.size();
            std::vector<const impgpu_image*> im(m);
            std::vector<unsigned char*> outs(m);
            std::vector<size_t> caps(m), lens(m, 0);
            std::vector<int> cs(m, IMP_OK);
            for (size_t j = 0; j < m; j++) {
                Req& r = reqs[kv.second[j]];
                const impb_slot_fields* s = &S.slots[r.slot].f;
                im[j] = r.img;
                outs[j] = S.slot_data(r.slot) + s->out_offset;
                caps[j] = (size_t)(S.slot_bytes - s->out_offset);
            }
            const int rc = impgpu_batch_encode_jpeg(im.data(), (int)m, kv.first, outs.data(), caps.data(), lens.data(), cs.data());
            for (size_t j = 0; j < m; j++) {
                Req& r = reqs[kv.second[j]];
                const int c = rc != IMP_OK ? rc : cs[j];
                if (c != IMP_OK) { fail(r, c, IMP_STEP_ENCODE, impgpu_last_error()); continue; }
                S.slots[r.slot].f.out_bytes = lens[j];
                r.code = IMP_OK; r.step = IMP_STEP_ENCODE; r.done = true;
            }
        }
        if (!raw.empty()) {
            const size_t m = raw.size();
            std::vector<const impgpu_image*> im(m);
            std::vector<unsigned char*> outs(m);
            std::vector<int> steps(m);
            for (size_t j = 0; j < m; j++) {
                Req& r = reqs[raw[j]];
                const impb_slot_fields* s = &S.slots[r.slot].f;
                im[j] = r.img; outs[j] = S.slot_data(r.slot) + s->out_offset; steps[j] = s->out_step;
            }
            const int rc = impgpu_batch_download(im.data(), (int)m, outs.data(), steps.data());
            for (size_t j = 0; j < m; j++) {
                Req& r = reqs[raw[j]];
                if (rc != IMP_OK) { fail(r, rc, IMP_STEP_ENCODE, impgpu_last_error()); continue; }
                r.code = IMP_OK; r.step = IMP_STEP_ENCODE; r.done = true;
            }
        }
        B.enc = nullptr;
        B.enc_who.clear();
        if (most) {
            B.enc_who = by_quality[async_quality];
            B.enc_im.resize(B.enc_who.size());
            for (size_t j = 0; j < B.enc_who.size(); j++) B.enc_im[j] = reqs[B.enc_who[j]].img;
            const int rc = impgpu_batch_encode_jpeg_begin(B.enc_im.data(), (int)B.enc_im.size(), async_quality, &B.enc);
            if (rc != IMP_OK) {
                B.enc = nullptr;
                for (size_t j = 0; j < B.enc_who.size(); j++) fail(reqs[B.enc_who[j]], rc, IMP_STEP_ENCODE, impgpu_last_error());
                B.enc_who.clear();
            }
        }
        g_us_answer += (uint64_t)(now_us() - t3);
    }

    void end(Batch& B) {
        std::vector<Req>& reqs = B.reqs;
        const double t0 = now_us();
        if (B.enc) {
            const size_t m = B.enc_who.size();
            std::vector<unsigned char*> outs(m);
            std::vector<size_t> caps(m), lens(m, 0);
            std::vector<int> cs(m, IMP_OK);
            for (size_t j = 0; j < m; j++) {
                Req& r = reqs[B.enc_who[j]];
                const impb_slot_fields* s = &S.slots[r.slot].f;
                outs[j] = S.slot_data(r.slot) + s->out_offset;
                caps[j] = (size_t)(S.slot_bytes - s->out_offset);
            }
            const int rc = impgpu_batch_encode_jpeg_finish(&B.enc, outs.data(), caps.data(), lens.data(), cs.data());
            B.enc = nullptr;
            for (size_t j = 0; j < m; j++) {
                Req& r = reqs[B.enc_who[j]];
                const int c = rc != IMP_OK ? rc : cs[j];
                if (c != IMP_OK) { fail(r, c, IMP_STEP_ENCODE, impgpu_last_error()); continue; }
                S.slots[r.slot].f.out_bytes = lens[j];
                r.code = IMP_OK; r.step = IMP_STEP_ENCODE; r.done = true;
            }
        }
        for (size_t k = 0; k < reqs.size(); k++) finish(reqs[k]);
        B.live = false;
        g_us_answer += (uint64_t)(now_us() - t0);
    }

    void loop() {
        impb_header_fields* h = S.h;
        std::vector<int> mine;
        int start = id * 7;
        Batch* P = nullptr;                                 // the batch whose decode is on the stream
        while (!g_stop || P) {
            if (P) middle(*P);                              // its verdicts, its operators, its answers enqueued
            mine.clear();
            if (!g_stop) {
                const uint32_t bell = __atomic_load_n(&h->doorbell, __ATOMIC_SEQ_CST);
                take(mine, start);
                if (mine.empty() && !P) {
                    const double idle0 = now_us();
                    __atomic_add_fetch(&h->sleepers, 1u, __ATOMIC_SEQ_CST);
                    take(mine, start);                      // (a submit between the scan and the count)
                    if (mine.empty()) {
                        timespec tick{0, 100 * 1000 * 1000};
                        futex(&h->doorbell, FUTEX_WAIT, bell, &tick);
                    }
                    __atomic_sub_fetch(&h->sleepers, 1u, __ATOMIC_SEQ_CST);
                    g_us_idle += (uint64_t)(now_us() - idle0);
                    if (mine.empty()) continue;
                }
                if (!P && O.gather_us > 0 && (int)mine.size() < O.batch) {
                    // a few workers answered together come back together: give the stragglers of that wave a moment
                    const double until = now_us() + O.gather_us;
                    while (now_us() < until && (int)mine.size() < O.batch) {
                        if (!take(mine, start)) { timespec nap{0, 5000}; nanosleep(&nap, nullptr); }
                    }
                }
            }
            Batch* B = nullptr;
            if (!mine.empty()) {
                // what is queued NOW is unpacked and its decode goes on the stream behind P's answers, while those are written
                B = P == &slots_[0] ? &slots_[1] : &slots_[0];
                start = (start + 1) % (int)h->nslots;
                begin(*B, mine);
            }
            if (P) end(*P);                                 // fetch P's files, wake its workers (B's decode runs meanwhile)
            P = B;
            if (P && !O.pipeline) { middle(*P); end(*P); P = nullptr; }      // (A/B: one batch at a time, as round 5 began)
        }
    }
};

// slots of workers that are gone (killed between requests, or timed out and left): free again
void reap(const Segment& S) {
    for (uint32_t i = 0; i < S.h->nslots; i++) {
        impb_slot_fields* s = &S.slots[i].f;
        const uint32_t st = aload(&s->state), owner = aload(&s->owner_pid);
        if (st == IMPB_FREE || st == IMPB_TAKEN) continue;
        if (owner && !pid_alive(owner)) {
            uint32_t expect = owner;
            if (__atomic_compare_exchange_n(&s->owner_pid, &expect, 0u, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED) && st != IMPB_SUBMITTED)
                astore(&s->state, (uint32_t)IMPB_FREE);
            // (SUBMITTED: a broker thread takes it, finds no owner when it is done and frees it)
        } else if (!owner && st == IMPB_DONE) {
            astore(&s->state, (uint32_t)IMPB_FREE);
        }
    }
}

int serve(const Options& o) {
    Segment S;
    if (!open_segment(o, &S)) return 3;
    // a hardware queue per lane (the runtime's default is four per process; lanes that share a queue run one behind the
    // other: four threads served 16 workers at 10.6 k requests/s on four queues and at 12.4 k on eight)
    if (!getenv("GPU_MAX_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", o.threads > 4 ? "16" : "8", 1);
    if (impgpu_env_start(o.device) != IMP_OK) {
        std::fprintf(stderr, "impgpu_broker: impgpu_env_start(%d): %s\n", o.device, impgpu_last_error());
        shm_unlink(o.name.c_str());                          // never served: workers must not find a segment nobody will answer on
        return 4;
    }
    (void)impgpu_env_bind_thread();
    // The front of every slot is page-locked: a JPEG a worker has unstuffed into it (glue/imp_gpu_client.c) goes to the device
    // from where it lies.  Files that reach past it are staged like any caller's.  (Locking allocates the pages: 8 MB a slot.)
    {
        const uint64_t want = std::min<uint64_t>(S.slot_bytes, (uint64_t)o.register_mb << 20);
        bool ok = want > 0;
        for (uint32_t i = 0; ok && i < S.h->nslots; i++) ok = impgpu_host_register(S.slot_data((int)i), (size_t)want) == IMP_OK;
        if (ok) S.registered = want;
        else if (want) std::fprintf(stderr, "impgpu_broker: the slots could not be page-locked (%s): prepared files are staged\n", impgpu_last_error());
    }
    std::vector<std::thread> threads;
    std::vector<Worker*> workers;
    for (int i = 0; i < o.threads; i++) {
        workers.push_back(new Worker(S, o, i));
        threads.emplace_back([w = workers.back()] { (void)impgpu_env_bind_thread(); w->loop(); });
    }
    astore(&S.h->broker_pid, (uint32_t)getpid());            // open for business
    if (!o.ready_file.empty()) { FILE* f = std::fopen(o.ready_file.c_str(), "w"); if (f) { std::fprintf(f, "%d\n", (int)getpid()); std::fclose(f); } }
    std::fprintf(stderr, "impgpu_broker: pid %d serves %s on device %d: %d slots of %ld MB, %d threads, epoch %u\n", (int)getpid(),
                 o.name.c_str(), impgpu_env_device(), o.slots, o.slot_mb, o.threads, S.h->epoch);
    int beats = 0;
    while (!g_stop) {
        timespec nap{0, 100 * 1000 * 1000};
        nanosleep(&nap, nullptr);
        __atomic_add_fetch(&S.h->heartbeat, 1u, __ATOMIC_RELAXED);
        if (++beats % 10 == 0) reap(S);
    }
    astore(&S.h->broker_pid, 0u);                            // closed: workers' next requests fail fast, waiting ones at their next tick
    futex(&S.h->doorbell, FUTEX_WAKE, 1 << 30, nullptr);
    for (auto& t : threads) t.join();
    for (Worker* w : workers) delete w;
    {
        const double nb = (double)(S.h->batches ? S.h->batches : 1);
        std::fprintf(stderr, "impgpu_broker: per batch, us: copy + validate %.0f, decode %.0f, operators %.0f, answers %.0f; idle per thread %.0f ms\n",
                     g_us_prepare.load() / nb, g_us_decode.load() / nb, g_us_ops.load() / nb, g_us_answer.load() / nb, g_us_idle.load() / 1e3 / o.threads);
    }
    std::fprintf(stderr, "impgpu_broker: served %llu requests in %llu batches\n", (unsigned long long)S.h->served, (unsigned long long)S.h->batches);
    {
        std::lock_guard<std::mutex> lk(g_marks.mu);
        for (impgpu_image*& m : g_marks.imgs) impgpu_image_release(&m);
    }
    impgpu_env_destroy();
    shm_unlink(o.name.c_str());                              // a clean stop leaves nothing in /dev/shm (a crash leaves the segment for the next child to adopt)
    return 0;
}

int supervise(const Options& o) {
    // this process never initialises HIP: every broker is a fork()ed child that starts from a clean slate
    int fast_deaths = 0;
    while (!g_stop) {
        const double t0 = now_us();
        const pid_t child = fork();
        if (child < 0) { std::perror("impgpu_broker: fork"); return 5; }
        if (child == 0) {
            Options c = o;
            c.supervise = false;
            _exit(serve(c));
        }
        int status = 0;
        while (waitpid(child, &status, 0) < 0 && errno == EINTR) {
            if (g_stop) kill(child, SIGTERM);
        }
        if (g_stop) break;
        const bool clean = WIFEXITED(status) && WEXITSTATUS(status) == 0;
        std::fprintf(stderr, "impgpu_broker: child %d ended (%s %d); starting a fresh one\n", (int)child,
                     WIFSIGNALED(status) ? "signal" : "status", WIFSIGNALED(status) ? WTERMSIG(status) : WEXITSTATUS(status));
        if (clean) break;
        fast_deaths = now_us() - t0 < 2e6 ? fast_deaths + 1 : 0;
        if (fast_deaths >= 5) { std::fprintf(stderr, "impgpu_broker: five brokers in a row died at once; giving up\n"); return 6; }
        timespec nap{0, 200 * 1000 * 1000};
        nanosleep(&nap, nullptr);
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    Options o;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "impgpu_broker: %s needs a value\n", what); std::exit(2); }
            return argv[++i];
        };
        if (a == "--name") o.name = val("--name");
        else if (a == "--device") o.device = std::atoi(val("--device"));
        else if (a == "--slots") o.slots = std::atoi(val("--slots"));
        else if (a == "--slot-mb") o.slot_mb = std::atol(val("--slot-mb"));
        else if (a == "--split-kb") o.split_kb = std::atol(val("--split-kb"));
        else if (a == "--pipeline") o.pipeline = std::atoi(val("--pipeline")) != 0;
        else if (a == "--register-mb") o.register_mb = std::max(0l, std::atol(val("--register-mb")));
        else if (a == "--threads") o.threads = std::atoi(val("--threads"));
        else if (a == "--batch") o.batch = std::atoi(val("--batch"));
        else if (a == "--gather-us") o.gather_us = std::atoi(val("--gather-us"));
        else if (a == "--ready-file") o.ready_file = val("--ready-file");
        else if (a == "--supervise") o.supervise = true;
        else { std::fprintf(stderr, "usage: impgpu_broker [--name /impgpu-broker-0] [--device 0] [--slots 64] [--slot-mb 32] [--register-mb 8] [--pipeline 1] [--threads 2] [--batch 64] [--gather-us 0] [--supervise] [--ready-file PATH]\n"); return 2; }
    }
    if (o.slots < 1 || o.slots > IMPB_MAX_SLOTS || o.slot_mb < 1 || o.slot_mb > 4096 || o.threads < 1 || o.threads > 32 || o.batch < 1 || o.batch > 256 ||
        o.name.empty() || o.name[0] != '/') {
        std::fprintf(stderr, "impgpu_broker: bad option value\n");
        return 2;
    }
    struct sigaction sa {};
    sa.sa_handler = on_signal;
    sigaction(SIGTERM, &sa, nullptr);
    sigaction(SIGINT, &sa, nullptr);
    signal(SIGPIPE, SIG_IGN);
    return o.supervise ? supervise(o) : serve(o);
}
