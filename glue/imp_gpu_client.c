/*
 * imp_gpu_client.c -- the worker side of include/impgpu_broker.h: plain C (gnu99), no HIP, no C++ runtime, so it can be
 * compiled into the nginx module next to bridge.c (glue/config adds it) or into a small shared object of its own
 * (ngx_http_imgproc_amd/libimpgpu_client.so, what the tests and tests/c/worker_harness.c load).
 *
 * One client = one slot of the broker's segment = one request in flight, which is what an IMP worker has (RunJob is
 * synchronous, module.c:298).  Nothing here allocates per request.
 */
#define _GNU_SOURCE
#include "impgpu_broker.h"

#include <errno.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#define IMPC_MAX_WATERMARKS 16

typedef struct {
    const unsigned char* pixels;
    int w, h, c, step;
    int broker_id;
    unsigned epoch;
} impc_watermark;

struct impgpu_client {
    char          name[128];
    uint8_t*      base;
    size_t        bytes;
    impb_header*  hdr;
    impb_slot*    slot;
    uint8_t*      data;
    int           slot_index;
    pid_t         pid;                  /* the process that claimed the slot: a fork()ed copy claims its own */
    long          timeout_ms;
    impc_watermark marks[IMPC_MAX_WATERMARKS];
    int           nmarks;
};

static __thread char t_err[200];
const char* impgpu_client_last_error(void) { return t_err; }

static int fail(const char* what) {
    snprintf(t_err, sizeof t_err, "%s", what);
    return IMP_ERROR_DEVICE;
}

static long futex(volatile uint32_t* addr, int op, uint32_t val, const struct timespec* to) {
    return syscall(SYS_futex, addr, op, val, to, NULL, 0);         /* shared (not FUTEX_PRIVATE): the word lives in the segment */
}

static int alive(uint32_t pid) {
    return pid != 0 && (kill((pid_t)pid, 0) == 0 || errno == EPERM);
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}

static void unmap(impgpu_client* c) {
    if (c->base) {
        if (c->slot && c->pid == getpid()) {
            /* hand the slot back (only our own: after a fork the parent still owns it) */
            uint32_t mine = (uint32_t)c->pid;
            if (__atomic_load_n(&c->slot->f.owner_pid, __ATOMIC_ACQUIRE) == mine) {
                __atomic_store_n(&c->slot->f.owner_pid, 0u, __ATOMIC_RELEASE);
                __atomic_store_n(&c->slot->f.state, (uint32_t)IMPB_FREE, __ATOMIC_RELEASE);
            }
        }
        munmap(c->base, c->bytes);
    }
    c->base = NULL; c->hdr = NULL; c->slot = NULL; c->data = NULL; c->bytes = 0; c->slot_index = -1;
}

/* map the segment and claim a slot */
static int map(impgpu_client* c) {
    struct stat st;
    int fd = shm_open(c->name, O_RDWR, 0);
    if (fd < 0) {
        snprintf(t_err, sizeof t_err, "no broker segment %s (%s)", c->name, strerror(errno));
        return IMP_ERROR_DEVICE;
    }
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(impb_header)) { close(fd); return fail("broker segment too small"); }
    void* p = mmap(NULL, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail("cannot map the broker segment");
    c->base = (uint8_t*)p;
    c->bytes = (size_t)st.st_size;
    c->hdr = (impb_header*)p;
    const impb_header_fields* h = &c->hdr->f;
    if (h->magic != IMPB_MAGIC || h->version != IMPB_VERSION || h->nslots == 0 || h->nslots > IMPB_MAX_SLOTS ||
        h->slots_offset + (uint64_t)h->nslots * sizeof(impb_slot) > c->bytes ||
        h->data_offset + (uint64_t)h->nslots * h->slot_data_bytes > c->bytes) {
        c->slot = NULL;
        unmap(c);
        return fail("broker segment has another layout (version mismatch?)");
    }
    if (!alive(__atomic_load_n(&h->broker_pid, __ATOMIC_ACQUIRE))) {
        c->slot = NULL;
        unmap(c);
        return fail("no live broker serves the segment");
    }
    impb_slot* slots = (impb_slot*)(c->base + h->slots_offset);
    const uint32_t me = (uint32_t)getpid();
    for (uint32_t k = 0; k < h->nslots; k++) {
        /* start at a slot derived from the pid: workers that attach together do not all fight for slot 0 */
        const uint32_t i = (me + k) % h->nslots;
        uint32_t expect = IMPB_FREE;
        if (__atomic_compare_exchange_n(&slots[i].f.state, &expect, (uint32_t)IMPB_CLAIMED, 0, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) {
            __atomic_store_n(&slots[i].f.owner_pid, me, __ATOMIC_RELEASE);
            slots[i].f.epoch = h->epoch;
            c->slot = &slots[i];
            c->slot_index = (int)i;
            c->data = c->base + h->data_offset + (uint64_t)i * h->slot_data_bytes;
            c->pid = getpid();
            return IMP_OK;
        }
    }
    c->slot = NULL;
    unmap(c);
    return fail("every slot of the broker is taken (more workers than --slots)");
}

int impgpu_client_attach(const char* name, impgpu_client** out) {
    if (!out) return IMP_ERROR_INVALID_ARGS;
    *out = NULL;
    if (!name) name = getenv("IMPGPU_BROKER");
    if (!name || !*name) name = IMPB_DEFAULT_NAME;
    impgpu_client* c = (impgpu_client*)calloc(1, sizeof *c);
    if (!c) return IMP_ERROR_MALLOC_FAILED;
    snprintf(c->name, sizeof c->name, "%s", name);
    c->slot_index = -1;
    const char* t = getenv("IMPGPU_BROKER_TIMEOUT_MS");
    c->timeout_ms = t ? atol(t) : 10000;
    if (c->timeout_ms < 1) c->timeout_ms = 1;
    int rc = map(c);
    if (rc != IMP_OK) { free(c); return rc; }
    *out = c;
    return IMP_OK;
}

void impgpu_client_detach(impgpu_client** client) {
    if (!client || !*client) return;
    unmap(*client);
    free(*client);
    *client = NULL;
}

/* A usable slot: ours (this pid), in a segment a live broker serves.  Re-attaches after a fork, after the broker was
 * replaced together with its segment, or after an earlier failure dropped the mapping. */
static int ready(impgpu_client* c) {
    if (c->base && c->pid != getpid()) {            /* forked copy: the slot is the parent's */
        munmap(c->base, c->bytes);
        c->base = NULL; c->slot = NULL;
    }
    if (c->base) {
        const uint32_t bp = __atomic_load_n(&c->hdr->f.broker_pid, __ATOMIC_ACQUIRE);
        if (alive(bp) && __atomic_load_n(&c->slot->f.owner_pid, __ATOMIC_ACQUIRE) == (uint32_t)c->pid) return IMP_OK;
        unmap(c);                                    /* broker gone (a fresh one may have made a new segment), or it took the slot back */
    }
    return map(c);
}

void* impgpu_client_input_buffer(impgpu_client* c, size_t bytes) {
    if (!c || ready(c) != IMP_OK) return NULL;
    if (bytes > c->hdr->f.slot_data_bytes) return NULL;
    return c->data;
}

static int put_text(impb_slot_fields* s, size_t* at, const char* str) {
    if (!str) return -1;
    const size_t n = strlen(str) + 1;
    if (*at + n > IMPB_TEXT_BYTES) return -2;
    memcpy(s->text + *at, str, n);
    const int where = (int)*at;
    *at += n;
    return where;
}

/* submit what is in the slot and sleep until the broker has answered */
static int roundtrip(impgpu_client* c) {
    impb_slot_fields* s = &c->slot->f;
    impb_header_fields* h = &c->hdr->f;
    const uint32_t broker = __atomic_load_n(&h->broker_pid, __ATOMIC_ACQUIRE);
    const uint32_t epoch = __atomic_load_n(&h->epoch, __ATOMIC_ACQUIRE);
    __atomic_store_n(&s->state, (uint32_t)IMPB_SUBMITTED, __ATOMIC_SEQ_CST);
    __atomic_add_fetch(&h->doorbell, 1u, __ATOMIC_SEQ_CST);
    if (__atomic_load_n(&h->sleepers, __ATOMIC_SEQ_CST) > 0) futex(&h->doorbell, FUTEX_WAKE, 1, NULL);
    const double t0 = now_ms();
    for (;;) {
        const uint32_t v = __atomic_load_n(&s->state, __ATOMIC_ACQUIRE);
        if (v == IMPB_DONE) break;
        if (v != IMPB_SUBMITTED && v != IMPB_TAKEN) { unmap(c); return fail("the broker took the slot back"); }
        struct timespec tick = {0, 50 * 1000 * 1000};
        futex(&s->state, FUTEX_WAIT, v, &tick);
        if (__atomic_load_n(&s->state, __ATOMIC_ACQUIRE) == IMPB_DONE) break;
        if (!alive(broker) || __atomic_load_n(&h->broker_pid, __ATOMIC_ACQUIRE) != broker ||
            __atomic_load_n(&h->epoch, __ATOMIC_ACQUIRE) != epoch) {
            /* the request died with the broker; the slot stays ours only if a new broker adopts the segment */
            __atomic_store_n(&s->state, (uint32_t)IMPB_CLAIMED, __ATOMIC_RELEASE);
            return fail("the broker went away while it held the request");
        }
        if (now_ms() - t0 > (double)c->timeout_ms) {
            /* The broker may still write into the slot: it is abandoned (left TAKEN/SUBMITTED with our pid, which the
             * broker frees once it is done with it or when this worker exits) and a fresh one is claimed next time. */
            __atomic_store_n(&s->owner_pid, 0u, __ATOMIC_RELEASE);     /* "abandoned": the broker frees it when it is done with it */
            c->slot = NULL;
            unmap(c);
            return fail("the broker did not answer in time (IMPGPU_BROKER_TIMEOUT_MS)");
        }
    }
    __atomic_store_n(&s->state, (uint32_t)IMPB_CLAIMED, __ATOMIC_RELEASE);
    return IMP_OK;
}

static int register_watermark(impgpu_client* c, impc_watermark* m) {
    impb_slot_fields* s = &c->slot->f;
    const size_t bytes = (size_t)m->step * (size_t)m->h;
    if (bytes > c->hdr->f.slot_data_bytes) { snprintf(t_err, sizeof t_err, "watermark larger than a slot"); return IMP_ERROR_MALLOC_FAILED; }
    memcpy(c->data, m->pixels, bytes);
    s->in_kind = IMPB_IN_WATERMARK; s->out_kind = IMPB_OUT_INFO;
    s->quality = 0; s->simple = 0; s->need_flatten = 0;          /* (nothing of the slot's last request rides along) */
    s->in_bytes = bytes; s->in_w = m->w; s->in_h = m->h; s->in_c = m->c; s->in_step = m->step;
    s->in_head_bytes = s->in_scan_at = s->in_scan_bytes = 0;
    s->filter_count = 0; s->crop_at = s->gravity_at = s->resize_at = s->ascii_at = -1; s->watermark_id = 0;
    const unsigned epoch = c->hdr->f.epoch;
    int rc = roundtrip(c);
    if (rc != IMP_OK) return rc;
    if (s->code != IMP_OK) { snprintf(t_err, sizeof t_err, "watermark refused: %.100s", s->error); return s->code; }
    m->broker_id = s->out_w;
    m->epoch = epoch;
    return IMP_OK;
}

int impgpu_client_prepare_watermark(impgpu_client* c, const unsigned char* pixels, int width, int height, int channels,
                                    int step, int* id) {
    if (!c || !pixels || !id || width <= 0 || height <= 0 || (channels != 1 && channels != 3 && channels != 4) ||
        (long long)step < (long long)width * channels)
        return IMP_ERROR_INVALID_ARGS;
    if (c->nmarks >= IMPC_MAX_WATERMARKS) { snprintf(t_err, sizeof t_err, "too many watermarks for one worker"); return IMP_ERROR_INVALID_ARGS; }
    int rc = ready(c);
    if (rc != IMP_OK) return rc;
    impc_watermark* m = &c->marks[c->nmarks];
    m->pixels = pixels; m->w = width; m->h = height; m->c = channels; m->step = step;
    rc = register_watermark(c, m);
    if (rc != IMP_OK) return rc;
    *id = ++c->nmarks;                              /* the client's own numbering: survives a broker restart */
    return IMP_OK;
}

/* (the marker walk of csrc/imp_jpeg.cpp's jpeg_parse, reduced to what decides "prepare or not", and jpeg_prepare_scan's
 * copy for a file without restart intervals; tests/test_broker_host.py holds the two against each other) */
int impgpu_jpeg_unstuff(const unsigned char* f, size_t size, unsigned char* out, size_t cap, size_t* head_bytes, size_t* scan_at,
                        size_t* scan_bytes, size_t* total_bytes) {
    if (!f || !out || size < 4 || f[0] != 0xFF || f[1] != 0xD8) return 0;
    size_t at = 2, scan_begin = 0;
    int frames = 0;
    while (!scan_begin) {
        if (at + 2 > size || f[at] != 0xFF) return 0;
        while (at < size && f[at] == 0xFF) at++;
        if (at >= size) return 0;
        const int marker = f[at++];
        if (marker == 0xD8 || marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
        if (marker == 0xD9 || at + 2 > size) return 0;
        const size_t len = ((size_t)f[at] << 8) | f[at + 1];
        if (len < 2 || len > size - at) return 0;
        if (marker == 0xC0 || marker == 0xC1) { if (frames++) return 0; }
        else if (marker >= 0xC0 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) return 0;   /* another process */
        else if (marker == 0xDD) { if (len != 4 || f[at + 2] || f[at + 3]) return 0; }                  /* a restart interval */
        else if (marker == 0xDA) { if (!frames) return 0; scan_begin = at + len; }
        at += len;
    }
    const size_t sa = (scan_begin + 255) & ~(size_t)255;
    if (size - scan_begin < IMPB_PREPARE_MIN_SCAN || sa > cap || size - scan_begin > cap - sa || cap - sa - (size - scan_begin) < IMPGPU_JPEG_SCAN_TAIL) return 0;
    unsigned char* o = out + sa;
    size_t n = 0;
    at = scan_begin;
    for (;;) {
        const unsigned char* ff = at < size ? (const unsigned char*)memchr(f + at, 0xFF, size - at) : NULL;
        const size_t run = (ff ? (size_t)(ff - f) : size) - at;
        memcpy(o + n, f + at, run);
        n += run; at += run;
        if (!ff) break;                                  /* no EOI: the MCU count decides */
        size_t m = at + 1;
        while (m < size && f[m] == 0xFF) m++;            /* fill bytes */
        if (m >= size) break;
        const int code = f[m];
        if (code == 0x00 && m == at + 1) { o[n++] = 0xFF; at = m + 1; }
        else if (code == 0x00 || (code >= 0xD0 && code <= 0xD7)) return 0;   /* FF FF 00, or RSTn without an interval: the library's verdict, on the file as it is */
        else break;                                      /* EOI, or whatever follows the scan */
    }
    if (!n) return 0;
    memcpy(out, f, scan_begin);
    memset(out + scan_begin, 0, sa - scan_begin);
    memset(o + n, 0xFF, IMPGPU_JPEG_SCAN_TAIL);
    *head_bytes = scan_begin; *scan_at = sa; *scan_bytes = n; *total_bytes = sa + n + IMPGPU_JPEG_SCAN_TAIL;
    return 1;
}

int impgpu_client_run(impgpu_client* c, const impgpu_client_request* r, impgpu_client_answer* a) {
    if (!c || !r || !a) return IMP_ERROR_INVALID_ARGS;
    memset(a, 0, sizeof *a);
    a->error = "";
    if (r->in_kind != IMPB_IN_FILE && r->in_kind != IMPB_IN_FRAME) return IMP_ERROR_INVALID_ARGS;
    if (r->out_kind < IMPB_OUT_JPEG || r->out_kind > IMPB_OUT_ASCII) return IMP_ERROR_INVALID_ARGS;
    if (r->watermark_id < 0 || r->watermark_id > c->nmarks) return IMP_ERROR_INVALID_ARGS;
    int rc = ready(c);
    if (rc != IMP_OK) return rc;
    int mark = 0;
    if (r->watermark_id) {
        impc_watermark* m = &c->marks[r->watermark_id - 1];
        if (m->epoch != c->hdr->f.epoch) {          /* the broker that knew it is gone */
            rc = register_watermark(c, m);
            if (rc != IMP_OK) return rc;
        }
        mark = m->broker_id;
    }
    impb_slot_fields* s = &c->slot->f;
    if (r->input_bytes > c->hdr->f.slot_data_bytes) { snprintf(t_err, sizeof t_err, "input larger than a slot"); return IMP_ERROR_MALLOC_FAILED; }
    s->in_bytes = r->input_bytes;
    s->in_head_bytes = s->in_scan_at = s->in_scan_bytes = 0;
    if (r->input) {
        /* a JPEG goes in with its scan out of the byte stuffing (the broker's lanes then make no pass over it: its bytes
         * go to the device from this slot); $IMPGPU_BROKER_PREPARE=0 copies every file as it is */
        size_t head = 0, sat = 0, sn = 0, total = 0;
        static int prepare = -1;
        if (prepare < 0) { const char* e = getenv("IMPGPU_BROKER_PREPARE"); prepare = !(e && e[0] == '0'); }
        if (prepare && r->in_kind == IMPB_IN_FILE &&
            impgpu_jpeg_unstuff(r->input, r->input_bytes, c->data, (size_t)c->hdr->f.slot_data_bytes, &head, &sat, &sn, &total)) {
            s->in_bytes = total; s->in_head_bytes = head; s->in_scan_at = sat; s->in_scan_bytes = sn;
        } else memcpy(c->data, r->input, r->input_bytes);
    }
    s->in_kind = (uint32_t)r->in_kind; s->out_kind = (uint32_t)r->out_kind;
    s->in_w = r->width; s->in_h = r->height; s->in_c = r->channels; s->in_step = r->step;
    s->quality = r->quality;
    size_t at = 0;
    s->crop_at = s->gravity_at = s->resize_at = s->ascii_at = -1;
    s->simple = s->need_flatten = s->filter_count = 0;
    if (r->out_kind == IMPB_OUT_ASCII && r->ascii_args) {
        s->ascii_at = put_text(s, &at, r->ascii_args);
        if (s->ascii_at == -2) { snprintf(t_err, sizeof t_err, "request text longer than %d bytes", IMPB_TEXT_BYTES); return IMP_ERROR_INVALID_ARGS; }
    }
    if (r->job) {
        const impgpu_job* j = r->job;
        if (j->filter_count < 0) return IMP_ERROR_INVALID_ARGS;
        if (j->filter_count > IMPB_MAX_FILTERS) return IMP_ERROR_TOO_MUCH_FILTERS;
        s->crop_at = put_text(s, &at, j->crop);
        s->gravity_at = put_text(s, &at, j->gravity);
        s->resize_at = put_text(s, &at, j->resize);
        int bad = s->crop_at == -2 || s->gravity_at == -2 || s->resize_at == -2;
        for (int i = 0; i < j->filter_count && !bad; i++) {
            s->filter_at[i] = put_text(s, &at, j->filters[i] ? j->filters[i] : "");
            bad = s->filter_at[i] == -2;
        }
        if (bad) { snprintf(t_err, sizeof t_err, "request text longer than %d bytes", IMPB_TEXT_BYTES); return IMP_ERROR_INVALID_ARGS; }
        s->simple = j->simple; s->need_flatten = j->need_flatten; s->filter_count = j->filter_count;
    }
    s->max_target_w = s->max_target_h = 0; s->max_filters_count = 0; s->allow_experiments = 0;
    s->watermark_id = mark;
    if (r->config) {
        const impgpu_config* g = r->config;
        s->max_target_w = g->max_target_w; s->max_target_h = g->max_target_h;
        s->max_filters_count = g->max_filters_count; s->allow_experiments = g->allow_experiments;
        s->watermark_opacity = g->watermark_opacity;
        s->watermark_offset_x = g->watermark_offset_x; s->watermark_offset_y = g->watermark_offset_y;
        s->watermark_gravity_x = g->watermark_gravity_x; s->watermark_gravity_y = g->watermark_gravity_y;
    }
    rc = roundtrip(c);
    if (rc != IMP_OK) return rc;
    const uint64_t cap = c->hdr->f.slot_data_bytes;
    if (s->out_offset > cap || s->out_bytes > cap - s->out_offset) { unmap(c); return fail("the broker's answer does not fit its slot"); }
    a->code = s->code; a->step = s->step;
    a->data = c->data + s->out_offset; a->bytes = (size_t)s->out_bytes;
    a->width = s->out_w; a->height = s->out_h; a->channels = s->out_c; a->row_step = s->out_step;
    a->brightness = s->brightness; a->batch_size = s->batch_size; a->broker_us = s->broker_us;
    s->error[sizeof s->error - 1] = 0;
    a->error = s->error;
    return IMP_OK;
}

int impgpu_client_stats(impgpu_client* c, unsigned long long* served, unsigned long long* batches, unsigned* epoch,
                        unsigned* broker_pid) {
    if (!c || !c->base) return IMP_ERROR_DEVICE;
    if (served) *served = c->hdr->f.served;
    if (batches) *batches = c->hdr->f.batches;
    if (epoch) *epoch = c->hdr->f.epoch;
    if (broker_pid) *broker_pid = c->hdr->f.broker_pid;
    return IMP_OK;
}
