/*
 * mock_broker.c -- the BROKER side of include/impgpu_broker.h without a GPU, for the CPU tests of the worker side
 * (glue/imp_gpu_client.c): creates the segment, serves every submitted slot by answering with the input bytes reversed
 * (code IMP_OK, or what the request's "quality" field asks for: see below), and can misbehave on command:
 *   mock_broker <name> <slots> <slot_kb> [mode]
 *   mode = serve     (default) answer every request at once
 *          slow:<ms>           answer after <ms> milliseconds
 *          mute                take requests and never answer (the client must time out and abandon its slot)
 * quality 1000 + c in a request: answer with code c and step 4 instead of the bytes (an operator error coming back).
 * Writes "ready" to stdout when the segment is served.  SIGTERM: clean stop (broker_pid = 0, segment unlinked).
 */
#define _GNU_SOURCE
#include <impgpu_broker.h>
#include <errno.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

static volatile sig_atomic_t g_stop;
static void on_term(int s) { (void)s; g_stop = 1; }
static long futex(volatile uint32_t* a, int op, uint32_t v, const struct timespec* to) { return syscall(SYS_futex, a, op, v, to, NULL, 0); }

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s name slots slot_kb [serve|slow:ms|mute]\n", argv[0]); return 2; }
    const char* name = argv[1];
    const int nslots = atoi(argv[2]);
    const uint64_t slot_bytes = (uint64_t)atoi(argv[3]) << 10;
    const char* mode = argc > 4 ? argv[4] : "serve";
    const int mute = !strcmp(mode, "mute");
    const long slow_ms = !strncmp(mode, "slow:", 5) ? atol(mode + 5) : 0;
    const uint64_t slots_offset = sizeof(impb_header), data_offset = slots_offset + (uint64_t)nslots * sizeof(impb_slot);
    const uint64_t total = data_offset + (uint64_t)nslots * slot_bytes;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_term;
    sigaction(SIGTERM, &sa, NULL);
    shm_unlink(name);
    int fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)total) != 0) { perror("shm"); return 3; }
    uint8_t* base = (uint8_t*)mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { perror("mmap"); return 3; }
    impb_header_fields* h = &((impb_header*)base)->f;
    impb_slot* slots = (impb_slot*)(base + slots_offset);
    h->magic = IMPB_MAGIC; h->version = IMPB_VERSION; h->nslots = (uint32_t)nslots;
    h->slot_data_bytes = slot_bytes; h->slots_offset = slots_offset; h->data_offset = data_offset;
    h->epoch = 1;
    __atomic_store_n(&h->broker_pid, (uint32_t)getpid(), __ATOMIC_RELEASE);
    printf("ready\n");
    fflush(stdout);
    while (!g_stop) {
        const uint32_t bell = __atomic_load_n(&h->doorbell, __ATOMIC_SEQ_CST);
        int served = 0;
        for (int i = 0; i < nslots; i++) {
            impb_slot_fields* s = &slots[i].f;
            uint32_t expect = IMPB_SUBMITTED;
            if (!__atomic_compare_exchange_n(&s->state, &expect, (uint32_t)IMPB_TAKEN, 0, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) continue;
            served++;
            if (mute) continue;
            if (slow_ms) { struct timespec nap = {slow_ms / 1000, (slow_ms % 1000) * 1000000L}; nanosleep(&nap, NULL); }
            uint8_t* data = base + data_offset + (uint64_t)i * slot_bytes;
            const uint64_t n = s->in_bytes <= slot_bytes / 2 ? s->in_bytes : 0, at = (n + 63) & ~(uint64_t)63;
            s->step = IMP_STEP_ENCODE;
            if (s->quality >= 1000) { s->code = s->quality - 1000; s->step = IMP_STEP_RESIZE; s->out_bytes = 0; s->out_offset = 0; snprintf(s->error, sizeof s->error, "asked for"); }
            else if (s->in_kind == IMPB_IN_WATERMARK) { s->code = IMP_OK; s->out_w = 1 + i; s->out_bytes = 0; s->out_offset = 0; }
            else {
                for (uint64_t k = 0; k < n; k++) data[at + k] = data[n - 1 - k];
                s->code = IMP_OK; s->out_offset = at; s->out_bytes = n;
                s->out_w = s->filter_count; s->out_h = s->watermark_id; s->out_c = s->crop_at >= 0 ? (int)strlen(s->text + s->crop_at) : -1;
            }
            s->batch_size = 1;
            __atomic_add_fetch(&h->served, (uint64_t)1, __ATOMIC_RELAXED);
            if (__atomic_load_n(&s->owner_pid, __ATOMIC_ACQUIRE) == 0) { __atomic_store_n(&s->state, (uint32_t)IMPB_FREE, __ATOMIC_RELEASE); continue; }
            __atomic_store_n(&s->state, (uint32_t)IMPB_DONE, __ATOMIC_RELEASE);
            futex(&s->state, FUTEX_WAKE, 1, NULL);
        }
        if (!served) {
            __atomic_add_fetch(&h->sleepers, 1u, __ATOMIC_SEQ_CST);
            struct timespec tick = {0, 20 * 1000 * 1000};
            futex(&h->doorbell, FUTEX_WAIT, bell, &tick);
            __atomic_sub_fetch(&h->sleepers, 1u, __ATOMIC_SEQ_CST);
        }
    }
    __atomic_store_n(&h->broker_pid, 0u, __ATOMIC_RELEASE);
    shm_unlink(name);
    return 0;
}
