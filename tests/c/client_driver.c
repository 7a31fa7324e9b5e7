/*
 * client_driver.c -- glue/imp_gpu_client.c driven from C under AddressSanitizer / UBSan against tests/c/mock_broker.c:
 *   client_driver <name> <requests> [forks]
 * every request carries a different number of bytes, a job with strings and filters, and checks the mock's answer (the
 * bytes reversed, the filter count, the watermark id it was given, the crop string's length); `forks` children are forked
 * AFTER the attach and must claim slots of their own.  Prints "ok <requests done>".
 */
#define _POSIX_C_SOURCE 200809L
#include <impgpu_broker.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

static int drive(impgpu_client* c, int requests, unsigned seed) {
    unsigned char* buf = (unsigned char*)malloc(70000);
    const char* filters[3] = {"gamma=1.5", "rotate=90", "blur=2"};
    int done = 0;
    for (int i = 0; i < requests; i++) {
        const size_t n = (size_t)((seed * 2654435761u + (unsigned)i * 40503u) % 60000u) + 1;
        for (size_t k = 0; k < n; k++) buf[k] = (unsigned char)(k * 31 + (unsigned)i + seed);
        impgpu_job job;
        impgpu_config cfg;
        memset(&job, 0, sizeof job);
        memset(&cfg, 0, sizeof cfg);
        job.crop = "16,9,c,c"; job.resize = "224,0"; job.filters = filters; job.filter_count = 1 + i % 3;
        impgpu_client_request r;
        impgpu_client_answer a;
        memset(&r, 0, sizeof r);
        r.in_kind = IMPB_IN_FILE; r.input = buf; r.input_bytes = n; r.job = &job; r.config = &cfg; r.out_kind = IMPB_OUT_JPEG; r.quality = 86;
        if (impgpu_client_run(c, &r, &a) != IMP_OK) { fprintf(stderr, "run: %s\n", impgpu_client_last_error()); return -1; }
        if (a.code != IMP_OK || a.bytes != n || a.width != job.filter_count || a.channels != 8) { fprintf(stderr, "answer %d %zu %d %d\n", a.code, a.bytes, a.width, a.channels); return -1; }
        for (size_t k = 0; k < n; k++) if (a.data[k] != buf[n - 1 - k]) { fprintf(stderr, "byte %zu differs\n", k); return -1; }
        done++;
    }
    free(buf);
    return done;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int requests = atoi(argv[2]), forks = argc > 3 ? atoi(argv[3]) : 0;
    impgpu_client* c = NULL;
    if (impgpu_client_attach(argv[1], &c) != IMP_OK) { fprintf(stderr, "attach: %s\n", impgpu_client_last_error()); return 3; }
    int done = drive(c, requests, 1);
    if (done < 0) return 4;
    for (int f = 0; f < forks; f++) {
        const pid_t pid = fork();
        if (pid == 0) {                      /* a forked copy: the slot is the parent's, the child claims its own */
            const int d = drive(c, requests, 100u + (unsigned)f);
            impgpu_client_detach(&c);
            _exit(d == requests ? 0 : 5);
        }
    }
    int bad = 0;
    for (int f = 0; f < forks; f++) { int st = 0; wait(&st); if (!WIFEXITED(st) || WEXITSTATUS(st)) bad++; }
    const int again = drive(c, requests, 7);  /* the parent's slot is still its own */
    impgpu_client_detach(&c);
    if (bad || again != requests) { fprintf(stderr, "children failed: %d, parent again: %d\n", bad, again); return 6; }
    printf("ok %d\n", done + again);
    return 0;
}
