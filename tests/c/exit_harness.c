/*
 * exit_harness.c -- a worker that ends WITHOUT impgpu_env_destroy: a stream, blocking-sync events, pinned rings and pool
 * blocks of two lanes are alive when it leaves.  impgpu_env_start registered impgpu_env_destroy with atexit(), so they
 * go back before the HIP runtime's own exit handlers run (round 4's review item: a live env at exit).
 *   exit_harness return|exit|busy
 * return: main returns; exit: exit(0) from inside a helper; busy: exit right behind an enqueued resize of a 4K frame,
 * without waiting for it.  Prints a line per step (flushed), the last one is "leaving".
 */
#define _POSIX_C_SOURCE 200809L
#include <impgpu.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void step(const char* s) { printf("%s\n", s); fflush(stdout); }

static void* other_lane(void* arg) {
    unsigned char* px = (unsigned char*)malloc(256 * 256 * 3);
    impgpu_image* im = NULL;
    (void)arg;
    memset(px, 7, 256 * 256 * 3);
    if (impgpu_image_upload(px, 256, 256, 3, 256 * 3, &im) != IMP_OK) return (void*)1;
    if (impgpu_cv_resize(&im, 64, 64, IMP_INTER_AREA) != IMP_OK) return (void*)1;
    if (impgpu_image_download(im, px, 64 * 3) != IMP_OK) return (void*)1;
    impgpu_image_release(&im);
    free(px);
    return NULL;
}

int main(int argc, char** argv) {
    const char* how = argc > 1 ? argv[1] : "return";
    unsigned char* px;
    impgpu_image *im = NULL, *kept = NULL;
    pthread_t th;
    void* res = NULL;
    if (impgpu_env_start(-1) != IMP_OK) { fprintf(stderr, "impgpu_env_start: %s\n", impgpu_last_error()); return 3; }
    step("env started");
    pthread_create(&th, NULL, other_lane, NULL);
    pthread_join(th, &res);
    if (res) return 4;
    step("second lane used and its thread gone");
    px = (unsigned char*)malloc((size_t)3840 * 2160 * 4);
    memset(px, 9, (size_t)3840 * 2160 * 4);
    if (impgpu_image_upload(px, 3840, 2160, 4, 3840 * 4, &im) != IMP_OK) return 4;
    if (impgpu_image_clone(im, &kept) != IMP_OK) return 4;                 /* a frame that is never released */
    if (!strcmp(how, "busy")) {
        int i;
        for (i = 0; i < 8; i++) { impgpu_image* t = NULL; if (impgpu_image_clone(im, &t) != IMP_OK || impgpu_cv_resize(&t, 1920, 1080, IMP_INTER_LANCZOS4) != IMP_OK) return 4; impgpu_image_release(&t); }
        step("work enqueued, not waited for");
    } else {
        if (impgpu_cv_resize(&im, 1920, 1080, IMP_INTER_AREA) != IMP_OK || impgpu_sync() != IMP_OK) return 4;
        step("request done");
    }
    step("leaving");
    if (!strcmp(how, "return")) return 0;
    exit(0);
}
