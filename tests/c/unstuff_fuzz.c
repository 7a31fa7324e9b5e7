/*
 * unstuff_fuzz.c -- impgpu_jpeg_unstuff (glue/imp_gpu_client.c: what a worker does to a request body on its way into shared
 * memory) under AddressSanitizer / UBSan, on mutated files and output buffers of exactly the capacity it is told:
 *   unstuff_fuzz <file.jpg> <iterations> [seed]
 * Every iteration damages a copy of the file (cut short, bytes flipped, marker-like pairs dropped in, runs of FF), picks a
 * capacity between 0 and a little more than the file, and calls the function on heap blocks of exactly those sizes.  A
 * result of 1 is checked: offsets inside the capacity, the scan 256-byte aligned, the tail all FF, the head the file's own
 * bytes, and the scan equal to a byte-by-byte unstuffing written here.  Prints "ok <taken> of <iterations>".
 */
#include <impgpu_broker.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned long long s_rng = 88172645463325252ull;
static unsigned rnd(void) { s_rng ^= s_rng << 13; s_rng ^= s_rng >> 7; s_rng ^= s_rng << 17; return (unsigned)(s_rng >> 11); }

/* the scan bytes the plain way: from `begin` on, FF 00 -> FF, FF FF.. fill dropped, stop at any other marker or the end */
static size_t naive(const unsigned char* f, size_t size, size_t begin, unsigned char* out) {
    size_t n = 0, i = begin;
    while (i < size) {
        if (f[i] != 0xFF) { out[n++] = f[i++]; continue; }
        size_t j = i + 1;
        while (j < size && f[j] == 0xFF) j++;
        if (j >= size) break;
        if (f[j] == 0 && j == i + 1) { out[n++] = 0xFF; i = j + 1; }
        else break;
    }
    return n;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* fp = fopen(argv[1], "rb");
    if (!fp) return 3;
    fseek(fp, 0, SEEK_END);
    const long len = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    unsigned char* orig = (unsigned char*)malloc((size_t)len);
    if (fread(orig, 1, (size_t)len, fp) != (size_t)len) return 3;
    fclose(fp);
    const int iterations = atoi(argv[2]);
    if (argc > 3) s_rng ^= (unsigned long long)atoll(argv[3]) * 0x9E3779B97F4A7C15ull;
    int taken = 0;
    for (int it = 0; it < iterations; it++) {
        size_t size = (size_t)len;
        const unsigned how = rnd() % 8;
        if (how == 1) size = rnd() % ((unsigned)len + 1);                       /* cut anywhere */
        else if (how == 2) size = (size_t)len - rnd() % 4096;                   /* cut in the scan */
        unsigned char* f = (unsigned char*)malloc(size ? size : 1);             /* exactly `size` bytes: a read past it is caught */
        memcpy(f, orig, size);
        if (how == 3 && size > 700) for (int k = 0; k < 4; k++) f[600 + rnd() % (size - 600)] = (unsigned char)rnd();
        if (how == 4 && size > 8) for (int k = 0; k < 6; k++) f[rnd() % size] = (unsigned char)rnd();          /* headers too */
        if (how == 5 && size > 700) { const size_t at = 600 + rnd() % (size - 602); f[at] = 0xFF; f[at + 1] = (unsigned char)(0xC0 + rnd() % 64); }
        if (how == 6 && size > 700) { const size_t at = 600 + rnd() % (size - 640); memset(f + at, 0xFF, 1 + rnd() % 32); }
        if (how == 7 && size > 40) { f[20 + rnd() % 16] = 0xFF; f[2 + rnd() % 30] = (unsigned char)rnd(); }   /* segment lengths */
        const size_t cap = (rnd() % 4 == 0) ? rnd() % (size + 1) : size + rnd() % 2048;
        unsigned char* out = (unsigned char*)malloc(cap ? cap : 1);
        size_t head = 0, at = 0, n = 0, total = 0;
        const int r = impgpu_jpeg_unstuff(f, size, out, cap, &head, &at, &n, &total);
        if (r != 0 && r != 1) { fprintf(stderr, "iteration %d: returned %d\n", it, r); return 4; }
        if (r == 1) {
            taken++;
            int bad = !(head >= 4 && head <= at && (at & 255) == 0 && n >= 1 && total == at + n + IMPGPU_JPEG_SCAN_TAIL && total <= cap && head <= size);
            if (!bad) bad = memcmp(out, f, head) != 0;
            for (size_t k = 0; !bad && k < IMPGPU_JPEG_SCAN_TAIL; k++) bad = out[at + n + k] != 0xFF;
            if (!bad) {
                unsigned char* want = (unsigned char*)malloc(size + 1);
                const size_t wn = naive(f, size, head, want);
                bad = wn != n || memcmp(want, out + at, n) != 0;
                free(want);
            }
            if (bad) { fprintf(stderr, "iteration %d (mutation %u, size %zu, cap %zu): head %zu at %zu n %zu total %zu do not hold\n", it, how, size, cap, head, at, n, total); return 5; }
        }
        free(out);
        free(f);
    }
    printf("ok %d of %d\n", taken, iterations);
    free(orig);
    return 0;
}
