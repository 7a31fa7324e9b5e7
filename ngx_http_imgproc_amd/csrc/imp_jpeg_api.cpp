// imp_jpeg_api.cpp -- impgpu_image_decode_jpeg: the reference's cvDecodeImage(&rawencoded, -1) for a JPEG blob
// (bridge.c:545-552) with everything but marker parsing and FF00 unstuffing on the device (imp_jpeg.h).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "imp_jpeg.h"

using namespace imp;

namespace {

enum HuffMode { HUFF_DEVICE = 0, HUFF_HOST = 1 };
HuffMode huff_mode() {
    // A/B switch, read per call (a getenv is nothing next to a decode): "host" = entropy decoding on the calling thread
    const char* s = std::getenv("IMPGPU_JPEG_HUFF");
    return (s && !std::strcmp(s, "host")) ? HUFF_HOST : HUFF_DEVICE;
}

// IMPGPU_JPEG_TRACE=1: one line per decode on stderr with the host's share of it, in microseconds
struct Stopwatch {
    bool on;
    std::chrono::steady_clock::time_point t0;
    double marks[8] = {};
    int n = 0;
    Stopwatch() : on(std::getenv("IMPGPU_JPEG_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark() {
        if (!on || n >= 8) return;
        const auto t = std::chrono::steady_clock::now();
        marks[n++] = std::chrono::duration<double, std::micro>(t - t0).count();
        t0 = t;
    }
};

}  // namespace

extern "C" {

int impgpu_image_decode_jpeg(const unsigned char* blob, size_t size, impgpu_image** out) {
    if (!blob || !out) return IMP_ERROR_INVALID_ARGS;
    if (!env_ready()) { set_error("impgpu_env_start has not been called", hipErrorNotInitialized); return IMP_ERROR_DEVICE; }
    TraceRange tr("IMP_STEP_DECODE");
    IMP_FAULT_POINT(IMP_STEP_DECODE);
    Stopwatch sw;
    JpegHeader H;
    if (int rc = jpeg_parse(blob, size, &H)) return rc;
    if (!frame_fits(H.width, H.height, H.ncomp)) return IMP_ERROR_UNSUPPORTED;
    JpegFrame F;
    int dc_ids[2], ac_ids[2];
    if (int rc = jpeg_frame_setup(H, &F, dc_ids, ac_ids)) return rc;
    hipStream_t s = env_stream();
    if (!s) return IMP_ERROR_DEVICE;

    // quantisation tables of the three components, natural order
    uint16_t qt3[3][64] = {};
    for (int i = 0; i < H.ncomp; i++) std::memcpy(qt3[i], H.qt[H.comp[i].tq], sizeof qt3[i]);
    void* d_qt = nullptr;
    if (int rc = upload_small(qt3, sizeof qt3, &d_qt, s)) return rc;

    void* d_coef = nullptr;
    const size_t coef_bytes = (size_t)F.total_slots * sizeof(int16_t);
    int rc = dev_alloc(coef_bytes, &d_coef);
    if (rc) { dev_free(d_qt); return rc; }

    sw.mark();                                                      // [0] parse + small uploads
    uint32_t* status = lane_mailbox();                              // pinned: the verdict's copy stays asynchronous
    if (!status) { dev_free(d_coef); dev_free(d_qt); return IMP_ERROR_DEVICE; }
    status[0] = status[1] = status[2] = status[3] = 0;
    const bool on_device = huff_mode() == HUFF_DEVICE;
    if (!on_device) {
        // A/B path: entropy decoding on this thread, dense coefficient planes over the link
        void* host = nullptr;
        void* token = nullptr;
        rc = stage_begin(coef_bytes, &host, &token);
        if (!rc) {
            std::memset(host, 0, coef_bytes);
            rc = jpeg_host_entropy(blob, size, H, (int16_t*)host, F);
            const int rc2 = stage_upload(token, d_coef, rc ? 0 : coef_bytes);
            if (!rc) rc = rc2;
        }
    } else {
        void *d_tabs = nullptr, *d_words = nullptr, *d_meta = nullptr, *d_ctl = nullptr;
        {
            std::vector<JpegHuffDev> tabs(4);
            rc = jpeg_build_tables(H, dc_ids, ac_ids, tabs.data());
            if (!rc) rc = upload_small(tabs.data(), 4 * sizeof(JpegHuffDev), &d_tabs, s);
        }
        JpegScan scan;
        const size_t total_mcus = (size_t)H.mcux * H.mcuy;
        const size_t nsegs = H.restart_interval ? (total_mcus + H.restart_interval - 1) / H.restart_interval : 1;
        const size_t cap = jpeg_scan_capacity(size - H.scan_begin, nsegs);
        void* host = nullptr;
        void* token = nullptr;
        if (!rc) rc = stage_begin(cap, &host, &token);
        if (!rc) {
            // the only pass the host makes over the compressed bytes: FF00 unstuffing while they are copied into pinned memory
            sw.mark();                                              // [1] tables + staging
            rc = jpeg_prepare_scan(blob, size, H, (uint8_t*)host, cap, &scan);
            sw.mark();                                              // [2] unstuffing copy
            const size_t bytes = rc ? 0 : (scan.nchunks + 1) * JPEG_CHUNK_BYTES;
            if (!rc) rc = dev_alloc(bytes, &d_words);
            const int rc2 = stage_upload(token, d_words, rc ? 0 : bytes);
            if (!rc) rc = rc2;
        }
        if (!rc) {
            F.nchunks = (unsigned)scan.nchunks;
            F.nsegs = (unsigned)scan.seg_first_chunk.size();
            std::vector<uint32_t> meta;
            jpeg_scan_meta(scan, &meta);
            rc = upload_small(meta.data(), meta.size() * sizeof(uint32_t), &d_meta, s);
        }
        const size_t ctl_bytes = jpeg_control_bytes(F.nchunks);
        if (!rc) rc = dev_alloc(ctl_bytes, &d_ctl);
        if (!rc && hipMemsetAsync(d_ctl, 0, ctl_bytes, s) != hipSuccess) { set_error("hipMemsetAsync(jpeg control)", hipGetLastError()); rc = IMP_ERROR_DEVICE; }
        if (!rc && hipMemsetAsync(d_coef, 0, coef_bytes, s) != hipSuccess) { set_error("hipMemsetAsync(jpeg coefficients)", hipGetLastError()); rc = IMP_ERROR_DEVICE; }
        if (!rc) {
            JpegHuffArgs A;
            A.words = (const uint32_t*)d_words;
            A.chunk_seg = (const uint32_t*)d_meta;
            A.seg_first_chunk = A.chunk_seg + F.nchunks;
            A.seg_bits = A.seg_first_chunk + F.nsegs;
            A.tables = (const JpegHuffDev*)d_tabs;
            A.coef = (int16_t*)d_coef;
            A.control = (uint32_t*)d_ctl;
            rc = launch_jpeg_entropy(F, A, s);
        }
        if (!rc) {
            // the kernel's verdict (did every interval decode to exactly its MCUs?) is read before the frame is handed on
            const hipError_t e = hipMemcpyAsync(status, (uint32_t*)d_ctl, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
            if (e != hipSuccess) { set_error("hipMemcpyAsync(jpeg status)", e); rc = IMP_ERROR_DEVICE; }
        }
        dev_free(d_tabs);
        dev_free(d_words);
        dev_free(d_meta);
        dev_free(d_ctl);
    }
    impgpu_image* im = nullptr;
    if (!rc) rc = image_new(H.width, H.height, H.ncomp, &im);
    if (!rc) rc = launch_jpeg_pixels(F, (const int16_t*)d_coef, (const uint16_t*)d_qt, im->d, im->step, s);
    dev_free(d_coef);
    dev_free(d_qt);
    sw.mark();                                                      // [3] enqueue
    if (!rc && on_device) {
        rc = lane_wait();
        if (!rc && status[1]) {
            char text[96];
            std::snprintf(text, sizeof text, "jpeg entropy stage refused the scan (status 0x%x)", status[1]);
            set_error_text(text);
            rc = IMP_ERROR_DECODE_FAILED;
        }
    }
    sw.mark();                                                      // [4] wait for the verdict
    if (sw.on)
        std::fprintf(stderr, "jpeg %dx%d %zu B: parse %.0f tables %.0f unstuff %.0f enqueue %.0f wait %.0f us; rounds %u + %u, status %u\n", H.width,
                     H.height, size, sw.marks[0], sw.marks[1], sw.marks[2], sw.marks[3], sw.marks[4], status[2], status[3], status[1]);
    if (rc) { image_delete(im); return rc; }
    *out = im;
    return IMP_OK;
}

}  // extern "C"
