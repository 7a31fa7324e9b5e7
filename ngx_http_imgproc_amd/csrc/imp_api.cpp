// imp_api.cpp -- the operator entry points of include/impgpu.h: each takes the reference's
// argument string, decides everything the reference decides on the CPU before touching pixels
// (imp_args.cpp), then enqueues kernels on the env stream.  impgpu_run_ops is the operator
// segment of RunJob (bridge.c:574-656) with Crop folded into the next operator's source view
// and runs of pointwise filters fused into one launch.
#include <cstddef>
#include <cstring>
#include "imp_internal.h"

using namespace imp;

namespace {

int need_env() {
    if (!env_ready()) { set_error("impgpu_env_start has not been called", hipErrorNotInitialized); return IMP_ERROR_DEVICE; }
    return IMP_OK;
}

Frames one_frame(const View& v, impgpu_image* dst) {
    Frames f{};
    f.src = v.d; f.src_stride = 0; f.v = v;
    f.dst = dst->d; f.dst_stride = 0; f.dw = dst->w; f.dh = dst->h; f.dstep = dst->step;
    f.count = 1;
    return f;
}

// The image -- or album -- being worked on: `owner` holds the memory, `v` is the window of its frame 0 that is the
// current frame (smaller than owner after a folded Crop); frame i of an album is the same window `stride()` bytes on.
struct Work {
    impgpu_image* owner;
    View v;
    int count() const { return owner->frames; }
    long long stride() const { return (long long)owner->fstride; }
    bool is_view() const { return v.d != owner->d || v.w != owner->w || v.h != owner->h; }
    void adopt(impgpu_image* im) {
        image_delete(owner);
        owner = im;
        v = view_of(im);
    }
    // a fresh destination with this work's frame count
    int fresh(int w, int h, int c, impgpu_image** out) const { return image_new_album(w, h, c, owner->frames, out); }
    Frames to(impgpu_image* dst) const {
        Frames f = one_frame(v, dst);
        f.src_stride = stride(); f.dst_stride = (long long)dst->fstride; f.count = owner->frames;
        return f;
    }
    uint8_t* px() const { return const_cast<uint8_t*>(v.d); }
};

int materialize(Work& wk) {
    if (!wk.is_view()) return IMP_OK;
    impgpu_image* out = nullptr;
    if (int rc = wk.fresh(wk.v.w, wk.v.h, wk.v.c, &out)) return rc;
    if (int rc = launch_copy(wk.to(out), env_stream())) { image_delete(out); return rc; }
    wk.adopt(out);
    return IMP_OK;
}

int do_resize(Work& wk, int w, int h, int interp) {
    impgpu_image* out = nullptr;
    if (int rc = wk.fresh(w, h, wk.v.c, &out)) return rc;
    if (int rc = launch_cv_resize(wk.to(out), interp, env_stream())) { image_delete(out); return rc; }
    wk.adopt(out);
    return IMP_OK;
}

int flush_program(Work& wk, PixelProgram& prog) {
    if (prog.empty()) return IMP_OK;
    int rc = launch_pixel_program(wk.px(), wk.stride(), wk.v.w, wk.v.h, wk.v.c, wk.v.step, wk.count(), prog, env_stream());
    prog.clear();
    return rc;
}

int apply_plan(Work& wk, const FilterPlan& plan) {
    switch (plan.cls) {
        case FC_FLIP: {
            impgpu_image* out = nullptr;
            if (int rc = wk.fresh(wk.v.w, wk.v.h, wk.v.c, &out)) return rc;
            if (int rc = launch_flip(wk.to(out), plan.flip_mode, env_stream())) { image_delete(out); return rc; }
            wk.adopt(out);
            return IMP_OK;
        }
        case FC_ROTATE: {
            impgpu_image* out = nullptr;
            const bool swap = plan.rotate != 180;
            if (int rc = wk.fresh(swap ? wk.v.h : wk.v.w, swap ? wk.v.w : wk.v.h, wk.v.c, &out)) return rc;
            if (int rc = launch_rotate(wk.to(out), plan.rotate, env_stream())) { image_delete(out); return rc; }
            wk.adopt(out);
            return IMP_OK;
        }
        case FC_BLUR: {
            if (wk.v.c == 4 || wk.v.c == 3) {      // one-pass fused kernel into a fresh frame; falls through when it does not apply
                impgpu_image* out = nullptr;
                if (int rc = wk.fresh(wk.v.w, wk.v.h, wk.v.c, &out)) return rc;
                int rc = launch_gaussian_fused(wk.to(out), plan.sigma, env_stream());
                if (rc == IMP_OK) { wk.adopt(out); return IMP_OK; }
                image_delete(out);
                if (rc != IMP_ERROR_UNSUPPORTED) return rc;
            }
            return launch_gaussian(wk.px(), wk.stride(), wk.v.w, wk.v.h, wk.v.c, wk.v.step, wk.count(), plan.sigma, env_stream());
        }
        default:
            return IMP_OK;
    }
}

int do_filter(Work& wk, const char* request, int allow, PixelProgram& prog) {
    FilterPlan plan;
    if (int rc = filter_plan(request, allow, wk.v.c, wk.v.w, wk.v.h, &plan, &prog)) return rc;
    if (plan.cls == FC_POINTWISE || plan.cls == FC_NOOP) return IMP_OK;     // stays queued in prog
    if (int rc = flush_program(wk, prog)) return rc;
    return apply_plan(wk, plan);
}

int do_watermark(Work& wk, const impgpu_config* cfg) {
    const impgpu_image* ov = cfg->watermark;
    if (wk.v.c < 3 || ov->c < 3) return IMP_ERROR_INVALID_ARGS;   // reference indexes B,G,R unconditionally
    int rx, ry, maxcol, maxrow;
    if (int rc = watermark_rect(wk.v.w, wk.v.h, ov->w, ov->h, cfg, &rx, &ry, &maxcol, &maxrow)) return rc;
    const float opacity = (float)(cfg->watermark_opacity / 100.0);   // bridge.c:275
    const float alpha = 1 - opacity;                                  // filters.c:620
    return launch_blend_over(wk.px(), wk.stride(), wk.v.w, wk.v.h, wk.v.c, wk.v.step, wk.count(), ov,
                             rx, ry, maxcol, maxrow, alpha, env_stream());
}

}  // namespace

extern "C" {

int impgpu_crop_geometry(int width, int height, const char* args, const char* gravity, int* x, int* y, int* w, int* h) {
    if (!args || !x || !y || !w || !h) return IMP_ERROR_INVALID_ARGS;
    return crop_geometry(width, height, args, gravity, x, y, w, h);
}

int impgpu_resize_geometry(int width, int height, const char* args, const impgpu_config* config, int simple,
                           int* w, int* h, int* interpolation) {
    if (!args || !w || !h || !interpolation) return IMP_ERROR_INVALID_ARGS;
    return resize_geometry(width, height, args, config ? config->max_target_w : 0, config ? config->max_target_h : 0,
                           simple, w, h, interpolation);
}

int impgpu_filter_check(const char* request, int allow_experiments) {
    if (!request) return IMP_ERROR_INVALID_ARGS;
    FilterPlan plan;
    PixelProgram prog;
    return filter_plan(request, allow_experiments, 4, 64, 64, &plan, &prog);
}

int impgpu_check_destructive(const char* request) { return check_destructive(request); }

int impgpu_image_clone(const impgpu_image* src, impgpu_image** out) {
    if (!src || !out) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    impgpu_image* im = nullptr;
    Work wk{const_cast<impgpu_image*>(src), view_of(src)};
    if (int rc = wk.fresh(src->w, src->h, src->c, &im)) return rc;
    if (int rc = launch_copy(wk.to(im), env_stream())) { image_delete(im); return rc; }
    *out = im;
    return IMP_OK;
}

int impgpu_crop(impgpu_image** pointer, const char* args, const char* gravity) {
    if (!pointer || !*pointer || !args) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    int x, y, w, h;
    if (int rc = crop_geometry((*pointer)->w, (*pointer)->h, args, gravity, &x, &y, &w, &h)) return rc;
    Work wk{*pointer, view_sub(view_of(*pointer), x, y, w, h)};
    int rc = IMP_OK;
    if (wk.is_view()) rc = materialize(wk);
    else {   // full-frame crop still yields a fresh image in the reference; the pixels are identical
    }
    *pointer = wk.owner;
    return rc;
}

int impgpu_resize(impgpu_image** pointer, const char* args, const impgpu_config* config, int simple) {
    if (!pointer || !*pointer || !args) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    int w, h, interp;
    if (int rc = resize_geometry((*pointer)->w, (*pointer)->h, args, config ? config->max_target_w : 0,
                                 config ? config->max_target_h : 0, simple, &w, &h, &interp))
        return rc;
    Work wk{*pointer, view_of(*pointer)};
    int rc = do_resize(wk, w, h, interp);
    *pointer = wk.owner;
    return rc;
}

int impgpu_cv_resize(impgpu_image** pointer, int width, int height, int interpolation) {
    if (!pointer || !*pointer || width <= 0 || height <= 0) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    Work wk{*pointer, view_of(*pointer)};
    int rc = do_resize(wk, width, height, interpolation);
    *pointer = wk.owner;
    return rc;
}

int impgpu_filter(impgpu_image** pointer, const char* request, int allow_experiments) {
    if (!pointer || !*pointer || !request) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    Work wk{*pointer, view_of(*pointer)};
    PixelProgram prog;
    int rc = do_filter(wk, request, allow_experiments, prog);
    if (!rc) rc = flush_program(wk, prog);
    *pointer = wk.owner;
    return rc;
}

int impgpu_prepare_watermark(impgpu_config* config, const unsigned char* pixels, int width, int height,
                             int channels, int step) {
    if (!config || !pixels) return IMP_ERROR_NO_SUCH_WATERMARK;
    if (int rc = need_env()) return rc;
    impgpu_image* im = nullptr;
    if (int rc = impgpu_image_upload(pixels, width, height, channels, step, &im)) return rc == IMP_ERROR_INVALID_ARGS ? IMP_ERROR_NO_SUCH_WATERMARK : rc;
    if (config->watermark) image_delete(config->watermark);
    config->watermark = im;
    return impgpu_sync();   // the overlay is read by every lane's stream afterwards: make the upload complete now
}

int impgpu_watermark(impgpu_image* image, const impgpu_config* config) {
    if (!image || !config || !config->watermark) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    Work wk{image, view_of(image)};
    return do_watermark(wk, config);
}

int impgpu_blend_with_paper(impgpu_image* image) {
    if (!image) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    if (image->c != 4) return IMP_ERROR_INVALID_ARGS;   // reference reads channel 3 unconditionally; RunJob only calls it for 4 channels
    return launch_blend_paper(image->d, (long long)image->fstride, image->w, image->h, image->step, image->frames, env_stream());
}

static int gif_compose(const impgpu_gif_page* pages, int count, int destructive, int page, bool as_album, impgpu_image** frames) {
    if (!pages || !frames || count <= 0 || page < -1) return IMP_ERROR_INVALID_ARGS;   // page < -1: the reference reads Frames[page] below the array
    if (page != -1) {                                              // advancedio.c:111-116: a page request is always a
        destructive = 1;                                           // destructive walk, and a page past the end is page 0
        if (page > count - 1) page = 0;
    }
    if (int rc = need_env()) return rc;
    const int cw = pages[0].width, ch = pages[0].height;           // advancedio.c:133-136: canvas = first page
    if (cw <= 0 || ch <= 0) return IMP_ERROR_INVALID_ARGS;
    const int npages = page >= 0 ? page + 1 : count;               // the walk stops at the requested page (:249-251)
    const int nout = page >= 0 ? 1 : count;
    // one blob: page table | output pointers | palettes | index planes
    std::vector<GifPageDev> meta((size_t)npages);
    size_t off = (size_t)npages * sizeof(GifPageDev) + (size_t)nout * sizeof(uint8_t*);
    off = (off + 15) & ~size_t(15);
    for (int f = 0; f < npages; f++) {
        const impgpu_gif_page& p = pages[f];
        if (!p.indices || !p.palette || p.width <= 0 || p.height <= 0 || p.pitch < p.width) return IMP_ERROR_INVALID_ARGS;
        meta[f].pal_off = (long long)off; off += 1024;
        meta[f].idx_off = (long long)off; off += ((size_t)p.pitch * p.height + 15) & ~size_t(15);
        meta[f].w = p.width; meta[f].h = p.height; meta[f].pitch = p.pitch; meta[f].left = p.left; meta[f].top = p.top;
        meta[f].dispose = p.dispose; meta[f].key = p.transparency_key; meta[f].pad = 0;
    }
    std::vector<impgpu_image*> imgs((size_t)nout, nullptr);
    auto drop = [&]() { for (impgpu_image* im : imgs) if (im) image_delete(im); };
    if (as_album) {                                                // every output frame in one block behind one handle
        imgs.resize(1);
        if (int rc = image_new_album(cw, ch, 4, nout, &imgs[0])) return rc;
    } else {
        for (int i = 0; i < nout; i++)
            if (int rc = image_new(cw, ch, 4, &imgs[i])) { drop(); return rc; }
    }
    std::vector<uint8_t> blob(off, 0);
    std::memcpy(blob.data(), meta.data(), (size_t)npages * sizeof(GifPageDev));
    uint8_t** optr = (uint8_t**)(blob.data() + (size_t)npages * sizeof(GifPageDev));
    for (int i = 0; i < nout; i++) optr[i] = as_album ? imgs[0]->d + (size_t)i * imgs[0]->fstride : imgs[i]->d;
    for (int f = 0; f < npages; f++) {
        std::memcpy(blob.data() + meta[f].pal_off, pages[f].palette, 1024);
        std::memcpy(blob.data() + meta[f].idx_off, pages[f].indices, (size_t)pages[f].pitch * pages[f].height);
    }
    void* dev = nullptr;
    hipStream_t s = env_stream();
    if (int rc = upload_small(blob.data(), blob.size(), &dev, s)) { drop(); return rc; }
    const uint8_t* d = (const uint8_t*)dev;
    int rc = launch_gif_compose(d, (const GifPageDev*)d, (uint8_t* const*)(d + (size_t)npages * sizeof(GifPageDev)), npages,
                                cw, ch, imgs[0]->step, destructive ? 1 : 0, page, s);
    dev_free(dev);                                                 // stream-ordered: after the kernel
    if (rc) { drop(); return rc; }
    for (size_t i = 0; i < imgs.size(); i++) frames[i] = imgs[i];
    return IMP_OK;
}

int impgpu_gif_compose(const impgpu_gif_page* pages, int count, int destructive, int page, impgpu_image** frames) {
    return gif_compose(pages, count, destructive, page, false, frames);
}

int impgpu_gif_compose_album(const impgpu_gif_page* pages, int count, int destructive, int page, impgpu_image** album) {
    return gif_compose(pages, count, destructive, page, true, album);
}

int impgpu_calc_perceived_brightness(const impgpu_image* image, float* brightness) {
    if (!image || !brightness) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    TraceRange tr("IMP_STEP_INFO");                                    // bridge.c:659-666
    IMP_FAULT_POINT(IMP_STEP_INFO);
    return launch_brightness(view_of(image), brightness, env_stream());
}

int impgpu_ascii(impgpu_image* image, const char* args, unsigned char* out, long capacity, long* length) {
    static const unsigned char wide[] = "$@B%8&WM#*oahkbdpqwmZO0QLCJUYXzcvunxrjft/\\|()1{}[]?-_+~<>i!lI;:,\"^`'. ";
    static const unsigned char narrow[] = "@%8#*+=-:. ";
    if (!image || !out || !length) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    if (image->c < 3) return IMP_ERROR_INVALID_ARGS;
    const unsigned char* table = (args && !std::strcmp(args, "wide")) ? wide : narrow;
    const int tablelen = (int)std::strlen((const char*)table);
    const float factor = (float)(256.0 / tablelen);
    const long buflen = (long)(image->w + 1) * image->h - 1;
    if (capacity < buflen) return IMP_ERROR_INVALID_ARGS;
    void *dev_table = nullptr, *dev_out = nullptr;
    if (int rc = upload_small(table, (size_t)tablelen + 1, &dev_table, env_stream())) return rc;
    if (int rc = dev_alloc((size_t)buflen + 1, &dev_out)) { dev_free(dev_table); return rc; }
    int rc = launch_ascii(image->d, image->w, image->h, image->c, image->step, (const uint8_t*)dev_table, tablelen, factor,
                          (uint8_t*)dev_out, env_stream());
    if (!rc) {
        hipError_t e = hipMemcpyAsync(out, dev_out, (size_t)buflen, hipMemcpyDeviceToHost, env_stream());
        if (e == hipSuccess) e = hipStreamSynchronize(env_stream());
        if (e != hipSuccess) { set_error("ascii readback", e); rc = IMP_ERROR_DEVICE; }
    }
    dev_free(dev_table);
    dev_free(dev_out);
    if (!rc) *length = buflen;
    return rc;
}

int impgpu_gray2bgr(impgpu_image** pointer) {
    if (!pointer || !*pointer) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    if ((*pointer)->c != 1) return IMP_OK;
    impgpu_image* out = nullptr;
    Work wk{*pointer, view_of(*pointer)};
    if (int rc = wk.fresh(wk.v.w, wk.v.h, 3, &out)) return rc;
    if (int rc = launch_gray2bgr(wk.to(out), env_stream())) { image_delete(out); return rc; }
    wk.adopt(out);
    *pointer = wk.owner;
    return IMP_OK;
}

static int single_stage(impgpu_image* image, int kind) {
    if (!image) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    if (image->c < 3) return IMP_ERROR_INVALID_ARGS;
    PixelProgram prog;
    Stage s{};
    s.kind = kind;
    prog.stages.push_back(s);
    return launch_pixel_program(image->d, (long long)image->fstride, image->w, image->h, image->c, image->step, image->frames, prog, env_stream());
}
int impgpu_rgb2hsv(impgpu_image* image) { return single_stage(image, ST_RGB2HSV); }
int impgpu_hsv2rgb(impgpu_image* image) { return single_stage(image, ST_HSV2RGB); }

int impgpu_run_ops(impgpu_image** pointer, const impgpu_job* job, const impgpu_config* config, int* step) {
    int dummy;
    if (!step) step = &dummy;
    *step = IMP_STEP_START;
    if (!pointer || !*pointer || !job || !config) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    // bridge.c:361-363 rejects the request while parsing; same limit here
    if (config->max_filters_count > 0 && job->filter_count > config->max_filters_count) return IMP_ERROR_TOO_MUCH_FILTERS;
    Work wk{*pointer, view_of(*pointer)};
    int rc = IMP_OK;
    PixelProgram prog;
    int fused_filters = 0;                                             // leading filters the Resize launch already applied
    bool watermark_done = false;

    *step = IMP_STEP_CROP;                                             // bridge.c:575-586
    if (job->crop) {
        TraceRange tr("IMP_STEP_CROP");
        if (fault_hit(IMP_STEP_CROP)) { rc = IMP_ERROR_DEVICE; goto done; }
        int x, y, w, h;
        rc = crop_geometry(wk.v.w, wk.v.h, job->crop, job->gravity, &x, &y, &w, &h);
        if (rc) goto done;
        wk.v = view_sub(wk.v, x, y, w, h);                             // no copy: the next operator reads the window
    }
    *step = IMP_STEP_RESIZE;                                           // bridge.c:588-604
    if (job->resize) {
        TraceRange tr("IMP_STEP_RESIZE");
        if (fault_hit(IMP_STEP_RESIZE)) { rc = IMP_ERROR_DEVICE; goto done; }
        int w, h, interp;
        rc = resize_geometry(wk.v.w, wk.v.h, job->resize, config->max_target_w, config->max_target_h, job->simple, &w, &h, &interp);
        if (rc) goto done;
        // A thumbnail request whose first filter is a rotation -- and, when that is its only filter, its watermark -- in
        // ONE launch: both ride on the stores of the row-streaming AREA kernel (a lone request is launch-bound: this is a
        // third of its launches and two intermediate frames).  Anything the fused kernel does not take goes step by step.
        rc = IMP_ERROR_UNSUPPORTED;
        if (interp == IMP_INTER_AREA && (wk.v.c == 4 || wk.v.c == 3) && job->filter_count >= 1) {    // BGRA, and BGR: every JPEG
            FilterPlan first;
            PixelProgram none;
            if (filter_plan(job->filters[0], config->allow_experiments, wk.v.c, w, h, &first, &none) == IMP_OK && first.cls == FC_ROTATE) {
                const bool swap = first.rotate != 180;
                const int fw = swap ? h : w, fh = swap ? w : h;
                const impgpu_image* ov = config->watermark;
                OverlayArgs wm{};
                bool with_wm = job->filter_count == 1 && ov && ov->c == 4 && !(((uintptr_t)ov->d | (uintptr_t)ov->step) & 3);
                if (with_wm && watermark_rect(fw, fh, ov->w, ov->h, config, &wm.rx, &wm.ry, &wm.maxcol, &wm.maxrow) != IMP_OK) with_wm = false;
                if (with_wm) {
                    wm.ov = ov->d; wm.ostep = ov->step;
                    wm.alpha = 1 - (float)(config->watermark_opacity / 100.0);     // bridge.c:275, filters.c:620
                }
                impgpu_image* out = nullptr;
                rc = wk.fresh(fw, fh, wk.v.c, &out);
                if (rc) goto done;
                Frames f = wk.to(out);
                f.dw = w; f.dh = h;                                    // the resized geometry; `out` is the turned frame
                rc = launch_area_rotate(f, first.rotate, with_wm ? &wm : nullptr, env_stream());
                if (rc == IMP_OK) { wk.adopt(out); fused_filters = 1; watermark_done = with_wm; }
                else image_delete(out);
            }
        }
        if (rc == IMP_ERROR_UNSUPPORTED) rc = do_resize(wk, w, h, interp);
        if (rc) goto done;
    }
    *step = IMP_STEP_FILTERING;                                        // bridge.c:606-627
    trace_push("IMP_STEP_FILTERING");
    if ((wk.v.c == 1 || job->filter_count > 0) && fault_hit(IMP_STEP_FILTERING)) { rc = IMP_ERROR_DEVICE; trace_pop(); goto done; }
    if (wk.v.c == 1) {
        impgpu_image* out = nullptr;
        rc = wk.fresh(wk.v.w, wk.v.h, 3, &out);
        if (rc) { trace_pop(); goto done; }
        rc = launch_gray2bgr(wk.to(out), env_stream());
        if (rc) { image_delete(out); trace_pop(); goto done; }
        wk.adopt(out);
    }
    for (int i = fused_filters; i < job->filter_count && !rc; i++) rc = do_filter(wk, job->filters[i], config->allow_experiments, prog);
    if (!rc) rc = flush_program(wk, prog);
    trace_pop();
    if (rc) goto done;
    *step = IMP_STEP_WATERMARK;                                        // bridge.c:629-640
    if (config->watermark) {
        TraceRange tr("IMP_STEP_WATERMARK");
        if (fault_hit(IMP_STEP_WATERMARK)) { rc = IMP_ERROR_DEVICE; goto done; }
        if (!watermark_done) rc = do_watermark(wk, config);
        if (rc) goto done;
    }
    if (job->need_flatten && wk.v.c == 4) {                            // bridge.c:642-656
        rc = launch_blend_paper(wk.px(), wk.stride(), wk.v.w, wk.v.h, wk.v.step, wk.count(), env_stream());
        if (rc) goto done;
    }
    rc = materialize(wk);
    if (!rc) *step = IMP_STEP_INFO;
done:
    *pointer = wk.owner;
    return rc;
}

// ------------------------------------------------------------------ batch entry points
int impgpu_batch_cv_resize(const void* src, long long src_frame_stride, int src_width, int src_height, int src_step,
                           void* dst, long long dst_frame_stride, int dst_width, int dst_height, int dst_step,
                           int channels, int count, int interpolation, void* stream) {
    if (!src || !dst || (channels != 1 && channels != 3 && channels != 4)) return IMP_ERROR_INVALID_ARGS;
    if (!view_fits(src_width, src_height, channels, src_step) || !view_fits(dst_width, dst_height, channels, dst_step)) return IMP_ERROR_INVALID_ARGS;
    if (count < 0 || count > 65535 || (count > 1 && (src_frame_stride < 0 || dst_frame_stride < 0))) return IMP_ERROR_INVALID_ARGS;
    if (interpolation < IMP_INTER_NN || interpolation > IMP_INTER_LANCZOS4) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    Frames f{};
    f.src = (const uint8_t*)src; f.src_stride = src_frame_stride;
    f.v = View{(const uint8_t*)src, src_width, src_height, channels, src_step};
    f.dst = (uint8_t*)dst; f.dst_stride = dst_frame_stride; f.dw = dst_width; f.dh = dst_height; f.dstep = dst_step;
    f.count = count;
    return launch_cv_resize(f, interpolation, stream ? (hipStream_t)stream : env_stream());
}

int impgpu_batch_resize_mixed(const impgpu_resize_item* items, int count, int channels, int simple, void* stream) {
    if (count < 0 || (count > 0 && !items) || (channels != 1 && channels != 3 && channels != 4)) return IMP_ERROR_INVALID_ARGS;
    for (int i = 0; i < count; i++)                                     // (again in the launcher; here so that it answers without a device)
        if (!items[i].src || !items[i].dst || !view_fits(items[i].src_width, items[i].src_height, channels, items[i].src_step) ||
            !view_fits(items[i].dst_width, items[i].dst_height, channels, items[i].dst_step))
            return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    static_assert(sizeof(MixFrame) == sizeof(impgpu_resize_item) && offsetof(MixFrame, dst) == offsetof(impgpu_resize_item, dst),
                  "MixFrame mirrors impgpu_resize_item");
    return launch_resize_mixed(reinterpret_cast<const MixFrame*>(items), count, channels, simple,
                               stream ? (hipStream_t)stream : env_stream());
}

int impgpu_batch_resize_rotate_watermark(const void* src, long long src_frame_stride, int src_width, int src_height, int src_step,
                                         void* dst, long long dst_frame_stride, int dst_step,
                                         int resize_width, int resize_height, int rotate,
                                         const impgpu_config* config, int channels, int count, void* stream) {
    if (!src || !dst || !config || (channels != 3 && channels != 4)) return IMP_ERROR_INVALID_ARGS;
    if (rotate != 0 && rotate != 90 && rotate != 180 && rotate != 270) return IMP_ERROR_INVALID_ARGS;
    if (!view_fits(src_width, src_height, channels, src_step) || resize_width <= 0 || resize_height <= 0 ||
        !view_fits(rotate == 90 || rotate == 270 ? resize_height : resize_width, rotate == 90 || rotate == 270 ? resize_width : resize_height, channels, dst_step))
        return IMP_ERROR_INVALID_ARGS;
    if (count < 0 || count > 65535 || (count > 1 && (src_frame_stride < 0 || dst_frame_stride < 0))) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : env_stream();
    const int interp = (resize_width > src_width || resize_height > src_height) ? IMP_INTER_CUBIC : IMP_INTER_AREA;  // bridge.c:190
    const bool swap = rotate == 90 || rotate == 270;
    const int fw = swap ? resize_height : resize_width, fh = swap ? resize_width : resize_height;
    if (dst_step < fw * channels) return IMP_ERROR_INVALID_ARGS;

    Frames rs{};
    rs.src = (const uint8_t*)src; rs.src_stride = src_frame_stride;
    rs.v = View{(const uint8_t*)src, src_width, src_height, channels, src_step};
    rs.count = count;
    rs.dw = resize_width; rs.dh = resize_height;
    void* mid = nullptr;
    int rc;
    rc = IMP_ERROR_UNSUPPORTED;
    bool watermark_done = false;
    if (interp == IMP_INTER_AREA && swap && resize_width * 2 == src_width && resize_height * 2 == src_height) {
        // exact 2x2 box + quarter turn: one pass, the half-size intermediate never reaches HBM -- and with a BGRA overlay
        // (4-byte aligned rows) the Watermark step rides on the same kernel's stores
        Frames fz = rs;
        fz.dst = (uint8_t*)dst; fz.dst_stride = dst_frame_stride; fz.dstep = dst_step; fz.dw = fw; fz.dh = fh;
        OverlayArgs wm{};
        const impgpu_image* ov = config->watermark;
        const bool fuse = ov && channels == 4 && ov->c == 4 && !(((uintptr_t)ov->d | (uintptr_t)ov->step) & 3);
        if (fuse) {
            rc = watermark_rect(fw, fh, ov->w, ov->h, config, &wm.rx, &wm.ry, &wm.maxcol, &wm.maxrow);
            if (rc) return rc;
            wm.ov = ov->d; wm.ostep = ov->step;
            wm.alpha = 1 - (float)(config->watermark_opacity / 100.0);     // bridge.c:275, filters.c:620
        }
        rc = launch_area2x2_rotate(fz, rotate, fuse ? &wm : nullptr, s);
        watermark_done = fuse && rc == IMP_OK;
    }
    if (rc == IMP_ERROR_UNSUPPORTED && interp == IMP_INTER_AREA) {
        // any other shrink, BGRA or BGR: the rotate and the watermark ride on the store phase of the row-streaming AREA kernel
        Frames fz = rs;
        fz.dst = (uint8_t*)dst; fz.dst_stride = dst_frame_stride; fz.dstep = dst_step;
        OverlayArgs wm{};
        const impgpu_image* ov = config->watermark;
        const bool fuse = ov && ov->c == 4 && !(((uintptr_t)ov->d | (uintptr_t)ov->step) & 3);
        if (fuse) {
            rc = watermark_rect(fw, fh, ov->w, ov->h, config, &wm.rx, &wm.ry, &wm.maxcol, &wm.maxrow);
            if (rc) return rc;
            wm.ov = ov->d; wm.ostep = ov->step;
            wm.alpha = 1 - (float)(config->watermark_opacity / 100.0);     // bridge.c:275, filters.c:620
        }
        rc = launch_area_rotate(fz, rotate, fuse ? &wm : nullptr, s);
        watermark_done = fuse && rc == IMP_OK;
    }
    if (rc != IMP_ERROR_UNSUPPORTED) {
        // fused path taken (or failed with a device error)
    } else if (rotate == 0) {
        rs.dst = (uint8_t*)dst; rs.dst_stride = dst_frame_stride; rs.dstep = dst_step;
        rc = launch_cv_resize(rs, interp, s);
    } else {
        const int mstep = aligned_step(resize_width, channels);
        const long long mstride = ((long long)mstep * resize_height + 15) & ~15LL;
        rc = dev_alloc_on((size_t)mstride * count + 16, &mid, s);
        if (rc) return rc;
        rs.dst = (uint8_t*)mid; rs.dst_stride = mstride; rs.dstep = mstep;
        rc = launch_cv_resize(rs, interp, s);
        if (!rc) {
            Frames rt{};
            rt.src = (const uint8_t*)mid; rt.src_stride = mstride;
            rt.v = View{(const uint8_t*)mid, resize_width, resize_height, channels, mstep};
            rt.dst = (uint8_t*)dst; rt.dst_stride = dst_frame_stride; rt.dw = fw; rt.dh = fh; rt.dstep = dst_step;
            rt.count = count;
            rc = launch_rotate(rt, rotate, s);
        }
    }
    if (!rc && config->watermark && !watermark_done) {
        int rx, ry, maxcol, maxrow;
        rc = watermark_rect(fw, fh, config->watermark->w, config->watermark->h, config, &rx, &ry, &maxcol, &maxrow);
        if (!rc) {
            const float opacity = (float)(config->watermark_opacity / 100.0);
            rc = launch_blend_over((uint8_t*)dst, dst_frame_stride, fw, fh, channels, dst_step, count, config->watermark,
                                   rx, ry, maxcol, maxrow, 1 - opacity, s);
        }
    }
    if (mid) dev_free_on(mid, s);       // no host wait: parked behind an event when `s` is the caller's stream
    return rc;
}

int impgpu_batch_filters(void* frames, long long frame_stride, int width, int height, int channels, int step, int count,
                         const char* const* filters, int filter_count, int allow_experiments, void* stream) {
    if (!frames || !filters || filter_count < 0 || (channels != 3 && channels != 4) || !view_fits(width, height, channels, step)) return IMP_ERROR_INVALID_ARGS;
    if (count < 0 || count > 65535 || (count > 1 && frame_stride < 0)) return IMP_ERROR_INVALID_ARGS;
    if (int rc = need_env()) return rc;
    PixelProgram prog;
    for (int i = 0; i < filter_count; i++) {
        FilterPlan plan;
        if (int rc = filter_plan(filters[i], allow_experiments, channels, width, height, &plan, &prog)) return rc;
        if (plan.cls != FC_POINTWISE && plan.cls != FC_NOOP) return IMP_ERROR_UNSUPPORTED;
    }
    return launch_pixel_program((uint8_t*)frames, frame_stride, width, height, channels, step, count, prog,
                                stream ? (hipStream_t)stream : env_stream());
}

}  // extern "C"
