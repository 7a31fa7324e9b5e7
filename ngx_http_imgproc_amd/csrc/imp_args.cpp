// imp_args.cpp -- the reference's argument grammar, evaluated on the host before any launch.
//
// Crop / Resize geometry (bridge.c:18-128, :143-190), the filter-* table and each filter's
// own argument checks (filters.c:5-70 and the callbacks), and the translation of pointwise
// filters into stages of the fused pixel program.  Everything a filter computes once per
// call on the CPU in the reference (gamma / gradient LUTs, filters.c:561-593) or that is a
// pure per-channel function of one 8-bit value (ModulateHSV's S and V, AlphaBlendAddColor,
// BrightnessContrast, Lomo) is tabulated here in the reference's own C arithmetic, so the
// device only does table look-ups for those and the results are identical by construction.
// Built with -ffp-contract=off.
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "imp_internal.h"

namespace imp {

// ---- C conversion semantics the reference relies on (x86-64 gcc) ----
// (int)floating: truncation; NaN / out of range give INT_MIN (cvttss2si / cvttsd2si).
static inline int to_int(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}
// value assigned to a `char` pixel through cvSetComponent (helpers.h:2): low byte of the int.
static inline uint8_t low_byte(int v) { return (uint8_t)(v & 0xff); }
static inline uint8_t store_fp(double v) { return low_byte(to_int(v)); }

// strtok_r-compatible splitter over a private copy: skips runs of the delimiter.
class Splitter {
public:
    Splitter(const char* s, char delim) : buf_(s ? s : ""), delim_(delim), pos_(0) {}
    // returns nullptr when exhausted; the pointer stays valid for the Splitter's lifetime
    const char* next() {
        while (pos_ < buf_.size() && buf_[pos_] == delim_) pos_++;
        if (pos_ >= buf_.size()) return nullptr;
        size_t start = pos_;
        while (pos_ < buf_.size() && buf_[pos_] != delim_) pos_++;
        if (pos_ < buf_.size()) buf_[pos_++] = '\0';
        return buf_.c_str() + start;
    }
private:
    std::string buf_;
    char delim_;
    size_t pos_;
};

static unsigned parse_uint(const char* tok, const char** rest) {
    char* end;
    unsigned v = (unsigned)std::strtol(tok ? tok : "", &end, 10);
    *rest = end;
    return v;
}

// ---------------------------------------------------------------- Crop, bridge.c:18-128
static int gravity_origin(const char* tok, const char* lo, const char* hi, size_t full, unsigned win, int* origin) {
    if (!tok) return IMP_ERROR_INVALID_ARGS;    // reference dereferences NULL here
    if (!std::strcmp(tok, lo)) { *origin = 0; return IMP_OK; }
    if (!std::strcmp(tok, hi)) { *origin = (int)(full - win); return IMP_OK; }
    if (!std::strcmp(tok, "c")) { *origin = (int)std::round((full - win) / 2.0); return IMP_OK; }
    const char* mode;
    unsigned px = parse_uint(tok, &mode);
    if (std::strcmp("px", mode)) return IMP_ERROR_INVALID_ARGS;
    *origin = (int)px;
    return IMP_OK;
}

int crop_geometry(int icol, int irow, const char* args, const char* gravity, int* ox, int* oy, int* ow, int* oh) {
    if (icol <= 0 || irow <= 0) return IMP_ERROR_INVALID_ARGS;
    const size_t col = (size_t)icol, row = (size_t)irow;
    Splitter a(args, ',');
    const char *wmode, *hmode;
    unsigned ww = parse_uint(a.next(), &wmode);
    unsigned wh = parse_uint(a.next(), &hmode);

    bool use_gravity = false;                        // :37-45
    if (gravity) {
        if (std::strlen(gravity) > 2) use_gravity = true;
        else return IMP_ERROR_INVALID_ARGS;
    }
    if (!*wmode && !*hmode) {                        // W:H ratio, :47-57
        if (ww == 0 || wh == 0) return IMP_ERROR_INVALID_ARGS;   // reference ends at :65 via inf/nan
        float px = (float)col;
        float py = px / ww * wh;
        if (py > row) { py = (float)row; px = py / wh * ww; }
        ww = (unsigned)(int)std::round(px);
        wh = (unsigned)(int)std::round(py);
    } else if (std::strcmp(wmode, "px") || std::strcmp(hmode, "px")) {
        return IMP_ERROR_INVALID_ARGS;
    }
    if (ww == 0 || ww > col || wh == 0 || wh > row) return IMP_ERROR_INVALID_ARGS;   // :65-68

    Splitter g(use_gravity ? gravity : "", ',');
    int wx, wy;
    const char* tok = use_gravity ? g.next() : a.next();
    if (!use_gravity && !tok) tok = "c";
    if (int rc = gravity_origin(tok, "l", "r", col, ww, &wx)) return rc;
    tok = use_gravity ? g.next() : a.next();
    if (!use_gravity && !tok) tok = "t";
    if (int rc = gravity_origin(tok, "t", "b", row, wh, &wy)) return rc;

    if (wx + (int)ww > icol || wy + (int)wh > irow) return IMP_ERROR_INVALID_ARGS;   // :125-128
    if (wx < 0 || wy < 0) return IMP_ERROR_INVALID_ARGS;   // reference: OpenCV ROI/copy failure
    *ox = wx; *oy = wy; *ow = (int)ww; *oh = (int)wh;
    return IMP_OK;
}

// ---------------------------------------------------------------- Resize, bridge.c:143-190
int resize_geometry(int icol, int irow, const char* args, unsigned max_w, unsigned max_h, int simple,
                    int* ow, int* oh, int* interp) {
    if (icol <= 0 || irow <= 0) return IMP_ERROR_INVALID_ARGS;
    const size_t col = (size_t)icol, row = (size_t)irow;
    Splitter a(args, ',');
    const char* rest;
    unsigned width = parse_uint(a.next(), &rest);
    unsigned height = parse_uint(a.next(), &rest);
    if (width == 0 && height == 0) return IMP_ERROR_INVALID_ARGS;
    if (width == 0) width = (unsigned)(int)std::round((float)height / row * col);
    if (height == 0) height = (unsigned)(int)std::round((float)width / col * row);
    const char* opt = a.next();
    bool up = opt && !std::strcmp(opt, "up");
    if (!up) {
        width = (unsigned)std::fmin((double)width, (double)col);
        height = (unsigned)std::fmin((double)height, (double)row);
    }
    // :184 compares `width` against both limits (sic); kept for drop-in behaviour
    if ((max_w > 0 && width > max_w) || (max_h > 0 && width > max_h)) return IMP_ERROR_TOO_BIG_TARGET;
    if (width == 0 || height == 0 || width > 0x7fff0000u || height > 0x7fff0000u) return IMP_ERROR_INVALID_ARGS;
    *ow = (int)width;
    *oh = (int)height;
    *interp = simple ? IMP_INTER_NN : ((width > col || height > row) ? IMP_INTER_CUBIC : IMP_INTER_AREA);
    return IMP_OK;
}

// ---------------------------------------------------------------- pixel-program building
static void lut_identity(uint8_t* t) {
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 256; i++) t[c * 256 + i] = (uint8_t)i;
}

// Appends a per-channel table stage; folds it into a directly preceding table stage.
static void push_lut(PixelProgram* p, const uint8_t* t) {
    if (!p->stages.empty() && p->stages.back().kind == ST_LUT4) {
        uint8_t* prev = p->tables.data() + p->stages.back().lut_off;
        for (int c = 0; c < 4; c++)
            for (int i = 0; i < 256; i++) prev[c * 256 + i] = t[c * 256 + prev[c * 256 + i]];
        return;
    }
    Stage s{};
    s.kind = ST_LUT4;
    s.lut_off = (int)p->tables.size();
    p->tables.insert(p->tables.end(), t, t + 1024);
    p->stages.push_back(s);
}
static void push_simple(PixelProgram* p, int kind) {
    Stage s{};
    s.kind = kind;
    p->stages.push_back(s);
}

// ModulateHSV, filters.c:524-547
static void add_modulate(PixelProgram* p, const int* hsv) {
    uint8_t t[1024];
    lut_identity(t);
    if (hsv[0] != 0)
        for (int i = 0; i < 256; i++) {
            int hue = i + hsv[0];
            if (hue > 180) hue -= 180;
            t[i] = low_byte(hue);
        }
    for (int c = 1; c < 3; c++)
        for (int i = 0; i < 256; i++) t[c * 256 + i] = low_byte(to_int(std::fmin(i * hsv[c] / 100.0, 255)));
    push_simple(p, ST_RGB2HSV);
    push_lut(p, t);
    push_simple(p, ST_HSV2RGB);
}
// AlphaBlendAddColor, filters.c:608-616
static void add_color(PixelProgram* p, const int* rgb, float alpha) {
    uint8_t t[1024];
    lut_identity(t);
    float beta = 1 - alpha;
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 256; i++) t[c * 256 + i] = store_fp((beta * i) + (rgb[2 - c] * alpha));
    push_lut(p, t);
}
// ApplyGamma + CalculateGammaLUT, filters.c:549-570 (all channels, alpha included)
static void add_gamma(PixelProgram* p, float gamma) {
    uint8_t t[1024];
    float inverse = 1 / gamma;
    for (int i = 0; i < 256; i++) {
        uint8_t v = low_byte(to_int(std::pow(i / 255.0, (double)inverse) * 255.0));
        for (int c = 0; c < 4; c++) t[c * 256 + i] = v;
    }
    push_lut(p, t);
}
// BrightnessContrast, filters.c:595-605 (channels 0..2)
static void add_brightness_contrast(PixelProgram* p, float br, float ct) {
    uint8_t t[1024];
    lut_identity(t);
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 256; i++) {
            int val = to_int((ct * i) + (br * 255));
            val = to_int(std::fmax(std::fmin((double)val, 255), 0));
            t[c * 256 + i] = low_byte(val);
        }
    push_lut(p, t);
}

static int hex2(const char* s) {
    char t[3] = {s[0], s[1], 0};
    return (int)std::strtol(t, nullptr, 16);
}

static float corner_dist(int ax, int ay, int bx, int by) {   // helpers.c:46-48
    return (float)std::sqrt(std::pow((double)(float)(ax - bx), 2) + std::pow((double)(float)(ay - by), 2));
}

struct FilterEntry { const char* name; int experimental; int destructive; };
static const FilterEntry kFilters[] = {        // filters.c:10-28
    {"flip", 0, 0},     {"rotate", 0, 0},   {"modulate", 0, 0}, {"colorize", 0, 0}, {"blur", 0, 1},
    {"gamma", 0, 0},    {"contrast", 0, 0}, {"gradmap", 0, 0},  {"vignette", 1, 1}, {"gotham", 1, 0},
    {"lomo", 1, 0},     {"kelvin", 1, 0},   {"rainbow", 1, 0},  {"scanline", 1, 0},
};

int check_destructive(const char* request) {   // filters.c:32-40 (prefix compare)
    if (!request) return 0;
    for (const FilterEntry& f : kFilters)
        if (!std::strncmp(request, f.name, std::strlen(f.name))) return f.destructive;
    return 0;
}

int filter_plan(const char* request, int allow_experiments, int channels, int w, int h,
                FilterPlan* plan, PixelProgram* prog) {
    plan->cls = FC_NOOP;
    Splitter rq(request, '=');
    const char* type = rq.next();
    if (!type) return IMP_ERROR_NO_SUCH_FILTER;
    const char* args = rq.next();
    if (!args) return IMP_ERROR_INVALID_ARGS;
    const FilterEntry* fe = nullptr;
    for (const FilterEntry& f : kFilters)
        if (!std::strcmp(type, f.name) && (allow_experiments || !f.experimental)) { fe = &f; break; }
    if (!fe) return IMP_ERROR_NO_SUCH_FILTER;
    const std::string name = fe->name;
    // The reference's pointwise loops index channels 0..2 unconditionally; RunJob promotes
    // 1-channel frames before filtering (bridge.c:613-618). A direct call on a 1-channel image
    // would run off the pixel there; rejected here.
    const bool needs_bgr = name != "flip" && name != "rotate" && name != "blur" && name != "gamma" && name != "contrast";
    if (needs_bgr && channels < 3) return IMP_ERROR_INVALID_ARGS;

    if (name == "flip") {                                  // filters.c:72-109
        if (std::strlen(args) != 2) return IMP_ERROR_INVALID_ARGS;
        int hz = 0, vt = 0;
        if (args[0] == '1') hz = 1; else if (args[0] != '0') return IMP_ERROR_INVALID_ARGS;
        if (args[1] == '1') vt = 1; else if (args[1] != '0') return IMP_ERROR_INVALID_ARGS;
        if (hz || vt) { plan->cls = FC_FLIP; plan->flip_mode = (hz && vt) ? -1 : (hz ? 1 : 0); }
        return IMP_OK;
    }
    if (name == "rotate") {                                // filters.c:111-133
        int amount = (int)std::strtol(args, nullptr, 10);
        if (amount != 90 && amount != 180 && amount != 270) return IMP_ERROR_INVALID_ARGS;
        plan->cls = FC_ROTATE;
        plan->rotate = amount;
        return IMP_OK;
    }
    if (name == "modulate") {                              // filters.c:135-158
        Splitter a(args, ',');
        int hsv[3];
        for (int i = 0; i < 3; i++) {
            const char* tok = a.next();
            if (!tok) return IMP_ERROR_INVALID_ARGS;
            hsv[i] = (int)std::strtol(tok, nullptr, 10);
        }
        if (hsv[0] < 0 || hsv[0] > 180 || hsv[2] <= 0) return IMP_ERROR_INVALID_ARGS;
        add_modulate(prog, hsv);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "colorize") {                              // filters.c:160-190
        Splitter a(args, ',');
        const char* color = a.next();
        if (!color || std::strlen(color) != 6) return IMP_ERROR_INVALID_ARGS;
        int rgb[3] = {hex2(color), hex2(color + 2), hex2(color + 4)};
        const char* op = a.next();
        float opacity = op ? std::strtof(op, nullptr) : 0.5f;
        if (opacity < 0 || opacity > 1) return IMP_ERROR_INVALID_ARGS;
        add_color(prog, rgb, opacity);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "blur") {                                  // filters.c:192-207
        Splitter a(args, ',');
        const char* tok = a.next();
        if (!tok) return IMP_ERROR_INVALID_ARGS;
        float sigma = std::strtof(tok, nullptr);
        if (sigma < 0) return IMP_ERROR_INVALID_ARGS;
        // sigma == 0 trips an OpenCV assertion in the reference; defined as a no-op.
        if (sigma > 0 && gaussian_ksize(sigma) > 1) { plan->cls = FC_BLUR; plan->sigma = sigma; }
        return IMP_OK;
    }
    if (name == "gamma") {                                 // filters.c:209-212
        add_gamma(prog, std::strtof(args, nullptr));
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "contrast") {                              // filters.c:214-221
        float v = std::strtof(args, nullptr);
        if (v <= 0) return IMP_ERROR_INVALID_ARGS;
        add_brightness_contrast(prog, 0, v);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "gradmap") {                               // filters.c:223-286, 572-593
        Splitter a(args, ',');
        uint8_t colors[8][3];
        int n = 0;
        while (const char* cur = a.next()) {
            if (std::strlen(cur) != 6) return IMP_ERROR_INVALID_ARGS;
            if (n >= 8) return IMP_ERROR_INVALID_ARGS;     // reference: heap overflow past 8 slots
            for (int i = 0; i < 3; i++) colors[n][i] = (uint8_t)hex2(cur + 2 * i);
            n++;
        }
        if (n < 2) return IMP_ERROR_INVALID_ARGS;          // reference: reads an unfilled table
        uint8_t lut[768];
        int segments = n - 1, ptr = 0;
        float inner = 256 / (float)segments;
        for (int c = 0; c < segments; c++)
            for (int i = 0; i < (int)inner; i++) {
                float step = i / inner;
                for (int j = 0; j < 3; j++)
                    lut[ptr++] = store_fp(std::round((double)(colors[c][j] + step * (colors[c + 1][j] - colors[c][j]))));
            }
        while (ptr < 768) { lut[ptr] = colors[n - 1][ptr % 3]; ptr++; }   // reference leaves these uninitialised
        Stage s{};
        s.kind = ST_GRADMAP;
        s.lut_off = (int)prog->tables.size();
        prog->tables.insert(prog->tables.end(), lut, lut + 768);
        while (prog->tables.size() % 4) prog->tables.push_back(0);
        prog->stages.push_back(s);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "vignette") {                              // filters.c:295-323, 693-703; helpers.c:50-66
        Splitter a(args, ',');
        const char* t0 = a.next();
        float intensity = t0 ? std::strtof(t0, nullptr) : 0.5f;
        const char* t1 = a.next();
        float radius = t1 ? std::strtof(t1, nullptr) : 1.0f;
        int cx = w / 2, cy = h / 2;
        float maxdis = 0;
        const int corners[4][2] = {{0, 0}, {w, 0}, {0, h}, {w, h}};
        for (auto& c : corners) { float d = corner_dist(c[0], c[1], cx, cy); if (maxdis < d) maxdis = d; }
        Stage s{};
        s.kind = ST_VIGNETTE;
        s.i0 = cx; s.i1 = cy;
        s.f0 = radius * maxdis;
        s.f1 = intensity;
        push_simple(prog, ST_RGB2HSV);
        prog->stages.push_back(s);
        push_simple(prog, ST_HSV2RGB);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "gotham") {                                // filters.c:325-333
        const int hsv[3] = {120, 5, 100}, rgb[3] = {17, 27, 93};
        add_modulate(prog, hsv);
        add_color(prog, rgb, (float)0.15);
        add_gamma(prog, (float)0.3);
        add_brightness_contrast(prog, (float)-0.07, (float)1.5);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "lomo") {                                  // filters.c:335-346
        uint8_t t[1024];
        lut_identity(t);
        for (int c = 1; c < 3; c++)
            for (int i = 0; i < 256; i++) {
                float val = (float)i;
                val = (float)std::fmax(std::fmin(val * 1.5 - 50, 255), 0);
                t[c * 256 + i] = store_fp(val);
            }
        push_lut(prog, t);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "kelvin") {                                // filters.c:348-354
        const int hsv[3] = {120, 50, 100}, rgb[3] = {255, 153, 0};
        add_modulate(prog, hsv);
        add_color(prog, rgb, (float)0.5);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "rainbow") {                               // filters.c:356-403
        int sat = 255;
        if (!std::strcmp(args, "mid")) sat = 190;
        else if (!std::strcmp(args, "pale")) sat = 120;
        else if (std::strcmp(args, "full")) return IMP_ERROR_INVALID_ARGS;
        Stage s{};
        s.kind = ST_RAINBOW;
        s.i0 = sat;
        push_simple(prog, ST_RGB2HSV);
        prog->stages.push_back(s);
        push_simple(prog, ST_HSV2RGB);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    if (name == "scanline") {                              // filters.c:405-455
        Splitter a(args, ',');
        const char* t0 = a.next();
        if (!t0) return IMP_ERROR_INVALID_ARGS;
        float intensity = std::strtof(t0, nullptr);
        if (intensity < 0 || intensity > 1) return IMP_ERROR_INVALID_ARGS;
        const char* t1 = a.next();
        float opacity = t1 ? std::strtof(t1, nullptr) : 0;
        if (opacity < 0 || opacity > 1) return IMP_ERROR_INVALID_ARGS;
        const char* t2 = a.next();
        int freq = t2 ? (int)std::strtol(t2, nullptr, 10) : 1;
        if (freq < 1) return IMP_ERROR_INVALID_ARGS;
        const char* t3 = a.next();
        int width = t3 ? (int)std::strtol(t3, nullptr, 10) : 1;
        if (width < 1) return IMP_ERROR_INVALID_ARGS;
        Stage s{};
        s.kind = ST_SCANLINE;
        s.i0 = freq; s.i1 = width;
        s.i2 = store_fp(255 * opacity);
        s.i3 = store_fp(255 * intensity);
        push_simple(prog, ST_RGB2HSV);
        prog->stages.push_back(s);
        push_simple(prog, ST_HSV2RGB);
        plan->cls = FC_POINTWISE;
        return IMP_OK;
    }
    return IMP_ERROR_NO_SUCH_FILTER;
}

// ---------------------------------------------------------------- Watermark placement, bridge.c:254-274
int watermark_rect(int basew, int baseh, int overw, int overh, const impgpu_config* cfg,
                   int* rx, int* ry, int* maxcol, int* maxrow) {
    int left, top;
    if (cfg->watermark_gravity_x == 'c') left = (basew - overw) / 2 + cfg->watermark_offset_x;
    else if (cfg->watermark_gravity_x == 'r') left = basew - overw - cfg->watermark_offset_x;
    else left = cfg->watermark_offset_x;
    if (cfg->watermark_gravity_y == 'c') top = (baseh - overh) / 2 + cfg->watermark_offset_y;
    else if (cfg->watermark_gravity_y == 'b') top = baseh - overh - cfg->watermark_offset_y;
    else top = cfg->watermark_offset_y;
    // cvSetImageROI: assertion (aborts the reference; INVALID_ARGS here), then clip to the image.
    if (!(left < basew && top < baseh && left + overw >= (overw > 0) && top + overh >= (overh > 0)))
        return IMP_ERROR_INVALID_ARGS;
    *rx = left < 0 ? 0 : left;
    *ry = top < 0 ? 0 : top;
    *maxrow = (int)std::fmin((double)overh, (double)(baseh - *ry));   // filters.c:624-625
    *maxcol = (int)std::fmin((double)overw, (double)(basew - *rx));
    return IMP_OK;
}

}  // namespace imp
