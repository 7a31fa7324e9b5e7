// imp_request.cpp -- the front half of RunJob on the host: URI unescape + GET grammar
// (bridge.c:304-372) and the encoder choice that decides `simple` and `need_flatten`
// (bridge.c:413-466, :594, :642-648), producing the impgpu_job that impgpu_run_ops consumes.
// With this a literal request line drives the device chain end to end.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "imp_internal.h"

struct impgpu_request {
    std::string text;                        // unescaped copy; every field below points into it
    std::vector<const char*> filters;
    impgpu_job job{};
    const char* quality = nullptr;
    const char* format = nullptr;
    int page = -1;
    int mime = 0;
    int destructive = 0;
};

namespace {

enum FormatClass { F_UNKNOWN, F_ALPHA_OK, F_NO_ALPHA, F_GIF, F_NOT_IMPL };

bool ieq(const char* a, const char* b) {
    for (; *a && *b; a++, b++)
        if ((*a | 32) != (*b | 32) && *a != *b) return false;
    return *a == *b;
}

// The slice of FreeImage_GetFIFFromFilename that matters to the pixel path: encoders without
// 32-bit support (advancedio.c:43-63), GIF (bridge.c:594), formats IMP refuses (advancedio.c:8-31).
FormatClass classify(const char* ext) {
    static const char* const alpha_ok[] = {"bmp", "png", "tga", "targa", "tif", "tiff", "webp", "jng", "xpm"};
    static const char* const no_alpha[] = {"jpg", "jif", "jpeg", "jpe", "j2k", "j2c", "jp2", "pbm", "pgm", "ppm"};
    static const char* const not_impl[] = {"ico", "koa", "iff", "lbm", "mng", "pcd", "pcx", "ras", "wap", "wbmp", "wbm",
                                           "psd", "cut", "xbm", "dds", "hdr", "g3", "sgi", "exr", "pfm", "pct", "pict",
                                           "pic", "jxr", "wdp", "hdp"};
    if (!ext) return F_UNKNOWN;
    if (const char* dot = std::strrchr(ext, '.')) ext = dot + 1;
    if (ieq(ext, "gif")) return F_GIF;
    for (const char* e : alpha_ok) if (ieq(ext, e)) return F_ALPHA_OK;
    for (const char* e : no_alpha) if (ieq(ext, e)) return F_NO_ALPHA;
    for (const char* e : not_impl) if (ieq(ext, e)) return F_NOT_IMPL;
    return F_UNKNOWN;
}

int hexval(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    c |= 32;
    return (c >= 'a' && c <= 'f') ? c - 'a' + 10 : -1;
}

std::string unescape(const char* s) {       // ngx_unescape_uri, type 0
    std::string out;
    const size_t n = std::strlen(s);
    out.reserve(n);
    for (size_t i = 0; i < n; i++) {
        if (s[i] == '%' && i + 2 < n && hexval(s[i + 1]) >= 0 && hexval(s[i + 2]) >= 0) {
            out.push_back((char)(hexval(s[i + 1]) * 16 + hexval(s[i + 2])));
            i += 2;
        } else out.push_back(s[i]);
    }
    return out;
}

bool prefix(const char* tok, const char* key) { return std::strncmp(tok, key, std::strlen(key)) == 0; }   // StartsWith, helpers.c:4-6

}  // namespace

extern "C" {

int impgpu_parse_request(const char* uri, const char* extension, const impgpu_config* config, impgpu_request** out) {
    if (!uri || !out) return IMP_ERROR_INVALID_ARGS;
    impgpu_request* r = new impgpu_request();
    *out = r;
    r->text = unescape(uri);
    const int max_filters = config ? config->max_filters_count : 5;
    char* buf = &r->text[0];
    // strtok_r(request, "?"): skip leading '?', path up to the next '?', then the parameter block up to the one after
    char* p = buf;
    while (*p == '?') p++;
    char* q = std::strchr(p, '?');
    if (!q) return IMP_ERROR_INVALID_ARGS;                               // bridge.c:340-343
    while (*q == '?') q++;
    if (!*q) return IMP_ERROR_INVALID_ARGS;
    if (char* end = std::strchr(q, '?')) *end = '\0';
    for (char* tok = q; tok && *tok;) {                                  // bridge.c:346-372, '&'-separated, empty pieces skipped
        while (*tok == '&') tok++;
        if (!*tok) break;
        char* next = std::strchr(tok, '&');
        if (next) *next++ = '\0';
        auto value = [&](char stop) -> const char* { char* v = std::strchr(tok, stop); return v ? v + 1 : nullptr; };
        const char* v = nullptr;
        if (prefix(tok, "crop")) { if (!(v = value('='))) return IMP_ERROR_INVALID_ARGS; r->job.crop = v; }
        else if (prefix(tok, "gravity")) { if (!(v = value('='))) return IMP_ERROR_INVALID_ARGS; r->job.gravity = v; }
        else if (prefix(tok, "resize")) { if (!(v = value('='))) return IMP_ERROR_INVALID_ARGS; r->job.resize = v; }
        else if (prefix(tok, "quality")) { if (!(v = value('='))) return IMP_ERROR_INVALID_ARGS; r->quality = v; }
        else if (prefix(tok, "format")) { if (!(v = value('='))) return IMP_ERROR_INVALID_ARGS; r->format = v; }
        else if (prefix(tok, "page")) { if (!(v = value('='))) return IMP_ERROR_INVALID_ARGS; r->page = (int)std::strtol(v, nullptr, 10); }
        else if (prefix(tok, "filter")) {
            if ((int)r->filters.size() >= max_filters) return IMP_ERROR_TOO_MUCH_FILTERS;
            if (!(v = value('-'))) return IMP_ERROR_INVALID_ARGS;        // reference: RewindArgs runs off the end
            r->filters.push_back(v);
            if (!r->destructive) r->destructive = imp::check_destructive(v);
        }
        tok = next;
    }
    r->job.filters = r->filters.empty() ? nullptr : r->filters.data();
    r->job.filter_count = (int)r->filters.size();

    const char* format = r->format ? r->format : (extension ? extension : "");    // bridge.c:413-416
    FormatClass cls = F_UNKNOWN;
    if (!std::strcmp(format, "jpg")) r->mime = -1;
    else if (!std::strcmp(format, "png")) r->mime = -2;
    else if (!std::strcmp(format, "json")) r->mime = -3;
    else if (!std::strcmp(format, "text")) r->mime = -5;
    else {
        cls = classify(format);
        if (cls == F_UNKNOWN || cls == F_NOT_IMPL) return IMP_ERROR_UNSUPPORTED; // bridge.c:441-444
        r->mime = -4;
    }
    // bridge.c:433-435, :448-450: every encoder but GIF (and the json exit) takes one page, so an absent page= means
    // page 0 -- which also makes LoadGIF's walk destructive (advancedio.c:111-113)
    if (r->page == -1 && r->mime != -3 && cls != F_GIF) r->page = 0;
    if (r->quality && (r->mime == -1 || r->mime == -2)) {               // bridge.c:475-500: jpg 0..100, png 0..9
        const long qv = std::strtol(r->quality, nullptr, 10);
        if (qv < 0 || qv > (r->mime == -1 ? 100 : 9)) return IMP_ERROR_INVALID_ARGS;
    }
    if (r->quality && r->mime == -4) {                                  // bridge.c:511-519: FIF_J2K / FIF_JP2 / FIF_WEBP 0..512
        const char* e = format;
        if (const char* dot = std::strrchr(e, '.')) e = dot + 1;
        if (ieq(e, "j2k") || ieq(e, "j2c") || ieq(e, "jp2") || ieq(e, "webp")) {
            const long qv = std::strtol(r->quality, nullptr, 10);
            if (qv < 0 || qv > 512) return IMP_ERROR_INVALID_ARGS;
        }
    }
    r->job.simple = cls == F_GIF;                                        // bridge.c:594
    r->job.need_flatten = r->mime == -1 || (r->mime == -4 && cls == F_NO_ALPHA);   // bridge.c:643-647
    return IMP_OK;
}

const impgpu_job* impgpu_request_job(const impgpu_request* r) { return r ? &r->job : nullptr; }
const char* impgpu_request_quality(const impgpu_request* r) { return r ? r->quality : nullptr; }
const char* impgpu_request_format(const impgpu_request* r) { return r ? r->format : nullptr; }
int impgpu_request_page(const impgpu_request* r) { return r ? r->page : -1; }
int impgpu_request_mime(const impgpu_request* r) { return r ? r->mime : 0; }
int impgpu_request_destructive(const impgpu_request* r) { return r ? r->destructive : 0; }
void impgpu_request_free(impgpu_request** r) {
    if (!r || !*r) return;
    delete *r;
    *r = nullptr;
}

}  // extern "C"
