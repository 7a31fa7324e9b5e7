/*
 * orc_request.c -- restatement of RunJob's request parsing (bridge.c:304-372) and of the
 * encoder choice that decides `simple` and `needFlatten` (bridge.c:413-466, :594, :642-648).
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see imp_oracle.h).
 *
 * FreeImage_GetFIFFromFilename (FreeImage 3.x, absent here) is reduced to the extensions whose
 * format decides something on the pixel path: which encoders lack 32-bit support
 * (advancedio.c:43-63), which is GIF (bridge.c:594), which are not implemented (advancedio.c:8-31).
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include "imp_oracle.h"

enum { F_UNKNOWN = 0, F_ALPHA_OK, F_NO_ALPHA, F_GIF, F_NOT_IMPL };

static int fif_class(const char* ext) {
    static const char* alpha_ok[] = {"bmp", "png", "tga", "targa", "tif", "tiff", "webp", "jng", "xpm", NULL};
    static const char* no_alpha[] = {"jpg", "jif", "jpeg", "jpe", "j2k", "j2c", "jp2", "pbm", "pgm", "ppm", NULL};
    static const char* not_impl[] = {"ico", "koa", "iff", "lbm", "mng", "pcd", "pcx", "ras", "wap", "wbmp", "wbm", "psd",
                                     "cut", "xbm", "dds", "hdr", "g3", "sgi", "exr", "pfm", "pct", "pict", "pic", "jxr",
                                     "wdp", "hdp", NULL};
    if (!ext) return F_UNKNOWN;
    const char* dot = strrchr(ext, '.');
    if (dot) ext = dot + 1;
    if (strcasecmp(ext, "gif") == 0) return F_GIF;
    for (int i = 0; alpha_ok[i]; i++) if (strcasecmp(ext, alpha_ok[i]) == 0) return F_ALPHA_OK;
    for (int i = 0; no_alpha[i]; i++) if (strcasecmp(ext, no_alpha[i]) == 0) return F_NO_ALPHA;
    for (int i = 0; not_impl[i]; i++) if (strcasecmp(ext, not_impl[i]) == 0) return F_NOT_IMPL;
    return F_UNKNOWN;
}

/* ngx_unescape_uri(..., type 0): %XX -> byte, anything else copied */
static char* unescape(const char* s) {
    size_t n = strlen(s);
    char* out = (char*)malloc(n + 1);
    size_t o = 0;
    for (size_t i = 0; i < n; i++) {
        if (s[i] == '%' && i + 2 < n) {
            int hi = -1, lo = -1;
            char a = s[i + 1], b = s[i + 2];
            if (a >= '0' && a <= '9') hi = a - '0'; else if ((a | 32) >= 'a' && (a | 32) <= 'f') hi = (a | 32) - 'a' + 10;
            if (b >= '0' && b <= '9') lo = b - '0'; else if ((b | 32) >= 'a' && (b | 32) <= 'f') lo = (b | 32) - 'a' + 10;
            if (hi >= 0 && lo >= 0) { out[o++] = (char)(hi * 16 + lo); i += 2; continue; }
        }
        out[o++] = s[i];
    }
    out[o] = 0;
    return out;
}

static int starts(const char* h, const char* n) { return strstr(h, n) == h; }
/* RewindArgs (helpers.c:18-23); the reference runs off the end when the stopper is missing: defined NULL */
static char* rewind_args(char* p, char stop) {
    char* q = strchr(p, stop);
    return q ? q + 1 : NULL;
}

void orc_request_free(orc_request* r) {
    if (!r) return;
    free(r->buffer);
    free(r);
}

int orc_parse_request(const char* uri, const char* exten, int max_filters, orc_request** out) {
    orc_request* r = (orc_request*)calloc(1, sizeof(orc_request));
    r->page = -1;
    r->buffer = unescape(uri);
    *out = r;
    char* ctx = NULL;
    strtok_r(r->buffer, "?", &ctx);
    char* params = strtok_r(NULL, "?", &ctx);
    if (!params) return ORC_ERROR_INVALID_ARGS;                       /* bridge.c:340-343 */
    ctx = NULL;
    char* tok;
    while ((tok = strtok_r(params, "&", &ctx))) {                     /* bridge.c:346-372 */
        params = NULL;
        char* v;
        if (starts(tok, "crop")) { if (!(v = rewind_args(tok, '='))) return ORC_ERROR_INVALID_ARGS; r->crop = v; }
        else if (starts(tok, "gravity")) { if (!(v = rewind_args(tok, '='))) return ORC_ERROR_INVALID_ARGS; r->gravity = v; }
        else if (starts(tok, "resize")) { if (!(v = rewind_args(tok, '='))) return ORC_ERROR_INVALID_ARGS; r->resize = v; }
        else if (starts(tok, "quality")) { if (!(v = rewind_args(tok, '='))) return ORC_ERROR_INVALID_ARGS; r->quality = v; }
        else if (starts(tok, "format")) { if (!(v = rewind_args(tok, '='))) return ORC_ERROR_INVALID_ARGS; r->format = v; }
        else if (starts(tok, "page")) { if (!(v = rewind_args(tok, '='))) return ORC_ERROR_INVALID_ARGS; r->page = (int)strtol(v, NULL, 10); }
        else if (starts(tok, "filter")) {
            if (r->filter_count >= max_filters || r->filter_count >= ORC_MAX_FILTERS) return ORC_ERROR_TOO_MUCH_FILTERS;
            if (!(v = rewind_args(tok, '-'))) return ORC_ERROR_INVALID_ARGS;
            r->filters[r->filter_count++] = v;
        }
    }
    /* encoder choice, bridge.c:413-466 */
    const char* format = r->format ? r->format : (exten ? exten : "");
    r->mime = 0;
    if (!strcmp(format, "jpg")) r->mime = -1;
    else if (!strcmp(format, "png")) r->mime = -2;
    else if (!strcmp(format, "json")) r->mime = -3;
    else if (!strcmp(format, "text")) r->mime = -5;
    int cls = F_UNKNOWN;
    if (r->mime == 0) {
        cls = fif_class(format);
        if (cls == F_UNKNOWN || cls == F_NOT_IMPL) return ORC_ERROR_UNSUPPORTED;   /* bridge.c:441-444 */
        r->mime = -4;
    }
    if (r->page == -1 && r->mime != -3 && cls != F_GIF) r->page = 0;  /* bridge.c:433-435, :448-450 */
    if (r->quality && (r->mime == -1 || r->mime == -2)) {             /* bridge.c:475-500 */
        long qv = strtol(r->quality, NULL, 10);
        if (qv < 0 || qv > (r->mime == -1 ? 100 : 9)) return ORC_ERROR_INVALID_ARGS;
    }
    if (r->quality && r->mime == -4) {                                /* bridge.c:511-519 */
        const char* e = strrchr(format, '.') ? strrchr(format, '.') + 1 : format;
        if (!strcasecmp(e, "j2k") || !strcasecmp(e, "j2c") || !strcasecmp(e, "jp2") || !strcasecmp(e, "webp")) {
            long qv = strtol(r->quality, NULL, 10);
            if (qv < 0 || qv > 512) return ORC_ERROR_INVALID_ARGS;
        }
    }
    r->simple = cls == F_GIF;                                         /* bridge.c:594 */
    r->need_flatten = r->mime == -1 || (r->mime == -4 && cls == F_NO_ALPHA);   /* bridge.c:643-647 (if the frame has alpha) */
    return ORC_OK;
}
