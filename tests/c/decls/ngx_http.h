/* declarations only -- see ngx_config.h in this directory */
#ifndef DECLS_NGX_HTTP_H
#define DECLS_NGX_HTTP_H
#include <ngx_core.h>
typedef struct ngx_http_request_s {
    ngx_pool_t* pool;
    ngx_str_t   uri, unparsed_uri, args, exten;
} ngx_http_request_t;
void ngx_unescape_uri(u_char** dst, u_char** src, size_t size, ngx_uint_t type);
#endif
