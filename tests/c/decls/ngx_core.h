/* declarations only -- see ngx_config.h in this directory */
#ifndef DECLS_NGX_CORE_H
#define DECLS_NGX_CORE_H
#include <ngx_config.h>
typedef struct ngx_pool_s ngx_pool_t;
typedef struct { size_t len; u_char* data; } ngx_str_t;
void* ngx_palloc(ngx_pool_t* pool, size_t size);
void* ngx_pcalloc(ngx_pool_t* pool, size_t size);
void* ngx_pnalloc(ngx_pool_t* pool, size_t size);
ngx_int_t ngx_pfree(ngx_pool_t* pool, void* p);
u_char* ngx_cpymem(void* dst, const void* src, size_t n);
extern ngx_int_t  ngx_process_slot;
extern ngx_uint_t ngx_worker;
#endif
