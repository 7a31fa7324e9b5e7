/* tests/c/decls -- DECLARATIONS ONLY, written for one purpose: to let `gcc -fsyntax-only` read glue/imp_gpu_bridge.c and
 * the bridge.c that glue/apply_glue.sh produces, in an image that has no nginx, OpenCV or FreeImage headers.  Nothing here
 * is linked, run, or compared with anything: it is test scaffolding for a compile check of OUR glue, not an oracle, not a
 * stand-in build of the reference, and it pins nothing.  Only the names those two files use are declared. */
#ifndef DECLS_NGX_CONFIG_H
#define DECLS_NGX_CONFIG_H
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
typedef intptr_t  ngx_int_t;
typedef uintptr_t ngx_uint_t;
typedef intptr_t  ngx_flag_t;
#endif
