/*
 * impgpu_broker.h -- one GPU, many worker PROCESSES (round 5).
 *
 * IMP runs as `worker_processes N` (docs/02 - Configuration.md:18); every worker calls OnEnvStart after the fork
 * (module.c:100-107) and then answers one request at a time, synchronously, inside RunJob (module.c:298, bridge.c:302).
 * With libimpgpu.so linked into every worker that is N device contexts, N sets of pools and pinned rings, and N streams
 * of single-file launches that the device runs one after the other: the batched entry points (impgpu_batch_decode_jpeg,
 * impgpu_batch_resize_mixed, impgpu_batch_encode_jpeg), which is where the device's rate is, are out of a worker's reach.
 *
 * The broker is the piece that carries a worker's single request into a batch: ONE process per GPU (`impgpu_broker`,
 * ngx_http_imgproc_amd/csrc/imp_broker.cpp) owns the HIP context and links libimpgpu.so; workers are plain C, never touch
 * HIP, and hand {file, job, config} over through a shared-memory segment (POSIX shm, one fixed SLOT per worker) with a
 * futex doorbell.  A broker thread takes every request that is queued at that moment (up to 64) and runs them as one
 * batch; the answer bytes (the JPEG file, or the pixels for a host encoder) come back in the worker's slot.
 *
 *   worker (glue/imp_gpu_client.c)                        broker (imp_broker.cpp)
 *   impgpu_client_attach         claim a slot             creates the segment, N_SLOTS slots, T threads
 *   impgpu_client_run            fill slot, state = SUBMITTED, ring the doorbell, sleep on the slot's state word
 *                                                         scan: SUBMITTED -> TAKEN for all queued slots; decode batch ->
 *                                                         operators -> encode batch / download; state = DONE, wake
 *   (answer is read in place; valid until the next call)
 *
 * A worker never waits for ever: the wait is cut into ticks, each tick looks at the broker's pid and epoch, and a broker
 * that is gone answers IMP_ERROR_DEVICE (the request fails like a lost device; the supervisor -- `impgpu_broker
 * --supervise`, a parent that never touches the GPU -- starts a FRESH child, which re-creates the segment's state with a
 * new epoch; workers re-attach lazily).  Slots of workers that died are taken back by the broker (kill(pid, 0)).
 *
 * Everything in the segment is plain data at fixed offsets (no pointers): both sides map it wherever they like.
 */
#ifndef IMPGPU_BROKER_H
#define IMPGPU_BROKER_H

#include <stddef.h>
#include <stdint.h>
#include "impgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define IMPB_MAGIC        0x42504D49u     /* "IMPB" */
#define IMPB_VERSION      2u              /* 2: a JPEG file may arrive with its scan unstuffed (in_head_bytes, in_scan_at, in_scan_bytes) */
#define IMPB_MAX_SLOTS    256
#define IMPB_TEXT_BYTES   3072            /* the job's strings, NUL-separated */
#define IMPB_MAX_FILTERS  32
#define IMPB_DEFAULT_NAME "/impgpu-broker-0"

/* slot states (the futex word of a slot) */
enum { IMPB_FREE = 0, IMPB_CLAIMED = 1, IMPB_SUBMITTED = 2, IMPB_TAKEN = 3, IMPB_DONE = 4 };
/* what the worker hands over / wants back */
enum { IMPB_IN_FILE = 0 /* a JPEG or PNG file */, IMPB_IN_FRAME = 1 /* decoded pixels (host fallback decoders) */,
       IMPB_IN_WATERMARK = 2 /* pixels of a location's overlay: registered, answer = its id */ };
enum { IMPB_OUT_JPEG = 0 /* cvEncodeImage(".jpg"), bridge.c:704 */, IMPB_OUT_FRAME = 1 /* pixels for a host encoder */,
       IMPB_OUT_INFO = 2 /* width, height, brightness (bridge.c:283-300) */,
       IMPB_OUT_ASCII = 3 /* the text exit, ASCII() of filters.c:486-522 (bridge.c:669-670) */ };
/* answer codes besides IMP_*: the broker did not take the file (not a JPEG/PNG the device decodes, or damaged) -- the
 * worker decodes on the host as before and comes back with IMPB_IN_FRAME */
#define IMPB_NOT_TAKEN    (-1)

typedef struct {
    uint32_t magic, version;
    uint32_t nslots;
    uint32_t reserved0;
    uint64_t slot_data_bytes;           /* bytes of the data area of one slot (input at 0, answer behind it) */
    uint64_t slots_offset;              /* first impb_slot */
    uint64_t data_offset;               /* first data area */
    volatile uint32_t broker_pid;       /* 0 while no broker serves the segment */
    volatile uint32_t epoch;            /* changes with every broker start: watermark ids and claims of an older epoch are void */
    volatile uint32_t doorbell;         /* futex: bumped by every submit */
    volatile uint32_t sleepers;         /* broker threads asleep on the doorbell */
    volatile uint32_t device;           /* the GPU this broker drives */
    volatile uint32_t heartbeat;        /* bumped about every 100 ms by the broker */
    volatile uint64_t served;           /* requests answered since this broker started */
    volatile uint64_t batches;          /* launches of the batch path they went through */
} impb_header_fields;
typedef union { impb_header_fields f; uint8_t page[4096]; } impb_header;

typedef struct {
    volatile uint32_t state;            /* IMPB_* ; futex word */
    volatile uint32_t owner_pid;
    volatile uint32_t epoch;            /* header epoch when the slot was claimed */
    uint32_t reserved0;
    /* ---- request ---- */
    uint32_t in_kind, out_kind;
    uint64_t in_bytes;
    /* IMPB_IN_FILE, a JPEG the worker has prepared (impgpu_jpeg_unstuff; in_scan_bytes = 0: the file as it came): the data
     * area holds the file's head [0, in_head_bytes), then at in_scan_at (256-byte aligned) in_scan_bytes of entropy-coded data
     * out of their byte stuffing and IMPGPU_JPEG_SCAN_TAIL bytes of 0xFF; in_bytes covers all of it */
    uint64_t in_head_bytes, in_scan_at, in_scan_bytes;
    int32_t  in_w, in_h, in_c, in_step; /* IMPB_IN_FRAME / IMPB_IN_WATERMARK */
    int32_t  quality;                   /* IMPB_OUT_JPEG */
    int32_t  simple, need_flatten, filter_count;
    int32_t  crop_at, gravity_at, resize_at;        /* offsets into text[], -1 = absent */
    int32_t  ascii_at;                  /* IMPB_OUT_ASCII: the argument string of ASCII(), -1 = "" */
    int32_t  filter_at[IMPB_MAX_FILTERS];
    uint32_t max_target_w, max_target_h;
    int32_t  max_filters_count, allow_experiments;
    int32_t  watermark_id;              /* 0 = none; else the id IMPB_IN_WATERMARK returned in this epoch */
    int32_t  watermark_opacity, watermark_offset_x, watermark_offset_y;
    char     watermark_gravity_x, watermark_gravity_y;
    char     pad0[2];
    /* ---- answer ---- */
    int32_t  code, step;                /* IMP_* / IMPB_NOT_TAKEN, IMP_STEP_* */
    uint64_t out_offset, out_bytes;     /* within the slot's data area */
    int32_t  out_w, out_h, out_c, out_step;
    float    brightness;
    int32_t  batch_size;                /* how many requests shared the launches this one rode in (diagnostic) */
    uint32_t broker_us;                 /* TAKEN -> DONE, microseconds (diagnostic) */
    char     error[120];
    char     text[IMPB_TEXT_BYTES];
} impb_slot_fields;
typedef union { impb_slot_fields f; uint8_t page[4096]; } impb_slot;      /* one page each: no two workers share a line */
typedef char impb_header_is_one_page[sizeof(impb_header) == 4096 ? 1 : -1];
typedef char impb_slot_is_one_page[sizeof(impb_slot) == 4096 ? 1 : -1];

/* ---- worker side (glue/imp_gpu_client.c; C99, no HIP, no C++ runtime) ---- */
typedef struct impgpu_client impgpu_client;

typedef struct {
    int                  in_kind;       /* IMPB_IN_FILE / IMPB_IN_FRAME */
    const unsigned char* input;         /* NULL: the caller has already put the bytes where impgpu_client_input_buffer said */
    size_t               input_bytes;   /* IMPB_IN_FRAME: step * height */
    int                  width, height, channels, step;   /* IMPB_IN_FRAME */
    const impgpu_job*    job;           /* what RunJob parsed (bridge.c:335-372); NULL = no operators */
    const impgpu_config* config;        /* limits + watermark placement; ->watermark is ignored, watermark_id is used */
    int                  watermark_id;
    int                  out_kind;      /* IMPB_OUT_* */
    int                  quality;
    const char*          ascii_args;    /* IMPB_OUT_ASCII (NULL = "") */
} impgpu_client_request;

typedef struct {
    int                  code, step;    /* IMP_* (or IMPB_NOT_TAKEN) and the IMP_STEP_* it belongs to */
    const unsigned char* data;          /* inside the worker's slot: valid until this client's next call */
    size_t               bytes;
    int                  width, height, channels, row_step;
    float                brightness;
    int                  batch_size;
    unsigned             broker_us;
    const char*          error;
} impgpu_client_answer;

/* name = NULL: $IMPGPU_BROKER, else IMPB_DEFAULT_NAME.  IMP_ERROR_DEVICE when no live broker serves the segment (the
 * text of the reason: impgpu_client_last_error). */
int         impgpu_client_attach(const char* name, impgpu_client** out);
void        impgpu_client_detach(impgpu_client** client);
/* where a caller that can read its input straight into shared memory puts it (request.input = NULL then); NULL when
 * `bytes` does not fit a slot */
void*       impgpu_client_input_buffer(impgpu_client* client, size_t bytes);
/* One request, synchronously (RunJob, bridge.c:302): returns IMP_OK when the broker answered -- the request's own verdict
 * is answer->code -- and IMP_ERROR_DEVICE when it could not be reached, died meanwhile, or did not answer within
 * $IMPGPU_BROKER_TIMEOUT_MS (default 10000); IMP_ERROR_MALLOC_FAILED when the input does not fit a slot (the worker then takes its in-process path). */
int         impgpu_client_run(impgpu_client* client, const impgpu_client_request* request, impgpu_client_answer* answer);
/* PrepareWatermark's pixels (bridge.c:221-234), once per worker and broker epoch; *id goes into request.watermark_id.
 * The client keeps (pointer, geometry) and registers again by itself when the broker was replaced. */
int         impgpu_client_prepare_watermark(impgpu_client* client, const unsigned char* pixels, int width, int height,
                                            int channels, int step, int* id);
const char* impgpu_client_last_error(void);
/* What impgpu_client_run does with a JPEG file on its way into the slot (exported for tests and for callers that fill the
 * input buffer themselves): if `file` is a Huffman-coded sequential JPEG (SOF0 / SOF1) with no restart interval whose scan
 * has at least IMPB_PREPARE_MIN_SCAN bytes, writes into out[0, cap): the file up to and including its SOS header, then --
 * at *scan_at, the next multiple of 256 -- the entropy-coded bytes with FF 00 -> FF and fill bytes dropped, up to the marker
 * that ends them (the rules of jpeg_prepare_scan, csrc/imp_jpeg.cpp), then IMPGPU_JPEG_SCAN_TAIL bytes of 0xFF; returns 1.
 * Returns 0 -- nothing of `out` is meaningful, send the file as it is -- for every other file (another process, restart
 * markers, a marker sequence the library calls damaged, no room).  The pass costs what the memcpy into the slot cost. */
#define IMPB_PREPARE_MIN_SCAN 40960     /* (what impgpu.h says of small launches: their Huffman stage stays on the host, which reads the file as it is) */
int         impgpu_jpeg_unstuff(const unsigned char* file, size_t size, unsigned char* out, size_t cap,
                                size_t* head_bytes, size_t* scan_at, size_t* scan_bytes, size_t* total_bytes);
/* diagnostics: the broker's counters as the segment shows them now */
int         impgpu_client_stats(impgpu_client* client, unsigned long long* served, unsigned long long* batches,
                                unsigned* epoch, unsigned* broker_pid);

#ifdef __cplusplus
}
#endif
#endif
