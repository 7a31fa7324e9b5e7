/*
 * rogue_worker.c -- a worker that does NOT go through impgpu_client_run: it fills its slot's request record by hand, with
 * offsets a correct client never writes, and rings the doorbell (the client's own map / roundtrip, this file includes its
 * source).  What the broker must do with each: answer with an error code, touch nothing outside the slot, and serve the next
 * request as if nothing had happened.
 *   rogue_worker <segment name> <file.jpg>
 * Prints one line per case: "<case> code <answer code> step <step>"; the last case is the same file handed over correctly.
 */
#include "../../glue/imp_gpu_client.c"

static int submit(impgpu_client* c, const char* what) {
    impb_slot_fields* s = &c->slot->f;
    s->in_kind = IMPB_IN_FILE; s->out_kind = IMPB_OUT_JPEG; s->quality = 86;
    s->crop_at = s->gravity_at = s->ascii_at = -1; s->resize_at = 0; memcpy(s->text, "224,0", 6);
    s->simple = s->need_flatten = s->filter_count = 0; s->watermark_id = 0;
    s->max_target_w = s->max_target_h = 0; s->max_filters_count = 0; s->allow_experiments = 0;
    const int rc = roundtrip(c);
    printf("%s rc %d code %d step %d bytes %llu\n", what, rc, rc ? 0 : s->code, rc ? 0 : s->step, rc ? 0ull : (unsigned long long)s->out_bytes);
    return rc;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* fp = fopen(argv[2], "rb");
    if (!fp) return 3;
    fseek(fp, 0, SEEK_END); const size_t size = (size_t)ftell(fp); fseek(fp, 0, SEEK_SET);
    unsigned char* file = (unsigned char*)malloc(size);
    if (fread(file, 1, size, fp) != size) return 3;
    fclose(fp);
    impgpu_client* c = NULL;
    if (impgpu_client_attach(argv[1], &c) != IMP_OK || ready(c) != IMP_OK) { fprintf(stderr, "attach: %s\n", impgpu_client_last_error()); return 4; }
    impb_slot_fields* s = &c->slot->f;
    const size_t cap = (size_t)c->hdr->f.slot_data_bytes;
    size_t head = 0, at = 0, n = 0, total = 0;
    if (!impgpu_jpeg_unstuff(file, size, c->data, cap, &head, &at, &n, &total)) { fprintf(stderr, "the file is not one a worker prepares\n"); return 5; }
    /* 1: the scan does not start on a 256-byte boundary */
    s->in_bytes = total; s->in_head_bytes = head; s->in_scan_at = at + 4; s->in_scan_bytes = n - 4;
    if (submit(c, "unaligned")) return 6;
    /* 2: the scan reaches past the bytes handed over (no room for the tail) */
    s->in_bytes = total; s->in_head_bytes = head; s->in_scan_at = at; s->in_scan_bytes = n + 600;
    if (submit(c, "overlong")) return 6;
    /* 3: the scan claims to lie past the slot altogether */
    s->in_bytes = total; s->in_head_bytes = head; s->in_scan_at = (cap + 4096) & ~(size_t)255; s->in_scan_bytes = n;
    if (submit(c, "outside")) return 6;
    /* 4: the head overlaps the scan */
    s->in_bytes = total; s->in_head_bytes = at + 300; s->in_scan_at = at; s->in_scan_bytes = n;
    if (submit(c, "overlap")) return 6;
    /* 5: in_bytes past the slot */
    s->in_bytes = cap + 1; s->in_head_bytes = head; s->in_scan_at = at; s->in_scan_bytes = n;
    if (submit(c, "toolong")) return 6;
    /* 6: offsets that hold, a head cut short of its scan (the library's own check) */
    s->in_bytes = total; s->in_head_bytes = head - 3; s->in_scan_at = at; s->in_scan_bytes = n;
    if (submit(c, "shorthead")) return 6;
    /* 7: counts in the 2^63 range */
    s->in_bytes = total; s->in_head_bytes = head; s->in_scan_at = at; s->in_scan_bytes = ~(uint64_t)0 - 100;
    if (submit(c, "huge")) return 6;
    /* and the same file as a client hands it over */
    s->in_bytes = total; s->in_head_bytes = head; s->in_scan_at = at; s->in_scan_bytes = n;
    if (submit(c, "correct")) return 6;
    impgpu_client_detach(&c);
    return 0;
}
