"""stdin: bench.py --stream --jpeg JSON lines -> one short line each"""
import json
import sys

for line in sys.stdin:
    try:
        d = json.loads(line)
    except ValueError:
        continue
    c = d["config"]
    print("threads %d batch %d %-8s %8.1f req/s %7.1f MB/s compressed %.2f bpp | answers: %s, %.0f B each" % (
        c["threads_per_gpu"], c["files_per_decode_call"], d["decoder"], d["value"], d["compressed_MB_per_sec"], d["bits_per_pixel"],
        d.get("answers", "raw"), d.get("answer_bytes_per_request", 0)), flush=True)
