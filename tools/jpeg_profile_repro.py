import sys, os, ctypes as C, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch
import bench
import ngx_http_imgproc_amd as imp
imp.env_start(0)
lib = imp.lib
files = bench.jpeg_pool(8)
print("profile ->", lib.impgpu_jpeg_profile(1), flush=True)
for n in (1, 8):
    blobs = (C.c_char_p * n)(*[b for _, _, b in files[:n]])
    sizes = (C.c_size_t * n)(*[len(b) for _, _, b in files[:n]])
    imgs = (C.c_void_p * n)(); codes = (C.c_int * n)()
    print("decode", n, flush=True)
    rc = lib.impgpu_batch_decode_jpeg(blobs, sizes, n, imgs, codes)
    print("rc", rc, list(codes), flush=True)
    st = (C.c_double * 16)(); lib.impgpu_jpeg_stage_times(st, 16); print([round(x, 1) for x in st], flush=True)
imp.env_destroy()
