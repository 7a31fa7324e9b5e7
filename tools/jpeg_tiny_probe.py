"""One small JPEG at a time: the device's entropy stage, the entropy stage on the calling thread (IMPGPU_JPEG_HUFF=host) with the
pixel kernel on the device, and Pillow (libjpeg-turbo) on one core; each call waits until the frame is complete."""
import io, os, sys, time
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/ngx_http_imgproc_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import ngx_http_imgproc_amd as gpu
from ngx_http_imgproc_amd.workloads import photo_like
from PIL import Image
gpu.env_start(0)
def timed(fn, reps=60):
    for _ in range(5): fn()
    t=[]
    for _ in range(reps):
        t0=time.perf_counter(); fn(); t.append(time.perf_counter()-t0)
    t.sort(); return t[len(t)//2]*1e3
for (w,h) in ((64,64),(160,120),(320,240),(480,360),(640,480),(800,600),(1024,768),(1280,720),(1920,1080)):
    b=io.BytesIO(); Image.fromarray(photo_like(h,w,seed=3)).save(b,format="JPEG",quality=90,subsampling=2); blob=b.getvalue()
    def dev():
        rc,im=gpu.Image.decode_jpeg(blob); assert rc==0; gpu.lib.impgpu_sync(); im.release()      # (the frame is complete)
    def pil():
        np.asarray(Image.open(io.BytesIO(blob)))
    os.environ["IMPGPU_JPEG_HUFF"]="device"; d=timed(dev)
    os.environ["IMPGPU_JPEG_HUFF"]="host"; hh=timed(dev)
    del os.environ["IMPGPU_JPEG_HUFF"]; au=timed(dev)
    print("%4dx%-4d %7d B | device entropy %.3f ms | host entropy + device pixels %.3f ms | default (by size) %.3f ms | Pillow %.3f ms" % (w,h,len(blob),d,hh,au,timed(pil,20)))
