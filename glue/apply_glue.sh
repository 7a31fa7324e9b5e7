#!/bin/bash
# apply_glue.sh <checkout of tommiv/ngx_http_imgproc> -- turns the reference module into a caller of libimpgpu.so.
#
# The edits are addressed by line number, so the script first checks that bridge.c / required.h / config are the
# revision those numbers belong to (the one SURVEY.md and every file:line citation in this repository refer to).
# It carries none of the reference's text: it deletes line ranges and inserts the calls into glue/imp_gpu_bridge.c.
# Edits go bottom-up so earlier line numbers stay valid.
set -euo pipefail
T=${1:?usage: apply_glue.sh <reference checkout>}
HERE=$(cd "$(dirname "$0")" && pwd)

check() {  # file sha256
    local have
    have=$(sha256sum "$T/$1" | cut -d' ' -f1)
    [ "$have" = "$2" ] || { echo "$1 is not the revision this glue was written against (sha256 $have)" >&2; exit 1; }
}
check bridge.c   f0bf6dfca6739f396636dd961677306d6b75481076328ca36e79a826f6290b24
check required.h 27c2f9c0a283e9cfeb740d8a7edb1392639da0f9986dd92759db8d2f28155b05
check config     c05ab98b00581607a549d347b8fb5a3927f349cec2a000959b8bc77dd2c54ae7

mkdir -p "$T/glue"
cp "$HERE/imp_gpu_bridge.c" "$HERE/imp_gpu_bridge.h" "$T/glue/"
cp "$HERE/config" "$T/config"                                   # config:1-5 -> glue sources + -limpgpu

B="$T/bridge.c"
# finalize: (bridge.c:714) -- device frames are released on every exit path past the decoder
sed -i '714a\
		ImpGpuRelease(\&gpu);' "$B"
# the basic encoder's JPEG case (bridge.c:703-709) is written on the device; its PNG case falls through to cvEncodeImage
sed -i '702a\
		if (answer->MIME == IMP_MIME_JPG) {\
			answer->Code = ImpGpuEncodeJpeg(\&gpu, basicCoderopt[1], req->pool, \&answer->EncodedBytes, \&answer->Length);\
			goto finalize;\
		}' "$B"
# every other encoder (bridge.c:683-710) reads IplImages: bring the results back right after Step = ENCODE / Code = OK (bridge.c:681)
sed -i '681a\
	if (encodeAdvancedIO || answer->MIME != IMP_MIME_JPG) {\
		answer->Code = ImpGpuDownload(\&gpu, \&album, req->pool);\
		if (answer->Code) {\
			goto finalize;\
		}\
	}' "$B"
# text exit (bridge.c:669-670): ASCII() on the device frame
sed -i '669,670c\
		Memory res = ImpGpuASCII(\&gpu, quality ? quality : "", req->pool);' "$B"
# json exit (bridge.c:661-662): Info() with the brightness reduction on the device; a device error is an error, not "brightness 0"
sed -i '661,662c\
		u_char* json = ImpGpuInfo(\&gpu, \&album, req->pool, \&answer->Code);\
		if (answer->Code) {\
			goto finalize;\
		}' "$B"
# Steps 3-7 (bridge.c:574-656): the crop / resize / filter / watermark / flatten loops
sed -i '574,656c\
	// Steps 3-7: main operators, on the GPU (glue/imp_gpu_bridge.c -> libimpgpu.so)\
	{\
		int lacksAlpha = answer->MIME == IMP_MIME_JPG;\
		#ifdef IMP_FEATURE_ADVANCED_IO\
			if (encodeAdvancedIO) {\
				lacksAlpha = !FiSupports32bit(encodeAdvancedIO);\
			}\
			int simple = album.Count > 0 \&\& encodeAdvancedIO == FIF_GIF;\
		#else\
			int simple = 0;\
		#endif\
		answer->Code = ImpGpuOperators(\&album, \&gpu, req->pool, crop, gravity, resize, simple, filters, filterCount, lacksAlpha, config, \&answer->Step);\
		if (answer->Code) {\
			goto finalize;\
		}\
	}\
' "$B"
# Step 2 (bridge.c:545): a JPEG is decoded on the device; whatever ImpGpuDecode does not take goes to cvDecodeImage as before
sed -i '545c\
	ImpGpuAlbum gpu = { NULL };\
	if (decodeBasicIo \&\& ImpGpuDecode(blob, size, \&album, \&gpu, req->pool)) {\
		// the frame is in HBM already\
	} else if (decodeBasicIo) {' "$B"
# worker lifecycle (bridge.c:10-16): the two "No op" bodies
sed -i '15c\
	ImpGpuEnvDestroy();' "$B"
sed -i '11c\
	ImpGpuEnvStart(IMP_GPU_WORKER_INDEX);' "$B"
sed -i '5a\
#include "glue/imp_gpu_bridge.h"' "$B"

# Config (required.h:108-118) gains the per-worker, per-location handle of the uploaded overlay (NULL from ngx_pcalloc,
# module.c:118; filled by the first request that needs it, glue/imp_gpu_bridge.c FillConfig)
sed -i '117a\
    void*        WatermarkDevice;' "$T/required.h"

echo "glue applied to $T: build nginx with --add-module=$T and IMPGPU_HOME=<this repository>"
