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
check advancedio.c a33b9fe31ad7f0f6a888c6c182fc8ee9959ddfcf82455776c44ea66f74af78be

mkdir -p "$T/glue"
cp "$HERE/imp_gpu_bridge.c" "$HERE/imp_gpu_bridge.h" "$HERE/imp_gpu_client.c" "$T/glue/"
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
# -- except FreeImage's single-frame encoders: SaveSingle fetches the frame straight into the bitmap it encodes (advancedio.c:428)
sed -i '681a\
	album.Device = NULL;\
	#ifdef IMP_FEATURE_ADVANCED_IO\
		if (encodeAdvancedIO \&\& encodeAdvancedIO != FIF_GIF \&\& album.Count == 1) {\
			album.Device = gpu.Handle;\
		}\
	#endif\
	if (!album.Device \&\& (encodeAdvancedIO || answer->MIME != IMP_MIME_JPG)) {\
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
# FreeImage's decoders (bridge.c:565) leave their frames in HBM (advancedio.c edits below): RunJob takes the handle over
sed -i '565a\
			gpu.Handle = album.Device;' "$B"
# Step 2 (bridge.c:545): a JPEG is decoded on the device; whatever ImpGpuDecode does not take goes to cvDecodeImage as before;
# a DEVICE failure fails the request at its DECODE step (bridge.c:568-571 returns album.Error)
sed -i '545c\
	ImpGpuAlbum gpu = { NULL };\
	album.Device = NULL;\
	int gpuDecoded = decodeBasicIo ? ImpGpuDecode(blob, size, \&album, \&gpu, req->pool) : 0;\
	if (gpuDecoded < 0) {\
		album.Error = -gpuDecoded;\
	} else if (gpuDecoded) {\
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

# Album (required.h:136-140) gains the device handle FiLoadFrames returns its frames in / SaveSingle fetches its frame from
sed -i '139a\
    void*  Device;' "$T/required.h"

# ---- advancedio.c: the FreeImage side hands frames to / takes them from the device without an IplImage in between
A="$T/advancedio.c"
# SaveSingle (advancedio.c:428-429): IplToFI32 / IplToFI24 on the device, into the bitmap FreeImage encodes
sed -i '428,429c\
    FIBITMAP* frame;\
    if (source->Device) {\
        int bpp = FiSupports32bit(format) ? 32 : 24;\
        frame = FreeImage_Allocate(ImpGpuFrameWidth(source->Device), ImpGpuFrameHeight(source->Device), bpp, 0, 0, 0);\
        if (!frame || ImpGpuFetchFi(source->Device, bpp, FreeImage_GetBits(frame), FreeImage_GetPitch(frame))) {\
            result->Error = IMP_ERROR_ENCODE_FAILED;\
            if (frame) {\
                FreeImage_Unload(frame);\
            }\
            return;\
        }\
    } else {\
        IplImage* image = source->Frames[0].Image;\
        frame = FiSupports32bit(format) ? IplToFI32(image) : IplToFI24(image);\
    }' "$A"
# FiLoadFrames (advancedio.c:325-326): no device frames yet
sed -i '326a\
    result.Device = NULL;' "$A"
# LoadSingle (advancedio.c:295-318): the bottom-up 32-bit bitmap goes to the device as it is; the flip happens there
sed -i '295,318c\
    result->Error = ImpGpuLoadSingle(result, pool, FreeImage_GetBits(fullcolor), w, h, FreeImage_GetPitch(fullcolor));' "$A"
# LoadGIF (advancedio.c:260-262, the release of `master`): all pages collected -> composited on the device in one call
sed -i '260,262c\
    if (!result->Error) {\
        result->Error = ImpGpuGifCompose(\&gif, isdestructive, page, result);\
    }' "$A"
# LoadGIF (advancedio.c:187-248): the per-page IplImage, the `master` canvas and the per-pixel compositing loop
sed -i '187,248c\
        result->Frames[frameid].Image = NULL;\
        if (!result->Error) {\
            result->Error = ImpGpuGifPage(\&gif, pool, frameid, framecount, FreeImage_GetBits(frame), w, h, FreeImage_GetPitch(frame), left, top,\
                                          result->Frames[frameid].Dispose, result->Frames[frameid].TransparencyKey, palette, canvasW, canvasH);\
        }' "$A"
# LoadGIF (advancedio.c:122): the index canvas lives on the device now; the pages are collected here
sed -i '122c\
    ImpGpuGif gif = { NULL, 0 };' "$A"
sed -i '6a\
#include "glue/imp_gpu_bridge.h"' "$A"

echo "glue applied to $T: build nginx with --add-module=$T and IMPGPU_HOME=<this repository>"
