/* declarations only -- see ngx_config.h in this directory */
#ifndef DECLS_FREEIMAGE_H
#define DECLS_FREEIMAGE_H
#define FREEIMAGE_MAJOR_VERSION 3
#define FREEIMAGE_MINOR_VERSION 17
typedef struct FIMEMORY FIMEMORY;
typedef struct FIBITMAP FIBITMAP;
typedef int FREE_IMAGE_FORMAT;
typedef unsigned char BYTE;
typedef unsigned int DWORD;
typedef int BOOL;
#define FIF_UNKNOWN (-1)
#define FIF_BMP 0
#define FIF_JPEG 2
#define FIF_TARGA 17
#define FIF_TIFF 18
#define FIF_GIF 25
#define FIF_J2K 30
#define FIF_JP2 31
#define FIF_WEBP 35
#define FIF_JXR 36
#define BMP_SAVE_RLE 1
#define TARGA_SAVE_RLE 2
#define TIFF_DEFLATE 0x0200
#define TIFF_LZW 0x1000
#define TIFF_JPEG 0x8000
#define TIFF_NONE 0x0800
/* what advancedio.c itself uses (the patched file is compile-checked too): declarations only */
typedef struct FIMULTIBITMAP FIMULTIBITMAP;
typedef struct FITAG FITAG;
typedef struct { BYTE rgbBlue, rgbGreen, rgbRed, rgbReserved; } RGBQUAD;
typedef int FREE_IMAGE_COLOR_TYPE;
typedef int FREE_IMAGE_MDMODEL;
typedef int FREE_IMAGE_MDTYPE;
typedef int FREE_IMAGE_QUANTIZE;
#define FIC_RGBALPHA 4
#define FIMD_ANIMATION 9
#define FIQ_NNQUANT 1
#define FIDT_BYTE 1
#define FIDT_SHORT 3
#define FIDT_LONG 4
#define FIF_ICO 1
#define FIF_JNG 3
#define FIF_KOALA 4
#define FIF_LBM 5
#define FIF_IFF 5
#define FIF_MNG 6
#define FIF_PBM 7
#define FIF_PBMRAW 8
#define FIF_PCD 9
#define FIF_PCX 10
#define FIF_PGM 11
#define FIF_PGMRAW 12
#define FIF_PNG 13
#define FIF_PPM 14
#define FIF_PPMRAW 15
#define FIF_RAS 16
#define FIF_WBMP 19
#define FIF_PSD 20
#define FIF_CUT 21
#define FIF_XBM 22
#define FIF_XPM 23
#define FIF_DDS 24
#define FIF_HDR 26
#define FIF_FAXG3 27
#define FIF_SGI 28
#define FIF_EXR 29
#define FIF_PFM 32
#define FIF_PICT 33
#define FIF_RAW 34
FIBITMAP* FreeImage_Allocate(int width, int height, int bpp, unsigned red_mask, unsigned green_mask, unsigned blue_mask);
void FreeImage_Unload(FIBITMAP* dib);
FIBITMAP* FreeImage_LoadFromMemory(FREE_IMAGE_FORMAT fif, FIMEMORY* stream, int flags);
BOOL FreeImage_SaveToMemory(FREE_IMAGE_FORMAT fif, FIBITMAP* dib, FIMEMORY* stream, int flags);
BOOL FreeImage_AcquireMemory(FIMEMORY* stream, BYTE** data, DWORD* size_in_bytes);
FIMULTIBITMAP* FreeImage_LoadMultiBitmapFromMemory(FREE_IMAGE_FORMAT fif, FIMEMORY* stream, int flags);
BOOL FreeImage_SaveMultiBitmapToMemory(FREE_IMAGE_FORMAT fif, FIMULTIBITMAP* bitmap, FIMEMORY* stream, int flags);
BOOL FreeImage_CloseMultiBitmap(FIMULTIBITMAP* bitmap, int flags);
int FreeImage_GetPageCount(FIMULTIBITMAP* bitmap);
void FreeImage_AppendPage(FIMULTIBITMAP* bitmap, FIBITMAP* data);
FIBITMAP* FreeImage_LockPage(FIMULTIBITMAP* bitmap, int page);
void FreeImage_UnlockPage(FIMULTIBITMAP* bitmap, FIBITMAP* data, BOOL changed);
unsigned FreeImage_GetWidth(FIBITMAP* dib);
unsigned FreeImage_GetHeight(FIBITMAP* dib);
unsigned FreeImage_GetPitch(FIBITMAP* dib);
unsigned FreeImage_GetBPP(FIBITMAP* dib);
BYTE* FreeImage_GetBits(FIBITMAP* dib);
BYTE* FreeImage_GetScanLine(FIBITMAP* dib, int scanline);
RGBQUAD* FreeImage_GetPalette(FIBITMAP* dib);
FREE_IMAGE_COLOR_TYPE FreeImage_GetColorType(FIBITMAP* dib);
int FreeImage_GetTransparentIndex(FIBITMAP* dib);
void FreeImage_SetTransparentIndex(FIBITMAP* dib, int index);
void FreeImage_SetTransparent(FIBITMAP* dib, BOOL enabled);
FIBITMAP* FreeImage_ConvertTo8Bits(FIBITMAP* dib);
FIBITMAP* FreeImage_ConvertTo24Bits(FIBITMAP* dib);
FIBITMAP* FreeImage_ConvertTo32Bits(FIBITMAP* dib);
FIBITMAP* FreeImage_ColorQuantizeEx(FIBITMAP* dib, FREE_IMAGE_QUANTIZE quantize, int PaletteSize, int ReserveSize, RGBQUAD* ReservePalette);
BOOL FreeImage_GetMetadata(FREE_IMAGE_MDMODEL model, FIBITMAP* dib, const char* key, FITAG** tag);
BOOL FreeImage_SetMetadata(FREE_IMAGE_MDMODEL model, FIBITMAP* dib, const char* key, FITAG* tag);
FITAG* FreeImage_CreateTag(void);
void FreeImage_DeleteTag(FITAG* tag);
const char* FreeImage_GetTagKey(FITAG* tag);
const void* FreeImage_GetTagValue(FITAG* tag);
BOOL FreeImage_SetTagKey(FITAG* tag, const char* key);
BOOL FreeImage_SetTagType(FITAG* tag, FREE_IMAGE_MDTYPE type);
BOOL FreeImage_SetTagCount(FITAG* tag, DWORD count);
BOOL FreeImage_SetTagLength(FITAG* tag, DWORD length);
BOOL FreeImage_SetTagValue(FITAG* tag, const void* value);
FIMEMORY* FreeImage_OpenMemory(BYTE* data, DWORD size);
void FreeImage_CloseMemory(FIMEMORY* stream);
FREE_IMAGE_FORMAT FreeImage_GetFileTypeFromMemory(FIMEMORY* stream, int size);
FREE_IMAGE_FORMAT FreeImage_GetFIFFromFilename(const char* filename);
#endif
