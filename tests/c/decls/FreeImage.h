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
FIMEMORY* FreeImage_OpenMemory(BYTE* data, DWORD size);
void FreeImage_CloseMemory(FIMEMORY* stream);
FREE_IMAGE_FORMAT FreeImage_GetFileTypeFromMemory(FIMEMORY* stream, int size);
FREE_IMAGE_FORMAT FreeImage_GetFIFFromFilename(const char* filename);
#endif
