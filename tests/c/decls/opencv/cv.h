/* declarations only -- see ../ngx_config.h */
#ifndef DECLS_OPENCV_CV_H
#define DECLS_OPENCV_CV_H
typedef struct { int width, height; } CvSize;
typedef struct { int x, y; } CvPoint;
typedef struct { int x, y, width, height; } CvRect;
typedef struct { double val[4]; } CvScalar;
typedef struct _IplImage {
    int nChannels, depth, width, height, widthStep, imageSize;
    char* imageData;
} IplImage;
typedef struct CvMat {
    int type, step, rows, cols;
    union { unsigned char* ptr; } data;
} CvMat;
typedef void CvArr;
#define IPL_DEPTH_8U 8
#define CV_8U 0
#define CV_8UC1 0
#define CV_INTER_NN 0
#define CV_INTER_LINEAR 1
#define CV_INTER_CUBIC 2
#define CV_INTER_AREA 3
#define CV_GRAY2BGR 8
#define CV_GAUSSIAN 2
CvSize    cvSize(int width, int height);
CvPoint   cvPoint(int x, int y);
CvRect    cvRect(int x, int y, int width, int height);
CvMat     cvMat(int rows, int cols, int type, void* data);
IplImage* cvCreateImage(CvSize size, int depth, int channels);
IplImage* cvCreateImageHeader(CvSize size, int depth, int channels);
void      cvReleaseImage(IplImage** image);
void      cvReleaseImageHeader(IplImage** image);
void      cvReleaseMat(CvMat** mat);
void      cvSetData(CvArr* arr, void* data, int step);
CvSize    cvGetSize(const CvArr* arr);
void      cvSetImageROI(IplImage* image, CvRect rect);
void      cvResetImageROI(IplImage* image);
CvRect    cvGetImageROI(const IplImage* image);
void      cvCopy(const CvArr* src, CvArr* dst, const CvArr* mask);
void      cvResize(const CvArr* src, CvArr* dst, int interpolation);
void      cvCvtColor(const CvArr* src, CvArr* dst, int code);
void      cvFlip(const CvArr* src, CvArr* dst, int mode);
void      cvTranspose(const CvArr* src, CvArr* dst);
void      cvSmooth(const CvArr* src, CvArr* dst, int type, int p1, int p2, double p3, double p4);
#endif
