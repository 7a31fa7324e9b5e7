/* declarations only -- see ../ngx_config.h */
#ifndef DECLS_OPENCV_HIGHGUI_H
#define DECLS_OPENCV_HIGHGUI_H
#include <opencv/cv.h>
#define CV_IMWRITE_JPEG_QUALITY 1
#define CV_IMWRITE_PNG_COMPRESSION 16
IplImage* cvDecodeImage(const CvMat* buf, int iscolor);
CvMat*    cvEncodeImage(const char* ext, const CvArr* image, const int* params);
#endif
