/* orc_internal.h -- helpers shared by the oracle's translation units. TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include <limits.h>
#include <math.h>
#include "imp_oracle.h"

/* C's float/double -> int conversion as x86-64 gcc performs it (cvttss2si / cvttsd2si):
 * truncation toward zero, and INT_MIN ("integer indefinite") for NaN or out-of-range.
 * The reference relies on it for (int)inf in CalculateGammaLUT and friends. */
static inline int orc_trunc(double v) {
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}
/* Store into `char`/`unsigned char` through cvSetComponent (helpers.h:2): the int value's low byte. */
static inline unsigned char orc_byte(int v) { return (unsigned char)(v & 0xff); }
/* float/double expression assigned to a char lvalue: convert (as above) then keep the low byte. */
static inline unsigned char orc_store(double v) { return orc_byte(orc_trunc(v)); }
/* OpenCV cvRound: round-half-to-even under the default rounding mode. */
static inline int orc_cvround(double v) { return (int)lrint(v); }
static inline unsigned char orc_sat_u8(int v) { return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }

orc_image* orc_cv_flip(const orc_image* src, int mode);
orc_image* orc_cv_transpose(const orc_image* src);

#endif
