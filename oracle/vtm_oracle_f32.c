/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  See vtm_oracle.h.
 *
 * fp32 instantiation of the restatement (vtm_oracle_body.inc): TFloat = float, i.e.
 * VocalTractModel0<float> (model 1), VocalTractModel2<float,D>, VocalTractModel4<float,1>;
 * std::pow/cos/sin/tan/sqrt/rint/abs on float resolve to the float overloads (powf, cosf, ...).
 */
#include "vtm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef float real;
#define RC(x) ((real) (x))      /* a literal of the reference written as TFloat */
#define PUB(name) name##_f32
#define R_POW powf
#define R_COS cosf
#define R_SIN sinf
#define R_TAN tanf
#define R_SQRT sqrtf
#define R_RINT rintf
#define R_FABS fabsf

#include "vtm_oracle_body.inc"
