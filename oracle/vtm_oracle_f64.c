/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  See vtm_oracle.h.
 *
 * fp64 instantiation of the restatement (vtm_oracle_body.inc): TFloat = double, i.e.
 * VocalTractModel0<double> (model 0), VocalTractModel2<double,D> (models 2, 3), VocalTractModel4<double,1> (model 4).
 */
#include "vtm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef double real;
#define RC(x) ((real) (x))      /* a literal of the reference written as TFloat */
#define PUB(name) name##_f64
#define R_POW pow
#define R_COS cos
#define R_SIN sin
#define R_TAN tan
#define R_SQRT sqrt
#define R_RINT rint
#define R_FABS fabs

#include "vtm_oracle_body.inc"
