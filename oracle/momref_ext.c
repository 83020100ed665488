/*
 * oracle/momref_ext.c -- TEST INFRASTRUCTURE ONLY: the EXTENDED-PRECISION build of oracle/momref.c.
 *
 * The same source text compiled with every `double` of the restatement replaced by the x87 `long double` (64-bit
 * mantissa, eps = 1.08e-19, 2048 times finer than Float64) and the libm calls by their `l` forms.  It is the arbiter of
 * tests/test_gpu_precision.py: for optically thick layers (many doublings) two correct Float64 implementations differ
 * by a multiple of 2^ndoubl eps, so "GPU vs Float64 oracle" alone cannot tell whose rounding is worse; against this
 * build (whose own error is 2^ndoubl x 1e-19, i.e. 1e-12 at ndoubl = 23) both can be measured.
 * All arrays of the C interface are `long double` here (numpy.longdouble on x86-64); oracle/cref.py converts.
 *
 * Build: gcc -O2 -fopenmp -fPIC -shared oracle/momref_ext.c -o oracle/libmomref_ext.so -lm   (oracle/Makefile)
 */
#include <float.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define ORA_EXT 1
#define double long double
#define exp expl
#define fabs fabsl
#define ldexp ldexpl
#define sqrt sqrtl
#include "momref.c"
