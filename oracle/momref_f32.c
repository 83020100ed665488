/*
 * oracle/momref_f32.c -- TEST INFRASTRUCTURE ONLY: the Float32 build of oracle/momref.c.
 *
 * The same source text compiled with every `double` of the restatement replaced by `float`, the libm calls by their `f`
 * forms and (-fsingle-precision-constant) every literal read as a Float32 literal: the reference's algorithm as it runs
 * with `float_type: Float32` (parameters_from_yaml.jl:160; every array of the model is FT).  It is the checker of the
 * library's dtype = 1 path (tests/test_gpu_rt_run.py::test_float32_*): against the Float64 oracle a Float32 run can only
 * be held to eps32 / dtau_elemental ~ 2e-2 ... 0.2; against this build the comparison is Float32 rounding against Float32
 * rounding of the same operations.  The doubling numbers and interface codes come from the caller, as in the Float64
 * build.  All arrays of the C interface are `float` here; oracle/cref.py converts (round to nearest).
 *
 * Build: gcc -O2 -fopenmp -fPIC -fsingle-precision-constant -ffp-contract=off -shared oracle/momref_f32.c
 *            -o oracle/libmomref_f32.so -lm   (oracle/Makefile)
 */
#include <float.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define ORA_EXT 1 /* plain column-sweep GEMM: the vector micro-kernel of the Float64 build is typed on double */
#define double float
#define exp expf
#define fabs fabsf
#define ldexp ldexpf
#define sqrt sqrtf
#include "momref.c"
