/* TEST INFRASTRUCTURE ONLY -- float32 build of ipm_ref.c (precision study of the HIP arithmetic). */
#define REAL float
#define FN(n) n##_f32
#include "ipm_ref.c"
