/* TEST INFRASTRUCTURE ONLY -- reduced-precision builds of ipm_ref.c (precision study of the HIP
 * arithmetic): _f32 = everything float; _mix = float I/O and matrices, double vectors and
 * double stage Hessian Quu + Cholesky (what the HIP kernels do). */
#ifdef CMPC_MIX
#define REAL double
#define MREAL float
#define PREAL float
#define CREAL double
#define FN(n) n##_mix
#else
#define REAL float
#define MREAL float
#define PREAL float
#define CREAL float
#define FN(n) n##_f32
#endif
#include "ipm_ref.c"
