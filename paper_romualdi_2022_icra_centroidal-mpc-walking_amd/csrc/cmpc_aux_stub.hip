// temporary stubs (replaced by cmpc_nlp_eval.hip)
#include "cmpc_device.h"
extern "C" int cmpc_launch_nlp_eval(const CmpcParams*, const float*, const float*, const float*, float, float*, float*, float*, float*, float*, hipStream_t) { return (int)hipErrorNotSupported; }
extern "C" int cmpc_launch_warm_shift(const CmpcParams*, const float*, float*, hipStream_t) { return (int)hipErrorNotSupported; }
extern "C" int cmpc_nlp_sparsity(int, int*, int*, int*, int*) { return -1; }
