/* TEST INFRASTRUCTURE ONLY -- see ipm_ref_f32.c */
#define CMPC_MIX 1
#include "ipm_ref_f32.c"
