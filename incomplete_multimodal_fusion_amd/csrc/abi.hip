#include "common.hpp"
#include "mmae_hip.h"
extern "C" int mmae_abi_version(void) { return MMAE_ABI_VERSION; }

#include "common.hpp"
thread_local int mmae_tls_last_hip_error = 0;
// hipError_t of the most recent launch of this thread that returned MMAE_ERR_LAUNCH (0 if none); reading it resets it.
extern "C" int mmae_last_hip_error(void) { const int e = mmae_tls_last_hip_error; mmae_tls_last_hip_error = 0; return e; }
extern "C" const char* mmae_hip_error_name(int code) { return hipGetErrorName((hipError_t)code); }
