#include "common.hpp"
#include "mmae_hip.h"
extern "C" int mmae_abi_version(void) { return MMAE_ABI_VERSION; }
