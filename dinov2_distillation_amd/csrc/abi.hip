#include "common.h"
#include "dkd_hip.h"
extern "C" int dkd_abi_version(void) { return DKD_ABI_VERSION; }
