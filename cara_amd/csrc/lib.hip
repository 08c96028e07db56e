// Library identification for libcara_hip.so.
#include "common.h"

extern "C" int cara_abi_version(void) { return 1; }
extern "C" const char* cara_build_arch(void) { return "gfx950"; }
