// ABI identification entry points of libfluxhip.
#include "../../include/fluxhip.h"

extern "C" int fluxhip_abi_version(void) { return FLUXHIP_ABI_VERSION; }
extern "C" const char* fluxhip_arch(void) { return "gfx950"; }
