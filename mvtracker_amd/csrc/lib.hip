// Library introspection entry points.
#include "common.h"

extern "C" int mvt_abi_version(void) { return 7; }
extern "C" const char* mvt_build_arch(void) { return "gfx950"; }
