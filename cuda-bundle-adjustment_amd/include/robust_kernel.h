// Include-compatibility shim: the reference spreads these types over several headers
// (ref: src/robust_kernel.h); this build keeps them in cugo_types.h.
#pragma once
#include "cugo_types.h"
