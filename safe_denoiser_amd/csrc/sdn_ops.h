// Internal (non-ABI) declarations shared between the translation units of libsdn.
#pragma once
#include "../../include/sdn.h"

int sdn_gemm_pick_nrep(int n_padded, int act);
