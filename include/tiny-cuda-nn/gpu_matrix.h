// gpu_matrix.h -- same include path as the reference (include/tiny-cuda-nn/gpu_matrix.h); the declarations live in tcnn_api.h.
#pragma once
#include "tcnn_api.h"
