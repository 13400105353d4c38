// optimizer.h -- same include path as the reference (include/tiny-cuda-nn/optimizer.h); the declarations live in tcnn_api.h.
#pragma once
#include "tcnn_api.h"
