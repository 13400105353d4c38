// network_with_input_encoding.h -- same include path as the reference (include/tiny-cuda-nn/network_with_input_encoding.h); the declarations live in tcnn_api.h.
#pragma once
#include "tcnn_api.h"
