// trainer.h -- same include path as the reference (include/tiny-cuda-nn/trainer.h); the declarations live in tcnn_api.h.
#pragma once
#include "tcnn_api.h"
