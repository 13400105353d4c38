// moved: the JSON value class is part of the public C++ surface (tcnn::json)
#pragma once
#include "../../include/tiny-cuda-nn/json_lite.h"
