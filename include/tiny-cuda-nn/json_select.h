// json_select.h -- which type tcnn::json is (tcnn_api.h, cpp_api.h).
// The reference hands nlohmann::json objects through its factories (config.h:53, cpp_api.h:113-115), vendored as
// dependencies/json/json.hpp and included by its callers as <json/json.hpp>.  Where that header is on the include path tcnn::json
// IS nlohmann::json -- iterators, items(), get<T>(), implicit conversions: everything callers use beyond the subset that
// json_lite.h implements.  Without it (or with -DTCNN_AMD_JSON_LITE) tcnn::json is the self-contained tcnn_amd::Json.
#pragma once

#include "json_lite.h"

#if !defined(TCNN_AMD_JSON_LITE) && defined(__has_include)
#if __has_include(<json/json.hpp>)
#include <json/json.hpp>
#define TCNN_AMD_HAVE_NLOHMANN_JSON 1
#endif
#endif

namespace tcnn {
#ifdef TCNN_AMD_HAVE_NLOHMANN_JSON
using json = nlohmann::json;
#else
using json = tcnn_amd::Json;
#endif
} // namespace tcnn
