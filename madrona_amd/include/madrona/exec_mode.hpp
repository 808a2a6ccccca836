// API contract: reference include/madrona/exec_mode.hpp
#pragma once

#include <cstdint>

namespace madrona {

// CUDA names "the GPU backend" in simulator code; here that is HIP on MI355X.
enum class ExecMode : uint32_t { CPU, CUDA };

}
