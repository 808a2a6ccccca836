// API contract: reference include/madrona/exec_mode.hpp
#pragma once

namespace madrona {

enum class ExecMode : uint32_t {
    CPU,
    CUDA, // the GPU backend: HIP on MI355X in this framework
};

}
