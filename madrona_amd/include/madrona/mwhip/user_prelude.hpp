// Included at the top of the generated wrapper around a simulator's sources,
// BEFORE `#pragma clang force_cuda_host_device begin`: pulls in the standard
// headers and the whole madrona overlay with their own explicit host/device
// annotations, so that only the simulator's own (unannotated) functions are
// affected by the pragma.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cassert>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <limits>
#include <memory>
#include <new>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include <madrona/macros.hpp>
#include <madrona/types.hpp>
#include <madrona/utils.hpp>
#include <madrona/span.hpp>
#include <madrona/optional.hpp>
#include <madrona/math.hpp>
#include <madrona/rand.hpp>
#include <madrona/ecs.hpp>
#include <madrona/ecs_flags.hpp>
#include <madrona/type_tracker.hpp>
#include <madrona/query.hpp>
#include <madrona/state.hpp>
#include <madrona/registry.hpp>
#include <madrona/context.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/taskgraph.hpp>
#include <madrona/taskgraph_builder.hpp>
#include <madrona/components.hpp>
#include <madrona/mw_gpu_entry.hpp>
