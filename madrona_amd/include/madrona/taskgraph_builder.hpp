// Simulators include <madrona/taskgraph_builder.hpp> on both backends
// (reference include/madrona/taskgraph_builder.hpp); in this backend the
// builder and the built-in nodes live in taskgraph.hpp.
#pragma once

#include <madrona/taskgraph.hpp>
