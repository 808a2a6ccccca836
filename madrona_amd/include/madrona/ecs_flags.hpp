// API contract: reference include/madrona/ecs_flags.hpp
#pragma once

#include <madrona/ecs.hpp>

#include <array>

namespace madrona {

enum class ArchetypeFlags : uint32_t {
    None = 0,
    ImportOffsets = 1u << 0,
};

enum class ComponentFlags : uint32_t {
    None = 0,
    ExportMemory = 1u << 0,
    ImportMemory = 1u << 1,
    CudaReserveMemory = 1u << 2,
    CudaAllocMemory = 1u << 3,
};

template <typename... ComponentTs>
struct ComponentMetadataSelector {
    std::array<ComponentFlags, sizeof...(ComponentTs)> flags;

    MADRONA_HD inline ComponentMetadataSelector() : flags {} {}

    MADRONA_HD inline ComponentMetadataSelector(ComponentFlags component_flags)
    {
        for (auto &f : flags) f = component_flags;
    }

    template <typename... FlagTs>
    MADRONA_HD inline ComponentMetadataSelector(FlagTs... in_flags)
        : flags { in_flags... }
    {}
};

#define MADRONA_FLAG_OPS(T) \
    MADRONA_HD inline T operator|(T a, T b) { return T(uint32_t(a) | uint32_t(b)); } \
    MADRONA_HD inline T operator&(T a, T b) { return T(uint32_t(a) & uint32_t(b)); } \
    MADRONA_HD inline T &operator|=(T &a, T b) { a = a | b; return a; } \
    MADRONA_HD inline T &operator&=(T &a, T b) { a = a & b; return a; }

MADRONA_FLAG_OPS(ArchetypeFlags)
MADRONA_FLAG_OPS(ComponentFlags)

#undef MADRONA_FLAG_OPS

}
