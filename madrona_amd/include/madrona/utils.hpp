// Small integer helpers.  API contract: reference include/madrona/utils.hpp
// (divideRoundUp, roundUp, roundUpPow2, int32NextPow2, int32Log2, u32mulhi).
#pragma once

#include <madrona/macros.hpp>
#include <madrona/types.hpp>

namespace madrona {
namespace utils {

template <typename T>
MADRONA_HD constexpr inline T divideRoundUp(T a, T b)
{
    return (a + (b - 1)) / b;
}

template <typename T>
MADRONA_HD constexpr inline T roundUp(T offset, T alignment)
{
    return divideRoundUp(offset, alignment) * alignment;
}

// alignment must be a power of two
template <typename T>
MADRONA_HD constexpr inline T roundUpPow2(T offset, T alignment)
{
    return (offset + alignment - 1) & ~(alignment - 1);
}

MADRONA_HD constexpr inline bool isPower2(uint64_t v)
{
    return v != 0 && (v & (v - 1)) == 0;
}

MADRONA_HD constexpr inline uint32_t int32NextPow2(uint32_t v)
{
    if (v <= 1) return 1;
    return 1u << (32 - __builtin_clz(v - 1));
}

MADRONA_HD constexpr inline uint32_t int32Log2(uint32_t v)
{
    return 31u - (uint32_t)__builtin_clz(v);
}

MADRONA_HD constexpr inline uint32_t u32mulhi(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

template <typename> struct PackDelegator;
template <template <typename...> typename T, typename... Args>
struct PackDelegator<T<Args...>> {
    template <typename Fn>
    static auto call(Fn &&fn) -> decltype(fn.template operator()<Args...>())
    {
        return fn.template operator()<Args...>();
    }
};

}
}
