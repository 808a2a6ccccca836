// API contract: reference include/madrona/types.hpp
#pragma once

#include <cstdint>
#include <type_traits>

namespace madrona {

using u64 = uint64_t;
using i64 = int64_t;
using u32 = uint32_t;
using i32 = int32_t;
using u16 = uint16_t;
using i16 = int16_t;
using u8 = uint8_t;
using i8 = int8_t;
using f32 = float;

constexpr inline u32 operator ""_u32(unsigned long long v) { return u32(v); }
constexpr inline u64 operator ""_u64(unsigned long long v) { return u64(v); }
constexpr inline i32 operator ""_i32(unsigned long long v) { return i32(v); }
constexpr inline i64 operator ""_i64(unsigned long long v) { return i64(v); }

// The reference uses a 32-bit CountT in MADRONA_GPU_MODE (types.hpp:38-42);
// device-resident structs keep that width here.
using CountT = int32_t;

template <typename T>
concept EnumType = std::is_enum_v<T>;

}
