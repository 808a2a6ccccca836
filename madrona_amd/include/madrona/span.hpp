// Non-owning view.  API contract: reference include/madrona/span.hpp
// (pointer + count, constructible from arrays / initializer lists).
#pragma once

#include <madrona/macros.hpp>
#include <madrona/types.hpp>

#include <array>
#include <cstddef>
#include <initializer_list>
#include <type_traits>

namespace madrona {

template <typename T>
class Span {
public:
    MADRONA_HD constexpr Span() : ptr_(nullptr), n_(0) {}
    MADRONA_HD constexpr Span(T *ptr, CountT num_elems)
        : ptr_(ptr), n_(num_elems) {}

    template <typename U = T,
              std::enable_if_t<std::is_const_v<U>, int> = 0>
    constexpr Span(std::initializer_list<std::remove_const_t<T>> init)
        : ptr_(init.begin()), n_((CountT)init.size()) {}

    template <size_t N>
    MADRONA_HD constexpr Span(T (&arr)[N]) : ptr_(arr), n_((CountT)N) {}

    template <typename U, size_t N>
    constexpr Span(const std::array<U, N> &arr)
        : ptr_(arr.data()), n_((CountT)N) {}

    template <typename U, size_t N>
    constexpr Span(std::array<U, N> &arr)
        : ptr_(arr.data()), n_((CountT)N) {}

    // Span<T> -> Span<const T>
    template <typename U,
              std::enable_if_t<std::is_same_v<const U, T>, int> = 0>
    MADRONA_HD constexpr Span(const Span<U> &o)
        : ptr_(o.data()), n_(o.size()) {}

    MADRONA_HD constexpr T *data() const { return ptr_; }
    MADRONA_HD constexpr CountT size() const { return n_; }
    MADRONA_HD constexpr bool empty() const { return n_ == 0; }

    MADRONA_HD constexpr T &operator[](CountT idx) const { return ptr_[idx]; }

    MADRONA_HD constexpr T *begin() const { return ptr_; }
    MADRONA_HD constexpr T *end() const { return ptr_ + n_; }

private:
    T *ptr_;
    CountT n_;
};

}
