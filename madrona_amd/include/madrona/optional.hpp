// Minimal Optional for trivially-destructible payloads (node ids, indices).
// API contract: reference include/madrona/optional.hpp (none(), has_value(),
// operator*, operator->, value_or).
#pragma once

#include <madrona/macros.hpp>

#include <utility>

namespace madrona {

template <typename T>
class Optional {
public:
    MADRONA_HD static constexpr Optional none() { return Optional(); }

    template <typename... Args>
    MADRONA_HD static constexpr Optional make(Args &&...args)
    {
        Optional o;
        o.value_ = T { std::forward<Args>(args)... };
        o.valid_ = true;
        return o;
    }

    MADRONA_HD constexpr Optional(const T &v) : value_(v), valid_(true) {}
    MADRONA_HD constexpr Optional(const Optional &) = default;
    MADRONA_HD constexpr Optional &operator=(const Optional &) = default;

    MADRONA_HD constexpr bool has_value() const { return valid_; }
    MADRONA_HD constexpr explicit operator bool() const { return valid_; }

    MADRONA_HD constexpr T &operator*() { return value_; }
    MADRONA_HD constexpr const T &operator*() const { return value_; }
    MADRONA_HD constexpr T *operator->() { return &value_; }
    MADRONA_HD constexpr const T *operator->() const { return &value_; }

    MADRONA_HD constexpr T value_or(const T &fallback) const
    {
        return valid_ ? value_ : fallback;
    }

    MADRONA_HD constexpr void reset() { valid_ = false; }

private:
    MADRONA_HD constexpr Optional() : value_(), valid_(false) {}

    T value_;
    bool valid_;
};

}
