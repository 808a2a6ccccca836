// Fixed-capacity arrays that live inside their owner (no heap).
// API contract: reference include/madrona/inline_array.hpp:16-131 (InlineArray:
// push_back / emplace_back / pop_back / clear / size / capacity / iteration;
// FixedInlineArray: emplace(i, ...) into N always-present slots).
#pragma once

#include <madrona/macros.hpp>

#include <cstddef>
#include <new>
#include <type_traits>
#include <utility>

namespace madrona {

namespace mwhip {

// raw, suitably aligned room for N objects of type T
template <typename T, size_t N>
struct InlineSlots {
    alignas(T) unsigned char bytes[sizeof(T) * (N > 0 ? N : 1)];

    MADRONA_HD T *ptr() { return std::launder(reinterpret_cast<T *>(bytes)); }
    MADRONA_HD const T *ptr() const
    {
        return std::launder(reinterpret_cast<const T *>(bytes));
    }
};

}

template <typename T, size_t N>
class InlineArray {
public:
    MADRONA_HD InlineArray() : count_(0) {}
    InlineArray(const InlineArray &) = delete;
    InlineArray &operator=(const InlineArray &) = delete;

    MADRONA_HD ~InlineArray() { clear(); }

    template <typename... Args>
    MADRONA_HD T &emplace_back(Args &&...args)
    {
        T *slot = slots_.ptr() + count_;
        new (slot) T(std::forward<Args>(args)...);
        count_ += 1;
        return *slot;
    }

    MADRONA_HD T &push_back(const T &v) { return emplace_back(v); }
    MADRONA_HD T &push_back(T &&v) { return emplace_back(std::move(v)); }

    MADRONA_HD void pop_back()
    {
        count_ -= 1;
        slots_.ptr()[count_].~T();
    }

    MADRONA_HD void clear()
    {
        if constexpr (!std::is_trivially_destructible_v<T>) {
            while (count_ > 0) pop_back();
        }
        count_ = 0;
    }

    MADRONA_HD T &operator[](size_t idx) { return slots_.ptr()[idx]; }
    MADRONA_HD const T &operator[](size_t idx) const { return slots_.ptr()[idx]; }

    MADRONA_HD T *data() { return slots_.ptr(); }
    MADRONA_HD const T *data() const { return slots_.ptr(); }
    MADRONA_HD T *begin() { return data(); }
    MADRONA_HD T *end() { return data() + count_; }
    MADRONA_HD const T *begin() const { return data(); }
    MADRONA_HD const T *end() const { return data() + count_; }

    MADRONA_HD size_t size() const { return count_; }
    MADRONA_HD constexpr size_t capacity() const { return N; }

private:
    mwhip::InlineSlots<T, N> slots_;
    size_t count_;
};

// N slots that always exist; emplace(i, ...) (re)constructs one of them.  Like
// the reference's, slots start out uninitialised for trivial T.
template <typename T, size_t N>
class FixedInlineArray {
public:
    MADRONA_HD FixedInlineArray()
    {
        if constexpr (!std::is_trivially_default_constructible_v<T>) {
            for (size_t i = 0; i < N; i++) new (slots_.ptr() + i) T();
        }
    }
    FixedInlineArray(const FixedInlineArray &) = delete;
    FixedInlineArray &operator=(const FixedInlineArray &) = delete;

    MADRONA_HD ~FixedInlineArray()
    {
        if constexpr (!std::is_trivially_destructible_v<T>) {
            for (size_t i = N; i > 0; i--) slots_.ptr()[i - 1].~T();
        }
    }

    template <typename... Args>
    MADRONA_HD T &emplace(size_t i, Args &&...args)
    {
        T *slot = slots_.ptr() + i;
        if constexpr (!std::is_trivially_destructible_v<T>) {
            slot->~T();
        }
        new (slot) T(std::forward<Args>(args)...);
        return *slot;
    }

    MADRONA_HD T &operator[](size_t idx) { return slots_.ptr()[idx]; }
    MADRONA_HD const T &operator[](size_t idx) const { return slots_.ptr()[idx]; }

    MADRONA_HD T *data() { return slots_.ptr(); }
    MADRONA_HD const T *data() const { return slots_.ptr(); }
    MADRONA_HD T *begin() { return data(); }
    MADRONA_HD T *end() { return data() + N; }
    MADRONA_HD const T *begin() const { return data(); }
    MADRONA_HD const T *end() const { return data() + N; }

    MADRONA_HD constexpr size_t size() const { return N; }

private:
    mwhip::InlineSlots<T, N> slots_;
};

}
