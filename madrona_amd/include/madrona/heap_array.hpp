// An array whose length is chosen once, at construction, on the heap.
// API contract: reference include/madrona/heap_array.hpp:23-137 (size fixed at
// construction; elements of non-trivial type are constructed by the user with
// emplace / insert and destroyed with destruct or at the end; release() hands
// the storage over as a Span).
#pragma once

#include <madrona/memory.hpp>
#include <madrona/span.hpp>
#include <madrona/types.hpp>

#include <initializer_list>
#include <new>
#include <type_traits>
#include <utility>

namespace madrona {

template <typename T, typename A = DefaultAlloc>
class HeapArray {
public:
    using RefT = std::add_lvalue_reference_t<T>;

    explicit HeapArray(CountT n, A alloc = A())
        : alloc_(std::move(alloc)),
          items_(n > 0 ? (T *)alloc_.alloc(sizeof(T) * (size_t)n) : nullptr),
          count_(n)
    {}

    HeapArray(std::initializer_list<T> init, A alloc = A())
        : HeapArray((CountT)init.size(), std::move(alloc))
    {
        CountT i = 0;
        for (const T &v : init) new (items_ + i++) T(v);
    }

    HeapArray(const HeapArray &) = delete;
    HeapArray &operator=(const HeapArray &) = delete;

    HeapArray(HeapArray &&o)
        : alloc_(std::move(o.alloc_)), items_(o.items_), count_(o.count_)
    {
        o.items_ = nullptr;
        o.count_ = 0;
    }

    HeapArray &operator=(HeapArray &&o)
    {
        if (this != &o) {
            clear();
            alloc_ = std::move(o.alloc_);
            items_ = o.items_;
            count_ = o.count_;
            o.items_ = nullptr;
            o.count_ = 0;
        }
        return *this;
    }

    ~HeapArray() { clear(); }

    // destroys every element and frees the storage
    void clear()
    {
        if (items_ == nullptr) return;
        if constexpr (!std::is_trivially_destructible_v<T>) {
            for (CountT i = count_; i > 0; i--) items_[i - 1].~T();
        }
        alloc_.dealloc(items_);
        items_ = nullptr;
        count_ = 0;
    }

    // the caller owns the storage (and its elements) from here on
    Span<T> release()
    {
        Span<T> out(items_, count_);
        items_ = nullptr;
        count_ = 0;
        return out;
    }

    template <typename... Args>
    RefT emplace(CountT i, Args &&...args)
    {
        new (items_ + i) T(std::forward<Args>(args)...);
        return items_[i];
    }

    RefT insert(CountT i, T &&v) { return emplace(i, std::move(v)); }
    RefT insert(CountT i, const T &v) { return emplace(i, v); }

    void destruct(CountT i) { items_[i].~T(); }

    RefT operator[](CountT idx) { return items_[idx]; }
    const T &operator[](CountT idx) const { return items_[idx]; }

    T *data() { return items_; }
    const T *data() const { return items_; }
    T *begin() { return items_; }
    T *end() { return items_ + count_; }
    const T *begin() const { return items_; }
    const T *end() const { return items_ + count_; }

    CountT size() const { return count_; }

private:
    [[no_unique_address]] A alloc_;
    T *items_;
    CountT count_;
};

}
