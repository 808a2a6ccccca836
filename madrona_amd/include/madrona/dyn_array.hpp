// Growable array.
// API contract: reference include/madrona/dyn_array.hpp:26-305 (push_back /
// emplace_back / pop_back / resize(n, init_fn) / reserve / clear / release /
// retrieve_ptr / uninit_back / insert / emplace / destruct, iteration, CountT
// sizes, allocator parameter with alloc / dealloc).
#pragma once

#include <madrona/memory.hpp>
#include <madrona/types.hpp>

#include <cstring>
#include <initializer_list>
#include <new>
#include <type_traits>
#include <utility>

namespace madrona {

template <typename T, typename A = DefaultAlloc>
class DynArray {
public:
    using RefT = std::add_lvalue_reference_t<T>;

    explicit DynArray(CountT init_capacity, A alloc = A())
        : alloc_(std::move(alloc)), items_(nullptr), count_(0), room_(0)
    {
        if (init_capacity > 0) {
            moveTo(init_capacity);
        }
    }

    DynArray(std::initializer_list<T> init, A alloc = A())
        : DynArray((CountT)init.size(), std::move(alloc))
    {
        for (const T &v : init) push_back(v);
    }

    DynArray(const DynArray &) = delete;
    DynArray &operator=(const DynArray &) = delete;

    DynArray(DynArray &&o)
        : alloc_(std::move(o.alloc_)), items_(o.items_), count_(o.count_),
          room_(o.room_)
    {
        o.items_ = nullptr;
        o.count_ = 0;
        o.room_ = 0;
    }

    DynArray &operator=(DynArray &&o)
    {
        if (this != &o) {
            release();
            alloc_ = std::move(o.alloc_);
            items_ = o.items_;
            count_ = o.count_;
            room_ = o.room_;
            o.items_ = nullptr;
            o.count_ = 0;
            o.room_ = 0;
        }
        return *this;
    }

    ~DynArray() { release(); }

    // destroys the elements, keeps the storage
    void clear()
    {
        if constexpr (!std::is_trivially_destructible_v<T>) {
            for (CountT i = count_; i > 0; i--) items_[i - 1].~T();
        }
        count_ = 0;
    }

    // destroys the elements and frees the storage
    void release()
    {
        clear();
        if (items_ != nullptr) {
            alloc_.dealloc(items_);
        }
        items_ = nullptr;
        room_ = 0;
    }

    // hands the storage (and the elements in it) to the caller
    T *retrieve_ptr()
    {
        T *out = items_;
        items_ = nullptr;
        count_ = 0;
        room_ = 0;
        return out;
    }

    void reserve(CountT new_capacity)
    {
        if (new_capacity > room_) {
            moveTo(new_capacity);
        }
    }

    void set_min_capacity(CountT capacity) { reserve(capacity); }

    // grows or shrinks to new_size; fn(T *slot) constructs each new element
    template <typename Fn>
    void resize(CountT new_size, Fn &&fn)
    {
        if (new_size > room_) {
            moveTo(growTarget(new_size));
        }
        while (count_ > new_size) pop_back();
        while (count_ < new_size) {
            fn(items_ + count_);
            count_ += 1;
        }
    }

    template <typename... Args>
    RefT emplace_back(Args &&...args)
    {
        makeRoomForOne();
        new (items_ + count_) T(std::forward<Args>(args)...);
        return items_[count_++];
    }

    RefT push_back(const T &v) { return emplace_back(v); }
    RefT push_back(T &&v) { return emplace_back(std::move(v)); }

    // index of a new, UNconstructed slot at the end
    CountT uninit_back()
    {
        makeRoomForOne();
        return count_++;
    }

    void pop_back()
    {
        count_ -= 1;
        items_[count_].~T();
    }

    // (re)construct / destroy one element in place, by index
    template <typename... Args>
    RefT emplace(CountT i, Args &&...args)
    {
        new (items_ + i) T(std::forward<Args>(args)...);
        return items_[i];
    }

    RefT insert(CountT i, const T &v) { return emplace(i, v); }
    RefT insert(CountT i, T &&v) { return emplace(i, std::move(v)); }
    void destruct(CountT i) { items_[i].~T(); }

    RefT operator[](CountT idx) { return items_[idx]; }
    const T &operator[](CountT idx) const { return items_[idx]; }

    T *data() { return items_; }
    const T *data() const { return items_; }
    T *begin() { return items_; }
    T *end() { return items_ + count_; }
    const T *begin() const { return items_; }
    const T *end() const { return items_ + count_; }

    RefT front() { return items_[0]; }
    const T &front() const { return items_[0]; }
    RefT back() { return items_[count_ - 1]; }
    const T &back() const { return items_[count_ - 1]; }

    CountT size() const { return count_; }

private:
    // 1.5x growth, at least 4 elements
    CountT growTarget(CountT at_least) const
    {
        CountT target = room_ + room_ / 2;
        if (target < 4) target = 4;
        return target > at_least ? target : at_least;
    }

    void makeRoomForOne()
    {
        if (count_ == room_) {
            moveTo(growTarget(count_ + 1));
        }
    }

    // moves the elements into fresh storage of `capacity` slots
    void moveTo(CountT capacity)
    {
        T *fresh = (T *)alloc_.alloc(sizeof(T) * (size_t)capacity);
        if (items_ != nullptr) {
            if constexpr (std::is_trivially_copyable_v<T>) {
                if (count_ > 0) {
                    memcpy((void *)fresh, (const void *)items_,
                           sizeof(T) * (size_t)count_);
                }
            } else {
                for (CountT i = 0; i < count_; i++) {
                    new (fresh + i) T(std::move(items_[i]));
                    items_[i].~T();
                }
            }
            alloc_.dealloc(items_);
        }
        items_ = fresh;
        room_ = capacity;
    }

    [[no_unique_address]] A alloc_;
    T *items_;
    CountT count_;
    CountT room_;
};

}
