// Host scratch allocator handed to the asset baker.
// API contract: reference include/madrona/stack_alloc.hpp (push / pop /
// alloc / allocN).  Minimal implementation: a list of malloc'd blocks.
#pragma once

#include <madrona/types.hpp>

#include <cstdlib>
#include <vector>

namespace madrona {

class StackAlloc {
public:
    struct Frame {
        size_t numAllocs;
    };

    inline StackAlloc(CountT = 0) {}
    StackAlloc(const StackAlloc &) = delete;
    inline ~StackAlloc() { release(); }

    inline Frame push() { return Frame { allocs_.size() }; }

    inline void pop(Frame frame)
    {
        while (allocs_.size() > frame.numAllocs) {
            free(allocs_.back());
            allocs_.pop_back();
        }
    }

    inline void *alloc(size_t num_bytes, size_t alignment)
    {
        if (alignment < 16) alignment = 16;
        size_t rounded = (num_bytes + alignment - 1) / alignment * alignment;
        void *ptr = aligned_alloc(alignment, rounded == 0 ? alignment : rounded);
        allocs_.push_back(ptr);
        return ptr;
    }

    template <typename T>
    inline T *alloc() { return (T *)alloc(sizeof(T), alignof(T)); }

    template <typename T>
    inline T *allocN(CountT num_elems)
    {
        return (T *)alloc(sizeof(T) * (size_t)num_elems, alignof(T));
    }

    inline void release() { pop(Frame { 0 }); }

private:
    std::vector<void *> allocs_;
};

}
