// Atomics and a spin lock with the reference's names (API contract:
// reference include/madrona/sync.hpp:42-382 -- Atomic<T>, AtomicRef<T>,
// SpinLock, the AtomicU32 ... AtomicFloat aliases, sync::relaxed ... seq_cst).
//
// The reference wraps cuda::atomic<T, thread_scope_device>.  Here the device
// side is the compiler's scoped atomics at AGENT scope: on MI355X the eight
// XCDs have private, mutually non-coherent L2s, agent-scope atomics are
// executed at the memory side (they bypass those L2s), and acquire / release
// orders add the L2 write-back / invalidate that makes plain data published
// through them visible across XCDs.  Host code gets the same classes over the
// __atomic builtins, so headers shared by both sides compile once.
#pragma once

#include <madrona/macros.hpp>
#include <madrona/types.hpp>

#include <cstdint>
#include <type_traits>

namespace madrona {

namespace sync {
enum memory_order : int {
    relaxed = __ATOMIC_RELAXED,
    acquire = __ATOMIC_ACQUIRE,
    release = __ATOMIC_RELEASE,
    acq_rel = __ATOMIC_ACQ_REL,
    seq_cst = __ATOMIC_SEQ_CST,
};
}

namespace mwhip {

// One implementation for Atomic<T> (owns the word) and AtomicRef<T> (borrows
// it).  T: 4 or 8 bytes, trivially copyable; arithmetic on integers and float.
template <typename T>
struct AtomicOps {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8);
    static_assert(std::is_trivially_copyable_v<T>);

    using Bits = std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t>;
    // the builtins take integers and floating point directly; anything else
    // goes through its bit pattern
    static constexpr bool direct =
        std::is_integral_v<T> || std::is_floating_point_v<T>;

    template <int order>
    MADRONA_HD static inline T load(const T *p)
    {
        constexpr int o = order == sync::release || order == sync::acq_rel ?
            (int)sync::acquire : order;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (direct) {
            return __hip_atomic_load(p, o, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            return __builtin_bit_cast(T, __hip_atomic_load(
                (const Bits *)p, o, __HIP_MEMORY_SCOPE_AGENT));
        }
#else
        return __builtin_bit_cast(T, __atomic_load_n((const Bits *)p, o));
#endif
    }

    template <int order>
    MADRONA_HD static inline void store(T *p, T v)
    {
        constexpr int o = order == sync::acquire || order == sync::acq_rel ?
            (int)sync::release : order;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (direct) {
            __hip_atomic_store(p, v, o, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store((Bits *)p, __builtin_bit_cast(Bits, v), o,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
#else
        __atomic_store_n((Bits *)p, __builtin_bit_cast(Bits, v), o);
#endif
    }

    template <int order>
    MADRONA_HD static inline T exchange(T *p, T v)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_bit_cast(T, __hip_atomic_exchange(
            (Bits *)p, __builtin_bit_cast(Bits, v), order,
            __HIP_MEMORY_SCOPE_AGENT));
#else
        return __builtin_bit_cast(T, __atomic_exchange_n(
            (Bits *)p, __builtin_bit_cast(Bits, v), order));
#endif
    }

    template <int success, int failure>
    MADRONA_HD static inline bool casWeak(T *p, T &expected, T desired)
    {
        Bits want = __builtin_bit_cast(Bits, expected);
#if defined(__HIP_DEVICE_COMPILE__)
        bool ok = __hip_atomic_compare_exchange_weak(
            (Bits *)p, &want, __builtin_bit_cast(Bits, desired), success,
            failure, __HIP_MEMORY_SCOPE_AGENT);
#else
        bool ok = __atomic_compare_exchange_n(
            (Bits *)p, &want, __builtin_bit_cast(Bits, desired), true, success,
            failure);
#endif
        expected = __builtin_bit_cast(T, want);
        return ok;
    }

    template <int order>
    MADRONA_HD static inline T fetchAdd(T *p, T v)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        // (float too: gfx950 has global_atomic_add_f32 with return)
        return __hip_atomic_fetch_add(p, v, order, __HIP_MEMORY_SCOPE_AGENT);
#else
        if constexpr (std::is_integral_v<T>) {
            return __atomic_fetch_add(p, v, order);
        } else {
            T seen = load<sync::relaxed>(p);
            while (!casWeak<order, sync::relaxed>(p, seen, seen + v)) {}
            return seen;
        }
#endif
    }

    template <int order>
    MADRONA_HD static inline T fetchOr(T *p, T v)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return __hip_atomic_fetch_or(p, v, order, __HIP_MEMORY_SCOPE_AGENT);
#else
        return __atomic_fetch_or(p, v, order);
#endif
    }
};

}

template <typename T>
class Atomic {
    using Ops = mwhip::AtomicOps<T>;
public:
    MADRONA_HD constexpr Atomic(T v) : value_(v) {}

    template <sync::memory_order order>
    MADRONA_HD inline T load() const { return Ops::template load<order>(&value_); }
    MADRONA_HD inline T load_relaxed() const { return load<sync::relaxed>(); }
    MADRONA_HD inline T load_acquire() const { return load<sync::acquire>(); }

    template <sync::memory_order order>
    MADRONA_HD inline void store(T v) { Ops::template store<order>(&value_, v); }
    MADRONA_HD inline void store_relaxed(T v) { store<sync::relaxed>(v); }
    MADRONA_HD inline void store_release(T v) { store<sync::release>(v); }

    template <sync::memory_order order>
    MADRONA_HD inline T exchange(T v)
    {
        return Ops::template exchange<order>(&value_, v);
    }

    template <sync::memory_order success_order,
              sync::memory_order failure_order>
    MADRONA_HD inline bool compare_exchange_weak(T &expected, T desired)
    {
        return Ops::template casWeak<success_order, failure_order>(
            &value_, expected, desired);
    }

    template <sync::memory_order order>
    MADRONA_HD inline T fetch_add(T v)
    {
        return Ops::template fetchAdd<order>(&value_, v);
    }
    MADRONA_HD inline T fetch_add_relaxed(T v) { return fetch_add<sync::relaxed>(v); }
    MADRONA_HD inline T fetch_add_acquire(T v) { return fetch_add<sync::acquire>(v); }
    MADRONA_HD inline T fetch_add_release(T v) { return fetch_add<sync::release>(v); }
    MADRONA_HD inline T fetch_add_acq_rel(T v) { return fetch_add<sync::acq_rel>(v); }

    template <sync::memory_order order>
    MADRONA_HD inline T fetch_sub(T v)
    {
        return Ops::template fetchAdd<order>(&value_, (T)(T(0) - v));
    }
    MADRONA_HD inline T fetch_sub_relaxed(T v) { return fetch_sub<sync::relaxed>(v); }
    MADRONA_HD inline T fetch_sub_acquire(T v) { return fetch_sub<sync::acquire>(v); }
    MADRONA_HD inline T fetch_sub_release(T v) { return fetch_sub<sync::release>(v); }
    MADRONA_HD inline T fetch_sub_acq_rel(T v) { return fetch_sub<sync::acq_rel>(v); }

private:
    mutable T value_;
};

using AtomicU32 = Atomic<uint32_t>;
using AtomicI32 = Atomic<int32_t>;
using AtomicU64 = Atomic<uint64_t>;
using AtomicI64 = Atomic<int64_t>;
using AtomicFloat = Atomic<float>;
using AtomicCount = Atomic<CountT>;

template <typename T>
class AtomicRef {
    using Ops = mwhip::AtomicOps<T>;
public:
    MADRONA_HD AtomicRef(T &ref) : addr_(&ref) {}

    template <sync::memory_order order>
    MADRONA_HD inline T load() const { return Ops::template load<order>(addr_); }
    MADRONA_HD inline T load_relaxed() const { return load<sync::relaxed>(); }
    MADRONA_HD inline T load_acquire() const { return load<sync::acquire>(); }

    template <sync::memory_order order>
    MADRONA_HD inline void store(T v) { Ops::template store<order>(addr_, v); }
    MADRONA_HD inline void store_relaxed(T v) { store<sync::relaxed>(v); }
    MADRONA_HD inline void store_release(T v) { store<sync::release>(v); }

    template <sync::memory_order order>
    MADRONA_HD inline T exchange(T v)
    {
        return Ops::template exchange<order>(addr_, v);
    }

    template <sync::memory_order success_order,
              sync::memory_order failure_order>
    MADRONA_HD inline bool compare_exchange_weak(T &expected, T desired)
    {
        return Ops::template casWeak<success_order, failure_order>(
            addr_, expected, desired);
    }

    template <sync::memory_order order>
    MADRONA_HD inline T fetch_add(T v)
    {
        return Ops::template fetchAdd<order>(addr_, v);
    }
    MADRONA_HD inline T fetch_add_relaxed(T v) { return fetch_add<sync::relaxed>(v); }

    template <sync::memory_order order>
    MADRONA_HD inline T fetch_sub(T v)
    {
        return Ops::template fetchAdd<order>(addr_, (T)(T(0) - v));
    }

    template <sync::memory_order order>
    MADRONA_HD inline T fetch_or(T v)
    {
        return Ops::template fetchOr<order>(addr_, v);
    }

private:
    T *addr_;
};

using AtomicI32Ref = AtomicRef<int32_t>;
using AtomicU32Ref = AtomicRef<uint32_t>;
using AtomicI64Ref = AtomicRef<int64_t>;
using AtomicU64Ref = AtomicRef<uint64_t>;
using AtomicFloatRef = AtomicRef<float>;

// Test-and-test-and-set lock.  On the device a holder and a waiter may be
// lanes of ONE wavefront: take it with the "try, do the work, release inside
// one loop iteration" shape (mwhip::withWorldCache, ecs_state.hpp) or through
// tryLock(); a bare lock() from divergent lanes of a wave can spin forever,
// exactly as on any SIMT machine without independent thread scheduling.
class SpinLock {
public:
    MADRONA_HD void lock()
    {
        while (lock_.exchange<sync::acquire>(1) == 1) {
            while (lock_.load_relaxed() == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_s_sleep(1);
#endif
            }
        }
    }

    MADRONA_HD bool tryLock()
    {
        if (lock_.load_relaxed() == 1) {
            return false;
        }
        return lock_.exchange<sync::acquire>(1) == 0;
    }

    MADRONA_HD void unlock() { lock_.store_release(0); }

private:
    AtomicI32 lock_ { 0 };
};

struct alignas(MADRONA_CACHE_LINE) CacheAlignedU32 {
    uint32_t v;
};

}
