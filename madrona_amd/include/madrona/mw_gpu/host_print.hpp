// mwGPU::HostPrint::log("fmt {} {}", a, b) -- formatted printing from device
// code (API contract: reference src/mw/device/include/madrona/mw_gpu/
// host_print.hpp:13-50, host_print.inl:9-56; host side src/mw/cuda_exec.cpp:
// HostPrintCPU).  "{}" placeholders, arguments of type int32_t / uint32_t /
// int64_t / uint64_t / float / pointer.
//
// The reference funnels every message through ONE channel behind a device-wide
// spin lock and a host thread that answers each message before the next can be
// written.  Here messages go into a ring of fixed-size records in pinned host
// memory: a writer claims a slot with one system-scope atomic, fills it and
// publishes it with a release store of its sequence number; the executor
// drains the ring after every replay it waits for (mwhip_run /
// mwhip_synchronize) and from a background thread in between, printing in
// sequence order.  Writers never wait for the host; when the ring is full the
// message takes no ticket, is counted as dropped and the drop count is reported.
#pragma once

#include <madrona/taskgraph.hpp>
#include <madrona/mwhip/ecs_state.hpp>

#include <cstdint>
#include <type_traits>

namespace madrona {
namespace mwGPU {

class HostPrint {
public:
    template <typename... Args>
    MADRONA_HD static inline void log(const char *str, Args &&...args)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        static_assert(sizeof...(Args) <= mwhip::HostPrintRecord::maxArgs,
                      "too many arguments for one HostPrint::log");
        StateManager *mgr = getStateManager();
        mwhip::HostPrintRing *ring = mgr->hostPrintRing;
        // (world constructors run twice -- a counting pass, then the real one,
        // ecs_state.hpp expandIdStore: only the real one speaks)
        if (ring == nullptr || mgr->initMode == 1u) {
            return;
        }

        // A ticket is only taken when the ring has room for it (compare-and-
        // swap on head against the host's tail): every ticket below head is a
        // record that is being, or has been, written, so the host can wait for
        // a slow writer without ever mistaking it for a dropped message.  (A
        // stale tail only makes the test conservative.)  Writers contend with
        // each other only: a lost race means another message got its ticket,
        // so the loop ends after at most as many rounds as the ring has room.
        unsigned long long seq = __hip_atomic_load(&ring->head,
            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        bool taken = false;
        while (!taken) {
            const uint64_t tail = __hip_atomic_load(&ring->tail,
                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (seq - tail >= mwhip::HostPrintRing::numRecords) {
                break;      // the host has not caught up
            }
            taken = __hip_atomic_compare_exchange_strong(&ring->head, &seq,
                seq + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (!taken) {
            __hip_atomic_fetch_add(&ring->dropped, 1ull, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        mwhip::HostPrintRecord &rec =
            ring->records[seq % mwhip::HostPrintRing::numRecords];

        int32_t n = 0;
        while (n < mwhip::HostPrintRecord::maxChars - 1 && str[n] != '\0') {
            rec.fmt[n] = str[n];
            n++;
        }
        rec.fmt[n] = '\0';
        rec.numArgs = (uint32_t)sizeof...(Args);
        int32_t i = 0;
        (storeArg(rec, i++, args), ...);
        (void)i;

        // the sequence number (+1, so that 0 means "never written") is the
        // record's valid flag
        __hip_atomic_store(&rec.seq, seq + 1ull, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
#else
        (void)str;
        ((void)args, ...);
#endif
    }

private:
#if defined(__HIP_DEVICE_COMPILE__)
    template <typename T>
    MADRONA_DEVICE static inline void storeArg(mwhip::HostPrintRecord &rec,
                                               int32_t i, const T &v)
    {
        using U = std::decay_t<T>;
        if constexpr (std::is_same_v<U, float>) {
            rec.types[i] = mwhip::HostPrintRecord::Float;
            rec.args[i] = (uint64_t)__builtin_bit_cast(uint32_t, v);
        } else if constexpr (std::is_pointer_v<U>) {
            rec.types[i] = mwhip::HostPrintRecord::Ptr;
            rec.args[i] = (uint64_t)v;
        } else if constexpr (std::is_integral_v<U> && sizeof(U) == 8) {
            rec.types[i] = std::is_signed_v<U> ? mwhip::HostPrintRecord::I64 :
                                                 mwhip::HostPrintRecord::U64;
            rec.args[i] = (uint64_t)v;
        } else if constexpr (std::is_integral_v<U> && sizeof(U) <= 4) {
            rec.types[i] = std::is_signed_v<U> ? mwhip::HostPrintRecord::I32 :
                                                 mwhip::HostPrintRecord::U32;
            rec.args[i] = std::is_signed_v<U> ?
                (uint64_t)(int64_t)(int32_t)v : (uint64_t)(uint32_t)v;
        } else {
            static_assert(!std::is_same_v<U, U>,
                          "HostPrint::log: unsupported argument type");
        }
    }
#endif
};

}
}
