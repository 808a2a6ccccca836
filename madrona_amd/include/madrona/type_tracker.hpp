// Type -> dense id registry, usable from host AND device code.
// API contract: reference include/madrona/type_tracker.hpp:14-35
// (typeID<T>(), registerType<T>(&counter), unassignedTypeID).
//
// MI355X design: ids are assigned on the host (registerTypes runs on the
// host in this backend) and mirrored into a per-type __device__ variable of
// the simulator's code object with hipMemcpyToSymbol, so device code resolves
// typeID<T>() with a single scalar load -- no NVRTC-time id patching
// (reference src/mw/cuda_exec.cpp:795-874) and no name hashing.
#pragma once

#include <madrona/macros.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace madrona {

namespace mwhip {

template <typename T>
struct HostTypeID {
    static inline uint32_t id = 0xFFFFFFFFu;
};

#if defined(__HIPCC__)
// `used`: the symbol must exist in the code object for hipMemcpyToSymbol even
// when no kernel happens to read this type's id.
// __constant__: written once from the host before any kernel runs, so the
// compiler may treat reads as invariant (scalar loads, hoisted above stores).
template <typename T>
__constant__ __attribute__((used)) uint32_t deviceTypeID = 0xFFFFFFFFu;
#endif

}

class TypeTracker {
public:
    static constexpr uint32_t unassignedTypeID = 0xFFFFFFFFu;

    template <typename T>
    MADRONA_HD static inline uint32_t typeID()
    {
#if defined(__HIP_DEVICE_COMPILE__)
        return mwhip::deviceTypeID<T>;
#else
        return mwhip::HostTypeID<T>::id;
#endif
    }

    // Referenced from the device copies of the registration functions so that
    // the device compilation pass instantiates (and emits) T's id variable.
    template <typename T>
    MADRONA_HD static inline void touchDeviceSymbol()
    {
#if defined(__HIP_DEVICE_COMPILE__)
        (void)*(volatile uint32_t *)&mwhip::deviceTypeID<T>;
#endif
    }

    // Host only. Assigns *next_id_ptr (post-incrementing it) unless T already
    // has an id, then (re)publishes the id to the current device.
    template <typename T>
    static void registerType(uint32_t *next_id_ptr)
    {
        uint32_t &id = mwhip::HostTypeID<T>::id;
        if (id == unassignedTypeID) {
            id = (*next_id_ptr)++;
        }
#if defined(__HIPCC__)
        hipError_t res = hipMemcpyToSymbol(
            HIP_SYMBOL(mwhip::deviceTypeID<T>), &id, sizeof(uint32_t));
        // A type whose id no device function reads has no device symbol
        // (nothing instantiated it in the device pass); nothing to publish.
        if (res != hipSuccess) {
            (void)hipGetLastError();
        }
#endif
    }
};

}
