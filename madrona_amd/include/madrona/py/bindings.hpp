// JAX / XLA custom-call bridge of a simulator's Manager.
// API contract: reference include/madrona/py/bindings.hpp:19-52 +
// bindings.inl:9-121 (JAXInterface::buildEntry<iface_fn, cpu_init_fn,
// cpu_step_fn, gpu_init_fn, gpu_step_fn, ...save / restore checkpoint fns>,
// cpuEntryFn / gpuEntryFn = the functions XLA calls) and
// src/python/bindings.cpp:222-283 (setup: the capsules, the opaque that carries
// the simulator's address, the interface pytrees).  Differences: pybind11
// instead of nanobind (not installed here), hipStream_t instead of
// cudaStream_t; the Python half -- primitives, lowerings, the ordering token
// -- is madrona_amd/jax_register.py (reference src/python/jax_register.py),
// imported when setup() runs, so that importing a Manager's module does not
// need jax.
//
//   .def("jax", JAXInterface::buildEntry<
//       &Manager::trainInterface, &Manager::cpuJAXInit, &Manager::cpuJAXStep,
//       &Manager::gpuJAXInit, &Manager::gpuJAXStep>())
//
// with   void Manager::cpuJAXStep(void **inputs, void **outputs);
//        void Manager::gpuJAXStep(hipStream_t strm, void **buffers);
#pragma once

#include <madrona/py/utils.hpp>

#include <pybind11/pybind11.h>

#include <cstddef>
#include <functional>
#include <type_traits>

// (== the HIP runtime's own declaration; a Manager's bindings that do not
// otherwise include the runtime still get the right signature)
struct ihipStream_t;
typedef struct ihipStream_t *hipStream_t;

namespace madrona::py {

namespace detail {
template <typename T> struct ClassOfMember;
template <typename C, typename R, typename... A>
struct ClassOfMember<R (C::*)(A...)> { using type = C; };
template <typename C, typename R, typename... A>
struct ClassOfMember<R (C::*)(A...) const> { using type = C; };

template <auto fn>
inline constexpr bool isNull = std::is_same_v<decltype(fn), std::nullptr_t>;
}

class JAXInterface {
public:
    // Returns the callable to bind as a method of the simulator class:
    // (sim, xla_gpu: bool) -> {"init": fn, "step": fn[, "save_ckpts",
    // "restore_ckpts"]}, the jitted functions of jax_register.py.
    template <auto iface_fn,
              auto cpu_init_fn,
              auto cpu_step_fn,
              auto gpu_init_fn = nullptr,
              auto gpu_step_fn = nullptr,
              auto cpu_save_ckpts_fn = nullptr,
              auto cpu_restore_ckpts_fn = nullptr,
              auto gpu_save_ckpts_fn = nullptr,
              auto gpu_restore_ckpts_fn = nullptr>
    static auto buildEntry();

    // What XLA calls.  CPU custom call: (out buffers, in buffers); in[0] holds
    // the simulator's address (an operand constant), in[1] the ordering token.
    template <typename SimT, auto fn>
    static void cpuEntryFn(void **out, void **in)
    {
        SimT *sim = *(SimT **)in[0];
        std::invoke(fn, *sim, in + 2, out);
    }

    // GPU custom call (API_VERSION_ORIGINAL: stream, buffers, opaque, length):
    // the opaque is the simulator's address; buffers[0] is the operand the CPU
    // flavour reads it from, buffers[1] the token.
    template <typename SimT, auto fn>
    static void gpuEntryFn(hipStream_t strm, void **buffers,
                           const char *opaque, size_t)
    {
        SimT *sim = *(SimT **)opaque;
        std::invoke(fn, *sim, strm, buffers + 2);
    }

    // The entry points without jax: {"init": capsule, "step": capsule, ...,
    // "sim_encode": bytes, "platform": "cpu" | "gpu"} -- what setup() hands to
    // jax_register.py, and what tests call the way XLA would.
    static inline pybind11::dict capsules(void *sim_ptr, void *init_fn,
                                          void *step_fn, void *save_ckpts_fn,
                                          void *restore_ckpts_fn, bool xla_gpu);

    static inline pybind11::dict setup(const TrainInterface &iface,
                                       pybind11::object sim_obj, void *sim_ptr,
                                       void *init_fn, void *step_fn,
                                       void *save_ckpts_fn,
                                       void *restore_ckpts_fn, bool xla_gpu);

    // {"actions": {name: (shape, dtype)}, "resets": ..., "sim_ctrl": ...,
    //  "pbt": {...}} and {"obs": {...}, "rewards": ..., "dones": ...,
    //  "stats": {...}, "pbt": {...}} with numpy dtype names: what
    // jax_register.py turns into jax.ShapeDtypeStruct
    static inline pybind11::dict inputsToPytree(const TrainInterface &iface);
    static inline pybind11::dict outputsToPytree(const TrainInterface &iface);
};

}

#include "bindings.inl"
