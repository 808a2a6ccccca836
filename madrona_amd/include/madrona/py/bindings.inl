#pragma once

#include <bit>

namespace madrona::py {

namespace detail {

inline const char *numpyName(TensorElementType t)
{
    switch (t) {
    case TensorElementType::UInt8: return "uint8";
    case TensorElementType::Int8: return "int8";
    case TensorElementType::Int16: return "int16";
    case TensorElementType::Int32: return "int32";
    case TensorElementType::Int64: return "int64";
    case TensorElementType::Float16: return "float16";
    case TensorElementType::Float32: return "float32";
    }
    return "uint8";
}

inline pybind11::tuple shapeDtype(const Tensor &t)
{
    pybind11::tuple shape((size_t)t.numDims());
    for (int64_t i = 0; i < t.numDims(); i++) {
        shape[(size_t)i] = pybind11::int_(t.dims()[i]);
    }
    return pybind11::make_tuple(shape, pybind11::str(numpyName(t.type())));
}

inline pybind11::dict namedToDict(Span<const NamedTensor> ts)
{
    pybind11::dict d;
    for (const NamedTensor &t : ts) {
        d[pybind11::str(t.name)] = shapeDtype(t.tensor);
    }
    return d;
}

template <typename T>
inline void *fnAddress(T fn)
{
    static_assert(sizeof(T) == sizeof(void *));
    return std::bit_cast<void *>(fn);
}

}

pybind11::dict JAXInterface::inputsToPytree(const TrainInterface &iface)
{
    TrainStepInputInterface in = iface.stepInputs();
    pybind11::dict d;
    d["actions"] = detail::namedToDict(in.actions);
    d["resets"] = detail::shapeDtype(in.resets);
    d["sim_ctrl"] = detail::shapeDtype(in.simCtrl);
    d["pbt"] = detail::namedToDict(in.pbt);
    return d;
}

pybind11::dict JAXInterface::outputsToPytree(const TrainInterface &iface)
{
    TrainStepOutputInterface out = iface.stepOutputs();
    pybind11::dict d;
    d["obs"] = detail::namedToDict(out.observations);
    d["rewards"] = detail::shapeDtype(out.rewards);
    d["dones"] = detail::shapeDtype(out.dones);
    d["stats"] = detail::namedToDict(out.stats);
    d["pbt"] = detail::namedToDict(out.pbt);
    return d;
}

pybind11::dict JAXInterface::capsules(void *sim_ptr, void *init_fn,
                                      void *step_fn, void *save_ckpts_fn,
                                      void *restore_ckpts_fn, bool xla_gpu)
{
    pybind11::dict d;
    // (the name XLA's registry checks: xla_client.register_custom_call_target)
    constexpr const char *capsule_name = "xla._CUSTOM_CALL_TARGET";
    d["init"] = pybind11::capsule(init_fn, capsule_name);
    d["step"] = pybind11::capsule(step_fn, capsule_name);
    if (save_ckpts_fn != nullptr && restore_ckpts_fn != nullptr) {
        d["save_ckpts"] = pybind11::capsule(save_ckpts_fn, capsule_name);
        d["restore_ckpts"] = pybind11::capsule(restore_ckpts_fn, capsule_name);
    }
    d["sim_ptr"] = pybind11::int_((uint64_t)(uintptr_t)sim_ptr);
    d["sim_encode"] = pybind11::bytes((const char *)&sim_ptr, sizeof(char *));
    d["platform"] = pybind11::str(xla_gpu ? "gpu" : "cpu");
    return d;
}

pybind11::dict JAXInterface::setup(const TrainInterface &iface,
                                   pybind11::object sim_obj, void *sim_ptr,
                                   void *init_fn, void *step_fn,
                                   void *save_ckpts_fn, void *restore_ckpts_fn,
                                   bool xla_gpu)
{
    pybind11::dict scope = capsules(sim_ptr, init_fn, step_fn, save_ckpts_fn,
                                    restore_ckpts_fn, xla_gpu);
    scope["sim_obj"] = sim_obj;
    scope["step_inputs_iface"] = inputsToPytree(iface);
    scope["step_outputs_iface"] = outputsToPytree(iface);
    if (iface.checkpointing().has_value()) {
        pybind11::dict ckpt;
        ckpt["data"] = detail::shapeDtype(iface.checkpointing()->checkpointData);
        scope["ckpt_iface"] = ckpt;
    } else {
        scope["ckpt_iface"] = pybind11::none();
    }
    // (imports jax; raises ImportError with a plain message when it is absent)
    pybind11::object reg =
        pybind11::module_::import("madrona_amd.jax_register").attr("register");
    return reg(scope).cast<pybind11::dict>();
}

template <auto iface_fn, auto cpu_init_fn, auto cpu_step_fn, auto gpu_init_fn,
          auto gpu_step_fn, auto cpu_save_ckpts_fn, auto cpu_restore_ckpts_fn,
          auto gpu_save_ckpts_fn, auto gpu_restore_ckpts_fn>
auto JAXInterface::buildEntry()
{
    using SimT =
        typename detail::ClassOfMember<decltype(cpu_step_fn)>::type;

    // (sim, xla_gpu, register = True): register = False returns the capsules
    // and the interface instead of going through jax
    return [](pybind11::object sim, bool xla_gpu, bool register_with_jax) {
        void *init_fn = nullptr, *step_fn = nullptr;
        void *save_fn = nullptr, *restore_fn = nullptr;
        if (xla_gpu) {
            if constexpr (!detail::isNull<gpu_init_fn> &&
                          !detail::isNull<gpu_step_fn>) {
                init_fn = detail::fnAddress(
                    &JAXInterface::gpuEntryFn<SimT, gpu_init_fn>);
                step_fn = detail::fnAddress(
                    &JAXInterface::gpuEntryFn<SimT, gpu_step_fn>);
            }
            if constexpr (!detail::isNull<gpu_save_ckpts_fn> &&
                          !detail::isNull<gpu_restore_ckpts_fn>) {
                save_fn = detail::fnAddress(
                    &JAXInterface::gpuEntryFn<SimT, gpu_save_ckpts_fn>);
                restore_fn = detail::fnAddress(
                    &JAXInterface::gpuEntryFn<SimT, gpu_restore_ckpts_fn>);
            }
        } else {
            init_fn = detail::fnAddress(
                &JAXInterface::cpuEntryFn<SimT, cpu_init_fn>);
            step_fn = detail::fnAddress(
                &JAXInterface::cpuEntryFn<SimT, cpu_step_fn>);
            if constexpr (!detail::isNull<cpu_save_ckpts_fn> &&
                          !detail::isNull<cpu_restore_ckpts_fn>) {
                save_fn = detail::fnAddress(
                    &JAXInterface::cpuEntryFn<SimT, cpu_save_ckpts_fn>);
                restore_fn = detail::fnAddress(
                    &JAXInterface::cpuEntryFn<SimT, cpu_restore_ckpts_fn>);
            }
        }
        if (init_fn == nullptr || step_fn == nullptr) {
            throw pybind11::value_error(
                "this simulator has no JAX entry points for the requested backend");
        }

        SimT *sim_ptr = sim.cast<SimT *>();
        TrainInterface iface = std::invoke(iface_fn, *sim_ptr);
        if (!register_with_jax) {
            pybind11::dict d = capsules((void *)sim_ptr, init_fn, step_fn,
                                        save_fn, restore_fn, xla_gpu);
            d["step_inputs_iface"] = inputsToPytree(iface);
            d["step_outputs_iface"] = outputsToPytree(iface);
            return d;
        }
        return setup(iface, sim, (void *)sim_ptr, init_fn, step_fn, save_fn,
                     restore_fn, xla_gpu);
    };
}

}
