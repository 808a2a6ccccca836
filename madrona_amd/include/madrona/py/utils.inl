// TrainInterface: storage and the copy routines (reference
// src/python/utils.cpp:403-560; header-only here).
#pragma once

namespace madrona::py {

struct TrainInterface::Impl {
    // (std::string keeps the names alive; NamedTensor::name points into them)
    std::vector<std::string> names;
    std::vector<NamedTensor> actions, inPBT, observations, stats, outPBT;
    Tensor resets, simCtrl, rewards, dones;
    Optional<TrainCheckpointingInterface> checkpointing;

    Impl(const TrainStepInputInterface &in, const TrainStepOutputInterface &out,
         Optional<TrainCheckpointingInterface> ckpt)
        : resets(in.resets), simCtrl(in.simCtrl), rewards(out.rewards),
          dones(out.dones), checkpointing(ckpt)
    {
        size_t total = (size_t)(in.actions.size() + in.pbt.size() +
            out.observations.size() + out.stats.size() + out.pbt.size());
        names.reserve(total);       // (no reallocation: c_str() stays put)
        auto copy = [this](Span<const NamedTensor> src,
                           std::vector<NamedTensor> &dst) {
            for (const NamedTensor &t : src) {
                names.emplace_back(t.name);
                dst.push_back(NamedTensor { names.back().c_str(), t.tensor });
            }
        };
        copy(in.actions, actions);
        copy(in.pbt, inPBT);
        copy(out.observations, observations);
        copy(out.stats, stats);
        copy(out.pbt, outPBT);
    }
};

void TrainInterface::ImplDeleter::operator()(Impl *p) const { delete p; }

TrainInterface::TrainInterface(TrainStepInputInterface step_inputs,
                               TrainStepOutputInterface step_outputs,
                               Optional<TrainCheckpointingInterface> checkpointing)
    : impl_(new Impl(step_inputs, step_outputs, checkpointing))
{}

namespace detail {
inline Span<const NamedTensor> spanOf(const std::vector<NamedTensor> &v)
{
    return Span<const NamedTensor>(v.data(), (CountT)v.size());
}
}

TrainStepInputInterface TrainInterface::stepInputs() const
{
    return TrainStepInputInterface {
        detail::spanOf(impl_->actions), impl_->resets, impl_->simCtrl,
        detail::spanOf(impl_->inPBT),
    };
}

TrainStepOutputInterface TrainInterface::stepOutputs() const
{
    return TrainStepOutputInterface {
        detail::spanOf(impl_->observations), impl_->rewards, impl_->dones,
        detail::spanOf(impl_->stats), detail::spanOf(impl_->outPBT),
    };
}

Optional<TrainCheckpointingInterface> TrainInterface::checkpointing() const
{
    return impl_->checkpointing;
}

// buffer order of the step's inputs: actions..., resets, simCtrl, pbt...
template <typename Fn>
void TrainInterface::forEachInput(Fn &&fn)
{
    for (const NamedTensor &t : impl_->actions) fn(t.tensor);
    fn(impl_->resets);
    fn(impl_->simCtrl);
    for (const NamedTensor &t : impl_->inPBT) fn(t.tensor);
}

// ... of its outputs: observations..., rewards, dones, stats..., pbt...
template <typename Fn>
void TrainInterface::forEachOutput(Fn &&fn, bool obs_only)
{
    for (const NamedTensor &t : impl_->observations) fn(t.tensor);
    if (obs_only) return;
    fn(impl_->rewards);
    fn(impl_->dones);
    for (const NamedTensor &t : impl_->stats) fn(t.tensor);
    for (const NamedTensor &t : impl_->outPBT) fn(t.tensor);
}

namespace detail {
// host <-> simulator tensor.  A device-resident tensor goes through the HIP
// runtime, which a host-only translation unit does not have: hip_copy.hpp
// installs the copy routine (the reference asserts without CUDA support,
// utils.cpp:409-416).
using GpuHostCopyFn = void (*)(void *dst, const void *src, uint64_t num_bytes,
                               bool to_device);
inline GpuHostCopyFn &gpuHostCopyHook()
{
    static GpuHostCopyFn fn = nullptr;
    return fn;
}

inline void gpuHostCopy(void *dst, const void *src, uint64_t n, bool to_device)
{
    GpuHostCopyFn fn = gpuHostCopyHook();
    if (fn == nullptr) {
        fprintf(stderr, "madrona::py::TrainInterface: a tensor lives on the GPU "
                "and this library was built without <madrona/py/hip_copy.hpp>\n");
        abort();
    }
    fn(dst, src, n, to_device);
}

inline void cpuToSim(const Tensor &dst, const void *src)
{
    uint64_t n = TrainInterface::numTensorBytes(dst);
    if (dst.isOnGPU()) {
        gpuHostCopy(dst.devicePtr(), src, n, true);
    } else {
        memcpy(dst.devicePtr(), src, n);
    }
}

inline void cpuFromSim(void *dst, const Tensor &src)
{
    uint64_t n = TrainInterface::numTensorBytes(src);
    if (src.isOnGPU()) {
        gpuHostCopy(dst, src.devicePtr(), n, false);
    } else {
        memcpy(dst, src.devicePtr(), n);
    }
}
}

void TrainInterface::cpuCopyStepInputs(void **buffers)
{
    forEachInput([&](const Tensor &t) { detail::cpuToSim(t, *buffers++); });
}

void TrainInterface::cpuCopyObservations(void **buffers)
{
    forEachOutput([&](const Tensor &t) { detail::cpuFromSim(*buffers++, t); }, true);
}

void TrainInterface::cpuCopyStepOutputs(void **buffers)
{
    forEachOutput([&](const Tensor &t) { detail::cpuFromSim(*buffers++, t); }, false);
}

}
