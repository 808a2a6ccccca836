// madrona::py::Tensor -- what a simulator's Manager hands to Python: a raw
// pointer (device or host) plus element type and shape.
// API contract: reference include/madrona/py/utils.hpp:58-141
// (TensorElementType, TensorInterface, Tensor(void *dev_ptr, type,
// Span<const int64_t> dims, Optional<int> gpu_id) and its accessors) and
// src/python/utils.cpp (numBytesPerItem, interface).  Header-only here; the
// pybind11 module that turns a Tensor into a PyTorch-ROCm tensor without
// copying is madrona_amd/csrc/py_bindings.cpp (DLPack, device kDLROCM), the
// counterpart of the reference's nanobind `tensor_to_pytorch`
// (src/python/bindings.cpp:52-68).
#pragma once

#include <madrona/macros.hpp>
#include <madrona/optional.hpp>
#include <madrona/span.hpp>

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace madrona::py {

enum class TensorElementType {
    UInt8,
    Int8,
    Int16,
    Int32,
    Int64,
    Float16,
    Float32,
};

struct TensorInterface {
    TensorElementType type;
    Span<const int64_t> dimensions;
};

class Tensor final {
public:
    static inline constexpr int64_t maxDimensions = 16;

    // gpu_id: Optional<int>::none() for host memory
    Tensor(void *dev_ptr, TensorElementType type,
           Span<const int64_t> dimensions, Optional<int> gpu_id)
        : dev_ptr_(dev_ptr),
          type_(type),
          gpu_id_(gpu_id.has_value() ? *gpu_id : -1),
          num_dimensions_((int64_t)dimensions.size()),
          dimensions_ {}
    {
        if (num_dimensions_ > maxDimensions) {
            num_dimensions_ = maxDimensions;
        }
        for (int64_t i = 0; i < num_dimensions_; i++) {
            dimensions_[(size_t)i] = dimensions[(CountT)i];
        }
    }

    // (not in the reference: an empty tensor, so that Optional<> of a struct
    // that holds one has something to default to)
    Tensor()
        : dev_ptr_(nullptr), type_(TensorElementType::UInt8), gpu_id_(-1),
          num_dimensions_(0), dimensions_ {}
    {}

    Tensor(const Tensor &o) = default;
    Tensor &operator=(const Tensor &o) = default;

    inline void *devicePtr() const { return dev_ptr_; }
    inline TensorElementType type() const { return type_; }
    inline bool isOnGPU() const { return gpu_id_ != -1; }
    inline int gpuID() const { return gpu_id_; }
    inline int64_t numDims() const { return num_dimensions_; }
    inline const int64_t *dims() const { return dimensions_.data(); }

    inline int64_t numBytesPerItem() const
    {
        switch (type_) {
        case TensorElementType::UInt8:
        case TensorElementType::Int8:
            return 1;
        case TensorElementType::Int16:
        case TensorElementType::Float16:
            return 2;
        case TensorElementType::Int32:
        case TensorElementType::Float32:
            return 4;
        case TensorElementType::Int64:
            return 8;
        }
        return 0;
    }

    inline int64_t numItems() const
    {
        int64_t n = 1;
        for (int64_t i = 0; i < num_dimensions_; i++) {
            n *= dimensions_[(size_t)i];
        }
        return n;
    }

    inline TensorInterface interface() const
    {
        return TensorInterface {
            type_,
            Span<const int64_t>(dimensions_.data(), (CountT)num_dimensions_),
        };
    }

private:
    void *dev_ptr_;
    TensorElementType type_;
    int gpu_id_;

    int64_t num_dimensions_;
    std::array<int64_t, maxDimensions> dimensions_;
};

struct NamedTensor {
    const char *name;
    Tensor tensor;
};

// What a training loop exchanges with a simulator per step, by name.
// API contract: reference include/madrona/py/utils.hpp:148-204
// (TrainStepInputInterface, TrainStepOutputInterface,
// TrainCheckpointingInterface, TrainInterface) and src/python/utils.cpp:403-560
// (the copy routines: actions..., resets, simCtrl, pbt... in; observations...,
// rewards, dones, stats..., pbt... out -- the order the XLA custom call's
// buffers arrive in).
struct TrainStepInputInterface {
    Span<const NamedTensor> actions;
    Tensor resets;
    Tensor simCtrl;
    Span<const NamedTensor> pbt = {};
};

struct TrainStepOutputInterface {
    Span<const NamedTensor> observations;
    Tensor rewards;
    Tensor dones;
    Span<const NamedTensor> stats = {};
    Span<const NamedTensor> pbt = {};
};

struct TrainCheckpointingInterface {
    Tensor checkpointData;
};

// Owns copies of the spans (and of the names) it was given: a Manager builds it
// from temporaries.
class TrainInterface final {
public:
    inline TrainInterface() = default;
    inline TrainInterface(TrainStepInputInterface step_inputs,
                          TrainStepOutputInterface step_outputs,
                          Optional<TrainCheckpointingInterface> checkpointing =
                              Optional<TrainCheckpointingInterface>::none());
    TrainInterface(TrainInterface &&o) = default;
    TrainInterface &operator=(TrainInterface &&o) = default;
    TrainInterface(const TrainInterface &) = delete;

    inline TrainStepInputInterface stepInputs() const;
    inline TrainStepOutputInterface stepOutputs() const;
    inline Optional<TrainCheckpointingInterface> checkpointing() const;

    // host buffers <-> the simulator's tensors (XLA's CPU custom call)
    inline void cpuCopyStepInputs(void **buffers);
    inline void cpuCopyObservations(void **buffers);
    inline void cpuCopyStepOutputs(void **buffers);

    // device buffers <-> the simulator's tensors on `strm` (XLA's GPU custom
    // call hands over its stream): the reference's cudaCopy* (utils.cpp:498-560)
    // on HIP.  `strm` is a hipStream_t; declared void * so that this header
    // does not need the HIP runtime -- defined in <madrona/py/hip_copy.hpp>.
    inline void **hipCopyStepInputs(void *strm, void **buffers);
    inline void hipCopyObservations(void *strm, void **buffers);
    inline void hipCopyStepOutputs(void *strm, void **buffers);

    static inline uint64_t numTensorBytes(const Tensor &t)
    {
        return (uint64_t)t.numItems() * (uint64_t)t.numBytesPerItem();
    }

private:
    struct Impl;
    struct ImplDeleter { inline void operator()(Impl *p) const; };
    std::unique_ptr<Impl, ImplDeleter> impl_;

    template <typename Fn> inline void forEachInput(Fn &&fn);
    template <typename Fn> inline void forEachOutput(Fn &&fn, bool obs_only);
};

}

#include "utils.inl"
