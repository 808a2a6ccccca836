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

}
