// pybind11 module `_madrona_amd_py`: madrona::py::Tensor -> torch.Tensor without
// a copy.  Counterpart of the reference's nanobind bindings
// (src/python/bindings.cpp:52-68 `tensor_to_pytorch`, :177-199 the Tensor
// class); nanobind is not available here, pybind11 is.
//
// The hand-over is a DLPack capsule built by hand: a NON-OWNING DLManagedTensor
// (the executor owns exported columns for its whole lifetime, reference
// mw_gpu.hpp:159-163) whose device is kDLROCM for device memory -- PyTorch-ROCm
// maps that to its "cuda" device -- and kDLCPU for host memory.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <madrona/py/utils.hpp>

#include <cstdint>
#include <string>
#include <vector>

namespace pyb = pybind11;
using madrona::py::Tensor;
using madrona::py::TensorElementType;

namespace {

// ---- DLPack v0.8 ABI (dlpack.h; stable C structs) ----
enum : int32_t { kDLCPU = 1, kDLROCM = 10 };
enum : uint8_t { kDLInt = 0, kDLUInt = 1, kDLFloat = 2 };

struct DLDevice { int32_t device_type; int32_t device_id; };
struct DLDataType { uint8_t code; uint8_t bits; uint16_t lanes; };
struct DLTensor {
    void *data;
    DLDevice device;
    int32_t ndim;
    DLDataType dtype;
    int64_t *shape;
    int64_t *strides;
    uint64_t byte_offset;
};
struct DLManagedTensor {
    DLTensor dl_tensor;
    void *manager_ctx;
    void (*deleter)(DLManagedTensor *self);
};

struct Managed {
    DLManagedTensor managed;
    int64_t shape[Tensor::maxDimensions];
};

DLDataType dlType(TensorElementType t)
{
    switch (t) {
    case TensorElementType::UInt8: return { kDLUInt, 8, 1 };
    case TensorElementType::Int8: return { kDLInt, 8, 1 };
    case TensorElementType::Int16: return { kDLInt, 16, 1 };
    case TensorElementType::Int32: return { kDLInt, 32, 1 };
    case TensorElementType::Int64: return { kDLInt, 64, 1 };
    case TensorElementType::Float16: return { kDLFloat, 16, 1 };
    case TensorElementType::Float32: return { kDLFloat, 32, 1 };
    }
    return { kDLUInt, 8, 1 };
}

void capsuleDestructor(PyObject *capsule)
{
    // still named "dltensor": nobody consumed it, so its deleter is ours to run
    if (PyCapsule_IsValid(capsule, "dltensor")) {
        auto *m = (DLManagedTensor *)PyCapsule_GetPointer(capsule, "dltensor");
        if (m != nullptr && m->deleter != nullptr) {
            m->deleter(m);
        }
    }
}

pyb::capsule toDLPack(const Tensor &t)
{
    auto *m = new Managed {};
    for (int64_t i = 0; i < t.numDims(); i++) {
        m->shape[i] = t.dims()[i];
    }
    DLTensor &dl = m->managed.dl_tensor;
    dl.data = t.devicePtr();
    dl.device = t.isOnGPU() ? DLDevice { kDLROCM, t.gpuID() } :
                              DLDevice { kDLCPU, 0 };
    dl.ndim = (int32_t)t.numDims();
    dl.dtype = dlType(t.type());
    dl.shape = m->shape;
    dl.strides = nullptr;       // compact row-major
    dl.byte_offset = 0;
    m->managed.manager_ctx = m;
    m->managed.deleter = [](DLManagedTensor *self) {
        delete (Managed *)self->manager_ctx;    // the data is not ours to free
    };
    return pyb::capsule(PyCapsule_New(&m->managed, "dltensor",
                                      &capsuleDestructor), false);
}

}

PYBIND11_MODULE(_madrona_amd_py, m)
{
    m.doc() = "madrona::py::Tensor for the MI355X backend (zero-copy DLPack "
              "export to PyTorch-ROCm)";

    pyb::enum_<TensorElementType>(m, "TensorElementType")
        .value("UInt8", TensorElementType::UInt8)
        .value("Int8", TensorElementType::Int8)
        .value("Int16", TensorElementType::Int16)
        .value("Int32", TensorElementType::Int32)
        .value("Int64", TensorElementType::Int64)
        .value("Float16", TensorElementType::Float16)
        .value("Float32", TensorElementType::Float32);

    pyb::class_<Tensor>(m, "Tensor")
        // (address, type, dims, gpu_id or None): what a Manager's exported
        // tensor accessor constructs on the C++ side
        .def(pyb::init([](uintptr_t ptr, TensorElementType type,
                          std::vector<int64_t> dims, pyb::object gpu_id) {
            madrona::Optional<int> gpu = madrona::Optional<int>::none();
            if (!gpu_id.is_none()) {
                gpu = madrona::Optional<int>::make(gpu_id.cast<int>());
            }
            return Tensor((void *)ptr, type,
                madrona::Span<const int64_t>(dims.data(),
                                             (madrona::CountT)dims.size()),
                gpu);
        }), pyb::arg("ptr"), pyb::arg("type"), pyb::arg("dims"),
            pyb::arg("gpu_id") = pyb::none())
        .def_property_readonly("device_ptr", [](const Tensor &t) {
            return (uintptr_t)t.devicePtr();
        })
        .def_property_readonly("type", &Tensor::type)
        .def_property_readonly("is_on_gpu", &Tensor::isOnGPU)
        .def_property_readonly("gpu_id", &Tensor::gpuID)
        .def_property_readonly("shape", [](const Tensor &t) {
            return std::vector<int64_t>(t.dims(), t.dims() + t.numDims());
        })
        .def_property_readonly("bytes_per_item", &Tensor::numBytesPerItem)
        .def("__dlpack__", [](const Tensor &t, pyb::object) {
            return toDLPack(t);
        }, pyb::arg("stream") = pyb::none())
        .def("__dlpack_device__", [](const Tensor &t) {
            return pyb::make_tuple(t.isOnGPU() ? (int)kDLROCM : (int)kDLCPU,
                                   t.isOnGPU() ? t.gpuID() : 0);
        })
        // == the reference's Tensor.to_torch()
        .def("to_torch", [](const Tensor &t) {
            pyb::object from_dlpack =
                pyb::module_::import("torch.utils.dlpack").attr("from_dlpack");
            return from_dlpack(toDLPack(t));
        });
}
