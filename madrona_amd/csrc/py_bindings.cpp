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

#include <madrona/py/bindings.hpp>
#include <madrona/py/hip_copy.hpp>
#include <madrona/py/utils.hpp>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace pyb = pybind11;
using madrona::py::JAXInterface;
using madrona::py::NamedTensor;
using madrona::py::Tensor;
using madrona::py::TensorElementType;
using madrona::py::TrainInterface;

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


// ---- a Manager-shaped class with the JAX entry points of a real simulator ----
// (reference madrona_escape_room src/mgr.cpp: trainInterface(), cpuJAXInit /
// cpuJAXStep(void **inputs, void **outputs), gpuJAXInit / gpuJAXStep(stream,
// void **buffers)).  N "worlds", state in host memory (gpu_id < 0) or in device
// memory; step(): obs[w] = { action[w][0] + action[w][1], resets[w], step count },
// reward[w] = obs[w][0] / 2, done[w] = resets[w].  Used by tests/test_py_bridge.py
// to drive TrainInterface and the XLA entry points without a simulator build.
// with_extras: one more tensor of every optional kind -- a pbt input "policy"
// (int32 [N]), a stats output "episodes" (int32 [N] = 10 * step + w) and a pbt
// output "fitness" (float32 [N] = policy[w] - w) -- so that the buffer order
// actions..., resets, simCtrl, pbt... / observations..., rewards, dones,
// stats..., pbt... is exercised end to end.
class DemoTrainSim {
public:
    DemoTrainSim(int64_t num_worlds, int gpu_id, bool with_extras = false)
        : n_(num_worlds), gpu_(gpu_id), extras_(with_extras)
    {
        policy_ = alloc(n_ * 4);
        episodes_ = alloc(n_ * 4);
        fitness_ = alloc(n_ * 4);
        action_ = alloc(n_ * 2 * 4);
        resets_ = alloc(n_ * 4);
        ctrl_ = alloc(4);
        obs_ = alloc(n_ * 3 * 4);
        rewards_ = alloc(n_ * 4);
        dones_ = alloc(n_ * 4);
    }
    ~DemoTrainSim()
    {
        for (void *p : owned_) {
            if (gpu_ >= 0) (void)hipFree(p); else free(p);
        }
    }
    DemoTrainSim(const DemoTrainSim &) = delete;

    Tensor tensor(void *p, TensorElementType t, std::vector<int64_t> dims) const
    {
        return Tensor(p, t, madrona::Span<const int64_t>(dims.data(),
                                                         (madrona::CountT)dims.size()),
                      gpu_ >= 0 ? madrona::Optional<int>::make(gpu_) :
                                  madrona::Optional<int>::none());
    }
    Tensor actionTensor() const { return tensor(action_, TensorElementType::Int32, { n_, 2 }); }
    Tensor obsTensor() const { return tensor(obs_, TensorElementType::Float32, { n_, 3 }); }

    TrainInterface trainInterface() const
    {
        NamedTensor actions[] = { { "move", actionTensor() } };
        NamedTensor obs[] = { { "self", obsTensor() } };
        if (extras_) {
            NamedTensor pbt_in[] = {
                { "policy", tensor(policy_, TensorElementType::Int32, { n_ }) } };
            NamedTensor stats[] = {
                { "episodes", tensor(episodes_, TensorElementType::Int32, { n_ }) } };
            NamedTensor pbt_out[] = {
                { "fitness", tensor(fitness_, TensorElementType::Float32, { n_ }) } };
            return TrainInterface(
                { madrona::Span<const NamedTensor>(actions, 1),
                  tensor(resets_, TensorElementType::Int32, { n_ }),
                  tensor(ctrl_, TensorElementType::Int32, { 1 }),
                  madrona::Span<const NamedTensor>(pbt_in, 1) },
                { madrona::Span<const NamedTensor>(obs, 1),
                  tensor(rewards_, TensorElementType::Float32, { n_ }),
                  tensor(dones_, TensorElementType::Int32, { n_ }),
                  madrona::Span<const NamedTensor>(stats, 1),
                  madrona::Span<const NamedTensor>(pbt_out, 1) });
        }
        return TrainInterface(
            { madrona::Span<const NamedTensor>(actions, 1),
              tensor(resets_, TensorElementType::Int32, { n_ }),
              tensor(ctrl_, TensorElementType::Int32, { 1 }) },
            { madrona::Span<const NamedTensor>(obs, 1),
              tensor(rewards_, TensorElementType::Float32, { n_ }),
              tensor(dones_, TensorElementType::Int32, { n_ }) });
    }

    void step()
    {
        steps_++;
        std::vector<int32_t> a((size_t)n_ * 2), r((size_t)n_), d((size_t)n_);
        std::vector<float> o((size_t)n_ * 3), rew((size_t)n_);
        get(a.data(), action_, a.size() * 4);
        get(r.data(), resets_, r.size() * 4);
        for (int64_t w = 0; w < n_; w++) {
            o[w * 3] = (float)(a[w * 2] + a[w * 2 + 1]);
            o[w * 3 + 1] = (float)r[w];
            o[w * 3 + 2] = (float)steps_;
            rew[w] = o[w * 3] / 2.f;
            d[w] = r[w];
        }
        put(obs_, o.data(), o.size() * 4);
        put(rewards_, rew.data(), rew.size() * 4);
        put(dones_, d.data(), d.size() * 4);
        if (extras_) {
            std::vector<int32_t> pol((size_t)n_), ep((size_t)n_);
            std::vector<float> fit((size_t)n_);
            get(pol.data(), policy_, pol.size() * 4);
            for (int64_t w = 0; w < n_; w++) {
                ep[w] = (int32_t)(10 * steps_ + w);
                fit[w] = (float)(pol[w] - (int32_t)w);
            }
            put(episodes_, ep.data(), ep.size() * 4);
            put(fitness_, fit.data(), fit.size() * 4);
        }
    }

    void cpuJAXInit(void **, void **outputs)
    {
        trainInterface().cpuCopyObservations(outputs);
    }
    void cpuJAXStep(void **inputs, void **outputs)
    {
        TrainInterface iface = trainInterface();
        iface.cpuCopyStepInputs(inputs);
        step();
        iface.cpuCopyStepOutputs(outputs);
    }
    void gpuJAXInit(hipStream_t strm, void **buffers)
    {
        trainInterface().hipCopyObservations(strm, buffers);
    }
    void gpuJAXStep(hipStream_t strm, void **buffers)
    {
        TrainInterface iface = trainInterface();
        void **outputs = iface.hipCopyStepInputs(strm, buffers);
        // (a real Manager replays its step graph on `strm` here; the demo's
        // arithmetic runs on the host: wait for the inputs first)
        (void)hipStreamSynchronize(strm);
        step();
        iface.hipCopyStepOutputs(strm, outputs);
    }
    int64_t steps() const { return steps_; }

private:
    void *alloc(int64_t bytes)
    {
        void *p = nullptr;
        if (gpu_ >= 0) {
            madrona::py::detail::reqHip(hipSetDevice(gpu_), "hipSetDevice");
            madrona::py::detail::reqHip(hipMalloc(&p, (size_t)bytes), "hipMalloc");
            (void)hipMemset(p, 0, (size_t)bytes);
        } else {
            p = calloc(1, (size_t)bytes);
        }
        owned_.push_back(p);
        return p;
    }
    void get(void *dst, const void *src, size_t n) const
    {
        if (gpu_ >= 0) (void)hipMemcpy(dst, src, n, hipMemcpyDeviceToHost);
        else memcpy(dst, src, n);
    }
    void put(void *dst, const void *src, size_t n) const
    {
        if (gpu_ >= 0) (void)hipMemcpy(dst, src, n, hipMemcpyHostToDevice);
        else memcpy(dst, src, n);
    }

    int64_t n_;
    int gpu_;
    bool extras_;
    int64_t steps_ = 0;
    void *action_, *resets_, *ctrl_, *obs_, *rewards_, *dones_;
    void *policy_, *episodes_, *fitness_;
    std::vector<void *> owned_;
};

std::vector<void *> pointerList(const std::vector<uintptr_t> &v)
{
    std::vector<void *> out;
    for (uintptr_t p : v) out.push_back((void *)p);
    return out;
}

void *capsulePointer(const pyb::capsule &c)
{
    return PyCapsule_GetPointer(c.ptr(), "xla._CUSTOM_CALL_TARGET");
}

TensorElementType fromDLType(uint8_t code, uint8_t bits)
{
    if (code == kDLInt) {
        switch (bits) {
        case 8: return TensorElementType::Int8;
        case 16: return TensorElementType::Int16;
        case 32: return TensorElementType::Int32;
        case 64: return TensorElementType::Int64;
        }
    } else if (code == kDLUInt && bits == 8) {
        return TensorElementType::UInt8;
    } else if (code == kDLFloat) {
        if (bits == 16) return TensorElementType::Float16;
        if (bits == 32) return TensorElementType::Float32;
    }
    throw pyb::type_error("madrona::py::Tensor: unsupported dtype");
}

// Tensor(torch_tensor): reference bindings.cpp:313-338 (nanobind ndarray);
// here through the tensor's own __dlpack__ (device kDLCPU, kDLROCM or kDLCUDA
// -- PyTorch-ROCm reports its devices as either)
Tensor tensorFromDLPack(pyb::object src)
{
    pyb::capsule cap = src.attr("__dlpack__")();
    auto *m = (DLManagedTensor *)PyCapsule_GetPointer(cap.ptr(), "dltensor");
    if (m == nullptr) {
        throw pyb::type_error("madrona::py::Tensor: not a DLPack tensor");
    }
    const DLTensor &dl = m->dl_tensor;
    if (dl.strides != nullptr) {
        int64_t expect = 1;
        for (int32_t i = dl.ndim - 1; i >= 0; i--) {
            if (dl.shape[i] != 1 && dl.strides[i] != expect) {
                throw pyb::value_error("madrona::py::Tensor: tensor is not contiguous");
            }
            expect *= dl.shape[i];
        }
    }
    if (dl.ndim > Tensor::maxDimensions) {
        throw pyb::value_error("madrona::py::Tensor: too many dimensions");
    }
    madrona::Optional<int> gpu = madrona::Optional<int>::none();
    if (dl.device.device_type == kDLROCM || dl.device.device_type == 2 /* kDLCUDA */) {
        gpu = madrona::Optional<int>::make(dl.device.device_id);
    } else if (dl.device.device_type != kDLCPU) {
        throw pyb::type_error("madrona::py::Tensor: unknown device type");
    }
    // (the view does not own: the caller keeps `src` alive, as in the reference;
    // the capsule's deleter runs when `cap` dies, the storage stays with `src`)
    return Tensor((char *)dl.data + dl.byte_offset,
                  fromDLType(dl.dtype.code, dl.dtype.bits),
                  madrona::Span<const int64_t>(dl.shape, (madrona::CountT)dl.ndim),
                  gpu);
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
        // Tensor(torch_tensor): a view of the tensor's storage
        .def(pyb::init(&tensorFromDLPack), pyb::arg("tensor"))
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

    pyb::class_<TrainInterface>(m, "TrainInterface")
        // reference bindings.cpp:365-374 (pytrees of shapes and dtypes; numpy
        // dtype names here, jax.ShapeDtypeStruct is built in jax_register.py)
        .def("step_inputs", [](const TrainInterface &iface) {
            return JAXInterface::inputsToPytree(iface);
        })
        .def("step_outputs", [](const TrainInterface &iface) {
            return JAXInterface::outputsToPytree(iface);
        });

    pyb::class_<DemoTrainSim>(m, "_DemoTrainSim")
        .def(pyb::init<int64_t, int, bool>(), pyb::arg("num_worlds"),
             pyb::arg("gpu_id") = -1, pyb::arg("with_extras") = false)
        .def("train_interface", &DemoTrainSim::trainInterface)
        .def("action_tensor", &DemoTrainSim::actionTensor)
        .def("obs_tensor", &DemoTrainSim::obsTensor)
        .def("step", &DemoTrainSim::step)
        .def_property_readonly("steps", &DemoTrainSim::steps)
        .def("jax", JAXInterface::buildEntry<
                &DemoTrainSim::trainInterface,
                &DemoTrainSim::cpuJAXInit, &DemoTrainSim::cpuJAXStep,
                &DemoTrainSim::gpuJAXInit, &DemoTrainSim::gpuJAXStep>(),
             pyb::arg("xla_gpu"), pyb::arg("register") = true);

    // what XLA does with a registered target, for tests (no jax here):
    // CPU custom call: fn(void **out, void **in)
    m.def("_call_cpu_custom_call", [](pyb::capsule target,
                                      std::vector<uintptr_t> outs,
                                      std::vector<uintptr_t> ins) {
        auto fn = (void (*)(void **, void **))capsulePointer(target);
        std::vector<void *> o = pointerList(outs), i = pointerList(ins);
        fn(o.data(), i.data());
    });
    // GPU custom call: fn(stream, void **buffers, opaque, opaque_len)
    m.def("_call_gpu_custom_call", [](pyb::capsule target, uintptr_t stream,
                                      std::vector<uintptr_t> buffers,
                                      pyb::bytes opaque) {
        auto fn = (void (*)(hipStream_t, void **, const char *, size_t))
            capsulePointer(target);
        std::vector<void *> b = pointerList(buffers);
        std::string o = opaque;
        fn((hipStream_t)stream, b.data(), o.data(), o.size());
    });
}
