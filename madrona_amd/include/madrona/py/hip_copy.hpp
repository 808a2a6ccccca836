// The HIP half of TrainInterface (reference src/python/utils.cpp:498-560, the
// cudaCopy* routines): asynchronous copies between XLA's device buffers and the
// simulator's exported tensors on the stream the custom call was handed, and
// the synchronous host <-> device copies of the CPU custom call when a tensor
// lives on the GPU.  Include from ONE translation unit that links the HIP
// runtime (a Manager's bindings, csrc/py_bindings.cpp).
#pragma once

#include <madrona/py/utils.hpp>

#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>

namespace madrona::py {

namespace detail {

inline void reqHip(hipError_t err, const char *what)
{
    if (err != hipSuccess) {
        fprintf(stderr, "madrona::py: %s failed: %s\n", what, hipGetErrorString(err));
        abort();
    }
}

// (installed when the library that includes this header is loaded)
inline const bool gpuHostCopyInstalled = [] {
    gpuHostCopyHook() = [](void *dst, const void *src, uint64_t n, bool to_device) {
        reqHip(hipMemcpy(dst, src, n, to_device ? hipMemcpyHostToDevice :
                                                  hipMemcpyDeviceToHost),
               "hipMemcpy (host <-> tensor)");
    };
    return true;
}();

}

void **TrainInterface::hipCopyStepInputs(void *strm, void **buffers)
{
    forEachInput([&](const Tensor &t) {
        detail::reqHip(hipMemcpyAsync(t.devicePtr(), *buffers++, numTensorBytes(t),
            t.isOnGPU() ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
            (hipStream_t)strm), "hipMemcpyAsync (step input)");
    });
    return buffers;
}

void TrainInterface::hipCopyObservations(void *strm, void **buffers)
{
    forEachOutput([&](const Tensor &t) {
        detail::reqHip(hipMemcpyAsync(*buffers++, t.devicePtr(), numTensorBytes(t),
            t.isOnGPU() ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
            (hipStream_t)strm), "hipMemcpyAsync (observation)");
    }, true);
}

void TrainInterface::hipCopyStepOutputs(void *strm, void **buffers)
{
    forEachOutput([&](const Tensor &t) {
        detail::reqHip(hipMemcpyAsync(*buffers++, t.devicePtr(), numTensorBytes(t),
            t.isOnGPU() ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
            (hipStream_t)strm), "hipMemcpyAsync (step output)");
    }, false);
}

}
