// Microbenchmark: what does one agent-scope atomic per workgroup on ONE address
// cost on MI355X?  (Decides how tickets / arrival counters may be used in the
// sort and ParallelFor kernels.)  hipcc --offload-arch=gfx950 -O3 -o atomic_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_none(unsigned *ctr, unsigned *out, int work)
{
    unsigned acc = 0;
    for (int i = 0; i < work; i++) acc += out[(blockIdx.x * 256 + threadIdx.x + i * 65536) & 0xFFFFF];
    if (acc == 0xdeadbeef) out[0] = acc;
}

template <int MODE>
__global__ void k_atomic(unsigned *ctr, unsigned *out, int work)
{
    __shared__ unsigned t;
    unsigned acc = 0;
    if (MODE == 0 || MODE == 3) {           // returning atomic at the start
        if (threadIdx.x == 0) {
            unsigned *p = MODE == 3 ? ctr + (blockIdx.x % 64) * 64 : ctr;
            t = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        acc = t;
    }
    for (int i = 0; i < work; i++) acc += out[(blockIdx.x * 256 + threadIdx.x + i * 65536) & 0xFFFFF];
    if (MODE == 1) {                        // returning atomic at the end
        __syncthreads();
        if (threadIdx.x == 0) {
            t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        acc += t;
    }
    if (MODE == 2) {                        // non-returning atomic at the end
        if (threadIdx.x == 0) {
            (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (MODE == 4) {                        // workgroup-scope... no: system scope
        if (threadIdx.x == 0) {
            t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        acc += t;
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}

template <typename F>
static float timeIt(F &&launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / 20;
}

int main()
{
    unsigned *ctr, *out;
    hipMalloc(&ctr, 1 << 20); hipMemset(ctr, 0, 1 << 20);
    hipMalloc(&out, 4 << 20); hipMemset(out, 0, 4 << 20);
    const char *names[] = {"ret@start", "ret@end", "noret@end", "ret@start,64 addrs", "ret@end,system"};
    for (int work : {0, 16}) {
        for (int blocks : {128, 1024, 4096, 16384}) {
            float base = timeIt([&] { hipLaunchKernelGGL(k_none, dim3(blocks), dim3(256), 0, 0, ctr, out, work); });
            float t0 = timeIt([&] { hipLaunchKernelGGL(k_atomic<0>, dim3(blocks), dim3(256), 0, 0, ctr, out, work); });
            float t1 = timeIt([&] { hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(256), 0, 0, ctr, out, work); });
            float t2 = timeIt([&] { hipLaunchKernelGGL(k_atomic<2>, dim3(blocks), dim3(256), 0, 0, ctr, out, work); });
            float t3 = timeIt([&] { hipLaunchKernelGGL(k_atomic<3>, dim3(blocks), dim3(256), 0, 0, ctr, out, work); });
            float t4 = timeIt([&] { hipLaunchKernelGGL(k_atomic<4>, dim3(blocks), dim3(256), 0, 0, ctr, out, work); });
            printf("work %2d blocks %5d  none %7.2f us | %s %7.2f | %s %7.2f | %s %7.2f | %s %7.2f | %s %7.2f\n",
                   work, blocks, base, names[0], t0, names[1], t1, names[2], t2, names[3], t3, names[4], t4);
        }
    }
    return 0;
}
