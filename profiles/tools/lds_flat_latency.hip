// Dependent-load latency on gfx950: ds_read_b32 against flat_load_dword that
// lands in LDS against global_load_dword that hits L2 (one wavefront, a
// pointer chase of 2048 steps; cycles = s_memtime).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_flat profiles/tools/lds_flat_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) chase(const uint32_t *perm, uint32_t *out,
                                            unsigned long long *cycles, int waves_busy)
{
    __shared__ uint32_t lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = perm[i];
    __syncthreads();
    uint32_t at = threadIdx.x;

    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 2048; i++) at = lds[at];
    unsigned long long t1 = __builtin_readcyclecounter();

    const uint32_t *generic = lds;
    asm volatile("" : "+v"(generic));     // (the compiler no longer knows it is LDS)
    uint32_t at2 = at;
    for (int i = 0; i < 2048; i++) at2 = generic[at2];
    unsigned long long t2 = __builtin_readcyclecounter();

    uint32_t at3 = at2;
    for (int i = 0; i < 2048; i++) at3 = __builtin_nontemporal_load(perm + at3);
    unsigned long long t3 = __builtin_readcyclecounter();

    // 12-byte record through each path (a Vector3): one dependent step = three words
    uint32_t at4 = at3;
    for (int i = 0; i < 1024; i++) {
        const uint32_t *r = generic + (at4 % 680u) * 3u;
        at4 = (r[0] ^ r[1] ^ r[2]) & 2047u;
    }
    unsigned long long t4 = __builtin_readcyclecounter();
    uint32_t at5 = at4;
    for (int i = 0; i < 1024; i++) {
        const uint32_t *r = lds + (at5 % 680u) * 3u;
        at5 = (r[0] ^ r[1] ^ r[2]) & 2047u;
    }
    unsigned long long t5 = __builtin_readcyclecounter();

    out[blockIdx.x * 64 + threadIdx.x] = at5;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        cycles[0] = t1 - t0; cycles[1] = t2 - t1; cycles[2] = t3 - t2;
        cycles[3] = t4 - t3; cycles[4] = t5 - t4;
    }
}

int main()
{
    std::vector<uint32_t> perm(2048);
    for (int i = 0; i < 2048; i++) perm[i] = (uint32_t)((i * 1237 + 331) % 2048);
    uint32_t *d_perm, *d_out; unsigned long long *d_cyc;
    hipMalloc(&d_perm, 2048 * 4); hipMalloc(&d_out, 64 * 4096 * 4); hipMalloc(&d_cyc, 64);
    hipMemcpy(d_perm, perm.data(), 2048 * 4, hipMemcpyHostToDevice);
    for (int blocks : { 1, 2048, 4096 }) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d_perm, d_out, d_cyc, 0);
            hipDeviceSynchronize();
        }
        unsigned long long c[8];
        hipMemcpy(c, d_cyc, 64, hipMemcpyDeviceToHost);
        printf("%d workgroups: cycles per dependent load: ds_read %.1f, flat->LDS %.1f, global (L2/MALL) %.1f; "
               "12-byte record: flat %.1f, ds %.1f\n", blocks,
               c[0] / 2048.0, c[1] / 2048.0, c[2] / 2048.0, c[3] / 1024.0, c[4] / 1024.0);
    }
    return 0;
}
