// Part of the fused rigid-body step (phys_impl/world_step.inl includes the parts
// in order, inside namespace madrona::phys::kernels): physicsOrderKernel: heaviest worlds first + the frame + the fallback list.

// Heaviest worlds first (longest-processing-time order): a counting sort of the
// worlds by what they cost last step, 256 buckets, one 1024-thread workgroup.
// The order only decides WHEN a world is stepped, never the result.
__global__ void __launch_bounds__(1024)
physicsOrderKernel(EcsState *S, void *node_data, uint32_t, uint32_t)
{
    mwhip::TraceScope trace_scope(S);
    const PhysicsStepParams params = *(const PhysicsStepParams *)node_data;
    const int32_t num_worlds = S->numWorlds;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        params.fallbackList[0] = 0;     // (worlds the LDS step hands to the HBM one)
    }
    if (tid >= 960u) {
        // (the last wavefront: its share of the cost scan starts a moment later)
        fillPhysicsFrame(S, detail::scratch(S),
                         &((PhysicsStepNode *)node_data)->frame, tid - 960u, 64u);
    }

    __shared__ uint32_t hist[256];
    __shared__ uint32_t wave_max[16];
    __shared__ uint32_t most_shared;

    // (up to 8192 worlds: a thread's costs stay in its registers -- the two
    // later passes would each wait for the same loads again)
    constexpr int32_t kept_costs = 8;
    const bool costs_kept = num_worlds <= kept_costs * 1024;
    uint32_t kept[kept_costs];
    uint32_t most = 0;
    if (costs_kept) {
#pragma unroll
        for (int32_t i = 0; i < kept_costs; i++) {
            const int32_t w = i * 1024 + (int32_t)tid;
            kept[i] = w < num_worlds ? params.worldCost[w] : 0u;
        }
#pragma unroll
        for (int32_t i = 0; i < kept_costs; i++) {
            most = kept[i] > most ? kept[i] : most;
        }
    } else {
        for (int32_t w = (int32_t)tid; w < num_worlds; w += 1024) {
            const uint32_t c = params.worldCost[w];
            most = c > most ? c : most;
        }
    }
    most = wave::maxReduce<64>(most);
    if (tid < 256) hist[tid] = 0;
    if (tid % 64 == 0) wave_max[tid / 64] = most;
    __syncthreads();
    if (tid == 0) {
        uint32_t m = 1;
        for (int i = 0; i < 16; i++) m = wave_max[i] > m ? wave_max[i] : m;
        most_shared = m;
    }
    __syncthreads();
    most = most_shared;

    // 256 buckets, heaviest first: any function of the cost that never
    // decreases will do (the order only decides when a world is stepped), so
    // no 64-bit division per world -- costs shifted down to 24 bits, one
    // multiplication by 255 / most
    const uint32_t cost_shift = most >= (1u << 24) ?
        8u - (uint32_t)__builtin_clz(most) : 0u;
    const float to_bucket = 255.f / (float)(most >> cost_shift);
    auto bucket_of = [cost_shift, to_bucket](uint32_t cost) {
        const uint32_t b = (uint32_t)((float)(cost >> cost_shift) * to_bucket);
        return 255u - (b < 255u ? b : 255u);
    };
    if (costs_kept) {
#pragma unroll
        for (int32_t i = 0; i < kept_costs; i++) {
            kept[i] = bucket_of(kept[i]);       // (from here on: its bucket)
            if (i * 1024 + (int32_t)tid < num_worlds) {
                atomicAdd(&hist[kept[i]], 1u);
            }
        }
    } else {
        for (int32_t w = (int32_t)tid; w < num_worlds; w += 1024) {
            atomicAdd(&hist[bucket_of(params.worldCost[w])], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the 256 buckets by the first four wavefronts (one thread
    // walking them: 256 dependent LDS round trips; 10.9 -> 9.0 us)
    {
        const uint32_t lane = tid & 63u;
        const uint32_t v = tid < 256u ? hist[tid] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (tid < 256u && lane == 63u) wave_max[tid >> 6] = incl;
        __syncthreads();
        if (tid < 256u) {
            uint32_t base = 0;
            for (uint32_t wv = 0; wv < (tid >> 6); wv++) base += wave_max[wv];
            hist[tid] = base + incl - v;
        }
    }
    __syncthreads();
    if (costs_kept) {
#pragma unroll
        for (int32_t i = 0; i < kept_costs; i++) {
            const int32_t w = i * 1024 + (int32_t)tid;
            if (w < num_worlds) {
                const uint32_t at = atomicAdd(&hist[kept[i]], 1u);
                params.worldOrder[at] = w;
            }
        }
    } else {
        for (int32_t w = (int32_t)tid; w < num_worlds; w += 1024) {
            const uint32_t at = atomicAdd(&hist[bucket_of(params.worldCost[w])], 1u);
            params.worldOrder[at] = w;
        }
    }
}

