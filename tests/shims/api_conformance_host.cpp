// TEST INFRASTRUCTURE.  Host half of the API-conformance check: the headers a
// real simulator's Manager (mgr.cpp) and Python binding include -- executor,
// physics loader, importer structs, containers, py::Tensor -- compiled as plain
// host C++ and exercised through a few C entry points (tests/test_api_
// conformance.py; nothing here needs a GPU until conf_run is called).
#define MADRONA_TRACING 1
#include <madrona/tracing.hpp>
#include <madrona/mw_gpu.hpp>
#include <madrona/exec_mode.hpp>
#include <madrona/importer.hpp>
#include <madrona/physics_loader.hpp>
#include <madrona/physics_assets.hpp>
#include <madrona/heap_array.hpp>
#include <madrona/dyn_array.hpp>
#include <madrona/inline_array.hpp>
#include <madrona/sync.hpp>
#include <madrona/py/utils.hpp>

#include <mwhip.h>

#include <string>

using namespace madrona;

namespace {

struct Tracked {
    static inline int live = 0;
    int v;
    explicit Tracked(int x) : v(x) { live++; }
    Tracked(const Tracked &o) : v(o.v) { live++; }
    Tracked(Tracked &&o) : v(o.v) { live++; }
    ~Tracked() { live--; }
};

}

extern "C" {

#define API __attribute__((visibility("default")))

// containers + host atomics: returns 0 if every invariant holds
API int conf_containers()
{
    {
        DynArray<Tracked> arr(0);
        for (int i = 0; i < 100; i++) arr.emplace_back(i);
        if (arr.size() != 100 || arr[57].v != 57 || arr.back().v != 99) return 1;
        arr.pop_back();
        arr.resize(150, [](Tracked *slot) { new (slot) Tracked(-1); });
        if (arr.size() != 150 || arr[149].v != -1 || arr[98].v != 98) return 2;
        arr.resize(10, [](Tracked *) {});
        if (arr.size() != 10 || Tracked::live != 10) return 3;
        DynArray<Tracked> moved(std::move(arr));
        if (moved.size() != 10 || arr.size() != 0) return 4;
        CountT slot = moved.uninit_back();
        moved.emplace(slot, 77);
        if (moved.back().v != 77) return 5;
        int sum = 0;
        for (const Tracked &t : moved) sum += t.v;
        if (sum != 45 + 77) return 6;
    }
    if (Tracked::live != 0) return 7;

    {
        HeapArray<Tracked> heap(5);
        for (CountT i = 0; i < heap.size(); i++) heap.emplace(i, (int)i * 2);
        if (heap[4].v != 8 || Tracked::live != 5) return 8;
        HeapArray<int32_t> ints { 1, 2, 3 };
        Span<int32_t> taken = ints.release();
        if (taken.size() != 3 || taken[2] != 3 || ints.size() != 0) return 9;
        rawDealloc(taken.data());
    }
    if (Tracked::live != 0) return 10;

    {
        InlineArray<Tracked, 8> small;
        small.emplace_back(1);
        small.push_back(Tracked(2));
        if (small.size() != 2 || small.capacity() != 8 || small[1].v != 2) return 11;
        small.pop_back();
        FixedInlineArray<int32_t, 3> fixed;
        fixed.emplace(2, 9);
        if (fixed.size() != 3 || fixed[2] != 9) return 12;
    }
    if (Tracked::live != 0) return 13;

    AtomicU32 counter { 5 };
    counter.fetch_add_relaxed(3);
    uint32_t expected = 8;
    while (!counter.compare_exchange_weak<sync::acq_rel, sync::relaxed>(
               expected, 1u)) {
        if (expected != 8) return 14;
    }
    AtomicFloat f { 1.f };
    f.fetch_add_relaxed(0.25f);
    if (f.load_acquire() != 1.25f || counter.load_relaxed() != 1u) return 15;
    SpinLock lock;
    lock.lock();
    if (lock.tryLock()) return 16;
    lock.unlock();
    return 0;
}

// host tracing with the reference's interface: three events, then the log file
// (<dir><trace name>_madrona_host_tracing.bin: N codes, then N time stamps)
API int conf_tracing(const char *dir)
{
    HOST_TRACING.events.clear();
    HOST_TRACING.time_stamps.clear();
    HostEventLogging(HostEvent::initStart);
    HostEventLogging(HostEvent::initEnd);
    HostEventLogging(HostEvent::megaKernelStart);
    if (HOST_TRACING.events.size() != 3 ||
            HOST_TRACING.time_stamps[2] < HOST_TRACING.time_stamps[0]) {
        return 1;
    }
    FinalizeLogging(dir);
    return 0;
}

// py::Tensor with the reference's constructor; returns bytes of the tensor
API int64_t conf_tensor_bytes(void *ptr, int32_t type, const int64_t *dims,
                              int32_t num_dims, int32_t gpu_id)
{
    py::Tensor t(ptr, (py::TensorElementType)type,
                 Span<const int64_t>(dims, (CountT)num_dims),
                 gpu_id >= 0 ? Optional<int>::make(gpu_id) :
                               Optional<int>::none());
    py::TensorInterface iface = t.interface();
    if (t.devicePtr() != ptr || t.isOnGPU() != (gpu_id >= 0) ||
            iface.dimensions.size() != (CountT)num_dims ||
            t.numDims() != num_dims) {
        return -1;
    }
    return t.numItems() * t.numBytesPerItem();
}

// Drives the device half (the simulator in api_conformance.hip) through the
// executor shim exactly as a Manager does; copies the exported Counter column
// (4 words per world) to counters_out after `steps` steps.  0 on success.
API int conf_run(uint32_t num_worlds, uint32_t steps, uint32_t *counters_out)
{
    struct Cfg { uint32_t unused; } cfg { 0 };
    std::string inits(num_worlds, '\0');

    MWCudaExecutor exec({
        .worldInitPtr = inits.data(),
        .numWorldInitBytes = 1,
        .userConfigPtr = (void *)&cfg,
        .numUserConfigBytes = (uint32_t)sizeof(Cfg),
        .numWorldDataBytes = 16,
        .worldDataAlignment = 8,
        .numWorlds = num_worlds,
        .numTaskGraphs = 1,
        .numExportedBuffers = 1,
    }, {
        {}, {}, CompileConfig::OptMode::LTO,
    }, MWCudaExecutor::initCUDA(0));

    MWCudaLaunchGraph graph = exec.buildLaunchGraphAllTaskGraphs();
    for (uint32_t i = 0; i < steps; i++) {
        exec.run(graph);
    }
    return mwhip_memcpy_d2h(counters_out, exec.getExported(0),
                            (uint64_t)num_worlds * 16);
}

}
