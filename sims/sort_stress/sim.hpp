// Sort/compaction stress simulator (test workload, not a BASELINE config):
// per-world random create/destroy churn over an archetype whose columns cover
// every gather width of the sort node (1, 2, 4, 8, 12, 16, 20, 240 bytes), plus
// a temporary archetype (makeTemporary / ClearTmpNode / world sort).  Compiled
// unchanged against the reference CPU backend (oracle) and the HIP backend.
#pragma once

#include <madrona/taskgraph_builder.hpp>
#include <madrona/custom_context.hpp>
#include <madrona/rand.hpp>

namespace sortstress {

using madrona::Entity;
using madrona::RandKey;
using madrona::RNG;

namespace consts {
inline constexpr int32_t maxItems = 40;
inline constexpr int32_t maxChurn = 6;
inline constexpr int32_t maxScratch = 12;
}

enum class ExportID : uint32_t {
    Churn,
    NumExports,
};

struct Tag8 { uint8_t v; };
struct Half { uint16_t v; };
struct Key { uint32_t v; };
struct Pair { uint64_t v; };
struct Vec3 { float v[3]; };
struct Quad { float v[4]; };
struct Blob20 { uint32_t v[5]; };
struct Wide { float v[60]; };

struct Churn {
    uint32_t step;
    uint32_t numItems;
    // written by a system that works in Context::tmpAlloc scratch
    uint32_t scratchSum;
};

// Task graphs.  Step = the graph every parity test replays.  The other three
// exist on the HIP backend only and split a step so that a test can look at
// the table between nodes: churn WITHOUT compaction (destroyed rows stay),
// SortArchetypeNode<Item, Key> alone (a sort by a non-WorldID key: the
// reference CPU backend's sortArchetype is not an oracle for it),
// and the compaction that has to follow it.
enum class TaskGraphID : uint32_t {
    Step,
#ifdef MADRONA_GPU_MODE
    ChurnOnly,
    SortByKey,
    CompactOnly,
#endif
    NumTaskGraphs,
};

struct Item : public madrona::Archetype<
    Tag8, Half, Key, Pair, Vec3, Quad, Blob20, Wide
> {};

struct Scratch : public madrona::Archetype<
    Key, Vec3
> {};

class Engine;

struct Sim : public madrona::WorldBase {
    struct Config {
        uint32_t seed;
        uint32_t worldBase;
        // 1: some worlds start empty and take their first block of entity ids
        // at run time (ids are then executor-specific; values avoid them)
        uint32_t coldStart;
        // 1: every world starts with one item and only creates until it is
        // full (a population that builds up at run time: table growth)
        uint32_t rampUp;
        // 1: every world logs a message at its second step (more messages in
        // one replay than the executor's message ring holds)
        uint32_t chatty;
        // 1: every world starts with one item; worlds [0, 100) create 30 items
        // at once at their 3rd and 30th step (and destroy them again ten steps
        // later), worlds [200, 264) churn as usual, the rest stand still -- a
        // short tail behind a long sorted prefix in which ONE scatter tile of
        // the compaction chain owns more new rows than it orders in LDS
        uint32_t burst;
        // 1: three more nodes behind the step's last one, all naming the SAME
        // dependency: siblingWriteSystem writes Quad, siblingReadSystem reads
        // Quad (it only comes out right if it runs behind the first, which is
        // the registration order the reference's executors follow),
        // siblingOtherSystem touches Blob20 alone
        uint32_t siblings;
    };

    struct WorldInit {};

    static void registerTypes(madrona::ECSRegistry &registry,
                              const Config &cfg);
    static void setupTasks(madrona::TaskGraphManager &taskgraph_mgr,
                           const Config &cfg);

    Sim(Engine &ctx, const Config &cfg, const WorldInit &init);

    RNG rng;
    uint32_t mixIds;
    uint32_t rampUp;
    uint32_t chatty;
    uint32_t burst;         // 0: off, 1: a bursting world, 2: churns, 3: stands still
    int32_t numItems;
    Entity items[consts::maxItems];
};

class Engine : public madrona::CustomContext<Engine, Sim> {
public:
    using CustomContext::CustomContext;
};

}
