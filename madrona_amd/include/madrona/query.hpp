// API contract: reference include/madrona/query.hpp:17-60 (QueryRef, Query<>,
// ResultRef<>).  A query is a run of [archetype id, column index per
// component] records inside EcsState::queryData; the host creates it through
// mwhip_make_query when the task graph is built.
#pragma once

#include <madrona/macros.hpp>
#include <madrona/types.hpp>

namespace madrona {

struct QueryRef {
    uint32_t offset;
    uint32_t numMatchingArchetypes;
    uint32_t numComponents;
    uint32_t numReferences;
    uint32_t flags;             // MWHIP_QUERY_*
};

template <typename... ComponentTs>
class Query {
public:
    MADRONA_HD Query() : ref_ { 0, 0xFFFFFFFFu, 0, 0, 0 } {}
    MADRONA_HD Query(QueryRef ref) : ref_(ref) {}

    MADRONA_HD inline uint32_t numMatchingArchetypes() const
    {
        return ref_.numMatchingArchetypes;
    }

    MADRONA_HD inline const QueryRef *getSharedRef() const { return &ref_; }
    MADRONA_HD inline bool initialized() const
    {
        return ref_.numMatchingArchetypes != 0xFFFFFFFFu;
    }

private:
    QueryRef ref_;
};

template <typename T>
class ResultRef {
public:
    MADRONA_HD inline ResultRef(T *ptr) : ptr_(ptr) {}

    MADRONA_HD inline T &value() { return *ptr_; }
    MADRONA_HD inline bool valid() const { return ptr_ != nullptr; }

private:
    T *ptr_;
};

}
