// API contract: reference include/madrona/custom_context.hpp:14-28
#pragma once

#include <madrona/context.hpp>

namespace madrona {

// class MyContext : public CustomContext<MyContext, MyPerWorldState> {}
template <typename ContextT, typename DataT>
class CustomContext : public Context {
public:
    MADRONA_HD inline CustomContext(DataT *world_data,
                                    const WorkerInit &worker_init)
        : Context(world_data, worker_init)
    {}

    MADRONA_HD inline DataT &data() const
    {
        return *static_cast<DataT *>(data_);
    }

    using WorldDataT = DataT;
};

}
