// forward declarations shared by the API headers
#pragma once

namespace madrona {
class Context; class StateManager; class ECSRegistry; class TaskGraphManager;
struct WorkerInit;
}
