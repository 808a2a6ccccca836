#pragma once

namespace madrona {

class StateManager;
class ECSRegistry;
class Context;
class TaskGraphManager;
struct WorkerInit;

}
