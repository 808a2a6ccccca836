#include "escape_room_render/sim.hpp"
#include "../escape_room_phys/mgr.cpp"
