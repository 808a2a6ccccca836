#include "sim.hpp"
#include "../escape_room_phys/sim.cpp"
