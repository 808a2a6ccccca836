// BASELINE config 5: the Escape Room with physics (sims/escape_room_phys) + the
// batch ray caster -- every rigid body and button drawable, a camera on each
// agent, one directional light.  Same sources, compiled with ESCPHYS_RENDER.
#pragma once
#define ESCPHYS_RENDER 1
#include "../escape_room_phys/sim.hpp"
