#include <madrona/mwhip/user_prelude.hpp>
#pragma clang force_cuda_host_device begin
#include "/root/repo/sims/escape_room_phys/sim.cpp"
#pragma clang force_cuda_host_device end
