// TEST INFRASTRUCTURE ONLY.  Forwards to the reference's device-side header
// where it lies (src/mw/device/include/madrona/bvh.hpp): that directory cannot
// be put on the include path of a host build, it shadows the host headers.
#pragma once
#include REF_DEVICE_BVH_HPP
