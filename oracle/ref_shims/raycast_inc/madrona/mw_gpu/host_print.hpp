// TEST INFRASTRUCTURE ONLY.  Host stand-in for the device-side HostPrint the
// reference's ray caster logs through: messages are dropped.
#pragma once
namespace madrona {
namespace mwGPU {
struct HostPrint {
    template <typename... Args>
    static void log(const char *, Args &&...) {}
};
}
}
