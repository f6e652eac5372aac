// pumipic_kktypes.hpp -- src/pumipic_kktypes.hpp:10-32 and the hostToDevice / deviceToHost helpers of
// support/SupportKK.h:55-100 on the mirror's View.
#pragma once
#include <chrono>
#include <thread>
#include "pumipic_adjacency.hpp"  // fp_t, Vector3d
namespace pumipic {
typedef DeviceSpace exe_space;
typedef DeviceSpace device_type;
typedef View<lid_t> kkLidView;
typedef View<gid_t> kkGidView;
typedef View<fp_t> kkFpView;
inline void hostToDeviceLid(kkLidView d, lid_t* h) { d.from_host(h); }
inline void deviceToHostLid(kkLidView d, lid_t* h) {
  if (d.size()) pp_check(pp_memcpy_d2h(h, d.data(), d.size() * sizeof(lid_t)), "deviceToHostLid");
}
inline void hostToDeviceFp(kkFpView d, fp_t* h) { d.from_host(h); }
}  // namespace pumipic
