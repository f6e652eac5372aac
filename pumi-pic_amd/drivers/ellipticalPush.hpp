// ellipticalPush.hpp -- test/ellipticalPush.hpp:5-70 on the particle_structs mirror: the same two
// user lambdas, run through ps::parallel_for.  (The fused library path of the same arithmetic is
// pp_push_search / pp_elliptical_push; the lambdas use the device libm, the fused path the
// deterministic sincos shared with the CPU oracle -- DESIGN.md "Trig".)
#pragma once
#include <cmath>
#include "pseudoXGCmTypes.hpp"

namespace ellipticalPush {
inline double h;  // x coordinate of center
inline double k;  // y coordinate of center
inline double d;  // ratio of ellipse minor axis length (a) to major axis length (b)

inline void setup(PS* ptcls, const double h_in, const double k_in, const double d_in) {
  h = h_in;
  k = k_in;
  d = d_in;
  auto x_nm1 = ptcls->get<0>();
  auto ptcl_b = ptcls->get<3>();
  auto ptcl_phi = ptcls->get<4>();
  const auto h_d = h;
  const auto k_d = k;
  const auto d_d = d;
  auto setMajorAxis = PS_LAMBDA(const int&, const int& pid, const int& mask) {
    if (mask) {
      const auto w = x_nm1(pid, 0);
      const auto z = x_nm1(pid, 1);
      const auto phi = atan2(d_d * (z - k_d), w - h_d);
      const auto b = (z - k_d) / sin(phi);
      ptcl_phi(pid) = phi;
      ptcl_b(pid) = b;
    }
  };
  ps::parallel_for(ptcls, setMajorAxis);
}

inline void push(PS* ptcls, Omega_h::Mesh& m, const double deg, const int iter) {
  (void)iter;
  const auto btime = pumipic::pumipic_prebarrier();
  pp_range_push("ellipticalPush");
  pumipic::Timer timer;
  auto class_ids = m.get_array<Omega_h::ClassId>(m.dim(), "class_id");
  auto x_nm0 = ptcls->get<1>();
  auto ptcl_b = ptcls->get<3>();
  auto ptcl_phi = ptcls->get<4>();
  const auto h_d = h;
  const auto k_d = k;
  const auto d_d = d;
  auto setPosition = PS_LAMBDA(const int& e, const int& pid, const int& mask) {
    if (mask) {
      const double centerFactor = class_ids[e] == 1 ? 0.01 : 1.0;
      const double distByClass = centerFactor * (double)1.0 / class_ids[e];
      const auto degP = deg * distByClass;
      const auto phi = ptcl_phi(pid);
      const auto b = ptcl_b(pid);
      const auto a = b * d_d;
      const auto rad = phi + degP * M_PI / 180.0;
      const auto x = a * cos(rad) + h_d;
      const auto y = b * sin(rad) + k_d;
      x_nm0(pid, 0) = x;
      x_nm0(pid, 1) = y;
      ptcl_phi(pid) = rad;
    }
  };
  ps::parallel_for(ptcls, setPosition);
  pumipic::RecordTime("elliptical push", timer.seconds(), btime);
  pp_range_pop();
}
}  // namespace ellipticalPush
