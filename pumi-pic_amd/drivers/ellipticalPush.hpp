// ellipticalPush.hpp -- the pseudoXGCm push as USER code on the particle_structs mirror (the reference keeps its
// version next to the driver, test/ellipticalPush.hpp:5-70; same entry points: ellipticalPush::setup(ptcls, h, k, d),
// ellipticalPush::push(ptcls, mesh, deg, iter)).  A particle sits on the ellipse
//     x = h + d b cos(phi),   y = k + b sin(phi)
// of its own semi-axis b; setup() recovers (b, phi) from a position, push() advances phi by deg / class_id degrees
// (a hundredth of that in class 1) and writes the new position into member 1.  Both are functors handed to
// ps::parallel_for and use the device libm -- the fused library path of the same arithmetic is pp_push_search /
// pp_elliptical_push with the deterministic sincos shared with the CPU oracle (DESIGN.md "Trig").  The order of the
// floating-point operations is the reference's, so that the unchanged reference driver and this one end bit for bit
// in the same state (tests/test_gpu_refdrivers.py).
#pragma once
#include <cmath>
#include "pseudoXGCmTypes.hpp"

namespace ellipticalPush {
struct Ellipse {
  double h = 0;  // centre, x
  double k = 0;  // centre, y
  double d = 1;  // minor over major axis
};
inline Ellipse& shape() {
  static Ellipse e;
  return e;
}

// position (member 0) -> angle (member 4) and semi-axis (member 3)
template <class Pos, class Axis, class Angle>
struct AxisOfPosition {
  Pos pos;
  Axis semi_axis;
  Angle angle;
  Ellipse el;
  PP_INLINE void operator()(const int&, const int& slot, const int& live) const {
    if (!live) return;
    const double dx = pos(slot, 0) - el.h, dy = pos(slot, 1) - el.k;
    const double phi = atan2(el.d * dy, dx);
    angle(slot) = phi;
    semi_axis(slot) = dy / sin(phi);
  }
};
// angle += step of the element's class; target position (member 1) on the particle's ellipse
template <class Tgt, class Axis, class Angle, class Classes>
struct AdvanceOnEllipse {
  Tgt target;
  Axis semi_axis;
  Angle angle;
  Classes class_of_element;
  Ellipse el;
  double degrees;
  PP_INLINE void operator()(const int& elem, const int& slot, const int& live) const {
    if (!live) return;
    const int cls = class_of_element[elem];
    const double slow = cls == 1 ? 0.01 : 1.0;
    const double per_class = slow * (double)1.0 / cls;
    const double step = degrees * per_class;
    const double b = semi_axis(slot);
    const double a = b * el.d;
    const double phi = angle(slot) + step * M_PI / 180.0;
    target(slot, 0) = a * cos(phi) + el.h;
    target(slot, 1) = b * sin(phi) + el.k;
    angle(slot) = phi;
  }
};

inline void setup(PS* ptcls, const double h_in, const double k_in, const double d_in) {
  shape() = Ellipse{h_in, k_in, d_in};
  auto pos = ptcls->get<0>();
  auto b = ptcls->get<3>();
  auto phi = ptcls->get<4>();
  AxisOfPosition<decltype(pos), decltype(b), decltype(phi)> f{pos, b, phi, shape()};
  ps::parallel_for(ptcls, f);
}

inline void push(PS* ptcls, Omega_h::Mesh& m, const double deg, const int /*iter*/) {
  const double waited = pumipic::pumipic_prebarrier();
  pp_range_push("ellipticalPush");
  pumipic::Timer clock;
  auto classes = m.get_array<Omega_h::ClassId>(m.dim(), "class_id");
  auto tgt = ptcls->get<1>();
  auto b = ptcls->get<3>();
  auto phi = ptcls->get<4>();
  AdvanceOnEllipse<decltype(tgt), decltype(b), decltype(phi), decltype(classes)> f{tgt, b, phi, classes, shape(), deg};
  ps::parallel_for(ptcls, f);
  pumipic::RecordTime("elliptical push", clock.seconds(), waited);
  pp_range_pop();
}
}  // namespace ellipticalPush
