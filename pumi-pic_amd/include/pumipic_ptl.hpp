// pumipic_ptl.hpp -- the particle fixture files of the reference's structure tests
// (particle_structs/test/read_particles.hpp:8-118, written by write_particle_file.cpp):
//
//   <num_elems> <num_ptcls>
//   <elem gid> <nppe>              for each element
//   <particle_elem> <id> <v0> <v1> <v2> <short> <int>     for each particle
//
// for the test particle type MemberTypes<int, double[3], short, int> (test_types.hpp:12).  Host-only,
// no HIP: the arrays are what pp_ps_create_scs / pp_ps_create_csr take (ppe, gids, particle_elements,
// particle_info[m] component-major).
#pragma once
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

namespace pumipic {
namespace ptl {

struct Particles {
  int num_elems = 0, num_ptcls = 0;
  std::vector<int> ppe;            // particles per element
  std::vector<long> gids;          // element global ids
  std::vector<int> elem;           // parent element of every particle
  std::vector<int> ids;            // member 0
  std::vector<double> vals1;       // member 1, component-major [3][num_ptcls]
  std::vector<short> vals2;        // member 2
  std::vector<int> vals3;          // member 3
  // member tables for the C-ABI constructors
  static const int* member_bytes() {
    static const int b[4] = {4, 8, 2, 4};
    return b;
  }
  static const int* member_ncomp() {
    static const int c[4] = {1, 3, 1, 1};
    return c;
  }
  std::vector<const void*> info() const { return {ids.data(), vals1.data(), vals2.data(), vals3.data()}; }
};

// readParticles (read_particles.hpp:18-73)
inline bool read(const std::string& path, Particles& p, std::string* err = nullptr) {
  std::ifstream in(path);
  if (!in) {
    if (err) *err = "[ERROR] Cannot open file " + path;
    return false;
  }
  in >> p.num_elems >> p.num_ptcls;
  if (!in || p.num_elems < 0 || p.num_ptcls < 0) {
    if (err) *err = "bad header in " + path;
    return false;
  }
  p.ppe.resize((size_t)p.num_elems);
  p.gids.resize((size_t)p.num_elems);
  for (int i = 0; i < p.num_elems; ++i) in >> p.gids[(size_t)i] >> p.ppe[(size_t)i];
  const size_t n = (size_t)p.num_ptcls;
  p.elem.resize(n);
  p.ids.resize(n);
  p.vals1.resize(3 * n);
  p.vals2.resize(n);
  p.vals3.resize(n);
  for (size_t i = 0; i < n; ++i) {
    in >> p.elem[i] >> p.ids[i];
    for (int j = 0; j < 3; ++j) in >> p.vals1[(size_t)j * n + i];
    in >> p.vals2[i] >> p.vals3[i];
  }
  if (!in) {
    if (err) *err = "truncated particle file " + path;
    return false;
  }
  return true;
}

// writeParticles (read_particles.hpp:75-118)
inline bool write(const std::string& path, const Particles& p) {
  std::ofstream out(path);
  if (!out) return false;
  out.precision(17);
  out << p.num_elems << ' ' << p.num_ptcls << '\n';
  for (int i = 0; i < p.num_elems; ++i) out << p.gids[(size_t)i] << ' ' << p.ppe[(size_t)i] << '\n';
  out << '\n';
  const size_t n = (size_t)p.num_ptcls;
  for (size_t i = 0; i < n; ++i) {
    out << p.elem[i] << ' ' << p.ids[i] << ' ';
    for (int j = 0; j < 3; ++j) out << p.vals1[(size_t)j * n + i] << ' ';
    out << p.vals2[i] << ' ' << p.vals3[i] << '\n';
  }
  return (bool)out;
}

}  // namespace ptl
}  // namespace pumipic
