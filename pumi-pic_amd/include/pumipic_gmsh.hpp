// pumipic_gmsh.hpp -- host-side reader for Gmsh ASCII .msh files (format 2.2 and 4.1), the mesh
// input the reference drivers accept through Omega_h::gmsh::read (test/pseudoXGCm.cpp:306-315,
// test/moller_trumbore_line_tri_test.cpp:45).  Omega_h itself is not part of the reference tree,
// so this follows the published MSH file format: $MeshFormat / $Nodes / $Elements sections.
//
// Output is what pp_mesh_create takes: vertex coordinates (dim per vertex), element->vertex ids
// of the top-dimensional simplices (triangles: type 2, tetrahedra: type 4) and one class id per
// element = the elementary (geometric) entity tag of the element, which is what Omega_h stores as
// `class_id`.  Vertex ids are renumbered densely in file order.  The elements one dimension below (triangles of a tet
// mesh, lines of a triangle mesh) are kept as SIDE classification: their vertices and elementary tags
// (side_verts / side_class), which Omega_h::gmsh::read turns into the `class_id` array of the sides
// (test/pseudoPushAndSearch.cpp:231 picks the start elements by it); other lower-dimensional elements are skipped.
// 2-D meshes drop the z coordinate.  Binary .msh files are rejected.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace pumipic {
namespace gmsh {

struct MeshData {
  int dim = 0;
  std::vector<double> coords;
  std::vector<int> elem2verts;
  std::vector<int> class_id;
  std::vector<int> side_verts;  // dim vertices per listed boundary entity
  std::vector<int> side_class;  // its elementary tag
};

inline bool fail(const std::string& msg, std::string* err) {
  if (err) *err = msg;
  return false;
}

inline bool read(const std::string& path, MeshData& out, std::string* err = nullptr) {
  std::ifstream in(path);
  if (!in) return fail("cannot open " + path, err);
  double version = 0;
  std::map<long long, int> node_index;  // gmsh node tag -> dense id
  std::vector<double> xyz;              // 3 per node
  struct Elem {
    int type, tag;
    long long v[4];
  };
  std::vector<Elem> elems;
  std::string line;
  while (std::getline(in, line)) {
    if (line.rfind("$MeshFormat", 0) == 0) {
      int file_type = 0, data_size = 0;
      in >> version >> file_type >> data_size;
      if (file_type != 0) return fail("binary .msh files are not supported", err);
      if (!((version >= 2.0 && version < 3.0) || (version >= 4.0 && version < 5.0)))
        return fail("unsupported .msh version", err);
    } else if (line.rfind("$Nodes", 0) == 0) {
      if (version < 3.0) {
        long long n = 0;
        in >> n;
        for (long long i = 0; i < n; ++i) {
          long long tag;
          double x, y, z;
          if (!(in >> tag >> x >> y >> z)) return fail("truncated $Nodes section", err);
          node_index[tag] = (int)(xyz.size() / 3);
          xyz.insert(xyz.end(), {x, y, z});
        }
      } else {  // 4.1: entity blocks, tags first, then coordinates
        long long nblocks = 0, n = 0, mn = 0, mx = 0;
        in >> nblocks >> n >> mn >> mx;
        for (long long b = 0; b < nblocks; ++b) {
          int edim, etag, parametric;
          long long nb;
          if (!(in >> edim >> etag >> parametric >> nb) || nb < 0 || nb > n) return fail("bad $Nodes block", err);
          std::vector<long long> tags((size_t)nb);
          for (auto& t : tags) in >> t;
          for (long long i = 0; i < nb; ++i) {
            double x, y, z;
            if (!(in >> x >> y >> z)) return fail("truncated $Nodes block", err);
            for (int k = 0; k < (parametric ? edim : 0); ++k) {
              double u;
              in >> u;
            }
            node_index[tags[(size_t)i]] = (int)(xyz.size() / 3);
            xyz.insert(xyz.end(), {x, y, z});
          }
        }
      }
    } else if (line.rfind("$Elements", 0) == 0) {
      static const int nverts_of[16] = {0, 2, 3, 4, 4, 8, 6, 5, 3, 6, 9, 10, 27, 18, 14, 1};
      if (version < 3.0) {
        long long n = 0;
        in >> n;
        std::getline(in, line);
        for (long long i = 0; i < n; ++i) {
          if (!std::getline(in, line)) return fail("truncated $Elements section", err);
          if (line.rfind("$EndElements", 0) == 0) break;  // (fewer elements than announced)
          std::istringstream ls(line);
          long long id;
          int type = 0, ntags = 0;
          ls >> id >> type >> ntags;
          if (!ls || ntags < 0 || ntags > 64) return fail("bad element line in .msh", err);
          int elementary = 0;
          for (int t = 0; t < ntags; ++t) {
            int v;
            ls >> v;
            if (t == 1) elementary = v;  // tags: physical, elementary, ...
          }
          if (type == 1 || type == 2 || type == 4) {
            Elem e{type, elementary, {0, 0, 0, 0}};
            for (int k = 0; k < nverts_of[type]; ++k) ls >> e.v[k];
            elems.push_back(e);
          }
        }
      } else {
        long long nblocks = 0, n = 0, mn = 0, mx = 0;
        in >> nblocks >> n >> mn >> mx;
        for (long long b = 0; b < nblocks; ++b) {
          int edim, etag, type;
          long long nb;
          if (!(in >> edim >> etag >> type >> nb) || nb < 0 || nb > n) return fail("bad $Elements block", err);
          const int nv = (type > 0 && type < 16) ? nverts_of[type] : -1;
          if (nv < 0) return fail("unsupported element type in .msh", err);
          for (long long i = 0; i < nb; ++i) {
            long long id;
            if (!(in >> id)) return fail("truncated $Elements block", err);
            Elem e{type, etag, {0, 0, 0, 0}};
            for (int k = 0; k < nv; ++k) {
              long long v;
              in >> v;
              if (k < 4) e.v[k] = v;
            }
            if (type == 1 || type == 2 || type == 4) elems.push_back(e);
          }
        }
      }
    }
  }
  if (version == 0) return fail("no $MeshFormat section", err);
  bool has_tet = false;
  for (auto& e : elems) has_tet |= e.type == 4;
  out.dim = has_tet ? 3 : 2;
  const int want = has_tet ? 4 : 2, nv = out.dim + 1, side_type = has_tet ? 2 : 1;
  out.coords.clear();
  out.elem2verts.clear();
  out.class_id.clear();
  out.side_verts.clear();
  out.side_class.clear();
  for (auto& e : elems) {
    if (e.type != side_type) continue;
    for (int k = 0; k < out.dim; ++k) {
      auto it = node_index.find(e.v[k]);
      if (it == node_index.end()) return fail("boundary element refers to an unknown node", err);
      out.side_verts.push_back(it->second);
    }
    out.side_class.push_back(e.tag);
  }
  for (size_t i = 0; i < xyz.size() / 3; ++i)
    for (int c = 0; c < out.dim; ++c) out.coords.push_back(xyz[3 * i + c]);
  for (auto& e : elems) {
    if (e.type != want) continue;
    for (int k = 0; k < nv; ++k) {
      auto it = node_index.find(e.v[k]);
      if (it == node_index.end()) return fail("element refers to an unknown node", err);
      out.elem2verts.push_back(it->second);
    }
    out.class_id.push_back(e.tag);
  }
  if (out.elem2verts.empty()) return fail("no triangles or tetrahedra in " + path, err);
  return true;
}

}  // namespace gmsh
}  // namespace pumipic
