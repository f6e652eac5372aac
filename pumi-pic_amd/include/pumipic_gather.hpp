// pumipic_gather.hpp -- gather side of the particle<->mesh coupling: device-inline interpolation
// helpers a push functor calls per particle, restated on raw pointers so that the same header
// serves the library's kernels (csrc/pp_gather.hip) and USER lambdas run through
// ps::parallel_for.
//
//   findBCCoordsInTet / interpolateTetVtx / interpolate3dFieldTet   src/pumipic_adjacency.hpp:772-809
//   interpolate2dField, interpolate2d_base(g|d), interpolate2d,
//   interpolate2d_field, interpolate2d_wgrid                        src/pumipic_utils.hpp:186-373
//   interpolate3d_field                                             src/pumipic_utils.hpp:375-418
//   interp2dVector_wgrid / interp2dVector                           src/pumipic_utils.hpp:420-454
//
// Arithmetic follows the reference expression by expression (compile with -ffp-contract=off for
// bit-identical results).  One documented deviation: interpolateTetVtx with dof > 1 indexes the
// reference's 4-entry gather out of bounds (`fv4[d*dof+comp]`, adjacency.hpp:776-781); here the
// vertex field is read as field[vertex*dof + comp], which is what interpolate3dFieldTet means.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

#define PPG_INLINE __host__ __device__ inline

namespace pumipic {

// barycentric coordinates of xyz in tet `elem` (find_barycentric_tet, adjacency.hpp:97-133:
// face values over vol6 of face 0 and its opposite vertex); returns false for a degenerate tet
PPG_INLINE bool findBCCoordsInTet(const double* coords, const int* mesh2verts, const double xyz[3],
                                  int elem, double bcc[4]) {
  static const int face[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {2, 0, 3}};
  double M[4][3];
  for (int i = 0; i < 4; ++i)
    for (int c = 0; c < 3; ++c) M[i][c] = coords[3 * mesh2verts[4 * elem + i] + c];
  double vals[4];
  for (int f = 0; f < 4; ++f) {
    const double* a = M[face[f][0]];
    const double* b = M[face[f][1]];
    const double* c = M[face[f][2]];
    const double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    const double ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const double ap[3] = {xyz[0] - a[0], xyz[1] - a[1], xyz[2] - a[2]};
    const double n[3] = {ac[1] * ab[2] - ac[2] * ab[1], ac[2] * ab[0] - ac[0] * ab[2],
                         ac[0] * ab[1] - ac[1] * ab[0]};
    vals[f] = ap[0] * n[0] + ap[1] * n[1] + ap[2] * n[2];
    bcc[f] = -1;
  }
  const double* a = M[0];
  const double ab[3] = {M[2][0] - a[0], M[2][1] - a[1], M[2][2] - a[2]};
  const double ac[3] = {M[1][0] - a[0], M[1][1] - a[1], M[1][2] - a[2]};
  const double n[3] = {ac[1] * ab[2] - ac[2] * ab[1], ac[2] * ab[0] - ac[0] * ab[2],
                       ac[0] * ab[1] - ac[1] * ab[0]};
  const double d3[3] = {M[3][0] - a[0], M[3][1] - a[1], M[3][2] - a[2]};
  const double vol6 = d3[0] * n[0] + d3[1] * n[1] + d3[2] * n[2];
  if (!(vol6 > 1.0e-20)) return false;
  const double inv_vol = 1.0 / vol6;
  for (int i = 0; i < 4; ++i) bcc[i] = inv_vol * vals[i];
  return true;
}

// bcc-weighted vertex field; bcc[fi] belongs to the vertex opposite face fi (3,2,0,1)
PPG_INLINE double interpolateTetVtx(const int* mesh2verts, const double* field, int elem,
                                    const double bcc[4], int dof = 1, int comp = 0) {
  static const int opp[4] = {3, 2, 0, 1};
  double val = 0;
  for (int fi = 0; fi < 4; ++fi) {
    const int v = mesh2verts[4 * elem + opp[fi]];
    val = val + bcc[fi] * field[v * dof + comp];
  }
  return val;
}
PPG_INLINE void interpolate3dFieldTet(const int* mesh2verts, const double* field, int elem,
                                      const double bcc[4], double fv[3]) {
  for (int i = 0; i < 3; ++i) fv[i] = interpolateTetVtx(mesh2verts, field, elem, bcc, 3, i);
}

PPG_INLINE double interpolate2d_base(double d1, double d2, double grid1, double grid2, double v,
                                     double dv) {
  return (d1 * (grid2 - v) + d2 * (v - grid1)) / dv;
}
// regular grid, one component at a time (utils.hpp:186-241)
PPG_INLINE double interpolate2dField(const double* data, double gridx0, double gridz0, double dx,
                                     double dz, int nx, int nz, const double pos[3],
                                     bool cylSymm = true, int nComp = 1, int comp = 0) {
  if (nx * nz == 1) return data[comp];
  double fxz = 0, fx_z1 = 0, fx_z2 = 0;
  double dim1 = pos[0];
  const double z = pos[2];
  if (cylSymm) dim1 = sqrt(pos[0] * pos[0] + pos[1] * pos[1]);
  int i = (int)floor((dim1 - gridx0) / dx);
  int j = (int)floor((z - gridz0) / dz);
  if (i < 0) i = 0;
  if (j < 0) j = 0;
  const double gridXi = gridx0 + i * dx, gridXip1 = gridx0 + (i + 1) * dx;
  const double gridZj = gridz0 + j * dz, gridZjp1 = gridz0 + (j + 1) * dz;
  if (i >= nx - 1 && j >= nz - 1) {
    fxz = data[(nx - 1 + (nz - 1) * nx) * nComp + comp];
  } else if (i >= nx - 1) {
    fx_z1 = data[(nx - 1 + j * nx) * nComp + comp];
    fx_z2 = data[(nx - 1 + (j + 1) * nx) * nComp + comp];
    fxz = ((gridZjp1 - z) * fx_z1 + (z - gridZj) * fx_z2) / dz;
  } else if (j >= nz - 1) {
    fx_z1 = data[(i + (nz - 1) * nx) * nComp + comp];
    fx_z2 = data[(i + (nz - 1) * nx) * nComp + comp];
    fxz = ((gridXip1 - dim1) * fx_z1 + (dim1 - gridXi) * fx_z2) / dx;
  } else {
    fx_z1 = ((gridXip1 - dim1) * data[(i + j * nx) * nComp + comp] +
             (dim1 - gridXi) * data[(i + 1 + j * nx) * nComp + comp]) / dx;
    fx_z2 = ((gridXip1 - dim1) * data[(i + (j + 1) * nx) * nComp + comp] +
             (dim1 - gridXi) * data[(i + 1 + (j + 1) * nx) * nComp + comp]) / dx;
    fxz = ((gridZjp1 - z) * fx_z1 + (z - gridZj) * fx_z2) / dz;
  }
  return fxz;
}
// utils.hpp:258-297 (the edge branches keep the reference's argument order)
PPG_INLINE double interpolate2d(const double* data, double gridXi, double gridXip1, double gridZj,
                                double gridZjp1, double x0, double z, int nx, int nz, int i, int j,
                                double dx, double dz, double y = 0, bool cylSymm = true,
                                int nComp = 1, int comp = 0) {
  if (nx <= 1 && nz <= 1) return data[comp];
  double x = x0;
  if (cylSymm) x = sqrt(x * x + y * y);
  double fxz = 0;
  if (i >= nx - 1 && j >= nz - 1) {
    fxz = data[(nx - 1 + (nz - 1) * nx) * nComp + comp];
  } else if (i >= nx - 1) {
    fxz = interpolate2d_base(data[(nx - 1 + j * nx) * nComp + comp],
                             data[(nx - 1 + (j + 1) * nx) * nComp + comp], z - gridZj, gridZjp1 - z, z, dz);
  } else if (j >= nz - 1) {
    fxz = interpolate2d_base(data[(i + (nz - 1) * nx) * nComp + comp],
                             data[(i + (nz - 1) * nx) * nComp + comp], x - gridXi, gridXip1 - x, x, dx);
  } else {
    const double fx_z1 = interpolate2d_base(data[(i + j * nx) * nComp + comp],
                                            data[(i + 1 + j * nx) * nComp + comp], gridXi, gridXip1, x, dx);
    const double fx_z2 = interpolate2d_base(data[(i + (j + 1) * nx) * nComp + comp],
                                            data[(i + 1 + (j + 1) * nx) * nComp + comp], gridXi, gridXip1, x, dx);
    fxz = interpolate2d_base(fx_z1, fx_z2, gridZj, gridZjp1, z, dz);
  }
  return fxz;
}
PPG_INLINE double interpolate2d_field(const double* data, double gridx0, double gridz0, double dx,
                                      double dz, int nx, int nz, const double pos[3],
                                      bool cylSymm = true, int nComp = 1, int comp = 0) {
  if (nx <= 1 && nz <= 1) return data[comp];
  double x = pos[0];
  const double z = pos[2];
  if (cylSymm) x = sqrt(x * x + pos[1] * pos[1]);
  int i = (int)floor((x - gridx0) / dx);
  int j = (int)floor((z - gridz0) / dz);
  if (i < 0) i = 0;
  if (j < 0) j = 0;
  return interpolate2d(data, gridx0 + i * dx, gridx0 + (i + 1) * dx, gridz0 + j * dz,
                       gridz0 + (j + 1) * dz, x, z, nx, nz, i, j, dx, dz, 0, false, nComp, comp);
}
PPG_INLINE double interpolate2d_wgrid(const double* data, const double* gridx, int nx,
                                      const double* gridz, int nz, const double pos[3],
                                      bool cylSymm = true, int nComp = 1, int comp = 0) {
  if (nx <= 1 || nz <= 1) return data[comp];
  double x = pos[0];
  const double z = pos[2];
  x = cylSymm ? sqrt(x * x + pos[1] * pos[1]) : x;
  const double dx = gridx[1] - gridx[0], dz = gridz[1] - gridz[0];
  int i = (int)floor((x - gridx[0]) / dx);
  int j = (int)floor((z - gridz[0]) / dz);
  i = (i < 0) ? 0 : i;
  j = (j < 0) ? 0 : j;
  const double gridXi = (i >= nx) ? gridx[nx - 1] : gridx[i];
  const double gridXip1 = (i >= nx - 1) ? gridx[nx - 1] : gridx[i + 1];
  const double gridZj = (j >= nz) ? gridz[nz - 1] : gridz[j];
  const double gridZjp1 = (j >= nz - 1) ? gridz[nz - 1] : gridz[j + 1];
  return interpolate2d(data, gridXi, gridXip1, gridZj, gridZjp1, x, z, nx, nz, i, j, dx, dz, 0, false,
                       nComp, comp);
}
// tri-linear on a structured grid (utils.hpp:375-418)
PPG_INLINE double interpolate3d_field(double x, double y, double z, int nx, int ny, int nz,
                                      const double* gridx, const double* gridy, const double* gridz,
                                      const double* data) {
  const double dx = gridx[1] - gridx[0], dy = gridy[1] - gridy[0], dz = gridz[1] - gridz[0];
  int i = (int)floor((x - gridx[0]) / dx);
  int j = (int)floor((y - gridy[0]) / dy);
  int k = (int)floor((z - gridz[0]) / dz);
  i = (i < 0) ? 0 : ((i >= nx - 1) ? (nx - 2) : i);
  j = (j < 0 || ny <= 1) ? 0 : ((j >= ny - 1) ? (ny - 2) : j);
  k = (k < 0 || nz <= 1) ? 0 : ((k >= nz - 1) ? (nz - 2) : k);
  auto baseg = [&](int di) {
    return interpolate2d_base(data[di], data[di + 1], gridx[i], gridx[i + 1], x, dx);
  };
  const double fx_z0 = baseg(i + j * nx + k * nx * ny);
  // a degenerate axis (1 plane) never reads the neighbouring plane
  const double fx_z1 = nz > 1 ? baseg(i + j * nx + (k + 1) * nx * ny) : fx_z0;
  const double fxy_z0 = ny > 1 ? baseg(i + (j + 1) * nx + k * nx * ny) : fx_z0;
  const double fxy_z1 = (ny > 1 && nz > 1) ? baseg(i + (j + 1) * nx + (k + 1) * nx * ny) : fx_z0;
  double fxyz = fx_z0;
  if (nz > 1) {
    const double fxz0 = interpolate2d_base(fx_z0, fx_z1, gridz[k], gridz[k + 1], z, dz);
    fxyz = fxz0;
    if (ny > 1) {
      const double fxz1 = interpolate2d_base(fxy_z0, fxy_z1, gridz[k], gridz[k + 1], z, dz);
      fxyz = interpolate2d_base(fxz0, fxz1, gridy[j], gridy[j + 1], y, dy);
    }
  }
  return fxyz;
}
// 3-component field on an (R,Z) grid, rotated from (R,phi) to (x,y) when cylindrically symmetric
PPG_INLINE void interp2dVector(const double* data3, double gridx0, double gridz0, double dx,
                               double dz, int nx, int nz, const double pos[3], double field[3],
                               bool cylSymm = false) {
  for (int i = 0; i < 3; ++i)
    field[i] = interpolate2d_field(data3, gridx0, gridz0, dx, dz, nx, nz, pos, cylSymm, 3, i);
  if (cylSymm) {
    const double theta = atan2(pos[1], pos[0]);
    const double f0 = field[0], f1 = field[1];
    field[0] = cos(theta) * f0 - sin(theta) * f1;
    field[1] = sin(theta) * f0 + cos(theta) * f1;
  }
}
PPG_INLINE void interp2dVector_wgrid(const double* data3, const double* gridx, int nx,
                                     const double* gridz, int nz, const double pos[3],
                                     double field[3], bool cylSymm = false) {
  for (int i = 0; i < 3; ++i) field[i] = interpolate2d_wgrid(data3, gridx, nx, gridz, nz, pos, cylSymm, 3, i);
  if (nx > 1 && nz > 1 && cylSymm) {
    const double theta = atan2(pos[1], pos[0]);
    const double f0 = field[0], f1 = field[1];
    field[0] = cos(theta) * f0 - sin(theta) * f1;
    field[1] = sin(theta) * f0 + cos(theta) * f1;
  }
}

}  // namespace pumipic
