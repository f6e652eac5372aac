// pp_ps_build.hpp -- host-side construction of the particle structures (included by pp_ps.hip only).
//
// Reference: particle_structs/src/scs/SellCSigma.h:229-285 (constructor), SCS_buildFns.h (chooseChunkHeight,
// sigmaSort, constructChunks, createGlobalMapping, constructOffsets, setupParticleMask, initSCSData),
// csr/CSR.hpp + CSR_buildFns.hpp.  Construction is set-up, not the hot path: the layout is computed on the host
// (HostLayout / host_layout, the same arithmetic as the device re-layout) and uploaded; member arrays are
// allocated with the component-stride and start-offset rules measured in DESIGN.md section 4.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
#include "pp_internal.hpp"

namespace {

struct HostLayout {
  int C = 1, nchunks = 0, nslices = 0, capacity = 0, num_empty = 0;
  std::vector<int> chunk_widths, row_to_element, element_to_row, offsets, slice_to_chunk, ptcls,
      chunk_start;
};

int choose_chunk_height(int maxC, const int* ppe, int n) {
  int cnt = 0;
  for (int i = 0; i < n; ++i) cnt += ppe[i] > 0;
  if (cnt == 0) return 1;
  return cnt < maxC ? cnt : maxC;
}

void host_layout(HostLayout& L, int C, int V, int sigma, int ne, const int* ppe, int pad_strat,
                 double pad) {
  L.C = C;
  std::vector<int> index((size_t)ne);
  L.ptcls.assign(ppe, ppe + ne);
  std::iota(index.begin(), index.end(), 0);
  if (sigma > 1 && ne > 0) {
    const int sg = std::min(sigma, std::max(ne, 1));
    const int n_sigma = ne / sg;
    for (int w = 0; w < n_sigma; ++w) {
      const int start = w * sg, end = (w == n_sigma - 1) ? ne : start + sg;
      std::stable_sort(index.begin() + start, index.begin() + end,
                       [&](int a, int b) { return ppe[a] < ppe[b]; });
    }
    for (int i = 0; i < ne; ++i) L.ptcls[i] = ppe[index[i]];
  }
  L.nchunks = ne / C + (ne % C != 0);
  const int nrows = L.nchunks * C;
  L.row_to_element.assign((size_t)nrows, 0);
  L.element_to_row.assign((size_t)nrows, 0);
  L.num_empty = 0;
  for (int i = 0; i < ne; ++i) {
    L.row_to_element[i] = index[i];
    L.element_to_row[index[i]] = i;
    L.num_empty += (L.ptcls[i] == 0);
  }
  for (int i = ne; i < nrows; ++i) {
    L.row_to_element[i] = i;
    L.element_to_row[i] = i;
    L.num_empty += 1;
  }
  L.chunk_widths.assign((size_t)L.nchunks, 0);
  for (int c = 0; c < L.nchunks; ++c) {
    int w = 0;
    for (int r = 0; r < C; ++r) {
      const int row = c * C + r;
      if (row < ne) w = std::max(w, L.ptcls[row]);
    }
    L.chunk_widths[c] = w;
  }
  if (pad > 0) {
    int cw_sum = 0, cw_cnt = 0;
    double cw_inv = 0;
    for (int c = 0; c < L.nchunks; ++c) {
      cw_sum += L.chunk_widths[c];
      cw_cnt += L.chunk_widths[c] > 0;
      if (L.chunk_widths[c] > 0) cw_inv += 1.0 / L.chunk_widths[c];
    }
    if (cw_sum > 0) {
      const double cw_sum2 = cw_sum / cw_inv * pad;
      const int avg_pad = (int)(cw_sum * pad / cw_cnt);
      for (int c = 0; c < L.nchunks; ++c) {
        int& w = L.chunk_widths[c];
        if (pad_strat == PP_PAD_EVENLY) {
          if (w > 0) w += avg_pad;
        } else if (pad_strat == PP_PAD_PROPORTIONALLY) {
          w = (int)(w + w * pad);
        } else {
          if (w != 0) w = (int)(w + cw_sum2 / w);
        }
      }
    }
  }
  L.nslices = 0;
  for (int c = 0; c < L.nchunks; ++c) L.nslices += L.chunk_widths[c] / V + (L.chunk_widths[c] % V != 0);
  L.offsets.assign((size_t)L.nslices + 1, 0);
  L.slice_to_chunk.assign((size_t)L.nslices, 0);
  L.chunk_start.assign((size_t)L.nchunks, 0);
  int s = 0;
  for (int c = 0; c < L.nchunks; ++c) {
    const int w = L.chunk_widths[c];
    const int ns = w / V + (w % V != 0);
    L.chunk_start[c] = L.offsets[s];
    for (int j = 0; j < ns; ++j, ++s) {
      L.slice_to_chunk[s] = c;
      const int rem = w % V;
      const int val = rem + (rem == 0) * V;
      const int size = (j == ns - 1) ? val * C : V * C;
      L.offsets[s + 1] = L.offsets[s] + size;
    }
  }
  L.capacity = L.offsets[L.nslices];
}

template <class T>
int upload_vec(pp::DevBuf& d, const std::vector<T>& h) {
  PP_HIP_CHECK(d.reserve(std::max<size_t>(h.size() * sizeof(T), 16)));
  if (!h.empty())
    PP_HIP_CHECK(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice,
                                pp::stream()));
  return PP_OK;
}

// Component stride of the SoA member arrays.  The components of a member (and the members
// themselves) are streamed side by side by every particle kernel; when the stride in bytes is a
// multiple of a large power of two (an SCS capacity is a multiple of C = 64 slots and often of much
// more) all those streams sit at the same position of the HBM channel interleave.  The stride is
// therefore a multiple of 64 slots whose quotient is 17 mod 32: successive component arrays start
// 17 x 512 B apart modulo 16 KiB.
int64_t spread_stride(int64_t n) {
  if (n < 4096) return n;
  int64_t s = (n + 63) / 64;
  while (s % 32 != 17) ++s;
  return s * 64;
}
int alloc_members(pp_ps* ps, std::vector<pp::DevBuf>& bufs, int64_t stride, bool zero) {
  bufs.resize((size_t)ps->nmembers);
  for (int m = 0; m < ps->nmembers; ++m) {
    const size_t bytes = (size_t)stride * ps->member_ncomp[m] * ps->member_bytes[m];
    PP_HIP_CHECK(bufs[m].reserve(std::max<size_t>(bytes, 16), (size_t)((m + 1) * 5 % 32) * 512));
    if (zero && bytes) PP_HIP_CHECK(hipMemsetAsync(bufs[m].p, 0, bytes, pp::stream()));
  }
  return PP_OK;
}

int set_members(pp_ps* ps, int nmembers, const int* mb, const int* mc) {
  PP_REQUIRE(nmembers > 0 && nmembers <= 8 && mb && mc, "particle structure: 1..8 members required");
  ps->nmembers = nmembers;
  ps->member_bytes.assign(mb, mb + nmembers);
  ps->member_ncomp.assign(mc, mc + nmembers);
  ps->member_map.resize((size_t)nmembers);
  std::iota(ps->member_map.begin(), ps->member_map.end(), 0);
  for (int m = 0; m < nmembers; ++m) {
    PP_REQUIRE(mb[m] == 1 || mb[m] == 2 || mb[m] == 4 || mb[m] == 8,
               "member scalar size must be 1, 2, 4 or 8 bytes");
    PP_REQUIRE(mc[m] >= 1, "member needs at least one component");
  }
  return PP_OK;
}

// the pseudoXGCm particle type (test/pseudoXGCmTypes.hpp): double[3], double[3], three 4-byte scalars
bool xgcm_shape(const pp_ps* ps) {
  static const int want_b[5] = {8, 8, 4, 4, 4}, want_c[5] = {3, 3, 1, 1, 1};
  if (ps->nmembers != 5) return false;
  for (int m = 0; m < 5; ++m)
    if (ps->member_bytes[m] != want_b[m] || ps->member_ncomp[m] != want_c[m]) return false;
  return true;
}

// host-side initial placement: scatter particle_info (component-major [ncomp][np]) into a host
// staging image of the member buffers, then upload.
int upload_initial(pp_ps* ps, const std::vector<int>& slot_of_particle, int np,
                   const void* const* info) {
  for (int m = 0; m < ps->nmembers; ++m) {
    if (!info[m]) continue;
    const int b = ps->member_bytes[m], nc = ps->member_ncomp[m];
    std::vector<unsigned char> img((size_t)ps->stride * nc * b, 0);
    const unsigned char* src = (const unsigned char*)info[m];
    for (int c = 0; c < nc; ++c)
      for (int i = 0; i < np; ++i)
        memcpy(&img[((size_t)c * ps->stride + slot_of_particle[i]) * b],
               &src[((size_t)c * np + i) * b], (size_t)b);
    PP_HIP_CHECK(hipMemcpy(ps->data[m].p, img.data(), img.size(), hipMemcpyHostToDevice));
  }
  return PP_OK;
}

}  // namespace
