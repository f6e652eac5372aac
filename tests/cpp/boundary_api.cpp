// Driver-level checks of the mirror-API members added in round 5 (GPU): CSR_Input + CSR(Input_T&)
// (csr/CSR_input.hpp:10-42), ParticleStructure::getPIDs (ps_for.hpp:57-85), printFormat
// (scs/SellCSigma.h:403-463, csr/CSR.hpp:232-266) and ps::copy<HostSpace> (ps_for.hpp:33-55), on a Sell-C-sigma
// and a CSR structure built from the same population -- in the style of particle_structs/test/test_structure.cpp.
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
#include "../../pumi-pic_amd/include/pumipic_adjacency.hpp"

using particle_structs::lid_t;
using particle_structs::MemberTypes;
typedef MemberTypes<double[3], int, float> Types;
typedef ps::ParticleStructure<Types> PS;

static int fails = 0;
#define CHECK(c)                                    \
  do {                                              \
    if (!(c)) {                                     \
      printf("FAILED line %d: %s\n", __LINE__, #c); \
      ++fails;                                      \
    }                                               \
  } while (0)

static void checkStructure(PS* ptcls, const std::vector<int>& ppe_h, const char* what) {
  const int ne = (int)ppe_h.size();
  const int np = std::accumulate(ppe_h.begin(), ppe_h.end(), 0);
  CHECK(ptcls->nElems() == ne && ptcls->nPtcls() == np);
  // every live particle gets its slot as id, a position and a weight
  auto x = ptcls->get<0>();
  auto id = ptcls->get<1>();
  auto wgt = ptcls->get<2>();
  auto setAll = PS_LAMBDA(const int& e, const int& pid, const int& mask) {
    if (mask) {
      id(pid) = pid;
      for (int i = 0; i < 3; ++i) x(pid, i) = e + 0.25 * i;
      wgt(pid) = 0.5f * e;
    }
  };
  ps::parallel_for(ptcls, setAll, "setAll");

  // ---- getPIDs
  PS::kkLidView pids, offsets;
  ptcls->getPIDs(pids, offsets);
  CHECK((int)offsets.size() == ne + 1 && (int)pids.size() == np);
  std::vector<int> off = offsets.to_host(), pd = pids.to_host();
  CHECK(off[0] == 0 && off[(size_t)ne] == np);
  // ---- ps::copy<HostSpace>: the snapshot agrees with getPIDs and with what the device lambda wrote
  auto* host = ps::copy<ps::HostSpace>(ptcls);
  CHECK(host->nElems() == ne && host->nPtcls() == np && host->capacity() == ptcls->capacity() &&
        host->numRows() == ptcls->numRows());
  std::vector<char> seen((size_t)host->capacity(), 0);
  for (int e = 0; e < ne; ++e) {
    CHECK(off[(size_t)e + 1] - off[(size_t)e] == ppe_h[(size_t)e]);
    for (int j = off[(size_t)e]; j < off[(size_t)e + 1]; ++j) {
      const int p = pd[(size_t)j];
      const bool ok = p >= 0 && p < host->capacity() && host->particle_mask[(size_t)p] &&
                      host->slot_element[(size_t)p] == e && !seen[(size_t)p];
      CHECK(ok);
      if (ok) seen[(size_t)p] = 1;
    }
  }
  auto hx = host->get<0>();
  auto hid = host->get<1>();
  auto hw = host->get<2>();
  {
    long l = 0, b = 0;
    for (lid_t pid = 0; pid < host->capacity(); ++pid) {
      const int e = host->slot_element[(size_t)pid];
      if (e < 0 || !host->particle_mask[(size_t)pid]) continue;
      ++l;
      if (hid(pid) != pid || hx(pid, 0) != e || hx(pid, 1) != e + 0.25 || hx(pid, 2) != e + 0.5 || hw(pid) != 0.5f * e) ++b;
    }
    CHECK(l == np && b == 0);
  }
  // ps::parallel_for on the host copy visits every slot with the device structure's (element, pid, mask)
  std::vector<int> visits((size_t)host->capacity(), 0);
  int* vp = visits.data();
  const int* se = host->slot_element.data();
  int mismatches = 0;
  int* mm = &mismatches;
  auto visit = PS_LAMBDA(const int& e, const int& pid, const int& mask) {
    vp[pid] += 1 + (mask ? 1 : 0);
    if (se[pid] != e) ++*mm;
  };
  ps::parallel_for(host, visit, "visit");
  long ones = 0, twos = 0;
  for (int v : visits) {
    ones += v == 1;
    twos += v == 2;
  }
  CHECK(mismatches == 0 && twos == np);
  printf("%s: %d elements, %d particles, capacity %d: getPIDs and the host copy agree (%ld masked slots visited)\n", what,
         ne, np, ptcls->capacity(), ones);
  delete host;
}

int main() {
  p::pp_check(pp_init(0), "pp_init");
  const int ne = 150;
  std::mt19937 gen(7);
  std::vector<int> ppe_h((size_t)ne);
  for (int& v : ppe_h) v = (int)(gen() % 12);
  ppe_h[3] = 0;
  ppe_h[77] = 40;
  const int np = std::accumulate(ppe_h.begin(), ppe_h.end(), 0);
  PS::kkLidView ppe("ppe", (size_t)ne);
  ppe.from_host(ppe_h.data());
  std::vector<long> g_h((size_t)ne);
  for (int e = 0; e < ne; ++e) g_h[(size_t)e] = 3L * e + 5;
  PS::kkGidView gids("gids", (size_t)ne);
  gids.from_host(g_h.data());
  pumipic::TeamPolicy policy = pumipic::TeamPolicyAuto(4, 32);

  ps::SCS_Input<Types> scs_in(policy, INT_MAX, 1024, ne, np, ppe, gids);
  scs_in.name = "scs";
  PS* scs = new ps::SellCSigma<Types>(scs_in);
  checkStructure(scs, ppe_h, "SellCSigma");
  scs->printFormat("FORMAT scs");

  ps::CSR_Input<Types> csr_in(policy, ne, np, ppe, gids);
  csr_in.padding_amount = 1.2;
  csr_in.name = "csr from an Input";
  PS* csr = new ps::CSR<Types>(csr_in);
  CHECK(csr->getName() == "csr from an Input");
  CHECK(csr->capacity() == (int)(np * 1.2));
  checkStructure(csr, ppe_h, "CSR");
  csr->printFormat("FORMAT csr");

  // a rebuild keeps all of it working (every particle one element up, the last element's particles deleted)
  {
    PS::kkLidView new_elem("new_elem", (size_t)scs->capacity());
    auto move = PS_LAMBDA(const int& e, const int& pid, const int& mask) { new_elem(pid) = mask ? (e + 1 < ne ? e + 1 : -1) : -1; };
    ps::parallel_for(scs, move, "move");
    scs->rebuild(new_elem);
    std::vector<int> shifted((size_t)ne, 0);
    for (int e = 0; e + 1 < ne; ++e) shifted[(size_t)e + 1] = ppe_h[(size_t)e];
    checkStructure(scs, shifted, "SellCSigma after a rebuild");
  }
  delete scs;
  delete csr;
  printf(fails ? "BOUNDARY FAILED (%d)\n" : "BOUNDARY OK\n", fails);
  return fails ? 1 : 0;
}
