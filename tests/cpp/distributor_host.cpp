// Host-only checks of the mirror's Distributor (support/psDistributor.hpp:10-138): world form and rank-subset
// form -- num_ranks / rank_host / rank / index / isWorld as the reference defines them, setRanks from a pointer
// and from a container.  No HIP call is made (compiles with hipcc, runs without a GPU).
#include <cstdio>
#include <vector>
#include "../../pumi-pic_amd/include/particle_structs.hpp"

namespace p = pumipic;

static int fails = 0;
#define CHECK(c)                                             \
  do {                                                       \
    if (!(c)) {                                              \
      printf("FAILED line %d: %s\n", __LINE__, #c);          \
      ++fails;                                               \
    }                                                        \
  } while (0)

int main() {
  {  // test/pseudoXGCm.cpp:390-396: self first, then the buffered ranks
    int ranks[4] = {5, 2, 7, 11};
    p::Distributor<> d(4, ranks);
    CHECK(!d.isWorld());
    CHECK(d.num_ranks() == 4);
    for (int i = 0; i < 4; ++i) {
      CHECK(d.rank_host(i) == ranks[i]);
      CHECK(d.rank(i) == ranks[i]);
      CHECK(d.index(ranks[i]) == i);
    }
    CHECK(d.index(3) < 0);  // not listed: undefined in the reference, -1 here (migrate refuses such a particle)
    CHECK(d.index(0) < 0);
  }
  {  // particle_structs/test/test_migrate.cpp:60-64: min(comm_size, 3) neighbours, duplicates when comm_size < 3
    int neighbors[3] = {0, 1, 1};
    p::Distributor<> d(2, neighbors);
    CHECK(d.num_ranks() == 2 && d.index(1) == 1 && d.index(0) == 0);
    std::vector<int> v = {4, 9};
    d.setRanks(v);
    CHECK(d.num_ranks() == 2 && d.rank_host(1) == 9 && d.index(9) == 1 && d.index(1) < 0);
    p::Distributor<> from_container(v);
    CHECK(!from_container.isWorld() && from_container.index(4) == 0);
  }
  {  // the world form names no ranks: rank(i) = i, index(p) = p
    p::Distributor<> w;
    CHECK(w.isWorld());
    CHECK(w.rank_host(6) == 6 && w.rank(3) == 3 && w.index(12) == 12);
  }
  printf(fails ? "distributor_host: %d FAILED\n" : "distributor_host: all checks passed\n", fails);
  return fails;
}
