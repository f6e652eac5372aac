// Host-only checks of pumi-pic_amd/include/pumipic_mpi.hpp: the MPI names the reference's drivers spell
// (test/pseudoXGCm.cpp:16-62, 434-512; src/pumipic_lb.hpp:380-398) over the library's communicator.  Started as two
// processes by the test (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, PP_COMM=tcp); no HIP call is made.
#include <cstdio>
#include "../../pumi-pic_amd/include/pumipic_mpi.hpp"

static int fails = 0;
#define CHECK(c)                                             \
  do {                                                       \
    if (!(c)) {                                              \
      printf("FAILED line %d: %s\n", __LINE__, #c);          \
      ++fails;                                               \
    }                                                        \
  } while (0)

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank = -1, size = -1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  CHECK(size == 2 && (rank == 0 || rank == 1));
  // getPtclImbalance (test/pseudoXGCm.cpp:41-62): MAX and SUM of a long, SUM of an int
  long ptcls = rank == 0 ? 1000 : 250, mx = 0, tot = 0;
  int has = 1, with = 0;
  MPI_Allreduce(&ptcls, &mx, 1, MPI_LONG, MPI_MAX, MPI_COMM_WORLD);
  MPI_Allreduce(&ptcls, &tot, 1, MPI_LONG, MPI_SUM, MPI_COMM_WORLD);
  MPI_Allreduce(&has, &with, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  CHECK(mx == 1000 && tot == 1250 && with == 2);
  // several values at once, every type, MIN
  double d[3] = {rank + 0.5, -1.0 * rank, 7.0}, dmin[3] = {0, 0, 0}, dsum[3] = {0, 0, 0};
  MPI_Allreduce(d, dmin, 3, MPI_DOUBLE, MPI_MIN, MPI_COMM_WORLD);
  MPI_Allreduce(d, dsum, 3, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
  CHECK(dmin[0] == 0.5 && dmin[1] == -1.0 && dmin[2] == 7.0);
  CHECK(dsum[0] == 2.0 && dsum[1] == -1.0 && dsum[2] == 14.0);
  float f = rank ? 2.5f : -3.0f, fmx = 0;
  MPI_Allreduce(&f, &fmx, 1, MPI_FLOAT, MPI_MAX, MPI_COMM_WORLD);
  CHECK(fmx == 2.5f);
  long long ll = 1ll << (40 + rank), llsum = 0;
  MPI_Allreduce(&ll, &llsum, 1, MPI_LONG_LONG, MPI_SUM, MPI_COMM_WORLD);
  CHECK(llsum == (1ll << 40) + (1ll << 41));
  // printPtclImb's pattern (src/pumipic_lb.hpp:383-386): the result is defined on the root only
  int np = rank ? 30 : 10, mn = -7, mxp = -7, sum = -7;
  MPI_Reduce(&np, &mn, 1, MPI_INT, MPI_MIN, 0, MPI_COMM_WORLD);
  MPI_Reduce(&np, &mxp, 1, MPI_INT, MPI_MAX, 0, MPI_COMM_WORLD);
  MPI_Reduce(&np, &sum, 1, MPI_INT, MPI_SUM, 0, MPI_COMM_WORLD);
  if (rank == 0)
    CHECK(mn == 10 && mxp == 30 && sum == 40);
  else
    CHECK(mn == -7 && mxp == -7 && sum == -7);
  MPI_Barrier(MPI_COMM_WORLD);
  // the library's output streams (support/ppPrint.h)
  pumipic::setStdout(stderr);
  CHECK(pumipic::getStdout() == stderr);
  pumipic::setStdout(stdout);
  pumipic::printInfo("rank %d of %d\n", rank, size);
  MPI_Finalize();
  printf(fails ? "rank %d: FAILED\n" : "rank %d: all checks passed\n", rank);
  return fails ? 1 : 0;
}
