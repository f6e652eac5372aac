// ps_combo160 driver on the MI355X-native particle_structs mirror.
//
// Follows performance_tests/ps_combo160.cpp:15-241: distribute particles over the elements with
// one of the Distribute strategies, build one structure (0 = SCS "Sell-C-ne", 1 = CSR), run
// PS_ITERS pseudo-push passes (a USER lambda through ps::parallel_for writing the 160-byte
// PerfTypes160 record) and ITERS redistribute(percentMoved) + migrate(= rebuild on one rank)
// rounds, and print the timing table.  CabM / DPS (structures 2, 3) need Cabana and are not
// part of the hot path.
//
//   usage: ps_combo160 <num_elems> <num_ptcls> <distribution 0-4> <structure 0|1>
//                      [-p percentMoved] [-s team_size] [-v vertical_slice] [-i iterations]
// The random draws use fixed seeds (the reference seeds from the wall clock, Distribute.cpp:79).
#include <cmath>
#include <cstring>
#include <random>
#include "../include/pumipic_adjacency.hpp"

using particle_structs::lid_t;
using particle_structs::MemberTypes;
typedef MemberTypes<double[17], int[4], long> PerfTypes160;
typedef ps::ParticleStructure<PerfTypes160> PS160;
typedef PS160::kkLidView kkLidView;
typedef PS160::kkGidView kkGidView;

static const char* distribute_name(int s) {
  static const char* n[] = {"Evenly", "Uniform", "Gaussian", "Exponential", "GITRm-like"};
  return (s >= 0 && s < 5) ? n[s] : "Unknown";
}
// one element for one particle, strategies of particle_structs/test/Distribute.cpp:323-331
static int draw_element(int strat, int ne, long i, long np, std::mt19937_64& g) {
  switch (strat) {
    case 0: {  // even: first np%ne elements get one more
      const long p = np / ne, r = np % ne;
      return (int)(i < (p + 1) * r ? i / (p + 1) : r + (i - (p + 1) * r) / (p ? p : 1));
    }
    case 1:
      return (int)(g() % (unsigned long)ne);
    case 2: {
      std::normal_distribution<double> d(ne / 2.0, ne / 8.0);
      const long e = (long)d(g);
      return (int)std::min<long>(std::max<long>(e, 0), ne - 1);
    }
    case 3: {  // exponential-ish: inverse CDF of rate 1 scaled so that element ne-1 is the tail
      const double u = (double)(g() % (unsigned long)ne) / ne;
      const double t = -std::log(1 - u) / -std::log(1.0 / ne);
      const long e = (long)(t * ne);
      return (int)(e >= ne || e < 0 ? g() % (unsigned long)ne : e);
    }
    default: {  // 85 % of the particles in the first 40 % of the elements
      const int cutoff = std::max(2 * ne / 5, 1);
      if (i < (long)std::ceil(np * 0.85)) return (int)(g() % (unsigned long)cutoff);
      return cutoff + (int)(g() % (unsigned long)std::max(ne - cutoff, 1));
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "Usage: %s <num_elems> <num_ptcls> <distribution 0-4> <structure 0=SCS 1=CSR> "
                    "[-p percentMoved] [-s team_size] [-v vertical_slice] [-i iterations]\n", argv[0]);
    return EXIT_FAILURE;
  }
  const int num_elems = atoi(argv[1]);
  const long num_ptcls = atol(argv[2]);
  const int strat = atoi(argv[3]), structure = atoi(argv[4]);
  double percentMoved = 0.5;
  int team_size = 32, vert_slice = 1024, iters = 100;
  for (int i = 5; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "-p")) percentMoved = atof(argv[i + 1]);
    else if (!strcmp(argv[i], "-s")) team_size = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "-v")) vert_slice = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "-i")) iters = atoi(argv[i + 1]);
  }
  if (structure != 0 && structure != 1) {
    fprintf(stderr, "structure %d needs Cabana (CabM/DPS): not built\n", structure);
    return EXIT_FAILURE;
  }
  p::pp_check(pp_init(0), "pp_init");
  fprintf(stderr, "Test Command:\n");
  for (int i = 0; i < argc; i++) fprintf(stderr, " %s", argv[i]);
  fprintf(stderr, "\n");

  // ---- initial distribution (distribute_particles)
  printf("Generating particle distribution with strategy: %s\n", distribute_name(strat));
  std::mt19937_64 gen(0);
  std::vector<lid_t> ppe_h(num_elems, 0), pe_h((size_t)num_ptcls);
  for (long i = 0; i < num_ptcls; ++i) {
    pe_h[i] = draw_element(strat, num_elems, i, num_ptcls, gen);
    ++ppe_h[pe_h[i]];
  }
  kkLidView ppe("ptcls_per_elem", num_elems);
  ppe.from_host(ppe_h.data());
  kkGidView element_gids("element_gids", num_elems);
  {
    std::vector<pumipic::gid_t> g(num_elems);
    for (int i = 0; i < num_elems; ++i) g[i] = i;
    element_gids.from_host(g.data());
  }
  pumipic::TeamPolicy policy = pumipic::TeamPolicyAuto(num_elems, team_size);
  std::string name;
  PS160* ptcls;
  if (structure == 0) {
    name = "Sell-" + std::to_string(policy.team_size()) + "-ne";
    ps::SCS_Input<PerfTypes160> in(policy, num_elems, vert_slice, num_elems, (lid_t)num_ptcls, ppe,
                                   element_gids);
    in.name = name;
    ptcls = new ps::SellCSigma<PerfTypes160>(in);
  } else {
    name = "CSR";
    ptcls = new ps::CSR<PerfTypes160>(policy, num_elems, (lid_t)num_ptcls, ppe, element_gids);
  }
  ptcls->printMetrics();

  // ---- pseudo-push (ps_combo160.cpp:134-184)
  printf("Performing %d iterations of push on each structure\nBeginning push on structure %s\n", iters,
         name.c_str());
  pumipic::View<double> parentElmData("parentElmData", (size_t)ptcls->nElems());
  {
    std::vector<double> h((size_t)num_elems);
    for (int e = 0; e < num_elems; ++e) h[e] = std::sqrt((double)e) * e;
    parentElmData.from_host(h.data());
  }
  for (int it = 0; it < iters; ++it) {
    auto dbls = ptcls->get<0>();
    auto nums = ptcls->get<1>();
    auto lint = ptcls->get<2>();
    auto pseudoPush = PS_LAMBDA(const int& e, const int& p, const int& mask) {
      if (mask) {
        for (int i = 0; i < 17; i++) {
          dbls(p, i) = 10.3;
          dbls(p, i) = dbls(p, i) * dbls(p, i) * dbls(p, i) / sqrt((double)p) / sqrt((double)e) +
                       parentElmData(e);
        }
        for (int i = 0; i < 4; i++) nums(p, i) = 4 * p + i;
        lint(p) = p;
      } else {
        for (int i = 0; i < 17; i++) dbls(p, i) = 0;
        for (int i = 0; i < 4; i++) nums(p, i) = -1;
        lint(p) = 0;
      }
    };
    p::fence();  // Kokkos::fence() on both sides of the timed pass (ps_combo160.cpp:180-184)
    p::Timer t;
    ps::parallel_for(ptcls, pseudoPush, "pseudo push");
    p::fence();
    p::RecordTime(name + " pseudo-push", t.seconds());
  }

  // ---- PS_COMBO_CMP=<passes>: the user lambda through ps::parallel_for (what the reference's driver times,
  // ps_combo160.cpp:158-183) next to the library's own restatement of that pass (pp_pseudo_push160, what
  // bench.py --workload c4 times), HIP events on the library stream around `passes` back-to-back launches
  if (const char* cmp = getenv("PS_COMBO_CMP")) {
    const int passes = std::max(1, atoi(cmp));
    auto dbls = ptcls->get<0>();
    auto nums = ptcls->get<1>();
    auto lint = ptcls->get<2>();
    auto pseudoPush = PS_LAMBDA(const int& e, const int& p, const int& mask) {
      if (mask) {
        for (int i = 0; i < 17; i++) {
          dbls(p, i) = 10.3;
          dbls(p, i) = dbls(p, i) * dbls(p, i) * dbls(p, i) / sqrt((double)p) / sqrt((double)e) +
                       parentElmData(e);
        }
        for (int i = 0; i < 4; i++) nums(p, i) = 4 * p + i;
        lint(p) = p;
      } else {
        for (int i = 0; i < 17; i++) dbls(p, i) = 0;
        for (int i = 0; i < 4; i++) nums(p, i) = -1;
        lint(p) = 0;
      }
    };
    // the same values with ONE store per component: the reference's lambda stores 10.3 first and the compiler may
    // not drop that store (parentElmData(e), read in between, could alias it) -- 296 instead of 160 bytes written
    auto pseudoPushOnce = PS_LAMBDA(const int& e, const int& p, const int& mask) {
      if (mask) {
        const double pe = parentElmData(e);
        for (int i = 0; i < 17; i++) {
          const double d = 10.3;
          dbls(p, i) = d * d * d / sqrt((double)p) / sqrt((double)e) + pe;
        }
        for (int i = 0; i < 4; i++) nums(p, i) = 4 * p + i;
        lint(p) = p;
      } else {
        for (int i = 0; i < 17; i++) dbls(p, i) = 0;
        for (int i = 0; i < 4; i++) nums(p, i) = -1;
        lint(p) = 0;
      }
    };
    void *e0 = pp_event_create(), *e1 = pp_event_create(), *e2 = pp_event_create(), *e3 = pp_event_create();
    for (int w = 0; w < 3; ++w) {  // warm-up of the kernels
      ps::parallel_for(ptcls, pseudoPush, "pseudo push");
      ps::parallel_for(ptcls, pseudoPushOnce, "pseudo push");
      p::pp_check(pp_pseudo_push160(ptcls->handle(), parentElmData.data()), "pp_pseudo_push160");
    }
    p::pp_check(pp_event_record(e0), "event");
    for (int k = 0; k < passes; ++k) ps::parallel_for(ptcls, pseudoPush, "pseudo push");
    p::pp_check(pp_event_record(e1), "event");
    for (int k = 0; k < passes; ++k)
      p::pp_check(pp_pseudo_push160(ptcls->handle(), parentElmData.data()), "pp_pseudo_push160");
    p::pp_check(pp_event_record(e2), "event");
    for (int k = 0; k < passes; ++k) ps::parallel_for(ptcls, pseudoPushOnce, "pseudo push");
    p::pp_check(pp_event_record(e3), "event");
    p::pp_check(pp_sync(), "sync");
    printf("PUSHCMP structure %s particles %d passes %d parallel_for_lambda_ms %.6f library_kernel_ms %.6f "
           "parallel_for_single_store_lambda_ms %.6f\n",
           name.c_str(), ptcls->nPtcls(), passes, pp_event_elapsed_ms(e0, e1) / passes,
           pp_event_elapsed_ms(e1, e2) / passes, pp_event_elapsed_ms(e2, e3) / passes);
    pp_event_destroy(e0);
    pp_event_destroy(e1);
    pp_event_destroy(e2);
    pp_event_destroy(e3);
  }

  // ---- redistribute + migrate (ps_combo160.cpp:186-232; one rank: migrate == rebuild)
  printf("Performing %d iterations of migrate/rebuild on each structure\nBeginning migrate on structure %s\n",
         iters, name.c_str());
  long checksum = 0;
  for (int it = 0; it < iters; ++it) {
    const int cap = ptcls->capacity();
    kkLidView new_elms("new elems", (size_t)cap);
    p::Timer t;
    {  // redistribute_particles: every live particle moves with probability percentMoved
      pp_ps_layout_t L;
      p::pp_check(pp_ps_layout(ptcls->handle(), &L), "pp_ps_layout");
      std::vector<int> se((size_t)cap);
      std::vector<unsigned char> mk((size_t)cap);
      p::pp_check(pp_memcpy_d2h(se.data(), L.slot_elem, sizeof(int) * (size_t)cap), "slot_elem");
      p::pp_check(pp_memcpy_d2h(mk.data(), L.mask, (size_t)cap), "mask");
      std::vector<lid_t> ne_h((size_t)cap, -1);
      std::uniform_real_distribution<double> u(0, 1);
      for (int s = 0; s < cap; ++s) {
        if (!mk[s]) continue;
        ne_h[s] = u(gen) < percentMoved ? draw_element(strat == 0 ? 1 : strat, num_elems, s, cap, gen) : se[s];
      }
      new_elms.from_host(ne_h.data());
    }
    p::RecordTime("redistribute", t.seconds());
    kkLidView new_process("new_process", (size_t)cap);
    p::fence();
    p::Timer tm;
    ptcls->migrate(new_elms, new_process);
    p::fence();
    p::RecordTime(name + " migrate", tm.seconds());
    checksum += ptcls->nPtcls();
  }
  printf("RESULT structure %s particles %d rounds %d particle_rounds %ld\n", name.c_str(), ptcls->nPtcls(),
         iters, checksum);
  delete ptcls;
  p::SummarizeTime();
  return 0;
}
