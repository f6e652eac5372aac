/*
 * ppo.h -- ORACLE C API (test infrastructure, NOT product code).
 *
 * Plain-C, single-threaded restatement of the PUMI-PIC per-timestep particle hot loop with
 * Kokkos::Serial semantics (SCS chunk height C=1 by default, one unfused pass per kernel per
 * walk iteration).  Each function cites the reference file:line it follows.
 *
 * PARITY PINNING: the reference cannot be built here (Kokkos/Omega_h/EnGPar absent)
 * and its mesh fixtures are an empty submodule (F2).  The oracle is pinned against every
 * reference KAT that survives that: barycentric test1/test2 (src/unit_tests.hpp:101-177), the
 * gyro-scatter KAT (test/pseudoXGCm_scatter.cpp:115-178) and the 14 search2d case shapes
 * (test/search2d.cpp:205-308) on a reconstructed tri8_parDiag plate, the Moller-Trumbore
 * ray-vs-segment semantics (test/moller_trumbore_line_tri_test.cpp:51-162) on an own 6-tet cube,
 * and the mesh-agnostic property checks of test/test_adj.cpp and particle_structs/test/.
 * Anything keyed to xgc/24k.osh element numbering is "parity unpinned".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef PPO_H
#define PPO_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- mesh */
typedef struct ppo_mesh {
  int dim, nverts, nelems, nsides;
  double* coords;     /* nverts*dim, vertex-major (Omega_h coords layout) */
  int* elem2verts;    /* nelems*(dim+1) */
  int* class_id;      /* nelems */
  int* elem2sides;    /* nelems*(dim+1): Omega_h ask_down(dim,dim-1).ab2b */
  int* side2verts;    /* nsides*dim   : ask_verts_of(dim-1) */
  int* side2elems_off; /* nsides+1     : ask_up(dim-1,dim).a2ab */
  int* side2elems;    /*              : ask_up(dim-1,dim).ab2b (ascending elem id) */
  signed char* side_exposed; /* mark_exposed_sides */
  double* elem_measure;      /* measure_elements_real */
  int* dual_off;      /* nelems+1 : ask_dual().a2ab */
  int* dual_elems;    /*          : ask_dual().ab2b (per elem, in local side order, exposed skipped) */
  int* vert2elems_off; /* nverts+1 : ask_up(0,dim).a2ab */
  int* vert2elems;     /*          : ascending elem id */
} ppo_mesh;

ppo_mesh* ppo_mesh_create(int dim, int nverts, const double* coords, int nelems,
                          const int* elem2verts, const int* class_id);
void ppo_mesh_destroy(ppo_mesh* m);
/* adjacency.tpp:418-428 */
double ppo_compute_tolerance_from_area(const ppo_mesh* m);

/* ---------------------------------------------------------------- geometry KAT entry points */
void ppo_kat_barycentric_tet(const double M[12], const double p[3], double parentVol,
                             double bcc_new[4], double bcc_old[4], double bcc_coords[4]);
void ppo_kat_barycentric_tri(const double fc[6], const double p[2], double area, double bcc[3]);
int ppo_kat_ray_triangle(const double fv[9], const double o[3], const double d[3], double tol,
                         int flip, int segment, double xpoint[3], double* dproj,
                         double* closeness, double* param);
int ppo_kat_line_edge_2d(const double ev[4], const double o[2], const double d[2], double tol,
                         int flip, double xpoint[2]);
int ppo_kat_line_triangle_simple(const double abc[9], const double o[3], const double d[3],
                                 int reverse, double tol, double xpoint[3], double* dproj);
int ppo_kat_all_positive(const double* a, int n, double tol);
int ppo_kat_min3(const double* a);
int ppo_kat_min_index(const double* a, int n);
int ppo_kat_max_index(const double* a, int n);

/* ---------------------------------------------------------------- shared deterministic sincos */
/* fdlibm-style (Cody-Waite 3-term reduction + minimax kernels), evaluated with plain IEEE
 * +,-,* in a fixed order so the HIP kernel reproduces it bit for bit. <=1 ulp vs libm. */
void ppo_sincos(double x, double* s, double* c);

/* ---------------------------------------------------------------- particle structure */
enum { PPO_PAD_EVENLY = 0, PPO_PAD_PROPORTIONALLY = 1, PPO_PAD_INVERSELY = 2 };
enum { PPO_SCS = 0, PPO_CSR = 1 };

typedef struct ppo_ps {
  int kind;
  int num_elems, num_ptcls, capacity, num_rows;
  /* SCS (scs/SellCSigma.h:186-215) */
  int C, C_max, V, sigma, num_chunks, num_slices;
  int* offsets;        /* num_slices+1 (SCS) or num_elems+1 (CSR) */
  int* slice_to_chunk; /* num_slices */
  int* row_to_element; /* num_rows */
  int* element_to_row; /* num_rows */
  unsigned char* mask; /* capacity (SCS only) */
  long* element_to_gid;
  int pad_strat;
  double shuffle_padding, extra_padding, minimize_size, padding_amount;
  int always_realloc, try_shuffling;
  int num_empty_elements;
  /* members: SoA, component-major, stride = alloc (ppView.h:7-10 LayoutLeft) */
  int nmembers;
  int* member_bytes; /* bytes of the scalar type */
  int* member_ncomp;
  long alloc;        /* slots allocated per component */
  void** data;
  long swap_alloc;
  int last_rebuild_was_shuffle;
} ppo_ps;

ppo_ps* ppo_scs_create(int C_max, int sigma, int V, int ne, int np, const int* ppe,
                       const long* gids, int pad_strat, double shuffle_padding,
                       double extra_padding, int nmembers, const int* member_bytes,
                       const int* member_ncomp, const int* particle_elements,
                       const void* const* particle_info);
ppo_ps* ppo_csr_create(int ne, int np, const int* ppe, const long* gids, double padding_amount,
                       int nmembers, const int* member_bytes, const int* member_ncomp,
                       const int* particle_elements, const void* const* particle_info);
void ppo_ps_destroy(ppo_ps* ps);
/* per-slot parent element and mask in parallel_for order (SellCSigma.h:526-558, CSR.hpp:186-213) */
void ppo_ps_slot_info(const ppo_ps* ps, int* slot_elem, unsigned char* slot_mask);
void* ppo_ps_member(ppo_ps* ps, int m);
long ppo_ps_alloc(const ppo_ps* ps);
/* scs/SCS_rebuild.h:4-314, csr/CSR_rebuild.hpp:18-118 */
void ppo_ps_rebuild(ppo_ps* ps, const int* new_element, int n_new, const int* new_particle_elements,
                    const void* const* new_particle_info);
/* ps_for.hpp:65-85 */
void ppo_ps_get_pids(const ppo_ps* ps, int* offsets_out /*ne+1*/, int* pids_out /*np*/);
/* SellCSigma.h:465-524 */
void ppo_scs_metrics(const ppo_ps* ps, int* padded_cells, int* padded_slices, int* empty_rows);

/* ---------------------------------------------------------------- pushes */
/* test/ellipticalPush.hpp:10-33 ; members: x=double[3], b=float, phi=float */
void ppo_elliptical_setup(ppo_ps* ps, int m_x, int m_b, int m_phi, double h, double k, double d);
/* test/ellipticalPush.hpp:36-70 ; trig: 0 = libm (literal reference), 1 = ppo_sincos (shared) */
void ppo_elliptical_push(ppo_ps* ps, const ppo_mesh* mesh, int m_xtgt, int m_b, int m_phi,
                         double h, double k, double d, double deg, int trig);
/* 3-D tokamak restatement of the same push: ellipse in the local (R,Z) plane +
 * rigid toroidal rotation by deg/class about the Z axis; rotation cos/sin per class are inputs */
void ppo_toroidal_push(ppo_ps* ps, const ppo_mesh* mesh, int m_x, int m_xtgt, int m_b, int m_phi,
                       double h, double k, double d, double deg, int trig);
/* test/pseudoPushAndSearch.cpp:87-119 */
void ppo_linear_push(ppo_ps* ps, int m_x, int m_xtgt, double distance, double dx, double dy,
                     double dz);
/* src/pumipic_push.hpp:17-75, launched over n */
void ppo_push_boris(int n, double* x, double* y, double* z, double* xp, double* yp, double* zp,
                    double* vx, double* vy, double* vz, const double* ex, const double* ey,
                    const double* ez, const double* br, const double* bt, const double* bz,
                    double dt);
/* test/pseudoXGCm.cpp:102-114 */
void ppo_update_positions(ppo_ps* ps, int m_x, int m_xtgt);
/* performance_tests/ps_combo160.cpp:158-178 ; members double[17], int[4], long */
void ppo_pseudo_push160(ppo_ps* ps, const double* parentElmData);

/* ---------------------------------------------------------------- searches */
/* src/pumipic_adjacency.hpp:1011-1158 ; returns found flag; loops_out optional */
int ppo_search_mesh_2d(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt, int m_pid,
                       int* elem_ids, int looplimit, int* loops_out);
/* src/pumipic_adjacency.tpp:460-654 (2-D and 3-D, BCC or intersection mode).
 * elem_ids_seeded = 0 behaves as an empty elem_ids passed in (allocated & seeded, tpp:504-515) */
int ppo_search_mesh(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt, int m_pid,
                    int* elem_ids, int elem_ids_seeded, int requireIntersection, int* inter_faces,
                    double* inter_points, int looplimit, int* loops_out, int* num_not_in_elem);
/* src/pumipic_adjacency.tpp:460-615 with a caller-supplied functor (the `Func` template argument):
 * called once per walk iteration between find_exit_face and set_new_element (tpp:561-565) */
typedef void (*ppo_trace_functor)(void* ctx, const ppo_mesh* mesh, ppo_ps* ps, int* elem_ids,
                                  int* inter_faces, int* lastExit, double* inter_points,
                                  int* ptcl_done, int m_x, int m_xtgt);
int ppo_trace_particle_through_mesh(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt,
                                    int m_pid, int* elem_ids, int elem_ids_seeded,
                                    int requireIntersection, int* inter_faces,
                                    double* inter_points, int looplimit, int* loops_out,
                                    int* num_not_in_elem, ppo_trace_functor func, void* ctx);
/* src/pumipic_adjacency.hpp:558-768 (legacy 3-D).  Q3 fallbacks are NOT replicated. */
int ppo_search_mesh_legacy3d(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt, int m_pid,
                             int* elem_ids, int elem_ids_seeded, double* xpoints, int* xface,
                             int looplimit, int* loops_out);
/* src/pumipic_adjacency.hpp:314-555 (search_mesh_3d: tol 1e-20, dual-ordered neighbours, wall
 * hits recorded in xpoints/xface).  Returns found, -2 when checkParent aborts. */
int ppo_search_mesh_3d(const ppo_mesh* mesh, ppo_ps* ps, int m_x, int m_xtgt, int m_pid,
                       int* elem_ids, int elem_ids_seeded, double* xpoints, int* xface,
                       int looplimit, int* loops_out);
/* src/pumipic_adjacency.hpp:1160-1252 */
int ppo_search_mesh_2d_pt(const ppo_mesh* mesh, const double orig[2], const double dest[2],
                          int pid, int initial_elem, int* loops, int looplimit);
/* tet variant of the single-point walk (BCC mode of search_mesh, tpp:276-285,365-416) */
int ppo_search_mesh_3d_pt(const ppo_mesh* mesh, const double dest[3], int initial_elem,
                          int looplimit);

/* ---------------------------------------------------------------- scatter / gather */
/* test/gyroScatter.hpp:101-166 (+ searchAndBuildMap :28-95); maps are nverts*gnr*gppr*(dim+1)
 * ints.  dim 3 (documented tet variant): the ring lies in the vertex's poloidal
 * half-plane, R' = R + r cos, Z' = Z + r sin, point = ((R'/R) x, (R'/R) y, Z'); the map holds the 4
 * vertices of the tet that contains the point. */
/* trig: 0 = libm cos/sin (literal reference), 1 = ppo_sincos (shared with the device) */
void ppo_create_gyro_ring_mappings(const ppo_mesh* mesh, double rmax, int gnr, int gppr,
                                   double theta_deg, int trig, int* forward_map,
                                   int* backward_map);
/* test/gyroScatter.hpp:168-229 ; nvpe = dim+1 (3 literal; 4 = documented tet deviation) */
void ppo_gyro_scatter_radius(const ppo_mesh* mesh, const ppo_ps* ps, const double* radius, const double* weight,
                             const int* v2v, double rmax, int gnr, int gppr, double* scatter_w,
                             int* num_clipped);
void ppo_gyro_scatter(const ppo_mesh* mesh, const ppo_ps* ps, const int* v2v, double rmax, int gnr,
                      int gppr, double* scatter_w);
/* test/pseudoPushAndSearch.cpp:340-374 */
void ppo_avg_ptcl_density(const ppo_mesh* mesh, const ppo_ps* ps, double* elem_cnt,
                          double* vert_density);
/* src/pumipic_adjacency.hpp:772-809 */
double ppo_interpolate_tet_vtx(const ppo_mesh* mesh, const double* field, int elem,
                               const double bcc[4], int dof, int comp);

/* src/pumipic_ptcl_ops.hpp:32-52 */
void ppo_set_unsafe_procs(const ppo_ps* ps, const int* elems, const unsigned char* safe,
                          const int* owners, int comm_rank, int* new_elems, int* new_procs);

/* gather side: src/pumipic_adjacency.hpp:772-809, src/pumipic_utils.hpp:186-241,375-454
 * (out arrays are [component][capacity]) */
void ppo_gather_tet_vtx(const ppo_mesh* mesh, const ppo_ps* ps, int m_x, const int* elem_ids,
                        const double* field, int dof, double* out, int* num_degenerate);
void ppo_interp2d_field(const ppo_ps* ps, int m_x, const double* data, double gridx0, double gridz0,
                        double dx, double dz, int nx, int nz, int cyl, int ncomp, int comp, double* out);
void ppo_interp2d_vector(const ppo_ps* ps, int m_x, const double* data3, double gridx0, double gridz0,
                         double dx, double dz, int nx, int nz, int cyl, double* out);
void ppo_interp3d_field(const ppo_ps* ps, int m_x, int nx, int ny, int nz, const double* gridx,
                        const double* gridy, const double* gridz, const double* data, double* out);

/* src/pumipic_part_construct.cpp:387-468 (bridge_dim 0 = vertices, otherwise sides) */
void ppo_bfs_buffer_layers(const ppo_mesh* mesh, int bridge_dim, int rank, int comm_size,
                           int safe_layers, int ghost_layers, const int* owner,
                           unsigned char* is_safe, int* has_part);
void ppo_bfs_safe_inward(const ppo_mesh* mesh, int bridge_dim, int rank, int safe_layers,
                         const int* owner, const int* has_part, unsigned char* safe);
/* OpenMP threads of the per-particle loops (1 at load; results are thread-count independent) */
void ppo_set_threads(int n);
int ppo_max_threads(void);
/* particle_structs/test/Distribute.h:28-89 (uniform strategy, counter-based draws) */
void ppo_exponential_tables(int ne, int* exp_start, int* exp_end);
void ppo_redistribute_particles_dist(const ppo_ps* ps, int strat, double percent_moved, unsigned long long seed,
                                     int* new_elems);
void ppo_redistribute_particles(const ppo_ps* ps, double percent_moved, unsigned long long seed,
                                int* new_elems);
/* wall geometry: src/pumipic_adjacency.hpp:812-1009.  abc = 3 vertices x 3 coordinates; *reg is
 * a TriRegion (0 VTXA,1 VTXB,2 VTXC,3 EDGEAB,4 EDGEAC,5 EDGEBC,6 TRIFACE) and is left untouched in
 * the EDGEAB branch of the non-wnormal form, as in the reference (hpp:951-958). */
void ppo_closest_point_on_triangle(const double abc[9], const double p[3], int wnormal, double q[3],
                                   int* reg);

#ifdef __cplusplus
}
#endif

#endif
