/*
 * pumipic_hip.h -- C-ABI of libpumipic_hip.so: the MI355X (gfx950) implementation of PUMI-PIC's
 * per-timestep particle hot loop (push -> element-to-element adjacency search -> particle<->mesh
 * scatter -> Sell-C-sigma / CSR rebuild + migration pack).
 *
 * The reference has no FFI: its "API" is C++ templates over Kokkos views.  Each
 * entry point below names the reference interface it stands in for (file:line under
 * SCOREC/pumi-pic).  The C++ host mirror of that template API lives in pumi-pic_amd/include/ and
 * is a thin inline layer over these functions; INTEGRATION.md shows the binding a maintainer adds.
 *
 * Conventions
 *   - all functions are extern "C", take plain pointers/sizes, return 0 on success or a negative
 *     PP_E* code; pp_last_error() gives a message.  No C++/torch types cross this boundary.
 *   - *_host pointers are host memory, *_dev pointers are HIP device memory on the current
 *     device.  Handles own their device memory.  Pointers obtained from pp_ps_member_ptr /
 *     pp_ps_layout are invalidated by pp_ps_rebuild / pp_ps_migrate_* (same rule as the
 *     reference's Segment accessors, SURVEY 8(b) "Ownership").
 *   - work is enqueued on one HIP stream per process (pp_stream()); calls return after enqueue
 *     unless they produce a host-visible result (found flags, counts, capacities).
 *   - particle members are SoA, component-major: member m, component c, slot pid lives at
 *     base_m[c*stride + pid] with stride = pp_ps_member_stride() (reference: pumipic::View is
 *     LayoutLeft, support/ppView.h:7-10; Segment(pid,i), support/Segment.h:29-98).
 *   - ids are int32 (lid_t / Omega_h LO), global ids int64 (gid_t), reals are IEEE double.
 *     Device code is built with -ffp-contract=off: every sign test and arg-min follows the
 *     reference's operation order so element ids are bit-identical to a Kokkos::Serial run.
 */
#ifndef PUMIPIC_HIP_H
#define PUMIPIC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  PP_OK = 0,
  PP_EINVAL = -1,   /* bad argument */
  PP_EHIP = -2,     /* HIP runtime error (no device, launch failure, OOM) */
  PP_ENOTIMPL = -3,
  PP_ESTATE = -4
};
enum { PP_SCS = 0, PP_CSR = 1 };
enum { PP_PAD_EVENLY = 0, PP_PAD_PROPORTIONALLY = 1, PP_PAD_INVERSELY = 2 }; /* scs_input.hpp:4-11 */

typedef struct pp_mesh pp_mesh;
typedef struct pp_ps pp_ps;

/* ------------------------------------------------------------------ runtime */
const char* pp_last_error(void);
const char* pp_version(void);
/* Select the device and create the library stream. Fails loudly (PP_EHIP) without a GPU. */
int pp_init(int device);
void* pp_stream(void); /* hipStream_t */
int pp_sync(void);
/* the HIP runtime's sticky last error of the calling thread, not cleared (0 = none; message in *msg_out) */
int pp_peek_hip_error(const char** msg_out);
int pp_device_count(void);
/* plain device memory helpers so hosts without a HIP toolchain (ctypes, cgo, JNI) can drive it */
/* pp_malloc / pp_free are POOLED: a freed block is kept (up to pp_pool_set_limit bytes) and handed to the
 * next request it fits, without hipFree's device synchronisation -- the per-step arrays of the reference's drivers
 * (Omega_h::Write<LO> elem_ids(capacity, -1) per search, test/pseudoXGCm.cpp:142-146; new_elems / new_procs per
 * migration, src/pumipic_ptcl_ops.hpp:56-60) cost a free-list lookup.  Reuse is ordered by the library stream:
 * code that reads such a block from ANOTHER stream must finish before the block is freed. */
void* pp_malloc(size_t bytes);
int pp_free(void* dev);
int pp_pool_trim(void); /* hand every cached block back to the runtime */
/* cap of the cache in bytes (default: 1/64 of the device memory, between 1 and 4 GiB; PP_POOL_LIMIT_MB in the
 * environment; 0 = cache nothing).  pp_malloc memory must be released with pp_free (a block released with hipFree
 * is recognised when its address comes back and is not cached); call pp_pool_trim before handing the device to
 * another allocator that needs the memory (torch, RCCL, the caller's own hipMalloc). */
int pp_pool_set_limit(size_t bytes);
int pp_pool_stats(size_t* live_bytes, size_t* cached_bytes, long long* hits, long long* misses);
/* Kokkos::View<T*>(name, n) filled with a value / Omega_h::Write<T>(n, value): `count` items of pattern_bytes
 * (1, 2, 4, 8) each, on the library stream */
int pp_fill(void* dev, const void* pattern_host, int pattern_bytes, size_t count);
int pp_memcpy_h2d(void* dev, const void* host, size_t bytes);
int pp_memcpy_d2h(void* host, const void* dev, size_t bytes);
int pp_memset(void* dev, int value, size_t bytes);
/* HIP-event timing on the library stream (bench.py measures kernels with these) */
void* pp_event_create(void);
int pp_event_record(void* ev);
float pp_event_elapsed_ms(void* start, void* stop); /* synchronises on stop */
int pp_event_destroy(void* ev);

/* ------------------------------------------------------------------ mesh
 * Replaces the Omega_h mesh queries the hot path makes (adjacency.tpp:238-241,394-396,489,
 * 497-501; adjacency.hpp:568-574,1030-1036): ask_elem_verts, coords, ask_down(dim,dim-1),
 * ask_up(dim-1,dim), ask_verts_of(dim-1), mark_exposed_sides, measure_elements_real, ask_dual,
 * ask_up(0,dim), get_array<ClassId>(dim,"class_id").  Derived once and cached.
 * Side numbering is this library's canonical one (first-seen order). */
pp_mesh* pp_mesh_create(int dim, int nverts, const double* coords_host, int nelems,
                        const int* elem2verts_host, const int* class_id_host);
int pp_mesh_destroy(pp_mesh* m);
int pp_mesh_info(const pp_mesh* m, int* dim, int* nverts, int* nelems, int* nsides);
/* number of edges (Omega_h Mesh::nedges()): the sides of a triangle mesh, the derived edges of a tet mesh */
int pp_mesh_num_edges(const pp_mesh* m);
/* compute_tolerance_from_area, adjacency.tpp:418-428 */
double pp_mesh_tolerance(const pp_mesh* m);
enum {
  PP_MESH_COORDS = 0,      /* double nverts*dim */
  PP_MESH_ELEM2VERTS = 1,  /* int nelems*(dim+1) */
  PP_MESH_CLASS_ID = 2,    /* int nelems */
  PP_MESH_ELEM2SIDES = 3,  /* int nelems*(dim+1) */
  PP_MESH_SIDE2VERTS = 4,  /* int nsides*dim */
  PP_MESH_SIDE2ELEMS_OFF = 5, /* int nsides+1 */
  PP_MESH_SIDE2ELEMS = 6,  /* int */
  PP_MESH_SIDE_EXPOSED = 7, /* int8 nsides */
  PP_MESH_ELEM_MEASURE = 8, /* double nelems */
  PP_MESH_DUAL_OFF = 9,    /* int nelems+1 */
  PP_MESH_DUAL_ELEMS = 10, /* int */
  PP_MESH_VERT2ELEMS_OFF = 11, /* int nverts+1 */
  PP_MESH_VERT2ELEMS = 12, /* int */
  PP_MESH_ELEM_RECORDS = 13, /* packed per-element walk records (DESIGN.md "data layout") */
  /* edges of a TET mesh (entity dimension 1; Omega_h ask_down(3,1) / ask_up(1,3)), derived on first use,
   * local order (0,1),(1,2),(2,0),(0,3),(1,3),(2,3), numbered as first seen like the sides */
  PP_MESH_ELEM2EDGES = 14,     /* int nelems*6 */
  PP_MESH_EDGE2VERTS = 15,     /* int nedges*2 */
  PP_MESH_EDGE2ELEMS_OFF = 16, /* int nedges+1 */
  PP_MESH_EDGE2ELEMS = 17      /* int */
};
/* device pointer + item count of a mesh array (for user kernels written against the C++ mirror) */
const void* pp_mesh_array_dev(const pp_mesh* m, int which, size_t* count);
/* copy a mesh array to host memory (out must hold count items) */
int pp_mesh_array_to_host(const pp_mesh* m, int which, void* out_host);

/* ------------------------------------------------------------------ particle structures
 * SellCSigma ctor scs/SellCSigma.h:66-72 + SCS_Input scs_input.hpp:27-36; CSR ctor
 * csr/CSR.hpp:37-69 + CSR_Input.  member_bytes[m] = sizeof(scalar type), member_ncomp[m] =
 * number of components (MemberTypes<double[3],double[3],int,float,float> -> bytes {8,8,4,4,4},
 * ncomp {3,3,1,1,1}).  particle_info_host[m] is [ncomp][np] component-major, or NULL.
 * C is the chunk height (team size); 64 is native on CDNA wave64. */
pp_ps* pp_ps_create_scs(int C, int sigma, int V, int num_elems, int num_ptcls,
                        const int* ppe_host, const int64_t* gids_host, int pad_strat,
                        double shuffle_padding, double extra_padding, int nmembers,
                        const int* member_bytes, const int* member_ncomp,
                        const int* particle_elements_host, const void* const* particle_info_host);
pp_ps* pp_ps_create_csr(int num_elems, int num_ptcls, const int* ppe_host,
                        const int64_t* gids_host, double padding_amount, int nmembers,
                        const int* member_bytes, const int* member_ncomp,
                        const int* particle_elements_host, const void* const* particle_info_host);
int pp_ps_destroy(pp_ps* ps);
/* SellCSigma::copy<MSpace> / CSR::copy, scs/SellCSigma.h:336-391: a deep copy -- the same layout arrays, the same
 * slots, every member -- as a new, independent structure (ps::copy<Space>(ptcls) of ps_for.hpp:33-55 in the mirror) */
pp_ps* pp_ps_clone(pp_ps* ps);
typedef struct pp_ps_info_t {
  int kind, num_elems, num_ptcls, capacity, num_rows;
  int C, V, sigma, num_chunks, num_slices, nmembers;
  int64_t stride; /* allocated slots per component */
} pp_ps_info_t;
/* nElems/nPtcls/capacity/numRows particle_structure.hpp:80-83 */
int pp_ps_info(const pp_ps* ps, pp_ps_info_t* out);
/* ptcls->get<N>() particle_structure.hpp:90-104 */
void* pp_ps_member_ptr(pp_ps* ps, int m);
int64_t pp_ps_member_stride(const pp_ps* ps);
typedef struct pp_ps_layout_t {
  const int* offsets;        /* SCS: num_slices+1 ; CSR: num_elems+1 */
  const int* slice_to_chunk; /* SCS only */
  const int* row_to_element; /* SCS only */
  const int* element_to_row; /* SCS only */
  const unsigned char* mask; /* SCS only: particle_mask, 1 byte per slot */
  const int* slot_elem;      /* capacity: parent element of every slot (-1 = not iterated) */
} pp_ps_layout_t;
/* device pointers used by the header-only ps::parallel_for (SellCSigma.h:526-558, CSR.hpp:186-213) */
int pp_ps_layout(const pp_ps* ps, pp_ps_layout_t* out);
/* What the header-only ps::parallel_for needs to visit every slot (thread = slot): the mask and the slot's
 * parent element.  Sell-C-sigma with chunk height 64: 64 consecutive slots are the 64 rows of one column of ONE
 * chunk, so the element is row_to_element[64 * group_chunk[pid / 64] + pid % 64] -- one wave-uniform load from a
 * capacity/64 table plus a coalesced load from the 0.4 MB row table, instead of 4 B per slot from a capacity-long
 * slot -> element table (which a full re-layout leaves unwritten until somebody asks: pp_ps_layout).  Else
 * (CSR, other chunk heights): group_chunk is NULL and slot_elem is filled. */
typedef struct pp_ps_iter_t {
  int capacity;
  const unsigned char* mask;
  const int* group_chunk;    /* capacity/64 entries, or NULL */
  const int* row_to_element; /* with group_chunk */
  const int* slot_elem;      /* without group_chunk */
} pp_ps_iter_t;
int pp_ps_iteration(const pp_ps* ps, pp_ps_iter_t* out);
int pp_ps_layout_to_host(const pp_ps* ps, int* offsets, int* slice_to_chunk, int* row_to_element,
                         int* element_to_row, unsigned char* mask, int* slot_elem);
/* element_to_gid (the kkGidView the structure was built with, SellCSigma.h:209): copies num_elems gids to
 * out_host and returns num_elems; 0 when the structure has none; negative on error */
int pp_ps_gids_to_host(const pp_ps* ps, int64_t* out_host);
int pp_ps_member_to_host(pp_ps* ps, int m, void* out_host);   /* [ncomp][stride] */
int pp_ps_member_from_host(pp_ps* ps, int m, const void* in_host);
/* rebuild(new_element, new_particle_elements, new_particle_info) scs/SCS_rebuild.h:122-314,
 * csr/CSR_rebuild.hpp:18-118.  new_element_dev has capacity entries (-1 = delete).
 * new_info_dev[m] is a DEVICE array [ncomp][n_new].  In place when the layout can be kept
 * (pp_ps_set_shuffling), else the full counting-sort re-layout.
 * LIFETIME OF THE INPUTS: the call returns when the new totals are on the host (the one host wait
 * of a rebuild); the kernels that move the particles are still running then and read
 * new_element_dev, new_elems_dev and the new_info_dev arrays.  They are ordered on the library
 * stream (pp_stream), so any later library call and pp_free / hipFree are safe; a caller that
 * overwrites or frees those buffers from ANOTHER stream must first wait for the library stream
 * (pp_sync, or an event recorded on pp_stream).  The same holds for the migration entry points.
 * THREADS: for pp_ps_rebuild* and the migration entry points structures are independent (each owns its
 * scratch, its pinned landing zone and its stamps): two host threads may rebuild DIFFERENT structures, calls on
 * ONE structure must not overlap, and all calls share one stream.  The searches (pp_push_search, pp_search_mesh*)
 * and pp_gyro_scatter* keep process-wide scratch (the deferred-walk queue, alternating counter sets, the ring
 * accumulator): call them from one host thread at a time -- the reference's model, one host thread per rank
 * (SURVEY 8(b) "Threading"). */
int pp_ps_rebuild(pp_ps* ps, const int* new_element_dev, int n_new, const int* new_elems_dev,
                  const void* const* new_info_dev);
/* updatePtclPositions (test/pseudoXGCm.cpp:102-114: x <- x_tgt, x_tgt <- 0) fused into the
 * rebuild's single data-movement pass; result-identical to pp_update_positions + pp_ps_rebuild
 * for every live particle.  This is what the drivers' rebuild() helper does (pseudoXGCm.cpp:116-140) */
int pp_ps_rebuild_commit(pp_ps* ps, int m_x, int m_xtgt, const int* new_element_dev, int n_new,
                         const int* new_elems_dev, const void* const* new_info_dev);
/* The rebuild of a pseudoXGCm step followed by its gyroScatter calls (test/pseudoXGCm.cpp:116-140,
 * 529-530) as ONE call: result-identical to pp_ps_rebuild (m_x = m_xtgt = -1) or pp_ps_rebuild_commit
 * followed by pp_gyro_scatter(mesh, ps, v2v_dev[k], ..., scatter_w_dev[k]) for k < nmaps.  The
 * scatter depends on the new per-element counts only, so it is enqueued before the rebuild's one
 * host sync and the GPU keeps working while the host waits. */
int pp_ps_rebuild_scatter(pp_ps* ps, int m_x, int m_xtgt, const int* new_element_dev, int n_new,
                          const int* new_elems_dev, const void* const* new_info_dev,
                          const pp_mesh* mesh, int nmaps, const int* const* v2v_dev,
                          double* const* scatter_w_dev, double rmax, int gnr, int gppr);
/* SellCSigma::setShuffling scs/SellCSigma.h:92 (default on, :236).  mode:
 *   0  never in place (setShuffling(false)): every rebuild is the full counting-sort re-layout;
 *   1  (default) the reference's decision (SCS_rebuild.h:33-42,160-189): the layout is kept iff every
 *      row's new count fits its chunk width; then only the particles that change element move
 *      (reshuffle, SCS_rebuild.h:4-119).  offsets / slice_to_chunk / row_to_element / element_to_row
 *      equal the reference's after every rebuild.  The decision is evaluated on the histogram the
 *      full re-layout needs anyway, so it costs nothing when the layout cannot be kept -- at 10^5 rows
 *      some row overflows its padding in nearly every step (measured: 600-2500 rows per pseudoXGCm
 *      step), so large structures practically always take the full re-layout, as the reference does;
 * Rows stay prefix-compact in every mode (a shrinking row back-fills its holes from its own tail). */
int pp_ps_set_shuffling(pp_ps* ps, int mode);
/* how the rebuilds of this structure ended so far: kept layout / full re-layout / full re-layouts whose first
 * pass read the records of the re-layout before it instead of the member arrays (back-to-back rebuilds of a
 * particle type wider than 64 B with no member access in between: performance_tests/ps_combo160.cpp:205-232) */
int pp_ps_rebuild_stats(const pp_ps* ps, long long* n_in_place, long long* n_full, long long* n_from_records);
/* What the structure is still holding back (DESIGN "Rebuild: the record-fed push").  A rebuild may leave members in
 * its staging records (lazy_rec 1 / 2 / 3), a member only logically zero (zero_pending = its storage index, else
 * -1; zero_z_pending: the third component of x_tgt after a 2-D record-fed push), slot -> element unwritten.
 * EVERY entry point that reads or exposes member data runs the pending passes first, so none of this is
 * observable through the API -- tests/test_gpu_state_machine.py holds every export to that, against a twin that
 * pp_ps_materialize brought up to date before the call. */
typedef struct pp_ps_deferred_t {
  int lazy_rec, zero_pending, zero_z_pending, elem_count_valid, slot_elem_valid, hot_row;
} pp_ps_deferred_t;
int pp_ps_deferred_state(const pp_ps* ps, pp_ps_deferred_t* out);
/* run every pending pass now (members to the SoA arrays, pending zeros, slot -> element, group -> chunk) */
int pp_ps_materialize(pp_ps* ps);
/* getPIDs ps_for.hpp:65-85: offsets_dev[ne+1], pids_dev[nPtcls] */
int pp_ps_get_pids(const pp_ps* ps, int* offsets_dev, int* pids_dev);
/* printMetrics SellCSigma.h:465-524 */
int pp_ps_metrics(const pp_ps* ps, int* padded_cells, int* padded_slices, int* empty_rows);
/* swap which member index two same-typed members refer to (O(1); used to ping-pong x / x_tgt
 * when a driver skips rebuild -- BASELINE config 2) */
int pp_ps_swap_members(pp_ps* ps, int m_a, int m_b);

/* ------------------------------------------------------------------ pushes */
/* ellipticalPush::setup test/ellipticalPush.hpp:10-33 (x: double[3], b/phi: float) */
int pp_elliptical_setup(pp_ps* ps, int m_x, int m_b, int m_phi, double h, double k, double d);
/* ellipticalPush::push test/ellipticalPush.hpp:36-70 */
int pp_elliptical_push(pp_ps* ps, const pp_mesh* mesh, int m_xtgt, int m_b, int m_phi, double h,
                       double k, double d, double deg);
/* 3-D tokamak restatement of the same push; see DESIGN.md */
int pp_toroidal_push(pp_ps* ps, const pp_mesh* mesh, int m_x, int m_xtgt, int m_b, int m_phi,
                     double h, double k, double d, double deg);
/* push lambda test/pseudoPushAndSearch.cpp:104-115 */
int pp_linear_push(pp_ps* ps, int m_x, int m_xtgt, double distance, double dx, double dy,
                   double dz);
/* pushBoris src/pumipic_push.hpp:17-75, launched over n (the reference launches over 1) */
int pp_push_boris(int n, double* x, double* y, double* z, double* xp, double* yp, double* zp,
                  double* vx, double* vy, double* vz, const double* ex, const double* ey,
                  const double* ez, const double* br, const double* bt, const double* bz,
                  double dt);
/* updatePtclPositions test/pseudoXGCm.cpp:102-114 */
int pp_update_positions(pp_ps* ps, int m_x, int m_xtgt);
/* redistribute_particles particle_structs/test/Distribute.h:28-89 with the uniform strategy: every
 * live particle draws a new element with probability percent_moved, masked slots get -1.  The draws
 * are a counter-based hash of (seed, slot) instead of the reference's Kokkos random pool (which is
 * not reproducible), so tests can check them against the oracle. */
int pp_redistribute_particles(const pp_ps* ps, double percent_moved, unsigned long long seed,
                              int* new_elems_dev);
/* the same with the re-draw following distribution strategy `strat` of distribute_particles
 * (particle_structs/test/Distribute.cpp:76-253, what ps_combo160 passes, :210): 1 uniform, 2 gaussian(ne/2,
 * ne/8) truncated and clamped, 3 the uniform -> exponential conversion (lambda 1), 4 the GITRm approximation
 * (85 % of the movers into the first 2/5 of the elements).  Strategy 0 (even) is an index rule, not a draw:
 * not served.  Draws are hashes of (seed, slot); normal variates are Irwin-Hall sums (exact in double). */
int pp_redistribute_particles_dist(const pp_ps* ps, int strat, double percent_moved, unsigned long long seed,
                                   int* new_elems_dev);
/* pseudoPush performance_tests/ps_combo160.cpp:158-178 (members double[17], int[4], long) */
int pp_pseudo_push160(pp_ps* ps, const double* parent_elm_data_dev);

/* ------------------------------------------------------------------ searches
 * elem_ids_dev: capacity ints, in/out.  The whole walk runs inside one kernel (no per-iteration
 * host round trip); `found` is the reference's return value. */
/* search_mesh_2d src/pumipic_adjacency.hpp:1011-1158 */
int pp_search_mesh_2d(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_pid,
                      int* elem_ids_dev, int looplimit, int* found);
/* search_mesh src/pumipic_adjacency.hpp:37-45 / adjacency.tpp:641-654 (2-D and 3-D; BCC when
 * requireIntersection==0, Moller-Trumbore / 2-D segment otherwise).  elem_ids_seeded==0 behaves
 * like an empty elem_ids (adjacency.tpp:504-515).  inter_* may be NULL when !requireIntersection */
int pp_search_mesh(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_pid,
                   int* elem_ids_dev, int elem_ids_seeded, int requireIntersection,
                   int* inter_faces_dev, double* inter_points_dev, int looplimit, int* found,
                   int* num_not_in_elem);
/* legacy 3-D search_mesh src/pumipic_adjacency.hpp:558-768 */
int pp_search_mesh_legacy3d(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_pid,
                            int* elem_ids_dev, int elem_ids_seeded, double* xpoints_dev,
                            int* xface_dev, int looplimit, int* found);
/* search_mesh_3d src/pumipic_adjacency.hpp:314-555 (tol 1e-20, neighbours in ask_dual order, wall
 * hits into xpoints/xface; both are written only for particles that hit an exposed face).
 * *found = 1 all found, 0 loop limit exceeded, -2 a particle was outside its parent element
 * (the reference aborts there, hpp:373-379).  The fallback of hpp:510 is not replicated (it
 * indexes the dual value array by a face id): the neighbour across that face is taken. */
int pp_search_mesh_3d(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_pid,
                      int* elem_ids_dev, int elem_ids_seeded, double* xpoints_dev, int* xface_dev,
                      int looplimit, int* found);
/* trace_particle_through_mesh with a caller-supplied functor (adjacency.tpp:460-615): the walk
 * kernel by kernel, so that any device code of the caller can run where the reference calls
 * `func` (between find_exit_face and set_new_element, tpp:561-565).  Per-slot work arrays of
 * `capacity` ints are the caller's: ptcl_done (tpp:486), last_exit (tpp:488).
 *   pp_trace_begin                      setInitial, finishUnmoved, initializeIntersection,
 *                                       check_initial_parents (tpp:503-552, 72-145)
 *   pp_trace_find_exit_face             find_exit_face (tpp:231-363), use_bcc = !requireIntersection
 *   pp_trace_check_model_intersection   the default functor RemoveParticleOnGeometricModelExit
 *                                       (tpp:365-387, 617-639)
 *   pp_trace_set_new_element            set_new_element (tpp:389-416) + the min reduction of
 *                                       ptcl_done as *num_unfinished (0 <=> found, tpp:567-571)
 *   pp_trace_not_found                  loop-limit clean-up (tpp:583-600)
 * pumi-pic_amd/include/pumipic_adjacency.hpp::trace_particle_through_mesh drives them in the
 * reference's order.  pp_search_mesh is the same walk with the default functor, fused. */
int pp_trace_begin(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int* elem_ids_dev,
                   int elem_ids_seeded, int requireIntersection, int* inter_faces_dev,
                   double* inter_points_dev, int* ptcl_done_dev, int* last_exit_dev,
                   int* num_not_in_elem);
int pp_trace_find_exit_face(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt,
                            const int* elem_ids_dev, int* ptcl_done_dev, int* last_exit_dev,
                            double* inter_points_dev, int use_bcc);
int pp_trace_check_model_intersection(const pp_mesh* mesh, pp_ps* ps, int* elem_ids_dev,
                                      int* ptcl_done_dev, const int* last_exit_dev,
                                      int requireIntersection, int* inter_faces_dev);
int pp_trace_set_new_element(const pp_mesh* mesh, pp_ps* ps, int* elem_ids_dev,
                             const int* ptcl_done_dev, const int* last_exit_dev, int* num_unfinished);
int pp_trace_not_found(pp_ps* ps, int* elem_ids_dev, const int* ptcl_done_dev, int* num_not_found);
/* Fused hot path: elliptical (dim 2) / toroidal (dim 3) push + BCC walk: particle state is read
 * once and x_tgt, phi, elem_ids written once.  Result-identical to pp_elliptical_push +
 * pp_search_mesh_2d (dim 2) and to pp_toroidal_push + pp_search_mesh in BCC mode (dim 3:
 * finishUnmoved, check_initial_parents, walk, tpp:72-145,460-615).  dim 2: the seeds search_mesh_2d
 * reads (-1 = own element, -nelems = outside); elem_ids_seeded == 0 = "every seed is -1" without the
 * caller filling the array (the loop that rebuilds after every search).  SCS structures use the
 * row-tiled kernels (dim 3: column loop + deferred walk of crossing particles, two launches,
 * library-owned queue of 32 B per tile slot); CSR structures the slot-parallel kernel.  found may be
 * NULL (no host sync).
 * After pp_ps_rebuild_commit / _scatter / a committing migration of a structure of the pseudoXGCm
 * particle type this call reads the re-layout's staging records directly (the pass that would copy them
 * into the member arrays is deferred: DESIGN.md "The record-fed push"); any other entry point that touches
 * member data runs that pass first, so callers see the same member contents either way. */
int pp_push_search(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xtgt, int m_b, int m_phi,
                   double h, double k, double d, double deg, int* elem_ids_dev,
                   int elem_ids_seeded, int looplimit, int* found);

/* check_initial_parents (adjacency.tpp:72-145) re-tests every step that a particle lies in the element
 * its walk starts from.  When the starting positions ARE the destinations the previous walk accepted
 * in exactly those elements -- the structure was rebuilt from the previous search's ids
 * (pp_ps_rebuild_commit and friends), or the ids are re-used as seeds after pp_ps_swap_members, and
 * nothing but removals (-1) was written to the ids in between -- the test recomputes the barycentric
 * coordinates the walk just accepted with the tighter tolerance 1e-10 (tpp:224,255 vs the area
 * tolerance >= 1e-8 of tpp:418-428) on bit-identical inputs, so it passes by construction.
 * pp_ps_set_origin_trust(ps, 1) lets the caller vouch for that: the fused 3-D push skips the test
 * (~15 % of its FP64 work).  The one path that accepts a destination without a containment test is
 * finishUnmoved (|dest - orig| < tol, tpp:527-536); pp_push_search_counters reports how many trusted
 * particles ended that way (0 in the pseudoXGCm flows: the slowest particle moves 1e-5 per push), so
 * the exactness of a trusted run is checkable after the fact.  Off by default. */
int pp_ps_set_origin_trust(pp_ps* ps, int on);
/* Elements visited, summed over all particles, by the last pp_search_mesh call that ran in intersection
 * mode on a tet mesh (the ray of adjacency.tpp:284-361 is followed to the domain boundary: the walk
 * length, not the particle count, is what that search costs).  One host sync; 0 when no such call ran. */
int pp_search_walk_steps(unsigned long long* steps);
/* counters of the last pp_push_search (one host sync): particles cut off by the loop limit, particles
 * that failed check_initial_parents, trusted particles that finished as unmoved */
int pp_push_search_counters(int* not_found, int* not_in_elem, int* unmoved_trusted);
/* The `bool found` of the most recent pp_push_search (the value every search of the reference returns and its
 * drivers assert, test/pseudoXGCm.cpp:153-154) for callers that passed found = NULL to keep the step free of a
 * host wait: the full re-layout of pp_ps_rebuild* / the migration that follows the search carries the search's
 * not-found count to the host WITH ITS OWN TOTALS (the one host wait a rebuild has anyway), and this call returns
 * it without touching the device.  When the last rebuild of `ps` did not carry it (in-place rebuild, CSR) the
 * structure's counters are read with one host sync.  The count is the STRUCTURE's (every Sell-C-sigma structure owns
 * its counter sets: searching other structures in between changes nothing); a CSR structure counts in a process-wide
 * set, and the call fails with PP_ESTATE when another structure was searched after it. */
int pp_ps_last_search_found(const pp_ps* ps, int* found);

/* ------------------------------------------------------------------ scatter / gather */
/* createGyroRingMappings test/gyroScatter.hpp:101-166 (maps: nverts*gnr*gppr*(dim+1) ints, device).
 * dim 3 is the documented tet variant: rings in the vertex's poloidal half-plane, the 4 vertices
 * of the containing tet per ring point. */
int pp_create_gyro_ring_mappings(const pp_mesh* mesh, double rmax, int gnr, int gppr,
                                 double theta_deg, int* forward_map_dev, int* backward_map_dev);
/* gyroScatter test/gyroScatter.hpp:168-229: scatter_w_dev[nverts] (overwritten).  A map written by
 * pp_create_gyro_ring_mappings is a constant of the run; the library keeps its transpose and runs
 * the second stage as a per-vertex gather (no FP64 atomics, additions in the reference's loop
 * order).  The transpose is dropped when the map's memory is freed or written through this API
 * (pp_free / pp_memcpy_h2d / pp_memset); a map edited by a user kernel in place must be passed as
 * a copy.  Any other map pointer takes the atomic form. */
int pp_gyro_scatter(const pp_mesh* mesh, const pp_ps* ps, const int* v2v_dev, double rmax, int gnr,
                    int gppr, double* scatter_w_dev);
/* A caller that edits a ring map in place with its own kernel / hipMemcpy (outside pp_memcpy_h2d /
 * pp_memset / pp_free, which drop the transpose themselves) tells the library so: the map then takes
 * the atomic form.  The two maps of one pp_create_gyro_ring_mappings call share one transpose (the
 * reference's projection is the identity, gyroScatter.hpp:125-134); forgetting either leaves the
 * other served.
 * THE CONTRACT IS CHECKED: pp_create_gyro_ring_mappings keeps a sampled content stamp of both maps (4096 entries
 * at a fixed stride); every 8th scatter through a transpose re-computes it on the device (one 256-thread block,
 * no host wait) and raises a host-mapped flag when it differs; the next pp_gyro_scatter / pp_ps_rebuild_scatter /
 * pp_ps_migrate* call that scatters then fails with PP_ESTATE and a message naming the cause -- a map rewritten
 * wholesale by a caller kernel (what test/pseudoXGCm_scatter.cpp:58-81 does on the host) cannot go unnoticed for
 * more than eight calls.  A single edited entry may escape the sample: the contract stands, the check is a net. */
int pp_gyro_map_forget(const int* map_dev);
/* gyroScatter with the particle radius the reference leaves as a TODO (test/gyroScatter.hpp:184
 * "ptclRadius = ringWidth*1.125; //TODO compute the radius") taken PER PARTICLE, and an optional
 * per-particle weight in place of the literal 1 (charge deposition).  Same ring selection (:186-191)
 * and the same two additions per element vertex (:193-200) as the reference; a particle whose upper
 * ring would be >= gnr (the reference asserts ringUp < gnr) contributes to its lower ring only and is
 * counted in *num_clipped (may be NULL: no host sync).  radius_dev / weight_dev are slot-indexed
 * arrays of `capacity` doubles (a double member's pointer qualifies); weight_dev == NULL means 1.
 * The contention the reference resolves with 2*(dim+1) FP64 atomics per particle is resolved per
 * row: ring sums of a row's run accumulate in registers and leave as one atomic per (element, ring)
 * touched, the vertices then gather their elements' sums.  Sums are no longer exact integers, so the
 * result depends on the summation order at the 1e-16 level (tests: <= 1e-12 relative vs the oracle).
 * pp_gyro_scatter stays the fast path for the reference's constant radius (no particle data read). */
int pp_gyro_scatter_radius(const pp_mesh* mesh, const pp_ps* ps, const double* radius_dev,
                           const double* weight_dev, const int* v2v_dev, double rmax, int gnr, int gppr,
                           double* scatter_w_dev, int* num_clipped);
/* setSyncArray of gyroSync test/gyroScatter.hpp:245-249: out[2v]=fwd[v], out[2v+1]=bkwd[v];
 * the SUM all-reduce itself (reduceCommArray, pumipic_comm.cpp:234-246) is RCCL on this buffer */
int pp_gyro_sync_pack(int nverts, const double* fwd_dev, const double* bkwd_dev, double* out_dev);
/* computeAvgPtclDensity test/pseudoPushAndSearch.cpp:340-374 */
int pp_avg_ptcl_density(const pp_mesh* mesh, const pp_ps* ps, double* elem_cnt_dev,
                        double* vert_density_dev);

/* ------------------------------------------------------------------ gather side (mesh/grid -> particle) */
/* findBCCoordsInTet + interpolateTetVtx / interpolate3dFieldTet, src/pumipic_adjacency.hpp:772-809,
 * for every live particle: bcc of the position member in its element (elem_ids_dev, or the parent
 * element when NULL), out_dev[c*capacity + pid] = sum_f bcc[f] * field_dev[vertex_opposite(f)*dof + c].
 * Dead slots and slots with element < 0 read 0.  *num_degenerate (may be NULL: no host sync) counts
 * particles whose tet has vol6 <= 1e-20 (the reference aborts there, :805). */
int pp_gather_tet_vtx(const pp_mesh* mesh, const pp_ps* ps, int m_x, const int* elem_ids_dev,
                      const double* field_dev, int dof, double* out_dev, int* num_degenerate);
/* interpolate2dField src/pumipic_utils.hpp:186-241 (regular (R|x, z) grid, one component) */
int pp_interp2d_field(const pp_ps* ps, int m_x, const double* data_dev, double gridx0, double gridz0,
                      double dx, double dz, int nx, int nz, int cyl_symm, int ncomp, int comp,
                      double* out_dev);
/* interp2dVector src/pumipic_utils.hpp:437-454: 3-component grid field, rotated to (x,y) when
 * cyl_symm; out_dev[c*capacity + pid] */
int pp_interp2d_vector(const pp_ps* ps, int m_x, const double* data3_dev, double gridx0, double gridz0,
                       double dx, double dz, int nx, int nz, int cyl_symm, double* out_dev);
/* interpolate3d_field src/pumipic_utils.hpp:375-418 (tri-linear, grid coordinate arrays) */
int pp_interp3d_field(const pp_ps* ps, int m_x, int nx, int ny, int nz, const double* gridx_dev,
                      const double* gridy_dev, const double* gridz_dev, const double* data_dev,
                      double* out_dev);
/* Gather + pushBoris in one pass over the live particles: E = interpolate3dFieldTet
 * of a 3-dof vertex field in the particle's element (elem_ids_dev, or the row element when NULL),
 * B = interp2dVector of a 3-component (R,Z) grid, then pushBoris src/pumipic_push.hpp:17-75 on the
 * members x (position), x_prev (previous position) and v (velocity), all double[3].  Value for value
 * equal to pp_gather_tet_vtx(dof 3) + pp_interp2d_vector + pp_push_boris on those members. */
int pp_boris_push_fields(const pp_mesh* mesh, pp_ps* ps, int m_x, int m_xprev, int m_v,
                         const int* elem_ids_dev, const double* efield_vtx_dev, const double* bgrid_dev,
                         double gridx0, double gridz0, double dx, double dz, int nx, int nz, int cyl_symm,
                         double dt, int* num_degenerate);
/* The same helpers (plus the _wgrid forms) are device-inline in pumi-pic_amd/include/pumipic_gather.hpp
 * for use inside user lambdas. */

/* ------------------------------------------------------------------ wall geometry */
/* closest_point_on_triangle (wnormal == 0) / closest_point_on_triangle_wnormal
 * src/pumipic_adjacency.hpp:824-1009 over n (triangle, point) pairs: tris_dev holds 9 doubles per
 * triangle at stride tri_stride doubles (0 = the same triangle for every point), pts_dev / out_dev
 * 3 doubles per point.  region_dev (may be NULL) receives the TriRegion; the plain form leaves it
 * untouched in its EDGEAB branch, as the reference does (hpp:951-958).  Device-inline forms for
 * user lambdas: pumi-pic_amd/include/pumipic_wall.hpp. */
int pp_closest_point_on_triangle(int n, const double* tris_dev, int tri_stride,
                                 const double* pts_dev, int wnormal, double* out_dev,
                                 int* region_dev);

/* ------------------------------------------------------------------ migration glue */
/* setUnsafeProcs src/pumipic_ptcl_ops.hpp:32-52 */
int pp_set_unsafe_procs(const pp_ps* ps, const int* elems_dev, const unsigned char* safe_dev,
                        const int* owners_dev, int comm_rank, int* new_elems_dev,
                        int* new_procs_dev);
/* PICpart safe zone and buffer by breadth-first layers, src/pumipic_part_construct.cpp:387-468.
 * bfsBufferLayers: starting from the elements this rank owns, every layer adds the elements that
 * share a bridge entity (bridge_dim 0 = vertex, dim-1 = side) with a visited one; is_safe_dev[e] =
 * reached within safe_layers, has_part_host[r] = 1 when an element owned by r is reached within
 * ghost_layers (the parts this rank buffers).  bfsSafeInward (buffer BFS + "full" safe method,
 * :439-468): safe = own elements + buffered elements at least safe_layers away from the unbuffered
 * region.  safe arrays are what pp_set_unsafe_procs takes. */
int pp_bfs_buffer_layers(const pp_mesh* mesh, int bridge_dim, int comm_rank, int comm_size,
                         int safe_layers, int ghost_layers, const int* owner_dev,
                         unsigned char* is_safe_dev, int* has_part_host);
int pp_bfs_safe_inward(const pp_mesh* mesh, int bridge_dim, int comm_rank, int comm_size,
                       int safe_layers, const int* owner_dev, const int* has_part_host,
                       unsigned char* safe_dev);
/* SellCSigma::migrate scs/SCS_migrate.h:29-137 send side: count per destination rank
 * (send_counts_host[nranks]) ... */
int pp_ps_migrate_count(const pp_ps* ps, const int* new_element_dev, const int* new_process_dev,
                        int comm_rank, int nranks, int* send_counts_host);
/* ... then pack (element gid int64 + every member) per destination, rank-major, into
 * send_gid_dev[total] and send_info_dev[m] = [ncomp][total]; marks sent particles' new_element
 * as -1 (removeSentParticles, SCS_migrate.h:189-196) */
int pp_ps_migrate_pack(const pp_ps* ps, int* new_element_dev, const int* new_process_dev,
                       int comm_rank, int nranks, const int* send_counts_host,
                       int64_t* send_gid_dev, void* const* send_info_dev);

/* One-buffer form of the same exchange: every leaving particle is packed as ONE record
 * [element gid int64 | all members as 32-bit words], padded to 16 B, rank-major.  A single
 * all-to-all-v moves everything; the receiver feeds the records to pp_ps_rebuild_records, which
 * maps gid -> lid (gid2lid_dev[ngids]; NULL: the structure's own table, built at construction from its element
 * gids as the reference's SCS does, SCS_migrate.h:181-187 -- the identity for gids 0..ne-1) and rebuilds with
 * the received particles as new particles (SCS_migrate.h:181-213). */
int pp_ps_migrate_record_bytes(const pp_ps* ps);
int pp_ps_migrate_pack_records(const pp_ps* ps, int* new_element_dev, const int* new_process_dev,
                               int comm_rank, int nranks, const int* send_counts_host,
                               void* send_records_dev);
int pp_ps_rebuild_records(pp_ps* ps, const int* new_element_dev, int n_recv,
                          const void* recv_records_dev, const int* gid2lid_dev, int64_t ngids);
/* The same two calls for a step that commits positions and scatters (pseudoXGCm.cpp:116-140,
 * 527-530) without separate passes: the records carry the particle AFTER updatePtclPositions
 * (member m_x is read from m_xtgt, m_xtgt travels as zeros), and the receiver's rebuild commits
 * its own particles, takes the received ones as they are and enqueues the nmaps gyroScatter
 * calls behind it (pp_ps_rebuild_scatter).  m_x = m_xtgt = -1 / nmaps = 0 switch either part off. */
int pp_ps_migrate_pack_records_commit(const pp_ps* ps, int m_x, int m_xtgt, int* new_element_dev,
                                      const int* new_process_dev, int comm_rank, int nranks,
                                      const int* send_counts_host, void* send_records_dev);
int pp_ps_rebuild_records_scatter(pp_ps* ps, int m_x, int m_xtgt, const int* new_element_dev, int n_recv,
                                  const void* recv_records_dev, const int* gid2lid_dev, int64_t ngids,
                                  const pp_mesh* mesh, int nmaps, const int* const* v2v_dev,
                                  double* const* scatter_w_dev, double rmax, int gnr, int gppr);

/* ------------------------------------------------------------------ communicators
 * One process per GPU.  The reference passes an MPI_Comm (Distributor, support/psDistributor.hpp:10-40;
 * Mesh::comm(), src/pumipic_comm.cpp:222-246) and stages every message through the host; here a
 * pp_comm carries one of four transports behind the same calls:
 *   rccl   device-resident collectives over xGMI (production): the library opens librccl at run time,
 *          rank 0 draws a 128-byte unique id (pp_comm_unique_id) that the launcher broadcasts
 *          (MPI_Bcast / torch.distributed / pp_bootstrap_broadcast) and every rank joins with
 *          pp_comm_create_rccl.  All collectives are enqueued on the library stream (pp_stream()).
 *   tcp    built-in host-staged exchange through rank 0 over sockets: the reference's host-staged MPI
 *          pattern, for boxes where RCCL cannot form the job (several ranks on one GPU, no xGMI).
 *   host   the caller supplies the collectives on HOST buffers (MPI in a PUMI-PIC build, gloo in the
 *          tests); the library stages device data through pinned memory.
 *   local  N virtual ranks inside one process (tests, debugging on one GPU): the ranks' calls are made
 *          one after the other, so migration is split-phase there (pp_ps_migrate_begin on every
 *          rank, then pp_ps_migrate_end on every rank).
 * pp_comm_create_env reads the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT;
 * PP_COMM=rccl|tcp, default rccl; PP_COMM_PORT overrides MASTER_PORT+1): WORLD_SIZE unset or 1 gives
 * a single-rank communicator whose collectives are no-ops. */
typedef struct pp_comm pp_comm;
typedef struct pp_comm_host_ops {
  /* PS_Comm_Ialltoall (SCS_migrate.h:48): one int to / from every rank */
  int (*alltoall_int)(void* user, const int* send, int* recv);
  /* the particle exchange (SCS_migrate.h:143-178 as one message per peer): byte counts / displacements
   * per rank */
  int (*alltoallv_bytes)(void* user, const void* send, const int64_t* send_bytes,
                         const int64_t* send_displ, void* recv, const int64_t* recv_bytes,
                         const int64_t* recv_displ);
  /* MPI_Allreduce(SUM) of doubles (pumipic_comm.cpp:243) and of longs (pseudoXGCm.cpp:508,523) */
  int (*allreduce_sum_f64)(void* user, double* buf, int64_t n);
  int (*allreduce_sum_i64)(void* user, int64_t* buf, int64_t n);
} pp_comm_host_ops;
int pp_comm_unique_id(void* id128_out);
pp_comm* pp_comm_create_rccl(const void* id128, int rank, int nranks);
pp_comm* pp_comm_create_tcp(const char* root_addr, int port, int rank, int nranks);
pp_comm* pp_comm_create_host(const pp_comm_host_ops* ops, void* user, int rank, int nranks);
int pp_comm_create_local(int nranks, pp_comm** comms_out);
pp_comm* pp_comm_create_env(void);
int pp_comm_rank(const pp_comm* c);
int pp_comm_size(const pp_comm* c);
const char* pp_comm_kind(const pp_comm* c); /* "self", "rccl", "tcp", "host", "local" */
int pp_comm_destroy(pp_comm* c);
/* rank 0 sends nbytes to every other rank over TCP (root_addr:port); used to hand out the RCCL id
 * when the launcher has no broadcast of its own */
int pp_bootstrap_broadcast(const char* root_addr, int port, int rank, int nranks, void* buf,
                           int nbytes);
/* host-side pieces of SellCSigma::migrate, usable without a GPU: the count exchange
 * (SCS_migrate.h:40-64) and the offsets both sides derive from the counts (:66-72, :129-133).
 * Displacements are in particles, rank-major, the own rank's count is 0. */
int pp_comm_exchange_counts(pp_comm* c, const int* send_counts_host, int* recv_counts_host);
int pp_migrate_plan(int nranks, int rank, const int* send_counts, const int* recv_counts,
                    int64_t* send_displ, int64_t* recv_displ, int64_t* n_send, int64_t* n_recv);
/* Mesh::reduceCommArray(dim, SUM_OP, array) for a fully buffered (replicated) mesh,
 * src/pumipic_comm.cpp:234-246: in-place SUM over ranks of n doubles in device memory */
int pp_allreduce_sum(pp_comm* c, double* buf_dev, int64_t n);
/* MPI_Allreduce(MPI_LONG, SUM) of a few host values (particle totals, test/pseudoXGCm.cpp:508,523) */
int pp_allreduce_sum_host_i64(pp_comm* c, int64_t* vals_host, int n);
int pp_comm_barrier(pp_comm* c);
/* Checked exchange through the calls a migration step makes (count exchange, ONE grouped send / receive
 * per peer pair, the gyroSync all-reduce) with known contents; PP_ESTATE + a message naming the rank, the
 * peer and the word when anything arrives wrong.  bench.py runs it before the timed loop of a multi-rank
 * job, so that a fabric / bootstrap problem is a failure in seconds (no reference counterpart: the
 * reference trusts MPI). */
int pp_comm_selftest(pp_comm* c, int nrec);
/* MPI_Allgather of nbytes host bytes per rank (timing summaries: SummarizeTimeAcrossProcesses,
 * support/ppTiming.cpp:220-300 reduces max / min / average over ranks) */
int pp_comm_allgather_host(pp_comm* c, const void* send_host, void* recv_host, int nbytes);

/* SellCSigma::migrate / CSR::migrate (scs/SCS_migrate.h:5-222, particle_structure.hpp:97-101):
 * particles with new_process != rank leave (counted, packed as records, exchanged, removed), the
 * received ones and the caller's n_new new particles enter the rebuild as new particles.  One rank:
 * plain rebuild (SCS_migrate.h:20-25).  new_element_dev is modified (sent particles read -1
 * afterwards, SCS_migrate.h:189-196). */
int pp_ps_migrate(pp_ps* ps, int* new_element_dev, const int* new_process_dev, pp_comm* comm);
/* The full form: + updatePtclPositions folded into the records and the rebuild (m_x, m_xtgt; -1 =
 * off), + the caller's new particles, + element gid -> lid table of the receiver (NULL = identity,
 * full-mesh replica), + the step's gyroScatter calls behind the rebuild (nmaps = 0: none).  See
 * pp_ps_migrate_pack_records_commit / pp_ps_rebuild_records_scatter for the pieces. */
int pp_ps_migrate_scatter(pp_ps* ps, int m_x, int m_xtgt, int* new_element_dev,
                          const int* new_process_dev, pp_comm* comm, int n_new,
                          const int* new_elems_dev, const void* const* new_info_dev,
                          const int* gid2lid_dev, int64_t ngids, const pp_mesh* mesh, int nmaps,
                          const int* const* v2v_dev, double* const* scatter_w_dev, double rmax,
                          int gnr, int gppr);
/* split-phase form of pp_ps_migrate_scatter: begin = count + exchange of counts + pack, end =
 * exchange + rebuild.  Required on a `local` communicator (every virtual rank begins before any
 * ends); on the others pp_ps_migrate_scatter is begin + end. */
int pp_ps_migrate_begin(pp_ps* ps, int m_x, int m_xtgt, int* new_element_dev,
                        const int* new_process_dev, pp_comm* comm, int n_new,
                        const int* new_elems_dev, const void* const* new_info_dev,
                        const int* gid2lid_dev, int64_t ngids, const pp_mesh* mesh, int nmaps,
                        const int* const* v2v_dev, double* const* scatter_w_dev, double rmax,
                        int gnr, int gppr);
int pp_ps_migrate_end(pp_ps* ps, pp_comm* comm, int* n_sent, int* n_received);
/* migrate_ptcls / migrate_lb_ptcls (src/pumipic_ptcl_ops.hpp:53-85) as ONE call: setUnsafeProcs
 * (:32-52: a particle whose new element is not safe on this part goes to that element's owner) is
 * the routing rule of the count / pack passes instead of a pass of its own that writes new_elems /
 * new_procs, then SellCSigma::migrate as pp_ps_migrate_scatter does it.  elem_ids_dev (the search's
 * result) is in/out: sent particles read -1 afterwards.  The particle balancer of migrate_lb_ptcls
 * (pumipic_lb.hpp:352-362) is not built.  _begin + pp_ps_migrate_end is the split-phase form. */
int pp_migrate_ptcls(pp_ps* ps, int m_x, int m_xtgt, int* elem_ids_dev, const unsigned char* safe_dev,
                     const int* owners_dev, pp_comm* comm, const pp_mesh* mesh, int nmaps,
                     const int* const* v2v_dev, double* const* scatter_w_dev, double rmax, int gnr, int gppr);
int pp_migrate_ptcls_begin(pp_ps* ps, int m_x, int m_xtgt, int* elem_ids_dev, const unsigned char* safe_dev,
                           const int* owners_dev, pp_comm* comm, const pp_mesh* mesh, int nmaps,
                           const int* const* v2v_dev, double* const* scatter_w_dev, double rmax, int gnr,
                           int gppr);

/* ------------------------------------------------------------------ PICparts and comm arrays
 * pumipic::Mesh (src/pumipic_mesh.hpp:9-131): one rank's part of a partitioned mesh -- the elements of the
 * core (owned) plus the whole parts reached by the buffer rule -- with the numberings and the exchange
 * plan that Mesh::reduceCommArray needs.
 *
 * pp_picpart_create = Mesh::Mesh(Input&) + constructPICPart + setupComm
 * (src/pumipic_part_construct.cpp:75-275, src/pumipic_comm.cpp:11-191).  `full` is the whole mesh, loaded
 * on every rank as the reference's drivers do; elem_owner_host[nelems] its partition vector (ownership
 * by classification: pp_owner_by_classification first).  buffer_method / safe_method = Input::Method
 * (src/pumipic_input.hpp:33-39: PP_PART_FULL, _BFS, _MINIMUM, _NONE; a NONE buffer is MINIMUM),
 * bridge_dim 0 (vertices) or dim-1 (sides), *_layers = bufferBFSLayers / safeBFSLayers (MINIMUM: 0).
 * Entity dimensions served: every dimension 0..dim, as the reference loops them (vertices, edges, faces of
 * tets, elements; the edges of a tet mesh are numbered by pp_mesh_num_edges / PP_MESH_EDGE2VERTS).  Vertices
 * and elements of a part are the kept ones in full-mesh order (:181-194); its sides and edges are numbered by
 * the part's own pp_mesh (PP_MESH_SIDE2VERTS, PP_MESH_EDGE2VERTS ...), PP_PART_FULL_IDS maps them to the full
 * mesh.  What travels between ranks is defined on full-mesh ids.
 *
 * What the reference exchanges at construction (MPI_Ialltoall of boundary sizes, Isend/Irecv of the
 * boundary lids, :113-190) is recomputed locally instead: every rank holds the full mesh and the
 * partition vector, so it runs the other ranks' buffer BFS itself (comm_size - 1 more sweeps on the
 * device) and needs no message.  Two things the reference leaves open are fixed: entities of a partially
 * buffered part are numbered in increasing part-local id (renumberBoundaryLids uses atomics, "the order
 * doesn't need to be consistent", :66-76), and the owner combines the fan-in contributions in increasing
 * rank (the reference: arrival order).
 *
 * `comm` is borrowed (must outlive the part); `full` is borrowed too when the buffer method is FULL
 * (pp_picpart_mesh returns it, as the reference points picpart at the mesh, :199-209). */
typedef struct pp_picpart pp_picpart;
enum { PP_PART_FULL = 0, PP_PART_BFS = 1, PP_PART_MINIMUM = 2, PP_PART_NONE = 3 };
enum { PP_OP_SUM = 0, PP_OP_MAX = 1, PP_OP_MIN = 2, PP_OP_BCAST = 3 }; /* Mesh::Op, pumipic_mesh.hpp:64-69 */
enum { PP_T_I32 = 0, PP_T_F64 = 1 };                                    /* INST(LO), INST(Real), pumipic_comm.cpp:443-447 */
/* setOwnerByClassification, src/pumipic_part_construct.cpp:278-302: owner[e] = class_owners[class_id[e]] */
int pp_owner_by_classification(const pp_mesh* full, const int* class_owners_host, int nclass, int comm_rank,
                               int* elem_owner_host_out);
pp_picpart* pp_picpart_create(const pp_mesh* full, const int* elem_owner_host, int buffer_method, int safe_method,
                              int bridge_dim, int buffer_layers, int safe_layers, pp_comm* comm);
int pp_picpart_destroy(pp_picpart* p);
const pp_mesh* pp_picpart_mesh(const pp_picpart* p); /* Mesh::mesh() */
/* isFullMesh, numBuffers(dim) (core parts held, self included), nents(0), nents(dim) */
int pp_picpart_info(const pp_picpart* p, int* is_full_mesh, int* num_buffers, int* nverts, int* nelems);
/* arrays of the part, entity dimension edim in {0, dim-1, dim}.  which:
 *   PP_PART_GIDS (int64, globalIds)  PP_PART_OWNERS (int, entOwners)  PP_PART_RANK_LIDS (int, rankLocalIndex)
 *   PP_PART_COMM_INDEX (int, commArrayIndex)  PP_PART_FULL_IDS (int: part entity -> full-mesh entity)
 *   PP_PART_ENT_IDS (int, sized by the FULL mesh: full entity -> part entity, -1 = not in the part)
 *   PP_PART_SAFE (unsigned char per element of the part, safeTag; edim ignored -- what pp_set_unsafe_procs takes) */
enum { PP_PART_GIDS = 0, PP_PART_OWNERS = 1, PP_PART_RANK_LIDS = 2, PP_PART_COMM_INDEX = 3, PP_PART_FULL_IDS = 4,
       PP_PART_ENT_IDS = 5, PP_PART_SAFE = 6 };
const void* pp_picpart_array_dev(const pp_picpart* p, int which, int edim, size_t* count);
int pp_picpart_array_to_host(const pp_picpart* p, int which, int edim, void* out_host);
/* nentsOffsets(edim): comm_size + 1 ints; bufferedRanks(edim): up to comm_size - 1 ranks, *n of them;
 * is_complete_part (2 complete, 1 partial, 0 absent) per rank */
int pp_picpart_nents_offsets(const pp_picpart* p, int edim, int* offsets_host);
int pp_picpart_buffered_ranks(const pp_picpart* p, int edim, int* ranks_host, int* n);
int pp_picpart_complete_parts(const pp_picpart* p, int edim, int* is_complete_host);
/* the owner's side of partially held parts: Mesh::boundary_parts / offset_bounded_per_dim / bounded_ent_ids
 * (src/pumipic_mesh.hpp:131-136, gathered in pumipic_comm.cpp:113-190) -- ranks that hold a boundary of this
 * rank's entities (ascending), prefix of their counts, and the rank-local ids they hold in sending order.
 * Null outputs are skipped (first call: sizes).  pp_picpart_num_global: Mesh::num_entites[edim]. */
int pp_picpart_bounded(const pp_picpart* p, int edim, int* n_boundaries, int* boundary_parts_host, int* offsets_host,
                       int* n_ids, int* ent_ids_host);
long long pp_picpart_num_global(const pp_picpart* p, int edim);
/* Mesh::reduceCommArray (src/pumipic_comm.cpp:249-440): array_dev holds nvals values per entity of the
 * part (entity-major, createCommArray's layout).  Fan-in: every part sends the segments of the parts it
 * buffers to their owners (device buffers; one grouped RCCL send/recv per peer); the owner combines its
 * own value and the contributions in increasing rank; fan-out: the owner returns its segment (the
 * boundary subset to parts that hold only a boundary).  PP_OP_BCAST skips the fan-in.  Full-mesh parts
 * take the same route (it is a reduce-scatter + all-gather by ownership, the volume of a ring all-reduce).
 * Local virtual ranks (pp_comm_create_local) call the three phases on every rank in turn. */
int pp_picpart_reduce(pp_picpart* p, int edim, int op, int dtype, int nvals, void* array_dev);
int pp_picpart_reduce_begin(pp_picpart* p, int edim, int op, int dtype, int nvals, void* array_dev);
int pp_picpart_reduce_mid(pp_picpart* p);
int pp_picpart_reduce_end(pp_picpart* p);

/* ------------------------------------------------------------------ particle load balancer
 * pumipic::ParticleBalancer (src/pumipic_lb.hpp:33-118, src/pumipic_lb.cpp).  pp_balancer_create =
 * ParticleBalancer(Mesh&) (pumipic_lb.cpp:23-84): the "sbar" of every element -- the set of parts on which
 * it is safe, its owner included -- as a 64-bit mask (bit r = part r; at most 64 ranks), derived from the
 * Input on every rank without a message; the sorted list of distinct masks is the same on every rank.
 * pp_balancer_sbar_ids_dev = getSbarIDs (index into that list per element of the part).
 *
 * pp_balancer_repartition = ParticleBalancer::repartition (pumipic_lb.hpp:352-362): addWeights (device
 * histogram of the particles staying here by the sbar of their new element, and of the particles already
 * leaving by destination), balance, selectParticles (rewrites new_procs_dev of the particles that move;
 * new_elems_dev are elements of the part, -1 = leaving the domain; particles pushed out of the safe zone
 * must already carry their owner, as in the reference).  `balance` is EnGPar's weight diffusion in the
 * reference (engpar::balanceWeights, scorec/EnGPar >= 1.1.0, not part of the reference tree); here one
 * all-gather of the weight rows and the same integer diffusion on every rank: a part above the average
 * sends step_factor x (its weight - the neighbour's) / (number of lighter neighbours) towards every
 * lighter neighbour through the sbars they share, only from what it holds at the start, until
 * max weight <= tol x average (tol 1.05 = 5 %).
 * pp_balancer_partition = ParticleBalancer::partition (:364-377): the same from a host array of particles
 * per element; new_procs_host (sum of the counts, element-major) receives the destinations.
 * Virtual ranks of one process: _begin on every rank, then _end on every rank. */
typedef struct pp_balancer pp_balancer;
pp_balancer* pp_balancer_create(pp_picpart* part);
int pp_balancer_destroy(pp_balancer* b);
int pp_balancer_num_sbars(const pp_balancer* b);
int pp_balancer_sbars(const pp_balancer* b, unsigned long long* masks_host);
const int* pp_balancer_sbar_ids_dev(const pp_balancer* b, size_t* n);
int pp_balancer_repartition(pp_balancer* b, const pp_ps* ps, double tol, const int* new_elems_dev, int* new_procs_dev,
                            double step_factor);
int pp_balancer_repartition_begin(pp_balancer* b, const pp_ps* ps, const int* new_elems_dev, int* new_procs_dev);
int pp_balancer_repartition_end(pp_balancer* b, double tol, double step_factor);
int pp_balancer_partition(pp_balancer* b, const int* ptcls_per_elem_host, double tol, double step_factor,
                          int* new_procs_host);
int pp_balancer_partition_begin(pp_balancer* b, const int* ptcls_per_elem_host);
int pp_balancer_partition_end(pp_balancer* b, double tol, double step_factor, int* new_procs_host);
/* the plan of the last call on this rank: *n entries (sbar index, target rank, particles), sbar-major;
 * weights_after_host (comm size values, may be NULL): every rank's weight once the plan is carried out */
int pp_balancer_last_plan(const pp_balancer* b, int* n, int* sbar_host, int* target_host, long long* amount_host,
                          long long* weights_after_host);

/* ------------------------------------------------------------------ tracing
 * Kokkos::Profiling::pushRegion / popRegion of the reference (e.g. adjacency.tpp:480,613,
 * SCS_rebuild.h:126,311) map to roctx ranges (rocprofv3 --marker-trace).  The library wraps its own
 * entry points (search, push, rebuild, migrate, scatter) when PP_ROCTX=1; these two let a driver add
 * its regions through the same switch. */
int pp_range_push(const char* name);
int pp_range_pop(void);

/* ray_intersects_triangle (segment == 0) / line_segment_intersects_triangle (segment == 1),
 * src/pumipic_adjacency.tpp:152-201, over n (triangle, origin, destination) triples: tris_dev holds 9
 * doubles per triangle at stride tri_stride doubles (0 = one triangle for all), orig / dest 3 doubles per
 * ray; flip_dev (per ray) or flip_all is the orientation flag of the face (utils.hpp:501-507).  hit_dev[i]
 * = the reference's return value; xpoint_dev (3 per ray) and dproj_closeness_param_dev (3 per ray) may
 * be NULL.  The walk (find_exit_face, tpp:340,357) uses the ray form, as the reference does; the
 * segment form is what a wall model calls when only hits inside the step count. */
int pp_ray_intersects_triangle(int n, const double* tris_dev, int tri_stride, const double* orig_dev,
                               const double* dest_dev, double tol, const int* flip_dev, int flip_all, int segment,
                               int* hit_dev, double* xpoint_dev, double* dproj_closeness_param_dev);

#ifdef __cplusplus
}
#endif
#endif
