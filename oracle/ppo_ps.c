/*
 * ppo_ps.c -- ORACLE (test infrastructure, NOT product code).
 *
 * Sell-C-sigma and CSR particle structures with Kokkos::Serial ordering semantics.
 * Follows particle_structs/src/scs/{SellCSigma.h,SCS_sort.h,SCS_buildFns.h,SCS_rebuild.h},
 * particle_structs/src/csr/{CSR.hpp,CSR_buildFns.hpp,CSR_rebuild.hpp},
 * particle_structs/src/ps_for.hpp:65-85 and support/psMemberType.h:72-112.
 *
 * Documented deviation: Kokkos' sort_by_key_thread (SCS_sort.h:36-47) is a bitonic network in
 * a third-party dependency that is not under /root/reference; its permutation of EQUAL-count
 * elements is not restated.  Ties are broken by ascending element id (stable).  No result of the
 * hot path depends on that order (everything is compared by the particle-id member).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "ppo.h"
#include <math.h>

static void* xcalloc(size_t n, size_t s) {
  void* p = calloc(n ? n : 1, s ? s : 1);
  if (!p) {
    fprintf(stderr, "ppo: out of memory\n");
    exit(EXIT_FAILURE);
  }
  return p;
}

/* ------------------------------------------------------------------ SCS layout pieces */
/* SCS_buildFns.h:3-16 */
static int choose_chunk_height(int maxC, const int* ppe, int n) {
  int cnt = 0;
  for (int i = 0; i < n; ++i) cnt += ppe[i] > 0;
  if (cnt == 0) return 1;
  if (cnt < maxC) return cnt;
  return maxC;
}

typedef struct {
  int key, idx;
} kv;
static int cmp_kv(const void* a, const void* b) {
  const kv* x = (const kv*)a;
  const kv* y = (const kv*)b;
  if (x->key != y->key) return (x->key < y->key) ? -1 : 1;
  return (x->idx < y->idx) ? -1 : (x->idx > y->idx);
}
/* SCS_sort.h:4-48 (non-CUDA branch): ascending by count inside windows of sigma elements */
static void sigma_sort(int* ptcls, int* index, int ne, const int* ppe, int sigma) {
  for (int i = 0; i < ne; ++i) {
    ptcls[i] = ppe[i];
    index[i] = i;
  }
  if (sigma > 1) {
    int mx = ne > 1 ? ne : 1;
    if (sigma > mx) sigma = mx;
    const int n_sigma = ne / sigma;
    kv* tmp = (kv*)xcalloc((size_t)ne, sizeof(kv));
    for (int w = 0; w < n_sigma; ++w) {
      const int start = w * sigma;
      const int end = (w == n_sigma - 1) ? ne : start + sigma;
      for (int i = start; i < end; ++i) {
        tmp[i].key = ptcls[i];
        tmp[i].idx = index[i];
      }
      qsort(tmp + start, (size_t)(end - start), sizeof(kv), cmp_kv);
      for (int i = start; i < end; ++i) {
        ptcls[i] = tmp[i].key;
        index[i] = tmp[i].idx;
      }
    }
    free(tmp);
  }
}

typedef struct {
  int C, nchunks, nslices, capacity, num_empty;
  int *chunk_widths, *row_to_element, *element_to_row, *offsets, *slice_to_chunk, *ptcls;
} scs_layout;

/* sigmaSort + constructChunks (SCS_buildFns.h:18-98) + constructOffsets (:115-153) */
static void build_layout(scs_layout* L, int C, int V, int sigma, int ne, const int* ppe,
                         int pad_strat, double shuffle_padding) {
  memset(L, 0, sizeof(*L));
  L->C = C;
  int* ptcls = (int*)xcalloc((size_t)ne, sizeof(int));
  int* index = (int*)xcalloc((size_t)ne, sizeof(int));
  sigma_sort(ptcls, index, ne, ppe, sigma);
  const int nchunks = ne / C + (ne % C != 0);
  L->nchunks = nchunks;
  L->chunk_widths = (int*)xcalloc((size_t)nchunks, sizeof(int));
  L->row_to_element = (int*)xcalloc((size_t)nchunks * C, sizeof(int));
  L->element_to_row = (int*)xcalloc((size_t)nchunks * C, sizeof(int));
  int empty = 0;
  for (int i = 0; i < ne; ++i) {
    const int element = index[i];
    L->row_to_element[i] = element;
    L->element_to_row[element] = i;
    empty += (ptcls[i] == 0);
  }
  for (int i = ne; i < nchunks * C; ++i) {
    L->row_to_element[i] = i;
    L->element_to_row[i] = i;
    empty += 1;
  }
  L->num_empty = empty;
  for (int c = 0; c < nchunks; ++c) {
    int width = 0;
    for (int r = 0; r < C; ++r) {
      const int row = c * C + r;
      if (row < ne && ptcls[row] > width) width = ptcls[row];
    }
    L->chunk_widths[c] = width;
  }
  if (shuffle_padding > 0) {
    int cw_sum = 0, cw_sum_count = 0;
    double cw_sum_inv = 0;
    for (int c = 0; c < nchunks; ++c) {
      cw_sum += L->chunk_widths[c];
      cw_sum_count += L->chunk_widths[c] > 0;
      if (L->chunk_widths[c] > 0) cw_sum_inv += 1.0 / L->chunk_widths[c];
    }
    if (cw_sum > 0) {
      const double cw_sum2 = cw_sum / cw_sum_inv * shuffle_padding;
      const int avg_pad = (int)(cw_sum * shuffle_padding / cw_sum_count);
      for (int c = 0; c < nchunks; ++c) {
        int* w = &L->chunk_widths[c];
        if (pad_strat == PPO_PAD_EVENLY) {
          if (*w > 0) *w += avg_pad;
        } else if (pad_strat == PPO_PAD_PROPORTIONALLY) {
          *w = (int)(*w + *w * shuffle_padding);
        } else if (pad_strat == PPO_PAD_INVERSELY) {
          if (*w != 0) *w = (int)(*w + cw_sum2 / *w);
        }
      }
    }
  }
  /* offsets */
  int nslices = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int w = L->chunk_widths[c];
    nslices += w / V + ((w % V) != 0);
  }
  L->nslices = nslices;
  L->offsets = (int*)xcalloc((size_t)nslices + 1, sizeof(int));
  L->slice_to_chunk = (int*)xcalloc((size_t)nslices, sizeof(int));
  int s = 0;
  const int nat_size = V * C;
  for (int c = 0; c < nchunks; ++c) {
    const int w = L->chunk_widths[c];
    const int ns = w / V + ((w % V) != 0);
    for (int j = 0; j < ns; ++j, ++s) {
      L->slice_to_chunk[s] = c;
      const int rem = w % V;
      const int val = rem + (rem == 0) * V;
      const int is_last = (j == ns - 1);
      const int size = (!is_last) * nat_size + is_last * val * C;
      L->offsets[s + 1] = L->offsets[s] + size;
    }
  }
  L->capacity = L->offsets[nslices];
  L->ptcls = ptcls;
  free(index);
}

/* first slot of each chunk (SCS_buildFns.h:160-176 chunk_starts) */
static int* chunk_starts_of(const scs_layout* L) {
  int* cs = (int*)xcalloc((size_t)L->nchunks, sizeof(int));
  for (int c = 0; c < L->nchunks; ++c) cs[c] = L->capacity;
  for (int s = L->nslices - 1; s >= 0; --s) cs[L->slice_to_chunk[s]] = L->offsets[s];
  return cs;
}

static size_t member_stride_bytes(const ppo_ps* ps, int m) { return (size_t)ps->member_bytes[m]; }

static void** alloc_members(const ppo_ps* ps, long alloc) {
  void** d = (void**)xcalloc((size_t)ps->nmembers, sizeof(void*));
  for (int m = 0; m < ps->nmembers; ++m)
    d[m] = xcalloc((size_t)alloc * ps->member_ncomp[m], member_stride_bytes(ps, m));
  return d;
}
static void free_members(const ppo_ps* ps, void** d) {
  if (!d) return;
  for (int m = 0; m < ps->nmembers; ++m) free(d[m]);
  free(d);
}
/* CopyViewToView for one member: dst[dst_idx] = src[src_idx] for every component */
static void copy_slot(const ppo_ps* ps, int m, void* dst, long dst_alloc, long dst_idx,
                      const void* src, long src_alloc, long src_idx) {
  const size_t b = member_stride_bytes(ps, m);
  for (int c = 0; c < ps->member_ncomp[m]; ++c)
    memcpy((char*)dst + ((size_t)c * dst_alloc + dst_idx) * b,
           (const char*)src + ((size_t)c * src_alloc + src_idx) * b, b);
}

static void set_members_meta(ppo_ps* ps, int nmembers, const int* member_bytes,
                             const int* member_ncomp) {
  ps->nmembers = nmembers;
  ps->member_bytes = (int*)xcalloc((size_t)nmembers, sizeof(int));
  ps->member_ncomp = (int*)xcalloc((size_t)nmembers, sizeof(int));
  memcpy(ps->member_bytes, member_bytes, sizeof(int) * (size_t)nmembers);
  memcpy(ps->member_ncomp, member_ncomp, sizeof(int) * (size_t)nmembers);
}

/* scs/SellCSigma.h:229-323 */
ppo_ps* ppo_scs_create(int C_max, int sigma, int V, int ne, int np, const int* ppe,
                       const long* gids, int pad_strat, double shuffle_padding,
                       double extra_padding, int nmembers, const int* member_bytes,
                       const int* member_ncomp, const int* particle_elements,
                       const void* const* particle_info) {
  ppo_ps* ps = (ppo_ps*)xcalloc(1, sizeof(ppo_ps));
  ps->kind = PPO_SCS;
  ps->num_elems = ne;
  ps->num_ptcls = np;
  ps->C_max = C_max;
  ps->V = V;
  ps->sigma = sigma;
  ps->pad_strat = pad_strat;
  ps->shuffle_padding = shuffle_padding;
  ps->extra_padding = extra_padding;
  ps->minimize_size = 0.8;
  ps->always_realloc = 0;
  ps->try_shuffling = 1;
  set_members_meta(ps, nmembers, member_bytes, member_ncomp);
  ps->C = choose_chunk_height(C_max, ppe, ne);
  scs_layout L;
  build_layout(&L, ps->C, V, sigma, ne, ppe, pad_strat, shuffle_padding);
  ps->num_chunks = L.nchunks;
  ps->num_rows = L.nchunks * ps->C;
  ps->num_slices = L.nslices;
  ps->capacity = L.capacity;
  ps->offsets = L.offsets;
  ps->slice_to_chunk = L.slice_to_chunk;
  ps->row_to_element = L.row_to_element;
  ps->element_to_row = L.element_to_row;
  ps->num_empty_elements = L.num_empty;
  if (gids) {
    ps->element_to_gid = (long*)xcalloc((size_t)ps->num_rows, sizeof(long));
    for (int i = 0; i < ne; ++i) ps->element_to_gid[i] = gids[i];
    for (int i = ne; i < ps->num_rows; ++i) ps->element_to_gid[i] = -1;
  }
  int cap = ps->capacity;
  ps->mask = (unsigned char*)xcalloc((size_t)cap, 1);
  if (extra_padding > 0) cap = (int)(cap * (1 + extra_padding));
  ps->alloc = cap;
  ps->swap_alloc = cap;
  ps->data = alloc_members(ps, ps->alloc);
  if (np > 0) {
    int* cs = chunk_starts_of(&L);
    /* setupParticleMask (SCS_buildFns.h:154-203) */
    for (int c = 0; c < L.nchunks; ++c)
      for (int r = 0; r < ps->C; ++r) {
        const int row = c * ps->C + r;
        const int elem = ps->row_to_element[row];
        for (int p = 0; p < L.chunk_widths[c]; ++p) {
          const int pid = cs[c] + r + p * ps->C;
          ps->mask[pid] = (elem < ne) ? (p < L.ptcls[row]) : 0;
        }
      }
    /* initSCSData (SCS_buildFns.h:205-232) */
    if (particle_elements && particle_info) {
      int* row_index = (int*)xcalloc((size_t)ps->num_rows, sizeof(int));
      for (int i = 0; i < ps->num_rows; ++i) row_index[i] = cs[i / ps->C] + i % ps->C;
      for (int i = 0; i < np; ++i) {
        const int row = ps->element_to_row[particle_elements[i]];
        const int idx = row_index[row];
        row_index[row] += ps->C;
        for (int m = 0; m < nmembers; ++m)
          copy_slot(ps, m, ps->data[m], ps->alloc, idx, particle_info[m], np, i);
      }
      free(row_index);
    }
    free(cs);
  }
  free(L.chunk_widths);
  free(L.ptcls);
  return ps;
}

/* csr/CSR.hpp:114-154, CSR_buildFns.hpp:60-103 */
ppo_ps* ppo_csr_create(int ne, int np, const int* ppe, const long* gids, double padding_amount,
                       int nmembers, const int* member_bytes, const int* member_ncomp,
                       const int* particle_elements, const void* const* particle_info) {
  ppo_ps* ps = (ppo_ps*)xcalloc(1, sizeof(ppo_ps));
  ps->kind = PPO_CSR;
  ps->num_elems = ne;
  ps->num_rows = ne;
  ps->num_ptcls = np;
  ps->always_realloc = 0;
  ps->minimize_size = 0.8;
  ps->padding_amount = padding_amount;
  set_members_meta(ps, nmembers, member_bytes, member_ncomp);
  ps->offsets = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  for (int e = 0; e < ne; ++e) ps->offsets[e + 1] = ps->offsets[e] + ppe[e];
  if (gids) {
    ps->element_to_gid = (long*)xcalloc((size_t)ne, sizeof(long));
    memcpy(ps->element_to_gid, gids, sizeof(long) * (size_t)ne);
  }
  ps->capacity = (int)(ps->offsets[ne] * padding_amount);
  ps->alloc = ps->capacity;
  ps->swap_alloc = ps->capacity;
  ps->data = alloc_members(ps, ps->alloc);
  if (particle_elements && particle_info && np > 0) {
    int* row_indices = (int*)xcalloc((size_t)ne + 1, sizeof(int));
    memcpy(row_indices, ps->offsets, sizeof(int) * ((size_t)ne + 1));
    for (int i = 0; i < np; ++i) {
      const int idx = row_indices[particle_elements[i]]++;
      for (int m = 0; m < nmembers; ++m)
        copy_slot(ps, m, ps->data[m], ps->alloc, idx, particle_info[m], np, i);
    }
    free(row_indices);
  }
  return ps;
}

void ppo_ps_destroy(ppo_ps* ps) {
  if (!ps) return;
  free_members(ps, ps->data);
  free(ps->offsets);
  free(ps->slice_to_chunk);
  free(ps->row_to_element);
  free(ps->element_to_row);
  free(ps->mask);
  free(ps->element_to_gid);
  free(ps->member_bytes);
  free(ps->member_ncomp);
  free(ps);
}

void* ppo_ps_member(ppo_ps* ps, int m) { return ps->data[m]; }
long ppo_ps_alloc(const ppo_ps* ps) { return ps->alloc; }

/* Iteration order of parallel_for: SellCSigma.h:526-558 / CSR.hpp:186-213.
 * visit(e, pid, mask, ctx) is called for every slot the reference functor would see. */
typedef void (*visit_fn)(int e, int pid, int mask, void* ctx);
static void ps_for(const ppo_ps* ps, visit_fn fn, void* ctx) {
  if (ps->num_ptcls == 0) return; /* SellCSigma.h:529, CSR.hpp:189 */
  if (ps->kind == PPO_SCS) {
    const int C = ps->C;
    for (int s = 0; s < ps->num_slices; ++s) {
      const int rowLen = (ps->offsets[s + 1] - ps->offsets[s]) / C;
      for (int r = 0; r < C; ++r) {
        const int row = ps->slice_to_chunk[s] * C + r;
        const int e = ps->row_to_element[row];
        const int start = ps->offsets[s] + r;
        for (int p = 0; p < rowLen; ++p) {
          const int pid = start + p * C;
          fn(e, pid, ps->mask[pid], ctx);
        }
      }
    }
  } else {
    for (int e = 0; e < ps->num_elems; ++e)
      for (int pid = ps->offsets[e]; pid < ps->offsets[e + 1]; ++pid)
        fn(e, pid, 1, ctx); /* CSR.hpp:203-207: mask effectively always true */
  }
}

typedef struct {
  int* slot_elem;
  unsigned char* slot_mask;
} slot_ctx;
static void slot_visit(int e, int pid, int mask, void* c) {
  slot_ctx* s = (slot_ctx*)c;
  s->slot_elem[pid] = e;
  s->slot_mask[pid] = (unsigned char)mask;
}
void ppo_ps_slot_info(const ppo_ps* ps, int* slot_elem, unsigned char* slot_mask) {
  for (int i = 0; i < ps->capacity; ++i) {
    slot_elem[i] = -1;
    slot_mask[i] = 0;
  }
  /* slot_info reports the layout even when num_ptcls==0 */
  ppo_ps tmp = *ps;
  tmp.num_ptcls = 1;
  slot_ctx c = {slot_elem, slot_mask};
  ps_for(&tmp, slot_visit, &c);
}

/* ------------------------------------------------------------------ rebuild */
typedef struct {
  const ppo_ps* ps;
  const int* new_element;
  int* a;
  int* b;
  int* c;
  int* d;
  int* e;
  int aux;
} rb_ctx;

static void count_new(int e, int pid, int mask, void* vc) {
  (void)e;
  rb_ctx* c = (rb_ctx*)vc;
  const int ne = c->new_element[pid];
  if (mask && ne != -1) c->a[ne]++;
}

/* reshuffle (SCS_rebuild.h:4-120) */
static void rs_count(int element_id, int pid, int mask, void* vc) {
  rb_ctx* c = (rb_ctx*)vc;
  ppo_ps* ps = (ppo_ps*)c->ps;
  const int new_elem = c->new_element[pid];
  const int row = ps->element_to_row[element_id];
  const int is_particle = mask && (new_elem != -1);
  const int is_moving = is_particle & (new_elem != element_id);
  if (is_moving && mask) c->a[ps->element_to_row[new_elem]]++;
  ps->mask[pid] = (unsigned char)is_particle;
  if (!is_particle) c->b[row]++;
}
static void rs_gather(int element_id, int pid, int mask, void* vc) {
  rb_ctx* c = (rb_ctx*)vc;
  const ppo_ps* ps = c->ps;
  const int new_elem = c->new_element[pid];
  const int is_moving = (new_elem != -1) & (new_elem != element_id) & mask;
  if (is_moving) {
    const int new_row = ps->element_to_row[new_elem];
    const int index = c->a[new_row]++; /* counting_offset_index */
    c->b[index] = pid;                 /* movingPtclIndices */
    c->c[index] = 1;                   /* isFromSCS */
  }
}
static void rs_holes(int element_id, int pid, int mask, void* vc) {
  rb_ctx* c = (rb_ctx*)vc;
  const ppo_ps* ps = c->ps;
  const int row = ps->element_to_row[element_id];
  if (!mask) {
    const int moving_index = c->a[row]++; /* offset_new_particles */
    const int max_index = c->b[row];      /* counting_offset_index */
    if (moving_index < max_index) c->c[moving_index] = pid; /* holes */
  }
}

static int scs_reshuffle(ppo_ps* ps, const int* new_element, int n_new,
                         const int* new_particle_elements, const void* const* new_particles) {
  const int nr = ps->num_rows;
  int* new_per_row = (int*)xcalloc((size_t)nr + 1, sizeof(int));
  int* holes_per_row = (int*)xcalloc((size_t)nr, sizeof(int));
  rb_ctx c = {ps, new_element, new_per_row, holes_per_row, NULL, NULL, NULL, 0};
  ps_for(ps, rs_count, &c);
  for (int i = 0; i < n_new; ++i) new_per_row[ps->element_to_row[new_particle_elements[i]]]++;
  int fail = 0;
  for (int i = 0; i < nr; ++i)
    if (new_per_row[i] > holes_per_row[i]) fail = 1;
  if (fail) {
    free(new_per_row);
    free(holes_per_row);
    return 0;
  }
  int* offset_new = (int*)xcalloc((size_t)nr + 1, sizeof(int));
  int* counting = (int*)xcalloc((size_t)nr + 1, sizeof(int));
  for (int i = 0; i < nr; ++i) offset_new[i + 1] = offset_new[i] + new_per_row[i];
  memcpy(counting, offset_new, sizeof(int) * ((size_t)nr + 1));
  const int num_moving = offset_new[nr];
  if (num_moving == 0) {
    int cnt = 0;
    for (int i = 0; i < ps->capacity; ++i) cnt += ps->mask[i];
    ps->num_ptcls = cnt;
    free(new_per_row);
    free(holes_per_row);
    free(offset_new);
    free(counting);
    return 1;
  }
  int* moving = (int*)xcalloc((size_t)num_moving, sizeof(int));
  int* fromSCS = (int*)xcalloc((size_t)num_moving, sizeof(int));
  rb_ctx g = {ps, new_element, counting, moving, fromSCS, NULL, NULL, 0};
  ps_for(ps, rs_gather, &g);
  for (int i = 0; i < n_new; ++i) {
    const int new_row = ps->element_to_row[new_particle_elements[i]];
    const int index = counting[new_row]++;
    moving[index] = i;
    fromSCS[index] = 0;
  }
  int* holes = (int*)xcalloc((size_t)num_moving, sizeof(int));
  rb_ctx h = {ps, new_element, offset_new, counting, holes, NULL, NULL, 0};
  ps_for(ps, rs_holes, &h);
  for (int i = 0; i < num_moving; ++i) {
    if (fromSCS[i] == 1) ps->mask[moving[i]] = 0;
    ps->mask[holes[i]] = 1;
  }
  /* ShuffleParticles (MemberTypeLibraries.h:217-253): one pass per member */
  for (int m = 0; m < ps->nmembers; ++m)
    for (int i = 0; i < num_moving; ++i) {
      if (fromSCS[i] == 1)
        copy_slot(ps, m, ps->data[m], ps->alloc, holes[i], ps->data[m], ps->alloc, moving[i]);
      else
        copy_slot(ps, m, ps->data[m], ps->alloc, holes[i], new_particles[m], n_new, moving[i]);
    }
  int cnt = 0;
  for (int i = 0; i < ps->capacity; ++i) cnt += ps->mask[i];
  ps->num_ptcls = cnt;
  free(new_per_row);
  free(holes_per_row);
  free(offset_new);
  free(counting);
  free(moving);
  free(fromSCS);
  free(holes);
  return 1;
}

typedef struct {
  const ppo_ps* ps;
  const int* new_element;
  const int* new_element_to_row;
  int* element_index;
  int* new_indices;
  unsigned char* new_mask;
  int new_C;
} copy_ctx;
static void copy_scs_visit(int e, int pid, int mask, void* vc) {
  (void)e;
  copy_ctx* c = (copy_ctx*)vc;
  const int new_elem = c->new_element[pid];
  if (mask && new_elem != -1) {
    const int new_row = c->new_element_to_row[new_elem];
    const int idx = c->element_index[new_row];
    c->element_index[new_row] += c->new_C;
    c->new_indices[pid] = idx;
    c->new_mask[idx] = 1;
  }
}
typedef struct {
  const ppo_ps* ps;
  const int* new_element;
  const int* new_indices;
  void** dst;
  long dst_alloc;
  int m;
} pstops_ctx;
static void pstops_visit(int e, int pid, int mask, void* vc) {
  (void)e;
  pstops_ctx* c = (pstops_ctx*)vc;
  if (mask && c->new_element[pid] != -1)
    copy_slot(c->ps, c->m, c->dst[c->m], c->dst_alloc, c->new_indices[pid], c->ps->data[c->m],
              c->ps->alloc, pid);
}
static void reset_mask_visit(int e, int pid, int mask, void* vc) {
  (void)e;
  (void)mask;
  ((ppo_ps*)vc)->mask[pid] = 0;
}

/* SCS_rebuild.h:122-314 */
static void scs_rebuild(ppo_ps* ps, const int* new_element, int n_new,
                        const int* new_particle_elements, const void* const* new_particles) {
  ps->last_rebuild_was_shuffle = 0;
  int* new_ppe = (int*)xcalloc((size_t)ps->num_rows, sizeof(int));
  rb_ctx c = {ps, new_element, new_ppe, NULL, NULL, NULL, NULL, 0};
  ps_for(ps, count_new, &c);
  for (int i = 0; i < n_new; ++i)
    if (new_particle_elements[i] == -1) {
      fprintf(stderr,
              "there are new particles being added that are marked"
              "as inactive (element id set to -1)\n");
      exit(EXIT_FAILURE);
    }
  for (int i = 0; i < n_new; ++i) new_ppe[new_particle_elements[i]]++;
  int active = 0;
  for (int i = 0; i < ps->num_rows; ++i) active += new_ppe[i];
  if (active == 0) {
    ps->num_ptcls = 0;
    /* resetMask goes through parallel_for, which is a no-op when nPtcls()==0 -- num_ptcls was
     * just zeroed (SCS_rebuild.h:169-176), so the mask is left untouched, as in the reference */
    ps_for(ps, reset_mask_visit, ps);
    free(new_ppe);
    return;
  }
  if (ps->try_shuffling &&
      scs_reshuffle(ps, new_element, n_new, new_particle_elements, new_particles)) {
    ps->last_rebuild_was_shuffle = 1;
    free(new_ppe);
    return;
  }
  const int new_C = choose_chunk_height(ps->C_max, new_ppe, ps->num_rows);
  scs_layout L;
  build_layout(&L, new_C, ps->V, ps->sigma, ps->num_elems, new_ppe, ps->pad_strat,
               ps->shuffle_padding);
  const int new_capacity = L.capacity;
  unsigned char* new_mask = (unsigned char*)xcalloc((size_t)new_capacity, 1);
  void** swap = NULL;
  long swap_alloc = ps->swap_alloc;
  if (ps->always_realloc || swap_alloc < new_capacity ||
      swap_alloc * ps->minimize_size < new_capacity) {
    swap_alloc = (long)(new_capacity * (1 + ps->extra_padding));
  }
  swap = alloc_members(ps, swap_alloc);
  /* element_index: first slot of every new row (SCS_rebuild.h:233-249) */
  int* cs = chunk_starts_of(&L);
  int* element_index = (int*)xcalloc((size_t)L.nchunks * new_C, sizeof(int));
  for (int ch = 0; ch < L.nchunks; ++ch)
    for (int e = 0; e < new_C; ++e)
      element_index[ch * new_C + e] = (L.chunk_widths[ch] > 0) ? cs[ch] + e : 0;
  int* new_indices = (int*)xcalloc((size_t)ps->capacity, sizeof(int));
  copy_ctx cc = {ps, new_element, L.element_to_row, element_index, new_indices, new_mask, new_C};
  ps_for(ps, copy_scs_visit, &cc);
  for (int m = 0; m < ps->nmembers; ++m) {
    pstops_ctx pc = {ps, new_element, new_indices, swap, swap_alloc, m};
    ps_for(ps, pstops_visit, &pc);
  }
  for (int i = 0; i < n_new; ++i) {
    const int new_row = L.element_to_row[new_particle_elements[i]];
    const int idx = element_index[new_row];
    element_index[new_row] += new_C;
    new_mask[idx] = 1;
    for (int m = 0; m < ps->nmembers; ++m)
      copy_slot(ps, m, swap[m], swap_alloc, idx, new_particles[m], n_new, i);
  }
  /* swap in */
  free(ps->offsets);
  free(ps->slice_to_chunk);
  free(ps->row_to_element);
  free(ps->element_to_row);
  free(ps->mask);
  long old_alloc = ps->alloc;
  free_members(ps, ps->data);
  ps->data = swap;
  ps->swap_alloc = old_alloc;
  ps->alloc = swap_alloc;
  ps->C = new_C;
  ps->num_ptcls = active;
  ps->num_chunks = L.nchunks;
  ps->num_slices = L.nslices;
  ps->capacity = new_capacity;
  ps->num_rows = L.nchunks * new_C;
  ps->row_to_element = L.row_to_element;
  ps->element_to_row = L.element_to_row;
  ps->offsets = L.offsets;
  ps->slice_to_chunk = L.slice_to_chunk;
  ps->mask = new_mask;
  ps->num_empty_elements = L.num_empty;
  if (ps->element_to_gid) {
    /* element_to_gid is indexed by element (SCS_buildFns.h:100-113); rows beyond ne are -1 */
    long* g = (long*)xcalloc((size_t)ps->num_rows, sizeof(long));
    for (int i = 0; i < ps->num_rows; ++i) g[i] = (i < ps->num_elems) ? ps->element_to_gid[i] : -1;
    free(ps->element_to_gid);
    ps->element_to_gid = g;
  }
  free(L.chunk_widths);
  free(L.ptcls);
  free(cs);
  free(element_index);
  free(new_indices);
  free(new_ppe);
}

/* CSR_rebuild.hpp:18-118 */
static void csr_rebuild(ppo_ps* ps, const int* new_element, int n_new,
                        const int* new_particle_elements, const void* const* new_particles) {
  const int ne = ps->num_elems;
  int* ppe = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  int num_removed = 0;
  if (ps->num_ptcls > 0)
    for (int pid = 0; pid < ps->offsets[ne]; ++pid) {
      if (new_element[pid] > -1)
        ppe[new_element[pid]]++;
      else
        num_removed++;
    }
  for (int i = 0; i < n_new; ++i) ppe[new_particle_elements[i]]++;
  int* offsets_new = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  for (int e = 0; e < ne; ++e) offsets_new[e + 1] = offsets_new[e] + ppe[e];
  int* row_indices = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  memcpy(row_indices, offsets_new, sizeof(int) * ((size_t)ne + 1));
  const int nold = (ps->num_ptcls > 0) ? ps->offsets[ne] : 0;
  int* new_indices = (int*)xcalloc((size_t)(nold > 0 ? nold : 1), sizeof(int));
  for (int pid = 0; pid < nold; ++pid) {
    const int new_elem = new_element[pid];
    new_indices[pid] = (new_elem != -1) ? row_indices[new_elem]++ : -1;
  }
  const int on_process = ps->num_ptcls - num_removed + n_new;
  long swap_alloc = ps->swap_alloc;
  if (ps->always_realloc || on_process > swap_alloc)
    swap_alloc = (long)(ps->padding_amount * on_process);
  else if (on_process < ps->minimize_size * swap_alloc)
    swap_alloc = (long)(ps->padding_amount * on_process);
  void** swap = alloc_members(ps, swap_alloc);
  for (int m = 0; m < ps->nmembers; ++m)
    for (int pid = 0; pid < nold; ++pid)
      if (new_element[pid] != -1)
        copy_slot(ps, m, swap[m], swap_alloc, new_indices[pid], ps->data[m], ps->alloc, pid);
  for (int i = 0; i < n_new; ++i) {
    const int idx = row_indices[new_particle_elements[i]]++;
    if (new_particles)
      for (int m = 0; m < ps->nmembers; ++m)
        copy_slot(ps, m, swap[m], swap_alloc, idx, new_particles[m], n_new, i);
  }
  long old_alloc = ps->alloc;
  free_members(ps, ps->data);
  ps->data = swap;
  ps->alloc = swap_alloc;
  ps->swap_alloc = old_alloc;
  ps->capacity = (int)swap_alloc;
  ps->num_ptcls = on_process;
  free(ps->offsets);
  ps->offsets = offsets_new;
  free(ppe);
  free(row_indices);
  free(new_indices);
}

void ppo_ps_rebuild(ppo_ps* ps, const int* new_element, int n_new, const int* new_particle_elements,
                    const void* const* new_particle_info) {
  if (ps->kind == PPO_SCS)
    scs_rebuild(ps, new_element, n_new, new_particle_elements, new_particle_info);
  else
    csr_rebuild(ps, new_element, n_new, new_particle_elements, new_particle_info);
}

/* ps_for.hpp:65-85 */
typedef struct {
  int* ppe;
  int* offsets;
  int* cur;
  int* pids;
} pid_ctx;
static void pid_count(int e, int pid, int mask, void* vc) {
  (void)pid;
  if (mask) ((pid_ctx*)vc)->ppe[e]++;
}
static void pid_set(int e, int pid, int mask, void* vc) {
  pid_ctx* c = (pid_ctx*)vc;
  if (mask) c->pids[c->offsets[e] + c->cur[e]++] = pid;
}
void ppo_ps_get_pids(const ppo_ps* ps, int* offsets_out, int* pids_out) {
  const int ne = ps->num_elems;
  int* ppe = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  int* cur = (int*)xcalloc((size_t)ne + 1, sizeof(int));
  pid_ctx c = {ppe, offsets_out, cur, pids_out};
  ps_for(ps, pid_count, &c);
  offsets_out[0] = 0;
  for (int e = 0; e < ne; ++e) offsets_out[e + 1] = offsets_out[e] + ppe[e];
  ps_for(ps, pid_set, &c);
  free(ppe);
  free(cur);
}

/* redistribute_particles particle_structs/test/Distribute.h:28-89, uniform re-draws from a
 * splitmix64 hash of (seed, slot) (the reference's Kokkos pool is not reproducible) */
static unsigned long long splitmix64(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
void ppo_redistribute_particles(const ppo_ps* ps, double percent_moved, unsigned long long seed,
                                int* new_elems) {
  const int cap = ps->capacity;
  int* slot_elem = (int*)xcalloc((size_t)cap + 1, sizeof(int));
  unsigned char* slot_mask = (unsigned char*)xcalloc((size_t)cap + 1, 1);
  ppo_ps_slot_info(ps, slot_elem, slot_mask);
  for (int pid = 0; pid < cap; ++pid) {
    if (slot_elem[pid] < 0 || !slot_mask[pid]) {
      new_elems[pid] = -1;
      continue;
    }
    const unsigned long long h0 = splitmix64(seed ^ (2ull * (unsigned long long)pid));
    const double prob = (double)(h0 >> 11) * (1.0 / 9007199254740992.0);
    if (prob <= percent_moved)
      new_elems[pid] = (int)(splitmix64(seed ^ (2ull * (unsigned long long)pid + 1ull)) %
                             (unsigned long long)ps->num_elems);
    else
      new_elems[pid] = slot_elem[pid];
  }
  free(slot_elem);
  free(slot_mask);
}

/* redistribute_particles with the re-draw following a distribution strategy (Distribute.h:28-89 calls
 * distribute_particles(numElems, total, strat, ...), particle_structs/test/Distribute.cpp:60-253, device
 * forms): 1 uniform, 2 gaussian(ne/2, ne/8) truncated to int and clamped (:106-121), 3 the uniform ->
 * exponential conversion (:155-199, lambda = 1), 4 the GITRm approximation (85 % into the first 2/5 of the
 * elements, :230-253).  The reference draws from a Kokkos pool; here every draw is a hash of (seed, slot):
 *   h1 = splitmix64(seed ^ (2 slot + 1)),  g_j = splitmix64(h1 + j)
 * and a normal variate is the Irwin-Hall sum of twelve 32-bit uniforms (sum / 2^32 - 6: exact in double), so
 * that host and device produce the same ids.  Strategy 3's two logarithms per element are evaluated ONCE on
 * the host into the tables exp_start / exp_end (ne ints each) -- the same tables feed the device. */
void ppo_exponential_tables(int ne, int* exp_start, int* exp_end) {
  const double lambda = 1.0f;
  const double freq_max = log(1.0 / ne) * -1;
  for (int uni = 0; uni < ne; ++uni) {
    const double percent_elem = ((double)uni) / ne;
    const double temp = -1 / lambda * log(1 - percent_elem) / freq_max;
    const double temp_next = -1 / lambda * log(1 - percent_elem - 1.0 / ne) / freq_max;
    const double a = temp * ne, b = temp_next * ne;
    /* (the last element's end is log(0) = inf, and the reference never uses it: uni == ne-1 -> element 0) */
    exp_start[uni] = (a >= 0 && a < 2147483647.0) ? (int)a : 2147483647;
    exp_end[uni] = (b >= 0 && b < 2147483647.0) ? (int)b : 2147483647;
  }
}
int ppo_draw_element(int strat, int ne, unsigned long long h1, const int* exp_start, const int* exp_end) {
  if (strat == 2) {
    double S = 0;
    for (int j = 0; j < 12; ++j) S += (double)(splitmix64(h1 + (unsigned long long)j) >> 32);
    const double z = S * (1.0 / 4294967296.0) - 6.0;
    const double v = ne / 2.0 + (ne / 8.0) * z;
    int elem = (int)v;
    if (elem < 0) elem = 0;
    if (elem >= ne) elem = ne - 1;
    return elem;
  }
  if (strat == 3) {
    const int uni = (int)(h1 % (unsigned long long)ne);
    if (uni == ne - 1) return 0;
    const int start = exp_start[uni];
    const long long length = (long long)exp_end[uni] - start;
    int inside = 0;
    if (length > 1) inside = (int)(splitmix64(h1 + 1ull) % (unsigned long long)length);
    long long e = (long long)start + inside;
    if (e >= ne) e = (long long)(splitmix64(h1 + 2ull) % (unsigned long long)ne);
    return (int)e;
  }
  if (strat == 4) {
    const int cutoff = 2 * ne / 5;
    const double u = (double)(splitmix64(h1 + 1ull) >> 11) * (1.0 / 9007199254740992.0);
    const unsigned long long g = splitmix64(h1 + 2ull);
    if (u < 0.85 && cutoff > 0) return (int)(g % (unsigned long long)cutoff);
    return cutoff + (int)(g % (unsigned long long)(ne - cutoff));
  }
  return (int)(h1 % (unsigned long long)ne);
}
void ppo_redistribute_particles_dist(const ppo_ps* ps, int strat, double percent_moved, unsigned long long seed,
                                     int* new_elems) {
  const int cap = ps->capacity, ne = ps->num_elems;
  int* slot_elem = (int*)xcalloc((size_t)cap + 1, sizeof(int));
  unsigned char* slot_mask = (unsigned char*)xcalloc((size_t)cap + 1, 1);
  int *es = NULL, *ee = NULL;
  if (strat == 3) {
    es = (int*)xcalloc((size_t)ne + 1, sizeof(int));
    ee = (int*)xcalloc((size_t)ne + 1, sizeof(int));
    ppo_exponential_tables(ne, es, ee);
  }
  ppo_ps_slot_info(ps, slot_elem, slot_mask);
  for (int pid = 0; pid < cap; ++pid) {
    if (slot_elem[pid] < 0 || !slot_mask[pid]) {
      new_elems[pid] = -1;
      continue;
    }
    const unsigned long long h0 = splitmix64(seed ^ (2ull * (unsigned long long)pid));
    const double prob = (double)(h0 >> 11) * (1.0 / 9007199254740992.0);
    if (prob <= percent_moved)
      new_elems[pid] = ppo_draw_element(strat, ne, splitmix64(seed ^ (2ull * (unsigned long long)pid + 1ull)), es, ee);
    else
      new_elems[pid] = slot_elem[pid];
  }
  free(slot_elem);
  free(slot_mask);
  free(es);
  free(ee);
}

/* SellCSigma.h:465-524 */
void ppo_scs_metrics(const ppo_ps* ps, int* padded_cells, int* padded_slices, int* empty_rows) {
  int pc = 0, psl = 0;
  const int C = ps->C;
  for (int s = 0; s < ps->num_slices; ++s) {
    const int rowLen = (ps->offsets[s + 1] - ps->offsets[s]) / C;
    int np_slice = 0;
    for (int r = 0; r < C; ++r)
      for (int p = 0; p < rowLen; ++p) np_slice += !ps->mask[ps->offsets[s] + r + p * C];
    pc += np_slice;
    psl += np_slice > 0;
  }
  *padded_cells = pc;
  *padded_slices = psl;
  *empty_rows = ps->num_empty_elements;
}

/* src/pumipic_ptcl_ops.hpp:32-52 */
typedef struct {
  const int* elems;
  const unsigned char* safe;
  const int* owners;
  int rank;
  int* new_elems;
  int* new_procs;
} unsafe_ctx;
static void unsafe_visit(int e, int ptcl, int mask, void* vc) {
  (void)e;
  unsafe_ctx* c = (unsafe_ctx*)vc;
  c->new_procs[ptcl] = c->rank;
  const int nelm = c->elems[ptcl];
  c->new_elems[ptcl] = nelm;
  if (mask && nelm != -1) {
    if (!c->safe[nelm]) c->new_procs[ptcl] = c->owners[nelm];
  }
}
void ppo_set_unsafe_procs(const ppo_ps* ps, const int* elems, const unsigned char* safe,
                          const int* owners, int comm_rank, int* new_elems, int* new_procs) {
  unsafe_ctx c = {elems, safe, owners, comm_rank, new_elems, new_procs};
  ps_for(ps, unsafe_visit, &c);
}
