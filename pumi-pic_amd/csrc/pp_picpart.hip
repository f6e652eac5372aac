// PICparts and comm arrays: pumipic::Mesh's construction (src/pumipic_part_construct.cpp:75-275), its
// exchange plan (Mesh::setupComm, src/pumipic_comm.cpp:11-191) and Mesh::reduceCommArray (:249-440).
//
// Construction is set-up work and runs on the host over the mesh's host tables (like pp_mesh_create),
// except for the breadth-first sweeps, which are the device kernels of pp_bfs_buffer_layers /
// pp_bfs_safe_inward.  Nothing is exchanged at construction: every rank holds the full mesh and the
// partition vector, so it derives what each other rank buffers by running that rank's BFS itself.
// The reduction is device work: pack into owner-major segments, one exchange of device buffers to the
// owners (fan-in), a gather-form combine in rank order (no atomics), one exchange back (fan-out), unpack.
#include <algorithm>
#include <array>
#include <cstring>
#include <numeric>

#include "pp_internal.hpp"

using pp::grid_for;
using pp::kBlock;

namespace {

struct DimData {
  int edim = 0, nfull = 0, nents = 0, my_count = 0;
  std::vector<int> goff;                           // global entities per rank, exclusive scan (P + 1)
  std::vector<int> ent_ids, full_ids;              // full -> part (-1), part -> full
  std::vector<int> owners, rank_lids, comm_index;  // per entity of the part
  std::vector<int64_t> gids;
  std::vector<int> poff, is_complete, buffered;  // nentsOffsets, is_complete_part, bufferedRanks
  std::vector<int> send_counts, recv_counts;     // fan-in: entities to every owner / from every holder
  std::vector<int> recv_ent;                     // my entity of every received entity, rank-major
  int nrecv = 0, nsend = 0;
  pp::DevBuf d_gids, d_owners, d_rank_lids, d_comm_index, d_full_ids, d_ent_ids, d_recv_ent, d_contrib_off,
      d_contrib_pos;
};

template <class T>
int upload(pp::DevBuf& d, const std::vector<T>& h) {
  PP_HIP_CHECK(d.reserve(std::max<size_t>(h.size(), 1) * sizeof(T)));
  if (!h.empty()) PP_HIP_CHECK(hipMemcpy(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return PP_OK;
}

// ---- reduceCommArray kernels.  Layout of the fan-in send buffer: the segments of the other owners in
// rank order (the comm-array order of pumipic_comm.cpp:280-288 with the own segment taken out -- it goes
// to `mine`), so that the exchange's per-rank displacements are the prefix of the counts.
template <class T>
__global__ void k_part_pack(long long n, int nvals, const int* __restrict__ ci, int p0, int mycnt,
                            const T* __restrict__ in, T* __restrict__ send, T* __restrict__ mine) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long id = i / nvals;
  const int v = (int)(i - id * nvals);
  const int j = ci[id];
  if (j < p0)
    send[(long long)j * nvals + v] = in[i];
  else if (j < p0 + mycnt)
    mine[(long long)(j - p0) * nvals + v] = in[i];
  else
    send[(long long)(j - mycnt) * nvals + v] = in[i];
}
// the owner's combine (:311-385), gather form: one thread per value of an owned entity walks the
// contributions it received in increasing rank
template <class T>
__global__ void k_part_reduce(long long n, int nvals, int op, T* __restrict__ mine, const T* __restrict__ recv,
                              const int* __restrict__ coff, const int* __restrict__ cpos) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long e = i / nvals;
  const int v = (int)(i - e * nvals);
  T x = mine[i];
  for (int k = coff[e]; k < coff[e + 1]; ++k) {
    const T y = recv[(long long)cpos[k] * nvals + v];
    if (op == PP_OP_SUM)
      x = x + y;
    else if (op == PP_OP_MAX)
      x = x > y ? x : y;  // maxReduce, pumipic_comm.cpp:222-227
    else
      x = x < y ? x : y;  // minReduce :228-233
  }
  mine[i] = x;
}
// fan-out send buffer (:386-421): every holder gets back the entities it sent, in its order
template <class T>
__global__ void k_part_fanout(long long n, int nvals, const int* __restrict__ recv_ent, const T* __restrict__ mine,
                              T* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long k = i / nvals;
  const int v = (int)(i - k * nvals);
  out[i] = mine[(long long)recv_ent[k] * nvals + v];
}
template <class T>
__global__ void k_part_unpack(long long n, int nvals, const int* __restrict__ ci, int p0, int mycnt,
                              const T* __restrict__ mine, const T* __restrict__ recv, T* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long id = i / nvals;
  const int v = (int)(i - id * nvals);
  const int j = ci[id];
  if (j < p0)
    out[i] = recv[(long long)j * nvals + v];
  else if (j < p0 + mycnt)
    out[i] = mine[(long long)(j - p0) * nvals + v];
  else
    out[i] = recv[(long long)(j - mycnt) * nvals + v];
}

}  // namespace

struct pp_picpart {
  int rank = 0, nranks = 1, dim = 0;
  bool is_full = false;
  const pp_mesh* full = nullptr;
  pp_mesh* part = nullptr;  // owned; null when the part is the full mesh
  pp_comm* comm = nullptr;
  std::vector<int> has_part;
  int num_buffers = 1;
  // the Input the part was built from (the balancer re-derives every rank's safe zone from it)
  std::vector<int> owner_e;
  int buffer_method = 0, safe_method = 0, bridge_dim = 0, buffer_layers = 0, safe_layers = 0;
  DimData D[4];  // 0: vertices, 1: elements, 2: sides (dimension dim-1), 3: edges of a tet mesh (dimension 1)
  int nslots = 3;  // 4 for tet meshes
  std::vector<unsigned char> safe;
  pp::DevBuf d_safe;
  // reduction scratch and the state between the phases
  pp::DevBuf d_send, d_mine, d_out;
  bool active = false;
  int k = 0, op = 0, dtype = 0, nvals = 0, phase = 0;
  void* array = nullptr;
};

namespace {

DimData* dim_slot(pp_picpart* p, int edim) {
  if (edim == 0) return &p->D[0];
  if (edim == p->dim) return &p->D[1];
  if (edim == p->dim - 1) return &p->D[2];
  if (edim == 1 && p->dim == 3) return &p->D[3];
  pp::set_error("PICpart: entity dimension must be between 0 (vertices) and the mesh dimension (elements)");
  return nullptr;
}
const DimData* dim_slot(const pp_picpart* p, int edim) { return dim_slot(const_cast<pp_picpart*>(p), edim); }

// Mesh::Mesh(Input&), pumipic_part_construct.cpp:75-118, for rank q: has_part (and is_safe when wanted)
int safe_and_buffer(const pp_mesh* full, int bridge_dim, int q, int P, int buffer_method, int safe_method,
                    int buffer_layers, int safe_layers, const int* d_owner, pp::DevBuf& d_safe_tmp,
                    std::vector<int>& has_part, std::vector<unsigned char>* is_safe) {
  const int ne = full->nelems;
  has_part.assign((size_t)P, 1);
  if (is_safe) is_safe->assign((size_t)ne, safe_method == PP_PART_FULL ? 1 : 0);
  const bool need_bfs = (safe_method != PP_PART_NONE && safe_method != PP_PART_FULL) || buffer_method != PP_PART_FULL;
  if (need_bfs) {
    std::vector<int> part((size_t)P, 0);
    int rc = pp_bfs_buffer_layers(full, bridge_dim, q, P, safe_layers, buffer_layers, d_owner,
                                  d_safe_tmp.as<unsigned char>(), part.data());
    if (rc) return rc;
    if (is_safe && (safe_method == PP_PART_BFS || safe_method == PP_PART_MINIMUM) && ne > 0)
      PP_HIP_CHECK(hipMemcpy(is_safe->data(), d_safe_tmp.p, (size_t)ne, hipMemcpyDeviceToHost));
    if (buffer_method == PP_PART_BFS || buffer_method == PP_PART_MINIMUM) has_part = part;
  }
  if (is_safe && buffer_method == PP_PART_BFS && safe_method == PP_PART_FULL && ne > 0) {
    int rc = pp_bfs_safe_inward(full, bridge_dim, q, P, safe_layers, d_owner, has_part.data(),
                                d_safe_tmp.as<unsigned char>());
    if (rc) return rc;
    PP_HIP_CHECK(hipMemcpy(is_safe->data(), d_safe_tmp.p, (size_t)ne, hipMemcpyDeviceToHost));
  }
  return PP_OK;
}

// createGlobalNumbering + rankLidNumbering (:336-386): owner-major, in entity order inside an owner
void global_numbering(const std::vector<int>& owner, int P, std::vector<int>& goff, std::vector<int64_t>& gid) {
  goff.assign((size_t)P + 1, 0);
  for (int o : owner) ++goff[(size_t)o + 1];
  for (int r = 0; r < P; ++r) goff[(size_t)r + 1] += goff[(size_t)r];
  std::vector<int> run(goff.begin(), goff.end() - 1);
  gid.resize(owner.size());
  for (size_t i = 0; i < owner.size(); ++i) gid[i] = run[(size_t)owner[i]]++;
}

// entities of the part for one dimension: part id -> full id as given (vertices / elements: the kept ones in
// full-mesh order, :181-194; sides: the order the part's own mesh derives)
void set_entities(DimData& d, int nfull, std::vector<int> full_ids) {
  d.nfull = nfull;
  d.full_ids = std::move(full_ids);
  d.nents = (int)d.full_ids.size();
  d.ent_ids.assign((size_t)nfull, -1);
  for (int i = 0; i < d.nents; ++i) d.ent_ids[(size_t)d.full_ids[(size_t)i]] = i;
}
std::vector<int> kept_list(const std::vector<char>& keep) {
  std::vector<int> ids;
  for (int i = 0; i < (int)keep.size(); ++i)
    if (keep[(size_t)i]) ids.push_back(i);
  return ids;
}
// the part's side of setupComm (pumipic_comm.cpp:11-111) for one entity dimension
void setup_holder(DimData& d, int P, int rank, const std::vector<int>& owner_full, const std::vector<int64_t>& gid_full) {
  d.owners.resize((size_t)d.nents);
  d.gids.resize((size_t)d.nents);
  d.rank_lids.resize((size_t)d.nents);
  d.poff.assign((size_t)P + 1, 0);
  for (int i = 0; i < d.nents; ++i) {
    const int f = d.full_ids[(size_t)i];
    d.owners[(size_t)i] = owner_full[(size_t)f];
    d.gids[(size_t)i] = gid_full[(size_t)f];
    d.rank_lids[(size_t)i] = (int)(gid_full[(size_t)f] - d.goff[(size_t)owner_full[(size_t)f]]);
    ++d.poff[(size_t)owner_full[(size_t)f] + 1];
  }
  for (int r = 0; r < P; ++r) d.poff[(size_t)r + 1] += d.poff[(size_t)r];
  d.is_complete.assign((size_t)P, 0);
  d.buffered.clear();
  d.send_counts.assign((size_t)P, 0);
  for (int r = 0; r < P; ++r) {
    const int gdiff = d.goff[(size_t)r + 1] - d.goff[(size_t)r], pdiff = d.poff[(size_t)r + 1] - d.poff[(size_t)r];
    d.is_complete[(size_t)r] = (gdiff == pdiff) + (pdiff != 0);  // :54-63
    if (pdiff != 0 && r != rank) d.buffered.push_back(r);         // :33-40
    if (r != rank) d.send_counts[(size_t)r] = pdiff;
  }
  d.my_count = d.poff[(size_t)rank + 1] - d.poff[(size_t)rank];
  // comm array index (:43-86): rank lid, renumbered in increasing FULL-mesh id for a partially held part (what
  // travels between ranks is defined on full-mesh ids: the owner lists the same entities in the same order)
  std::vector<int> order((size_t)d.nents);
  std::iota(order.begin(), order.end(), 0);
  if (!std::is_sorted(d.full_ids.begin(), d.full_ids.end()))
    std::sort(order.begin(), order.end(), [&](int a, int b) { return d.full_ids[(size_t)a] < d.full_ids[(size_t)b]; });
  std::vector<int> run((size_t)P, 0);
  d.comm_index.resize((size_t)d.nents);
  for (int i : order) {
    const int o = d.owners[(size_t)i];
    const int lid = d.is_complete[(size_t)o] == 1 ? run[(size_t)o]++ : d.rank_lids[(size_t)i];
    d.comm_index[(size_t)i] = lid + d.poff[(size_t)o];
  }
  d.nsend = d.nents - d.my_count;
}

int upload_dim(DimData& d) {
  int rc;
  if ((rc = upload(d.d_gids, d.gids))) return rc;
  if ((rc = upload(d.d_owners, d.owners))) return rc;
  if ((rc = upload(d.d_rank_lids, d.rank_lids))) return rc;
  if ((rc = upload(d.d_comm_index, d.comm_index))) return rc;
  if ((rc = upload(d.d_full_ids, d.full_ids))) return rc;
  if ((rc = upload(d.d_ent_ids, d.ent_ids))) return rc;
  if ((rc = upload(d.d_recv_ent, d.recv_ent))) return rc;
  // contributions of every owned entity, in the order they sit in the receive buffer (= rank order)
  std::vector<int> coff((size_t)d.my_count + 1, 0), cpos((size_t)d.nrecv);
  for (int k = 0; k < d.nrecv; ++k) ++coff[(size_t)d.recv_ent[(size_t)k] + 1];
  for (int e = 0; e < d.my_count; ++e) coff[(size_t)e + 1] += coff[(size_t)e];
  std::vector<int> run(coff.begin(), coff.end() - 1);
  for (int k = 0; k < d.nrecv; ++k) cpos[(size_t)run[(size_t)d.recv_ent[(size_t)k]]++] = k;
  if ((rc = upload(d.d_contrib_off, coff))) return rc;
  if ((rc = upload(d.d_contrib_pos, cpos))) return rc;
  return PP_OK;
}

template <class T>
int reduce_phase(pp_picpart* p, int phase) {
  DimData& d = p->D[p->k];
  hipStream_t st = pp::stream();
  pp_comm* c = p->comm;
  const int nv = p->nvals;
  const size_t es = sizeof(T);
  T* arr = (T*)p->array;
  const int p0 = d.poff[(size_t)p->rank];
  void* d_recv = nullptr;
  int rc;
  if (phase == 0) {  // pack + (local world) publish the fan-in
    PP_HIP_CHECK(p->d_send.reserve(std::max<size_t>((size_t)d.nsend * nv, 1) * es));
    PP_HIP_CHECK(p->d_mine.reserve(std::max<size_t>((size_t)d.my_count * nv, 1) * es));
    PP_HIP_CHECK(p->d_out.reserve(std::max<size_t>((size_t)d.nrecv * nv, 1) * es));
    const long long n = (long long)d.nents * nv;
    if (n > 0)
      k_part_pack<T><<<grid_for((size_t)n), kBlock, 0, st>>>(n, nv, d.d_comm_index.as<int>(), p0, d.my_count, arr,
                                                            p->d_send.as<T>(), p->d_mine.as<T>());
    PP_LAUNCH_CHECK();
    if (c->kind == 4 && p->op != PP_OP_BCAST)
      return pp::local_publish(c->world.get(), c->rank, d.send_counts, p->d_send.p, (int)(nv * es), 1);
    return PP_OK;
  }
  if (phase == 1) {  // fan-in exchange, combine, fan-out send buffer
    if (p->op != PP_OP_BCAST) {
      if (c->kind == 4 && (rc = pp::local_all_begun(c->world.get(), 1))) return rc;
      std::vector<int> rcnt = d.recv_counts;
      rc = pp::comm_exchange_records(c, p->d_send.p, d.send_counts, rcnt, (int)(nv * es), &d_recv, 1);
      if (c->kind == 4) pp::local_ended(c->world.get(), c->rank, 1);
      if (rc) return rc;
      if (rcnt != d.recv_counts) {
        pp::set_error("pp_picpart_reduce: the ranks disagree on the exchange plan (different partition vectors "
                      "or buffer rules?)");
        return PP_ESTATE;
      }
      const long long n = (long long)d.my_count * nv;
      if (n > 0 && d.nrecv > 0)
        k_part_reduce<T><<<grid_for((size_t)n), kBlock, 0, st>>>(n, nv, p->op, p->d_mine.as<T>(), (const T*)d_recv,
                                                                d.d_contrib_off.as<int>(), d.d_contrib_pos.as<int>());
    }
    const long long n = (long long)d.nrecv * nv;
    if (n > 0)
      k_part_fanout<T><<<grid_for((size_t)n), kBlock, 0, st>>>(n, nv, d.d_recv_ent.as<int>(), p->d_mine.as<T>(),
                                                              p->d_out.as<T>());
    PP_LAUNCH_CHECK();
    if (c->kind == 4) return pp::local_publish(c->world.get(), c->rank, d.recv_counts, p->d_out.p, (int)(nv * es), 2);
    return PP_OK;
  }
  // phase 2: fan-out exchange (the counts of the fan-in, reversed), unpack
  if (c->kind == 4 && (rc = pp::local_all_begun(c->world.get(), 2))) return rc;
  std::vector<int> rcnt = d.send_counts;
  rc = pp::comm_exchange_records(c, p->d_out.p, d.recv_counts, rcnt, (int)(nv * es), &d_recv, 2);
  if (c->kind == 4) pp::local_ended(c->world.get(), c->rank, 2);
  if (rc) return rc;
  if (rcnt != d.send_counts) {
    pp::set_error("pp_picpart_reduce: the ranks disagree on the exchange plan");
    return PP_ESTATE;
  }
  const long long n = (long long)d.nents * nv;
  if (n > 0)
    k_part_unpack<T><<<grid_for((size_t)n), kBlock, 0, st>>>(n, nv, d.d_comm_index.as<int>(), p0, d.my_count,
                                                            p->d_mine.as<T>(), (const T*)d_recv, arr);
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int run_phase(pp_picpart* p, int phase) {
  return p->dtype == PP_T_I32 ? reduce_phase<int>(p, phase) : reduce_phase<double>(p, phase);
}

}  // namespace

extern "C" {

int pp_owner_by_classification(const pp_mesh* full, const int* class_owners_host, int nclass, int comm_rank,
                               int* elem_owner_host_out) {
  PP_REQUIRE(full && class_owners_host && elem_owner_host_out && nclass > 0, "pp_owner_by_classification: bad argument");
  int mine = 0;
  for (int e = 0; e < full->nelems; ++e) {
    const int c = full->class_id[(size_t)e];
    if (c < 0 || c >= nclass) {  // the reference prints and reads out of range (:286-290); here an error
      pp::set_error("pp_owner_by_classification: class id of an element outside [0, nclass)");
      return PP_EINVAL;
    }
    elem_owner_host_out[e] = class_owners_host[c];
    mine += class_owners_host[c] == comm_rank;
  }
  if (!mine && full->nelems > 0) {  // the reference asserts (:297-301)
    pp::set_error("pp_owner_by_classification: this rank owns no element");
    return PP_EINVAL;
  }
  return PP_OK;
}

pp_picpart* pp_picpart_create(const pp_mesh* full, const int* elem_owner_host, int buffer_method, int safe_method,
                              int bridge_dim, int buffer_layers, int safe_layers, pp_comm* comm) {
  auto fail = [](const char* msg) -> pp_picpart* {
    pp::set_error(msg);
    return nullptr;
  };
  if (!full || !elem_owner_host) return fail("pp_picpart_create: null argument");
  if (buffer_method < PP_PART_FULL || buffer_method > PP_PART_NONE || safe_method < PP_PART_FULL ||
      safe_method > PP_PART_NONE)
    return fail("pp_picpart_create: unknown buffer / safe method");
  if (buffer_layers < 0 || safe_layers < 0) return fail("pp_picpart_create: negative layer count");
  if (bridge_dim != 0 && bridge_dim != full->dim - 1)
    return fail("pp_picpart_create: bridge_dim must be 0 (vertices) or dim-1 (sides)");
  if (buffer_method == PP_PART_NONE) buffer_method = PP_PART_MINIMUM;  // pumipic_input.cpp:96-100
  if (buffer_method == PP_PART_MINIMUM) buffer_layers = 0;             // :107-110
  if (safe_method == PP_PART_MINIMUM) safe_layers = 0;
  const int P = comm ? comm->nranks : 1, rank = comm ? comm->rank : 0;
  const int ne = full->nelems, nv = full->nverts, nvpe = full->dim + 1;
  for (int e = 0; e < ne; ++e)
    if (elem_owner_host[e] < 0 || elem_owner_host[e] >= P)
      return fail("pp_picpart_create: element owner outside [0, comm size)");
  pp_picpart* p = new pp_picpart();
  auto bail = [&](const char* msg) -> pp_picpart* {
    if (msg) pp::set_error(msg);
    if (p->part) pp_mesh_destroy(p->part);
    delete p;
    return nullptr;
  };
  p->rank = rank;
  p->nranks = P;
  p->dim = full->dim;
  p->full = full;
  p->comm = comm;
  p->is_full = buffer_method == PP_PART_FULL;
  p->D[0].edim = 0;
  p->D[1].edim = full->dim;
  p->D[2].edim = full->dim - 1;
  std::vector<int> owner_e(elem_owner_host, elem_owner_host + ne);
  p->owner_e = owner_e;
  p->buffer_method = buffer_method;
  p->safe_method = safe_method;
  p->bridge_dim = bridge_dim;
  p->buffer_layers = buffer_layers;
  p->safe_layers = safe_layers;
  pp::DevBuf d_owner, d_safe_tmp;
  if (upload(d_owner, owner_e) != PP_OK) return bail(nullptr);
  if (d_safe_tmp.reserve((size_t)std::max(ne, 1)) != hipSuccess) return bail("pp_picpart_create: out of device memory");
  // ---- this rank's safe zone and buffer
  std::vector<unsigned char> is_safe;
  if (safe_and_buffer(full, bridge_dim, rank, P, buffer_method, safe_method, buffer_layers, safe_layers,
                      d_owner.as<int>(), d_safe_tmp, p->has_part, &is_safe) != PP_OK)
    return bail(nullptr);
  p->num_buffers = 0;
  for (int r = 0; r < P; ++r) p->num_buffers += p->has_part[(size_t)r] > 0;
  // ---- ownership of the vertices (defineOwners :305-323) and the global numbering (:153-163)
  std::vector<int> owner_v((size_t)nv, P);
  for (int v = 0; v < nv; ++v)
    for (int k = full->vert2elems_off[(size_t)v]; k < full->vert2elems_off[(size_t)v + 1]; ++k)
      owner_v[(size_t)v] = std::min(owner_v[(size_t)v], owner_e[(size_t)full->vert2elems[(size_t)k]]);
  for (int v = 0; v < nv; ++v)
    if (owner_v[(size_t)v] >= P) return bail("pp_picpart_create: a vertex belongs to no element");
  std::vector<int64_t> gid_v, gid_e;
  global_numbering(owner_v, P, p->D[0].goff, gid_v);
  global_numbering(owner_e, P, p->D[1].goff, gid_e);
  // ---- entities of this part (setSafeEnts :470-494) and the holder side of setupComm
  auto kept = [&](const std::vector<int>& has_part, std::vector<char>& keep_e, std::vector<char>& keep_v) {
    keep_e.assign((size_t)ne, 0);
    keep_v.assign((size_t)nv, 0);
    for (int e = 0; e < ne; ++e)
      if (has_part[(size_t)owner_e[(size_t)e]]) {
        keep_e[(size_t)e] = 1;
        for (int k = 0; k < nvpe; ++k) keep_v[(size_t)full->elem2verts[(size_t)e * nvpe + k]] = 1;
      }
  };
  std::vector<char> keep_e, keep_v;
  kept(p->has_part, keep_e, keep_v);
  set_entities(p->D[0], nv, kept_list(keep_v));
  set_entities(p->D[1], ne, kept_list(keep_e));
  if (p->D[1].nents == 0 && ne > 0) return bail("pp_picpart_create: empty part on this rank (:232-235)");
  // ---- the part's mesh (:196-258): kept vertices and elements in full-mesh order
  if (!p->is_full) {
    const DimData &dv = p->D[0], &de = p->D[1];
    std::vector<double> coords((size_t)dv.nents * full->dim);
    for (int i = 0; i < dv.nents; ++i)
      for (int c = 0; c < full->dim; ++c)
        coords[(size_t)i * full->dim + c] = full->coords[(size_t)dv.full_ids[(size_t)i] * full->dim + c];
    std::vector<int> e2v((size_t)de.nents * nvpe), cls((size_t)de.nents);
    for (int i = 0; i < de.nents; ++i) {
      const int f = de.full_ids[(size_t)i];
      for (int k = 0; k < nvpe; ++k) e2v[(size_t)i * nvpe + k] = dv.ent_ids[(size_t)full->elem2verts[(size_t)f * nvpe + k]];
      cls[(size_t)i] = full->class_id[(size_t)f];
    }
    p->part = pp_mesh_create(full->dim, dv.nents, coords.data(), de.nents, e2v.data(), cls.data());
    if (!p->part) return bail(nullptr);
  }
  // ---- sides (and, for tets, edges): the part's own mesh numbers them (pp_mesh derives them from its
  // elements); part entity -> full entity by their vertices.  Owner = smallest owner of the elements around
  // it (defineOwners :305-323 over ask_up(d, dim)).  The reference loops every dimension 0..dim
  // (pumipic_part_construct.cpp:141-163, test/test_comm_array.cpp:48-66).
  p->nslots = full->dim == 3 ? 4 : 3;
  if (full->dim == 3) {
    if (pp::mesh_edges(full) != PP_OK) return bail(nullptr);
    if (p->part && pp::mesh_edges(p->part) != PP_OK) return bail(nullptr);
    p->D[3].edim = 1;
  }
  struct MidDim {
    int slot, nvpe, nfull;
    const std::vector<int>*ent2verts, *up_off, *up, *part_ent2verts;
    int npart;
  };
  std::vector<MidDim> mids;
  mids.push_back(MidDim{2, full->dim, full->nsides, &full->side2verts, &full->side2elems_off, &full->side2elems,
                        p->part ? &p->part->side2verts : nullptr, p->part ? p->part->nsides : 0});
  if (full->dim == 3)
    mids.push_back(MidDim{3, 2, full->nedges, &full->edge2verts, &full->edge2elems_off, &full->edge2elems,
                          p->part ? &p->part->edge2verts : nullptr, p->part ? p->part->nedges : 0});
  std::vector<int> owner_mid[2];
  std::vector<int64_t> gid_mid[2];
  for (size_t mi = 0; mi < mids.size(); ++mi) {
    const MidDim& M = mids[mi];
    const int ns = M.nfull, nvps = M.nvpe;
    std::vector<int>& owner_s = owner_mid[mi];
    owner_s.assign((size_t)ns, P);
    for (int sd = 0; sd < ns; ++sd)
      for (int k = (*M.up_off)[(size_t)sd]; k < (*M.up_off)[(size_t)sd + 1]; ++k)
        owner_s[(size_t)sd] = std::min(owner_s[(size_t)sd], owner_e[(size_t)(*M.up)[(size_t)k]]);
    std::vector<int> side_full;
    if (p->is_full) {
      side_full.resize((size_t)ns);
      std::iota(side_full.begin(), side_full.end(), 0);
    } else {
      typedef std::array<int, 4> Key;  // sorted vertices (full ids) + entity id
      auto key_of = [&](const int* v, const std::vector<int>* to_full, int id) {
        Key k{{0, 0, 0, id}};
        for (int j = 0; j < nvps; ++j) k[(size_t)j] = to_full ? (*to_full)[(size_t)v[j]] : v[j];
        std::sort(k.begin(), k.begin() + nvps);
        return k;
      };
      std::vector<Key> keys((size_t)ns);
      for (int sd = 0; sd < ns; ++sd) keys[(size_t)sd] = key_of(&(*M.ent2verts)[(size_t)sd * nvps], nullptr, sd);
      auto less3 = [](const Key& a, const Key& b) {
        return std::lexicographical_compare(a.begin(), a.begin() + 3, b.begin(), b.begin() + 3);
      };
      std::sort(keys.begin(), keys.end(), less3);
      side_full.resize((size_t)M.npart);
      for (int sd = 0; sd < M.npart; ++sd) {
        const Key k = key_of(&(*M.part_ent2verts)[(size_t)sd * nvps], &p->D[0].full_ids, -1);
        auto it = std::lower_bound(keys.begin(), keys.end(), k, less3);
        if (it == keys.end() || less3(k, *it))
          return bail("pp_picpart_create: a side / edge of the part is not one of the full mesh");
        side_full[(size_t)sd] = (*it)[3];
      }
    }
    set_entities(p->D[M.slot], ns, std::move(side_full));
    for (int sd = 0; sd < ns; ++sd)
      if (owner_s[(size_t)sd] >= P) return bail("pp_picpart_create: a side / edge belongs to no element");
    global_numbering(owner_s, P, p->D[M.slot].goff, gid_mid[mi]);
  }
  const std::vector<int>& owner_s = owner_mid[0];
  const std::vector<int64_t>& gid_s = gid_mid[0];
  const int ns = full->nsides;
  setup_holder(p->D[0], P, rank, owner_v, gid_v);
  setup_holder(p->D[1], P, rank, owner_e, gid_e);
  setup_holder(p->D[2], P, rank, owner_s, gid_s);
  if (p->nslots == 4) setup_holder(p->D[3], P, rank, owner_mid[1], gid_mid[1]);
  // ---- the owner side (what the reference learns from MPI_Ialltoall + Isend/Irecv, :113-190): which of my
  // entities every other rank holds, in increasing full-mesh id -- from that rank's own buffer rule
  for (int k = 0; k < p->nslots; ++k) {
    p->D[k].recv_counts.assign((size_t)P, 0);
    p->D[k].recv_ent.clear();
  }
  std::vector<int> part_q;
  std::vector<char> ke_q, kv_q, ks_q, kd_q;
  for (int q = 0; q < P; ++q) {
    if (q == rank) continue;
    if (buffer_method == PP_PART_FULL) {
      part_q.assign((size_t)P, 1);
    } else if (safe_and_buffer(full, bridge_dim, q, P, buffer_method, PP_PART_NONE, buffer_layers, 0,
                               d_owner.as<int>(), d_safe_tmp, part_q, nullptr) != PP_OK) {
      return bail(nullptr);
    }
    kept(part_q, ke_q, kv_q);
    ks_q.assign((size_t)ns, 0);
    for (int sd = 0; sd < ns; ++sd)
      for (int k = full->side2elems_off[(size_t)sd]; k < full->side2elems_off[(size_t)sd + 1]; ++k)
        if (ke_q[(size_t)full->side2elems[(size_t)k]]) ks_q[(size_t)sd] = 1;
    if (p->nslots == 4) {  // an edge is held when an element around it is
      kd_q.assign((size_t)full->nedges, 0);
      for (int ed = 0; ed < full->nedges; ++ed)
        for (int k = full->edge2elems_off[(size_t)ed]; k < full->edge2elems_off[(size_t)ed + 1]; ++k)
          if (ke_q[(size_t)full->edge2elems[(size_t)k]]) kd_q[(size_t)ed] = 1;
    }
    for (int k = 0; k < p->nslots; ++k) {
      DimData& d = p->D[k];
      const std::vector<int>& owner = k == 0 ? owner_v : (k == 1 ? owner_e : (k == 2 ? owner_s : owner_mid[1]));
      const std::vector<int64_t>& gid = k == 0 ? gid_v : (k == 1 ? gid_e : (k == 2 ? gid_s : gid_mid[1]));
      const std::vector<char>& keep = k == 0 ? kv_q : (k == 1 ? ke_q : (k == 2 ? ks_q : kd_q));
      const int g0 = d.goff[(size_t)rank];
      const size_t before = d.recv_ent.size();
      for (int i = 0; i < d.nfull; ++i)  // increasing full id: the order rank q numbers a partial part in
        if (keep[(size_t)i] && owner[(size_t)i] == rank) d.recv_ent.push_back((int)(gid[(size_t)i] - g0));
      const int cnt = (int)(d.recv_ent.size() - before);
      d.recv_counts[(size_t)q] = cnt;
      // a completely held part travels in rank-lid order (its lids are not renumbered, :66-76)
      if (cnt == d.my_count) std::iota(d.recv_ent.begin() + (long)before, d.recv_ent.end(), 0);
    }
  }
  for (int k = 0; k < p->nslots; ++k) p->D[k].nrecv = (int)p->D[k].recv_ent.size();
  p->safe.resize((size_t)p->D[1].nents);
  for (int i = 0; i < p->D[1].nents; ++i) p->safe[(size_t)i] = is_safe[(size_t)p->D[1].full_ids[(size_t)i]];
  if (upload(p->d_safe, p->safe) != PP_OK || upload_dim(p->D[0]) != PP_OK || upload_dim(p->D[1]) != PP_OK ||
      upload_dim(p->D[2]) != PP_OK || (p->nslots == 4 && upload_dim(p->D[3]) != PP_OK))
    return bail(nullptr);
  return p;
}

// the owner's side of a partially held part, as the reference keeps (and writes to its .ppm files,
// src/pumipic_file.cpp:80-115): boundary_parts (ranks that hold only a boundary of my entities, ascending),
// offset_bounded_per_dim (prefix over them) and bounded_ent_ids (my rank-local ids they hold, in the order
// they send them).  Null outputs are skipped: call once for the sizes.
int pp_picpart_bounded(const pp_picpart* p, int edim, int* n_boundaries, int* boundary_parts_host, int* offsets_host,
                       int* n_ids, int* ent_ids_host) {
  PP_REQUIRE(p, "pp_picpart_bounded: null part");
  const DimData* d = dim_slot(p, edim);
  if (!d) return PP_EINVAL;
  int nb = 0, nid = 0, pos = 0;
  if (offsets_host) offsets_host[0] = 0;
  for (int q = 0; q < p->nranks; ++q) {
    const int cnt = q == p->rank || d->recv_counts.empty() ? 0 : d->recv_counts[(size_t)q];
    if (cnt > 0 && cnt != d->my_count) {  // (all of mine = a completely buffered part: no list travels)
      if (boundary_parts_host) boundary_parts_host[nb] = q;
      if (ent_ids_host)
        for (int k = 0; k < cnt; ++k) ent_ids_host[nid + k] = d->recv_ent[(size_t)pos + k];
      nid += cnt;
      ++nb;
      if (offsets_host) offsets_host[nb] = nid;
    }
    pos += cnt;
  }
  if (n_boundaries) *n_boundaries = nb;
  if (n_ids) *n_ids = nid;
  return PP_OK;
}
// global number of entities of a dimension (Mesh::num_entites, pumipic_mesh.hpp:106)
long long pp_picpart_num_global(const pp_picpart* p, int edim) {
  if (!p) return PP_EINVAL;
  const DimData* d = dim_slot(p, edim);
  return d && !d->goff.empty() ? (long long)d->goff.back() : -1;
}

int pp_picpart_destroy(pp_picpart* p) {
  if (!p) return PP_OK;
  if (p->part) pp_mesh_destroy(p->part);
  delete p;
  return PP_OK;
}

const pp_mesh* pp_picpart_mesh(const pp_picpart* p) { return p ? (p->part ? p->part : p->full) : nullptr; }

int pp_picpart_info(const pp_picpart* p, int* is_full_mesh, int* num_buffers, int* nverts, int* nelems) {
  PP_REQUIRE(p, "pp_picpart_info: null part");
  if (is_full_mesh) *is_full_mesh = p->is_full ? 1 : 0;
  if (num_buffers) *num_buffers = p->num_buffers;
  if (nverts) *nverts = p->D[0].nents;
  if (nelems) *nelems = p->D[1].nents;
  return PP_OK;
}

const void* pp_picpart_array_dev(const pp_picpart* p, int which, int edim, size_t* count) {
  if (!p) {
    pp::set_error("pp_picpart_array_dev: null part");
    return nullptr;
  }
  if (which == PP_PART_SAFE) {
    if (count) *count = p->safe.size();
    return p->d_safe.p;
  }
  const DimData* d = dim_slot(p, edim);
  if (!d) return nullptr;
  const pp::DevBuf* b = nullptr;
  size_t n = (size_t)d->nents;
  switch (which) {
    case PP_PART_GIDS: b = &d->d_gids; break;
    case PP_PART_OWNERS: b = &d->d_owners; break;
    case PP_PART_RANK_LIDS: b = &d->d_rank_lids; break;
    case PP_PART_COMM_INDEX: b = &d->d_comm_index; break;
    case PP_PART_FULL_IDS: b = &d->d_full_ids; break;
    case PP_PART_ENT_IDS: b = &d->d_ent_ids; n = (size_t)d->nfull; break;
    default: pp::set_error("pp_picpart_array_dev: unknown array"); return nullptr;
  }
  if (count) *count = n;
  return b->p;
}

int pp_picpart_array_to_host(const pp_picpart* p, int which, int edim, void* out_host) {
  PP_REQUIRE(p && out_host, "pp_picpart_array_to_host: null argument");
  if (which == PP_PART_SAFE) {
    if (!p->safe.empty()) memcpy(out_host, p->safe.data(), p->safe.size());
    return PP_OK;
  }
  const DimData* d = dim_slot(p, edim);
  if (!d) return PP_EINVAL;
  auto cp = [&](const void* src, size_t bytes) {
    if (bytes) memcpy(out_host, src, bytes);
    return PP_OK;
  };
  switch (which) {
    case PP_PART_GIDS: return cp(d->gids.data(), d->gids.size() * sizeof(int64_t));
    case PP_PART_OWNERS: return cp(d->owners.data(), d->owners.size() * sizeof(int));
    case PP_PART_RANK_LIDS: return cp(d->rank_lids.data(), d->rank_lids.size() * sizeof(int));
    case PP_PART_COMM_INDEX: return cp(d->comm_index.data(), d->comm_index.size() * sizeof(int));
    case PP_PART_FULL_IDS: return cp(d->full_ids.data(), d->full_ids.size() * sizeof(int));
    case PP_PART_ENT_IDS: return cp(d->ent_ids.data(), d->ent_ids.size() * sizeof(int));
    default: break;
  }
  pp::set_error("pp_picpart_array_to_host: unknown array");
  return PP_EINVAL;
}

int pp_picpart_nents_offsets(const pp_picpart* p, int edim, int* offsets_host) {
  PP_REQUIRE(p && offsets_host, "pp_picpart_nents_offsets: null argument");
  const DimData* d = dim_slot(p, edim);
  if (!d) return PP_EINVAL;
  std::copy(d->poff.begin(), d->poff.end(), offsets_host);
  return PP_OK;
}

int pp_picpart_buffered_ranks(const pp_picpart* p, int edim, int* ranks_host, int* n) {
  PP_REQUIRE(p && n, "pp_picpart_buffered_ranks: null argument");
  const DimData* d = dim_slot(p, edim);
  if (!d) return PP_EINVAL;
  *n = (int)d->buffered.size();
  if (ranks_host) std::copy(d->buffered.begin(), d->buffered.end(), ranks_host);
  return PP_OK;
}

int pp_picpart_complete_parts(const pp_picpart* p, int edim, int* is_complete_host) {
  PP_REQUIRE(p && is_complete_host, "pp_picpart_complete_parts: null argument");
  const DimData* d = dim_slot(p, edim);
  if (!d) return PP_EINVAL;
  std::copy(d->is_complete.begin(), d->is_complete.end(), is_complete_host);
  return PP_OK;
}

int pp_picpart_reduce_begin(pp_picpart* p, int edim, int op, int dtype, int nvals, void* array_dev) {
  pp::Range rg_("picpart_reduce");
  PP_REQUIRE(p && (array_dev || nvals == 0), "pp_picpart_reduce: null argument");
  PP_REQUIRE(op >= PP_OP_SUM && op <= PP_OP_BCAST, "pp_picpart_reduce: unknown operation");
  PP_REQUIRE(dtype == PP_T_I32 || dtype == PP_T_F64, "pp_picpart_reduce: dtype must be PP_T_I32 or PP_T_F64");
  PP_REQUIRE(nvals >= 0, "pp_picpart_reduce: negative nvals");
  PP_REQUIRE(!p->active, "pp_picpart_reduce: a reduction of this part is still between its phases");
  DimData* d = dim_slot(p, edim);
  if (!d) return PP_EINVAL;
  p->k = (int)(d - p->D);
  p->op = op;
  p->dtype = dtype;
  p->nvals = nvals;
  p->array = array_dev;
  if (p->nranks == 1 || !p->comm || nvals == 0) {  // :259-260
    p->phase = 3;
    p->active = true;
    return PP_OK;
  }
  int rc = run_phase(p, 0);
  if (rc) return rc;
  p->phase = 1;
  p->active = true;
  return PP_OK;
}
int pp_picpart_reduce_mid(pp_picpart* p) {
  PP_REQUIRE(p && p->active, "pp_picpart_reduce_mid: no reduction in flight (call pp_picpart_reduce_begin)");
  if (p->phase == 3) return PP_OK;
  PP_REQUIRE(p->phase == 1, "pp_picpart_reduce_mid: called twice");
  int rc = run_phase(p, 1);
  if (rc) {
    p->active = false;
    return rc;
  }
  p->phase = 2;
  return PP_OK;
}
int pp_picpart_reduce_end(pp_picpart* p) {
  PP_REQUIRE(p && p->active, "pp_picpart_reduce_end: no reduction in flight");
  if (p->phase == 3) {
    p->active = false;
    return PP_OK;
  }
  PP_REQUIRE(p->phase == 2, "pp_picpart_reduce_end: call pp_picpart_reduce_mid first");
  p->active = false;
  return run_phase(p, 2);
}
int pp_picpart_reduce(pp_picpart* p, int edim, int op, int dtype, int nvals, void* array_dev) {
  PP_REQUIRE(p, "pp_picpart_reduce: null part");
  PP_REQUIRE(!p->comm || p->comm->kind != 4,
             "pp_picpart_reduce: virtual ranks of one process call pp_picpart_reduce_begin / _mid / _end on every "
             "rank in turn");
  int rc = pp_picpart_reduce_begin(p, edim, op, dtype, nvals, array_dev);
  if (rc) return rc;
  if ((rc = pp_picpart_reduce_mid(p))) return rc;
  return pp_picpart_reduce_end(p);
}

}  // extern "C"

// ================================================================================================
// Particle load balancer: pumipic::ParticleBalancer, src/pumipic_lb.hpp / pumipic_lb.cpp.
//
// sbars (ParticleBalancer::ParticleBalancer, pumipic_lb.cpp:23-135): for every element the set of parts
// on which it is safe, its owner included.  The reference collects them with three rounds of messages
// (safe flags of the core to the owners, the owners' sbar lists to the buffers, global numbers); here
// every rank derives all of them from the Input, as the PICpart construction does -- the set itself (a
// 64-bit mask, bit r = part r) is the job-wide name of an sbar, and the sorted list of distinct masks is
// the same on every rank without a message.
// Weights (addWeights, pumipic_lb.hpp:138-237): device histogram of the particles that stay here by the
// sbar of their new element, and of the particles already sent elsewhere by destination.
// Balance: the reference calls EnGPar (engpar::balanceWeights, pumipic_lb.cpp:495-531; scorec/EnGPar >=
// 1.1.0, not in the reference tree).  Here: one all-gather of the weight rows, then every rank runs the
// same integer diffusion on the job-wide table (`diffuse` below; DESIGN.md states the scheme) -- at node
// scale the table is a few hundred numbers, replicating the computation costs less than a second exchange.
// Selection (selectParticles, :239-299): per sbar a list of (target, amount) consumed particle by
// particle, particles whose new element is outside the core first; a particle that finds an exhausted
// entry moves on to the next one (the reference lets it go and relies on further iterations), so the
// plan's amounts are met exactly.
namespace {

__global__ void k_bal_weights(int capacity, const unsigned char* __restrict__ mask, const int* __restrict__ new_elems,
                              const int* __restrict__ new_procs, int rank, int nranks, int ne,
                              const int* __restrict__ sbar_me, int* __restrict__ w, int* __restrict__ forced,
                              int* __restrict__ bad) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid]) return;
  const int pr = new_procs[pid];
  if (pr == rank) {
    const int e = new_elems[pid];
    if (e == -1) return;
    if (e < 0 || e >= ne) {
      *bad = 1;
      return;
    }
    const int s = sbar_me[e];
    if (s >= 0) atomicAdd(&w[s], 1);
  } else if (pr >= 0 && pr < nranks) {
    atomicAdd(&forced[pr], 1);
  } else {
    *bad = 1;
  }
}
// plan of this rank: entries [off[s], off[s+1]) of sbar s, each (target, remaining); cursor[s] walks them
__global__ void k_bal_select(int capacity, const unsigned char* __restrict__ mask, const int* __restrict__ new_elems,
                             int* __restrict__ new_procs, int rank, const int* __restrict__ owners,
                             const int* __restrict__ sbar_me, int noncore_only, const int* __restrict__ off,
                             int* __restrict__ cursor, const int* __restrict__ target, int* __restrict__ remaining) {
  const int pid = blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= capacity || !mask[pid]) return;
  if (new_procs[pid] != rank) return;
  const int e = new_elems[pid];
  if (e < 0) return;
  if (noncore_only && owners[e] == rank) return;
  const int s = sbar_me[e];
  if (s < 0) return;
  const int end = off[s + 1];
  for (;;) {
    const int idx = __hip_atomic_load(&cursor[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (idx >= end) return;
    if (atomicSub(&remaining[idx], 1) > 0) {
      new_procs[pid] = target[idx];
      return;
    }
    atomicMax(&cursor[s], idx + 1);
  }
}

struct PlanEntry {
  int sbar, target;
  long long amount;
};

// The diffusion (what EnGPar's balanceWeights does in the reference): rounds until max weight <= tol x
// average or nothing moves (at most 50).  In a round, on the weights the round started with, every part p
// above the average visits its lighter neighbours q (parts it shares an sbar with, p still holding weight
// there) in increasing rank and asks each for step_factor x (W_p - W_q) / (number of lighter neighbours)
// particles, taken from the shared sbars in increasing order, never more than p held in an sbar at the
// start of the call (a one-step migration cannot forward what is still to arrive).
void diffuse(const std::vector<unsigned long long>& masks, const std::vector<std::vector<long long>>& weights,
             const std::vector<long long>& forced_in, double tol, double step_factor,
             std::vector<std::vector<PlanEntry>>& plan, std::vector<long long>& W) {
  const int P = (int)weights.size(), M = (int)masks.size();
  std::vector<std::vector<long long>> avail = weights;
  W.assign((size_t)P, 0);
  long long total = 0;
  for (int r = 0; r < P; ++r) {
    for (int i = 0; i < M; ++i) W[(size_t)r] += avail[(size_t)r][(size_t)i];
    W[(size_t)r] += forced_in[(size_t)r];
    total += W[(size_t)r];
  }
  std::vector<std::vector<long long>> send((size_t)P, std::vector<long long>((size_t)M * P, 0));
  for (int it = 0; it < 50; ++it) {
    const long long mx = *std::max_element(W.begin(), W.end());
    if ((double)mx * P <= tol * (double)total) break;
    const std::vector<long long> W0 = W;
    bool moved = false;
    for (int p = 0; p < P; ++p) {
      if (W0[(size_t)p] * P <= total) continue;
      std::vector<int> nbrs;
      for (int q = 0; q < P; ++q) {
        if (q == p || !(W0[(size_t)q] < W0[(size_t)p])) continue;
        bool shared = false;
        for (int i = 0; i < M && !shared; ++i)
          shared = avail[(size_t)p][(size_t)i] > 0 && ((masks[(size_t)i] >> q) & 1ull) && ((masks[(size_t)i] >> p) & 1ull);
        if (shared) nbrs.push_back(q);
      }
      for (int q : nbrs) {
        long long want = (long long)(step_factor * (double)(W0[(size_t)p] - W0[(size_t)q]) / (double)nbrs.size());
        for (int i = 0; i < M && want > 0; ++i) {
          if (!(((masks[(size_t)i] >> q) & 1ull) && ((masks[(size_t)i] >> p) & 1ull))) continue;
          const long long t = std::min(avail[(size_t)p][(size_t)i], want);
          if (t <= 0) continue;
          avail[(size_t)p][(size_t)i] -= t;
          send[(size_t)p][(size_t)i * P + q] += t;
          W[(size_t)p] -= t;
          W[(size_t)q] += t;
          want -= t;
          moved = true;
        }
      }
    }
    if (!moved) break;
  }
  plan.assign((size_t)P, {});
  for (int p = 0; p < P; ++p)
    for (int i = 0; i < M; ++i)
      for (int q = 0; q < P; ++q)
        if (send[(size_t)p][(size_t)i * P + q] > 0) plan[(size_t)p].push_back({i, q, send[(size_t)p][(size_t)i * P + q]});
}

}  // namespace

struct pp_balancer {
  pp_picpart* part = nullptr;
  int rank = 0, nranks = 1;
  std::vector<unsigned long long> masks;  // sorted distinct sbars of the job
  std::vector<int> sbar_ids;              // per element of the part: index into masks (getSbarIDs)
  std::vector<int> sbar_me;               // the same, -1 where the sbar does not contain this rank
  pp::DevBuf d_sbar_ids, d_sbar_me, d_w, d_forced, d_bad, d_off, d_cursor, d_target, d_remaining;
  std::vector<PlanEntry> last_plan;
  std::vector<long long> last_W;
  // between _begin and _end
  bool active = false;
  int mode = 0;  // 1: repartition (structure), 2: partition (array)
  const pp_ps* ps = nullptr;
  const int* new_elems = nullptr;
  int* new_procs = nullptr;
  std::vector<long long> row;  // this rank's weights per sbar, then forced per destination
  std::vector<int> ppe;
};

namespace pp {
// local world mailbox (pp_comm.hip)
int local_mail_put(LocalWorld* w, int rank, const void* data, size_t bytes);
int local_mail_get_all(LocalWorld* w, int rank, size_t bytes, void* out);
}  // namespace pp

namespace {

int gather_rows(pp_balancer* b, std::vector<long long>& table, bool put, bool get) {
  pp_comm* c = b->part->comm;
  const int P = b->nranks;
  const size_t n = b->row.size(), bytes = n * sizeof(long long);
  if (!c || c->kind == 0 || P == 1) {
    table = b->row;
    return PP_OK;
  }
  if (c->kind == 4) {
    if (put) {
      int rc = pp::local_mail_put(c->world.get(), c->rank, b->row.data(), bytes);
      if (rc) return rc;
    }
    if (get) {
      table.resize(n * (size_t)P);
      return pp::local_mail_get_all(c->world.get(), c->rank, bytes, table.data());
    }
    return PP_OK;
  }
  if (!get) return PP_OK;  // real transports exchange in one step, at `end`
  table.resize(n * (size_t)P);
  return pp_comm_allgather_host(c, b->row.data(), table.data(), (int)bytes);
}

int plan_from_table(pp_balancer* b, const std::vector<long long>& table, double tol, double step_factor) {
  const int P = b->nranks, M = (int)b->masks.size();
  std::vector<std::vector<long long>> w((size_t)P, std::vector<long long>((size_t)M, 0));
  std::vector<long long> forced((size_t)P, 0);
  for (int r = 0; r < P; ++r) {
    const long long* row = table.data() + (size_t)r * (M + P);
    for (int i = 0; i < M; ++i) w[(size_t)r][(size_t)i] = row[i];
    for (int q = 0; q < P; ++q) forced[(size_t)q] += row[M + q];
  }
  std::vector<std::vector<PlanEntry>> plan;
  diffuse(b->masks, w, forced, tol, step_factor, plan, b->last_W);
  b->last_plan = plan[(size_t)b->rank];
  return PP_OK;
}

}  // namespace

extern "C" {

pp_balancer* pp_balancer_create(pp_picpart* part) {
  if (!part) {
    pp::set_error("pp_balancer_create: null part");
    return nullptr;
  }
  const int P = part->nranks, rank = part->rank;
  if (P > 64) {
    pp::set_error("pp_balancer_create: an sbar is a 64-bit set of parts -- more than 64 ranks are not supported");
    return nullptr;
  }
  const pp_mesh* full = part->full;
  const int ne = full->nelems;
  pp_balancer* b = new pp_balancer();
  b->part = part;
  b->rank = rank;
  b->nranks = P;
  // every rank's safe zone and buffer from the Input (Mesh::Mesh(Input&) for rank q)
  std::vector<unsigned long long> mask((size_t)ne);
  for (int e = 0; e < ne; ++e) mask[(size_t)e] = 1ull << part->owner_e[(size_t)e];
  pp::DevBuf d_owner, d_safe_tmp;
  if (upload(d_owner, part->owner_e) != PP_OK || d_safe_tmp.reserve((size_t)std::max(ne, 1)) != hipSuccess) {
    delete b;
    return nullptr;
  }
  std::vector<int> has_part;
  std::vector<unsigned char> is_safe;
  for (int q = 0; q < P; ++q) {
    if (safe_and_buffer(full, part->bridge_dim, q, P, part->buffer_method, part->safe_method, part->buffer_layers,
                        part->safe_layers, d_owner.as<int>(), d_safe_tmp, has_part, &is_safe) != PP_OK) {
      delete b;
      return nullptr;
    }
    for (int e = 0; e < ne; ++e)
      if (is_safe[(size_t)e] && has_part[(size_t)part->owner_e[(size_t)e]]) mask[(size_t)e] |= 1ull << q;
  }
  b->masks = mask;
  std::sort(b->masks.begin(), b->masks.end());
  b->masks.erase(std::unique(b->masks.begin(), b->masks.end()), b->masks.end());
  const DimData& de = part->D[1];
  b->sbar_ids.resize((size_t)de.nents);
  b->sbar_me.resize((size_t)de.nents);
  for (int i = 0; i < de.nents; ++i) {
    const unsigned long long m = mask[(size_t)de.full_ids[(size_t)i]];
    const int idx = (int)(std::lower_bound(b->masks.begin(), b->masks.end(), m) - b->masks.begin());
    b->sbar_ids[(size_t)i] = idx;
    b->sbar_me[(size_t)i] = ((m >> rank) & 1ull) ? idx : -1;
  }
  if (upload(b->d_sbar_ids, b->sbar_ids) != PP_OK || upload(b->d_sbar_me, b->sbar_me) != PP_OK) {
    delete b;
    return nullptr;
  }
  return b;
}

int pp_balancer_destroy(pp_balancer* b) {
  delete b;
  return PP_OK;
}

int pp_balancer_num_sbars(const pp_balancer* b) { return b ? (int)b->masks.size() : 0; }

int pp_balancer_sbars(const pp_balancer* b, unsigned long long* masks_host) {
  PP_REQUIRE(b && masks_host, "pp_balancer_sbars: null argument");
  std::copy(b->masks.begin(), b->masks.end(), masks_host);
  return PP_OK;
}

const int* pp_balancer_sbar_ids_dev(const pp_balancer* b, size_t* n) {
  if (!b) return nullptr;
  if (n) *n = b->sbar_ids.size();
  return b->d_sbar_ids.as<int>();
}

int pp_balancer_repartition_begin(pp_balancer* b, const pp_ps* ps, const int* new_elems_dev, int* new_procs_dev) {
  pp::Range rg_("balancer_weights");
  PP_REQUIRE(b && ps && (ps->capacity == 0 || (new_elems_dev && new_procs_dev)), "pp_balancer_repartition: null argument");
  PP_REQUIRE(!b->active, "pp_balancer_repartition: the previous call is still between its two halves");
  PP_REQUIRE(ps->num_elems == b->part->D[1].nents, "pp_balancer_repartition: the structure is not over the part's elements");
  hipStream_t st = pp::stream();
  const int M = (int)b->masks.size(), P = b->nranks;
  PP_HIP_CHECK(b->d_w.reserve(sizeof(int) * (size_t)(M + P + 1)));
  PP_HIP_CHECK(hipMemsetAsync(b->d_w.p, 0, sizeof(int) * (size_t)(M + P + 1), st));
  int* w = b->d_w.as<int>();
  if (ps->capacity > 0 && ps->num_ptcls > 0)
    k_bal_weights<<<grid_for(ps->capacity), kBlock, 0, st>>>(ps->capacity, ps->d_mask.as<unsigned char>(), new_elems_dev,
                                                            new_procs_dev, b->rank, P, ps->num_elems,
                                                            b->d_sbar_me.as<int>(), w, w + M, w + M + P);
  PP_LAUNCH_CHECK();
  std::vector<int> h((size_t)(M + P + 1));
  PP_HIP_CHECK(hipMemcpyAsync(h.data(), w, sizeof(int) * h.size(), hipMemcpyDeviceToHost, st));
  PP_HIP_CHECK(hipStreamSynchronize(st));
  PP_REQUIRE(!h[(size_t)(M + P)], "pp_balancer_repartition: a new element or new process is out of range");
  b->row.assign(h.begin(), h.begin() + (M + P));
  b->mode = 1;
  b->ps = ps;
  b->new_elems = new_elems_dev;
  b->new_procs = new_procs_dev;
  std::vector<long long> none;
  int rc = gather_rows(b, none, true, false);
  if (rc) return rc;
  b->active = true;
  return PP_OK;
}

int pp_balancer_repartition_end(pp_balancer* b, double tol, double step_factor) {
  pp::Range rg_("balancer_select");
  PP_REQUIRE(b && b->active && b->mode == 1, "pp_balancer_repartition_end: call pp_balancer_repartition_begin first");
  b->active = false;
  if (b->nranks == 1) return PP_OK;  // pumipic_lb.hpp:356-357
  std::vector<long long> table;
  int rc = gather_rows(b, table, false, true);
  if (rc) return rc;
  if ((rc = plan_from_table(b, table, tol, step_factor))) return rc;
  if (b->last_plan.empty()) return PP_OK;
  const int M = (int)b->masks.size();
  std::vector<int> off((size_t)M + 1, 0), target, remaining;
  for (const PlanEntry& e : b->last_plan) ++off[(size_t)e.sbar + 1];
  for (int i = 0; i < M; ++i) off[(size_t)i + 1] += off[(size_t)i];
  for (const PlanEntry& e : b->last_plan) {  // already sbar-major, target ascending
    target.push_back(e.target);
    remaining.push_back((int)e.amount);
  }
  std::vector<int> cursor(off.begin(), off.end() - 1);
  if ((rc = upload(b->d_off, off)) || (rc = upload(b->d_cursor, cursor)) || (rc = upload(b->d_target, target)) ||
      (rc = upload(b->d_remaining, remaining)))
    return rc;
  hipStream_t st = pp::stream();
  const pp_ps* ps = b->ps;
  for (int pass = 0; pass < 2; ++pass)  // selectNonCoreParticles, then selectParticles (:262-298)
    if (ps->capacity > 0)
      k_bal_select<<<grid_for(ps->capacity), kBlock, 0, st>>>(
          ps->capacity, ps->d_mask.as<unsigned char>(), b->new_elems, b->new_procs, b->rank,
          b->part->D[1].d_owners.as<int>(), b->d_sbar_me.as<int>(), pass == 0 ? 1 : 0, b->d_off.as<int>(),
          b->d_cursor.as<int>(), b->d_target.as<int>(), b->d_remaining.as<int>());
  PP_LAUNCH_CHECK();
  return PP_OK;
}

int pp_balancer_repartition(pp_balancer* b, const pp_ps* ps, double tol, const int* new_elems_dev, int* new_procs_dev,
                            double step_factor) {
  PP_REQUIRE(b, "pp_balancer_repartition: null balancer");
  PP_REQUIRE(!b->part->comm || b->part->comm->kind != 4,
             "pp_balancer_repartition: virtual ranks of one process call _begin on every rank, then _end");
  int rc = pp_balancer_repartition_begin(b, ps, new_elems_dev, new_procs_dev);
  if (rc) return rc;
  return pp_balancer_repartition_end(b, tol, step_factor);
}

int pp_balancer_partition_begin(pp_balancer* b, const int* ptcls_per_elem_host) {
  PP_REQUIRE(b && ptcls_per_elem_host, "pp_balancer_partition: null argument");
  PP_REQUIRE(!b->active, "pp_balancer_partition: the previous call is still between its two halves");
  const int M = (int)b->masks.size(), P = b->nranks, ne = b->part->D[1].nents;
  b->ppe.assign(ptcls_per_elem_host, ptcls_per_elem_host + ne);
  b->row.assign((size_t)(M + P), 0);
  for (int e = 0; e < ne; ++e) {  // addWeights(picparts, ptcls_per_elem), pumipic_lb.hpp:219-237
    PP_REQUIRE(b->ppe[(size_t)e] >= 0, "pp_balancer_partition: negative particle count");
    if (b->sbar_me[(size_t)e] >= 0) b->row[(size_t)b->sbar_me[(size_t)e]] += b->ppe[(size_t)e];
  }
  b->mode = 2;
  std::vector<long long> none;
  int rc = gather_rows(b, none, true, false);
  if (rc) return rc;
  b->active = true;
  return PP_OK;
}

int pp_balancer_partition_end(pp_balancer* b, double tol, double step_factor, int* new_procs_host) {
  PP_REQUIRE(b && b->active && b->mode == 2 && new_procs_host, "pp_balancer_partition_end: call pp_balancer_partition_begin first");
  b->active = false;
  long long np = 0;
  for (int n : b->ppe) np += n;
  for (long long i = 0; i < np; ++i) new_procs_host[i] = b->rank;  // setSelf :392-395
  if (b->nranks == 1) return PP_OK;
  std::vector<long long> table;
  int rc = gather_rows(b, table, false, true);
  if (rc) return rc;
  if ((rc = plan_from_table(b, table, tol, step_factor))) return rc;
  // selectParticles over the array (:379-...): elements in order, the plan's entries of the element's sbar in order
  const int M = (int)b->masks.size();
  std::vector<std::vector<std::pair<int, long long>>> per((size_t)M);
  for (const PlanEntry& e : b->last_plan) per[(size_t)e.sbar].push_back({e.target, e.amount});
  std::vector<size_t> cur((size_t)M, 0);
  long long at = 0;
  for (size_t e = 0; e < b->ppe.size(); ++e) {
    const int s = b->sbar_me[e];
    for (int i = 0; i < b->ppe[e]; ++i, ++at) {
      if (s < 0) continue;
      auto& lst = per[(size_t)s];
      while (cur[(size_t)s] < lst.size() && lst[cur[(size_t)s]].second <= 0) ++cur[(size_t)s];
      if (cur[(size_t)s] >= lst.size()) continue;
      new_procs_host[at] = lst[cur[(size_t)s]].first;
      --lst[cur[(size_t)s]].second;
    }
  }
  return PP_OK;
}

int pp_balancer_partition(pp_balancer* b, const int* ptcls_per_elem_host, double tol, double step_factor,
                          int* new_procs_host) {
  PP_REQUIRE(b, "pp_balancer_partition: null balancer");
  PP_REQUIRE(!b->part->comm || b->part->comm->kind != 4,
             "pp_balancer_partition: virtual ranks of one process call _begin on every rank, then _end");
  int rc = pp_balancer_partition_begin(b, ptcls_per_elem_host);
  if (rc) return rc;
  return pp_balancer_partition_end(b, tol, step_factor, new_procs_host);
}

int pp_balancer_last_plan(const pp_balancer* b, int* n, int* sbar_host, int* target_host, long long* amount_host,
                          long long* weights_after_host) {
  PP_REQUIRE(b && n, "pp_balancer_last_plan: null argument");
  *n = (int)b->last_plan.size();
  for (size_t i = 0; i < b->last_plan.size(); ++i) {
    if (sbar_host) sbar_host[i] = b->last_plan[i].sbar;
    if (target_host) target_host[i] = b->last_plan[i].target;
    if (amount_host) amount_host[i] = b->last_plan[i].amount;
  }
  if (weights_after_host) std::copy(b->last_W.begin(), b->last_W.end(), weights_after_host);
  return PP_OK;
}

}  // extern "C"
