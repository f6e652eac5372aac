"""ctypes binding of libpumipic_hip.so (include/pumipic_hip.h).

Python is only the test/bench harness; this module adds no compute of its own and has NO CPU
fallback: if the HIP library is missing or no GPU is visible every entry point raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PUMIPIC_HIP_LIB: another build of the same library, e.g. the laboratory build `make -C csrc lab`)
LIB_PATH = os.environ.get("PUMIPIC_HIP_LIB") or os.path.join(_HERE, "libpumipic_hip.so")
CSRC = os.path.join(_HERE, "csrc")

c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)
c_void_pp = C.POINTER(C.c_void_p)

PAD_EVENLY, PAD_PROPORTIONALLY, PAD_INVERSELY = 0, 1, 2
SCS, CSR = 0, 1

(MESH_COORDS, MESH_ELEM2VERTS, MESH_CLASS_ID, MESH_ELEM2SIDES, MESH_SIDE2VERTS,
 MESH_SIDE2ELEMS_OFF, MESH_SIDE2ELEMS, MESH_SIDE_EXPOSED, MESH_ELEM_MEASURE, MESH_DUAL_OFF,
 MESH_DUAL_ELEMS, MESH_VERT2ELEMS_OFF, MESH_VERT2ELEMS, MESH_ELEM_RECORDS, MESH_ELEM2EDGES, MESH_EDGE2VERTS,
 MESH_EDGE2ELEMS_OFF, MESH_EDGE2ELEMS) = range(18)
_MESH_DTYPES = {MESH_COORDS: np.float64, MESH_ELEM2VERTS: np.int32, MESH_CLASS_ID: np.int32,
                MESH_ELEM2SIDES: np.int32, MESH_SIDE2VERTS: np.int32,
                MESH_SIDE2ELEMS_OFF: np.int32, MESH_SIDE2ELEMS: np.int32,
                MESH_SIDE_EXPOSED: np.int8, MESH_ELEM_MEASURE: np.float64, MESH_DUAL_OFF: np.int32,
                MESH_DUAL_ELEMS: np.int32, MESH_VERT2ELEMS_OFF: np.int32,
                MESH_VERT2ELEMS: np.int32, MESH_ELEM2EDGES: np.int32, MESH_EDGE2VERTS: np.int32,
                MESH_EDGE2ELEMS_OFF: np.int32, MESH_EDGE2ELEMS: np.int32}


class PPError(RuntimeError):
    pass


class PsInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("num_elems", C.c_int), ("num_ptcls", C.c_int),
                ("capacity", C.c_int), ("num_rows", C.c_int), ("C", C.c_int), ("V", C.c_int),
                ("sigma", C.c_int), ("num_chunks", C.c_int), ("num_slices", C.c_int),
                ("nmembers", C.c_int), ("stride", C.c_int64)]


class PsLayout(C.Structure):
    _fields_ = [("offsets", C.c_void_p), ("slice_to_chunk", C.c_void_p),
                ("row_to_element", C.c_void_p), ("element_to_row", C.c_void_p),
                ("mask", C.c_void_p), ("slot_elem", C.c_void_p)]


# every symbol include/pumipic_hip.h declares: name -> (restype, argtypes)
_V, _I, _D, _S = C.c_void_p, C.c_int, C.c_double, C.c_size_t
SYMBOLS = {
    "pp_last_error": (C.c_char_p, []),
    "pp_version": (C.c_char_p, []),
    "pp_init": (_I, [_I]),
    "pp_stream": (_V, []),
    "pp_sync": (_I, []),
    "pp_peek_hip_error": (_I, [C.POINTER(C.c_char_p)]),
    "pp_device_count": (_I, []),
    "pp_malloc": (_V, [_S]),
    "pp_free": (_I, [_V]),
    "pp_memcpy_h2d": (_I, [_V, _V, _S]),
    "pp_memcpy_d2h": (_I, [_V, _V, _S]),
    "pp_memset": (_I, [_V, _I, _S]),
    "pp_pool_trim": (_I, []),
    "pp_pool_set_limit": (_I, [_S]),
    "pp_pool_stats": (_I, [C.POINTER(_S), C.POINTER(_S), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "pp_fill": (_I, [_V, _V, _I, _S]),
    "pp_event_create": (_V, []),
    "pp_event_record": (_I, [_V]),
    "pp_event_elapsed_ms": (C.c_float, [_V, _V]),
    "pp_event_destroy": (_I, [_V]),
    "pp_mesh_create": (_V, [_I, _I, _V, _I, _V, _V]),
    "pp_mesh_destroy": (_I, [_V]),
    "pp_mesh_info": (_I, [_V, c_int_p, c_int_p, c_int_p, c_int_p]),
    "pp_mesh_num_edges": (_I, [_V]),
    "pp_mesh_tolerance": (_D, [_V]),
    "pp_mesh_array_dev": (_V, [_V, _I, C.POINTER(_S)]),
    "pp_mesh_array_to_host": (_I, [_V, _I, _V]),
    "pp_ps_create_scs": (_V, [_I, _I, _I, _I, _I, _V, _V, _I, _D, _D, _I, _V, _V, _V, _V]),
    "pp_ps_create_csr": (_V, [_I, _I, _V, _V, _D, _I, _V, _V, _V, _V]),
    "pp_ps_destroy": (_I, [_V]),
    "pp_ps_clone": (_V, [_V]),
    "pp_ps_info": (_I, [_V, C.POINTER(PsInfo)]),
    "pp_ps_member_ptr": (_V, [_V, _I]),
    "pp_ps_member_stride": (C.c_int64, [_V]),
    "pp_ps_layout": (_I, [_V, C.POINTER(PsLayout)]),
    "pp_ps_iteration": (_I, [_V, _V]),
    "pp_ps_layout_to_host": (_I, [_V, _V, _V, _V, _V, _V, _V]),
    "pp_ps_gids_to_host": (_I, [_V, _V]),
    "pp_ps_last_search_found": (_I, [_V, c_int_p]),
    "pp_ps_member_to_host": (_I, [_V, _I, _V]),
    "pp_ps_member_from_host": (_I, [_V, _I, _V]),
    "pp_ps_rebuild": (_I, [_V, _V, _I, _V, _V]),
    "pp_ps_rebuild_commit": (_I, [_V, _I, _I, _V, _I, _V, _V]),
    "pp_ps_get_pids": (_I, [_V, _V, _V]),
    "pp_ps_set_shuffling": (_I, [_V, _I]),
    "pp_ps_rebuild_stats": (_I, [_V, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "pp_ps_deferred_state": (_I, [_V, C.POINTER(C.c_int * 6)]),
    "pp_ps_materialize": (_I, [_V]),
    "pp_ps_metrics": (_I, [_V, c_int_p, c_int_p, c_int_p]),
    "pp_ps_swap_members": (_I, [_V, _I, _I]),
    "pp_elliptical_setup": (_I, [_V, _I, _I, _I, _D, _D, _D]),
    "pp_elliptical_push": (_I, [_V, _V, _I, _I, _I, _D, _D, _D, _D]),
    "pp_toroidal_push": (_I, [_V, _V, _I, _I, _I, _I, _D, _D, _D, _D]),
    "pp_linear_push": (_I, [_V, _I, _I, _D, _D, _D, _D]),
    "pp_push_boris": (_I, [_I] + [_V] * 15 + [_D]),
    "pp_update_positions": (_I, [_V, _I, _I]),
    "pp_pseudo_push160": (_I, [_V, _V]),
    "pp_search_mesh_2d": (_I, [_V, _V, _I, _I, _I, _V, _I, c_int_p]),
    "pp_search_mesh": (_I, [_V, _V, _I, _I, _I, _V, _I, _I, _V, _V, _I, c_int_p, c_int_p]),
    "pp_search_mesh_legacy3d": (_I, [_V, _V, _I, _I, _I, _V, _I, _V, _V, _I, c_int_p]),
    "pp_closest_point_on_triangle": (_I, [_I, _V, _I, _V, _I, _V, _V]),
    "pp_trace_begin": (_I, [_V, _V, _I, _I, _V, _I, _I, _V, _V, _V, _V, c_int_p]),
    "pp_trace_find_exit_face": (_I, [_V, _V, _I, _I, _V, _V, _V, _V, _I]),
    "pp_trace_check_model_intersection": (_I, [_V, _V, _V, _V, _V, _I, _V]),
    "pp_trace_set_new_element": (_I, [_V, _V, _V, _V, _V, c_int_p]),
    "pp_trace_not_found": (_I, [_V, _V, _V, c_int_p]),
    "pp_redistribute_particles": (_I, [_V, C.c_double, C.c_ulonglong, _V]),
    "pp_redistribute_particles_dist": (_I, [_V, _I, C.c_double, C.c_ulonglong, _V]),
    "pp_boris_push_fields": (_I, [_V, _V, _I, _I, _I, _V, _V, _V] + [C.c_double] * 4 + [_I, _I, _I, C.c_double, c_int_p]),
    "pp_bfs_buffer_layers": (_I, [_V, _I, _I, _I, _I, _I, _V, _V, c_int_p]),
    "pp_bfs_safe_inward": (_I, [_V, _I, _I, _I, _I, _V, c_int_p, _V]),
    "pp_ps_rebuild_scatter": (_I, [_V, _I, _I, _V, _I, _V, _V, _V, _I, _V, _V, C.c_double, _I, _I]),
    "pp_ps_migrate_pack_records_commit": (_I, [_V, _I, _I, _V, _V, _I, _I, c_int_p, _V]),
    "pp_ps_rebuild_records_scatter": (_I, [_V, _I, _I, _V, _I, _V, _V, C.c_int64, _V, _I, _V, _V,
                                           C.c_double, _I, _I]),
    "pp_search_mesh_3d": (_I, [_V, _V, _I, _I, _I, _V, _I, _V, _V, _I, c_int_p]),
    "pp_push_search": (_I, [_V, _V, _I, _I, _I, _I, _D, _D, _D, _D, _V, _I, _I, c_int_p]),
    "pp_ps_set_origin_trust": (_I, [_V, _I]),
    "pp_search_walk_steps": (_I, [_V]),
    "pp_push_search_counters": (_I, [c_int_p, c_int_p, c_int_p]),
    "pp_create_gyro_ring_mappings": (_I, [_V, _D, _I, _I, _D, _V, _V]),
    "pp_gyro_scatter": (_I, [_V, _V, _V, _D, _I, _I, _V]),
    "pp_gyro_sync_pack": (_I, [_I, _V, _V, _V]),
    "pp_gyro_map_forget": (_I, [_V]),
    "pp_ray_intersects_triangle": (_I, [_I, _V, _I, _V, _V, _D, _V, _I, _I, _V, _V, _V]),
    "pp_owner_by_classification": (_I, [_V, c_int_p, _I, _I, c_int_p]),
    "pp_picpart_create": (_V, [_V, c_int_p, _I, _I, _I, _I, _I, _V]),
    "pp_picpart_destroy": (_I, [_V]),
    "pp_picpart_mesh": (_V, [_V]),
    "pp_picpart_info": (_I, [_V, c_int_p, c_int_p, c_int_p, c_int_p]),
    "pp_picpart_array_dev": (_V, [_V, _I, _I, C.POINTER(C.c_size_t)]),
    "pp_picpart_array_to_host": (_I, [_V, _I, _I, _V]),
    "pp_picpart_nents_offsets": (_I, [_V, _I, c_int_p]),
    "pp_picpart_buffered_ranks": (_I, [_V, _I, c_int_p, c_int_p]),
    "pp_picpart_bounded": (_I, [_V, _I, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "pp_picpart_num_global": (C.c_longlong, [_V, _I]),
    "pp_picpart_complete_parts": (_I, [_V, _I, c_int_p]),
    "pp_picpart_reduce": (_I, [_V, _I, _I, _I, _I, _V]),
    "pp_picpart_reduce_begin": (_I, [_V, _I, _I, _I, _I, _V]),
    "pp_picpart_reduce_mid": (_I, [_V]),
    "pp_picpart_reduce_end": (_I, [_V]),
    "pp_balancer_create": (_V, [_V]),
    "pp_balancer_destroy": (_I, [_V]),
    "pp_balancer_num_sbars": (_I, [_V]),
    "pp_balancer_sbars": (_I, [_V, _V]),
    "pp_balancer_sbar_ids_dev": (_V, [_V, C.POINTER(C.c_size_t)]),
    "pp_balancer_repartition": (_I, [_V, _V, _D, _V, _V, _D]),
    "pp_balancer_repartition_begin": (_I, [_V, _V, _V, _V]),
    "pp_balancer_repartition_end": (_I, [_V, _D, _D]),
    "pp_balancer_partition": (_I, [_V, c_int_p, _D, _D, c_int_p]),
    "pp_balancer_partition_begin": (_I, [_V, c_int_p]),
    "pp_balancer_partition_end": (_I, [_V, _D, _D, c_int_p]),
    "pp_balancer_last_plan": (_I, [_V, c_int_p, c_int_p, c_int_p, _V, _V]),
    "pp_gyro_scatter_radius": (_I, [_V, _V, _V, _V, _V, _D, _I, _I, _V, c_int_p]),
    "pp_gather_tet_vtx": (_I, [_V, _V, _I, _V, _V, _I, _V, _V]),
    "pp_interp2d_field": (_I, [_V, _I, _V, _D, _D, _D, _D, _I, _I, _I, _I, _I, _V]),
    "pp_interp2d_vector": (_I, [_V, _I, _V, _D, _D, _D, _D, _I, _I, _I, _V]),
    "pp_interp3d_field": (_I, [_V, _I, _I, _I, _I, _V, _V, _V, _V, _V]),
    "pp_avg_ptcl_density": (_I, [_V, _V, _V, _V]),
    "pp_set_unsafe_procs": (_I, [_V, _V, _V, _V, _I, _V, _V]),
    "pp_ps_migrate_count": (_I, [_V, _V, _V, _I, _I, _V]),
    "pp_ps_migrate_pack": (_I, [_V, _V, _V, _I, _I, _V, _V, _V]),
    "pp_ps_migrate_record_bytes": (_I, [_V]),
    "pp_ps_migrate_pack_records": (_I, [_V, _V, _V, _I, _I, _V, _V]),
    "pp_ps_rebuild_records": (_I, [_V, _V, _I, _V, _V, C.c_int64]),
    "pp_comm_unique_id": (_I, [_V]),
    "pp_comm_create_rccl": (_V, [_V, _I, _I]),
    "pp_comm_create_tcp": (_V, [C.c_char_p, _I, _I, _I]),
    "pp_comm_create_host": (_V, [_V, _V, _I, _I]),
    "pp_comm_create_local": (_I, [_I, c_void_pp]),
    "pp_comm_create_env": (_V, []),
    "pp_comm_rank": (_I, [_V]),
    "pp_comm_size": (_I, [_V]),
    "pp_comm_kind": (C.c_char_p, [_V]),
    "pp_comm_destroy": (_I, [_V]),
    "pp_bootstrap_broadcast": (_I, [C.c_char_p, _I, _I, _I, _V, _I]),
    "pp_comm_exchange_counts": (_I, [_V, c_int_p, c_int_p]),
    "pp_migrate_plan": (_I, [_I, _I, c_int_p, c_int_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                             C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pp_allreduce_sum": (_I, [_V, _V, C.c_int64]),
    "pp_allreduce_sum_host_i64": (_I, [_V, C.POINTER(C.c_int64), _I]),
    "pp_comm_barrier": (_I, [_V]),
    "pp_comm_selftest": (_I, [_V, _I]),
    "pp_comm_allgather_host": (_I, [_V, _V, _V, _I]),
    "pp_ps_migrate": (_I, [_V, _V, _V, _V]),
    "pp_ps_migrate_scatter": (_I, [_V, _I, _I, _V, _V, _V, _I, _V, _V, _V, C.c_int64, _V, _I, _V, _V,
                                   C.c_double, _I, _I]),
    "pp_ps_migrate_begin": (_I, [_V, _I, _I, _V, _V, _V, _I, _V, _V, _V, C.c_int64, _V, _I, _V, _V,
                                 C.c_double, _I, _I]),
    "pp_ps_migrate_end": (_I, [_V, _V, c_int_p, c_int_p]),
    "pp_migrate_ptcls": (_I, [_V, _I, _I, _V, _V, _V, _V, _V, _I, _V, _V, C.c_double, _I, _I]),
    "pp_migrate_ptcls_begin": (_I, [_V, _I, _I, _V, _V, _V, _V, _V, _I, _V, _V, C.c_double, _I, _I]),
    "pp_range_push": (_I, [C.c_char_p]),
    "pp_range_pop": (_I, []),
}


def build(force=False):
    """Compile every HIP source for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j8"])
    return LIB_PATH


_lib = None


def lib():
    """Load the HIP library; loud failure when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PPError("libpumipic_hip.so is not built (run __graft_entry__.build()); "
                          "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError = missing export
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise PPError("pumipic_hip error %d: %s" % (rc, lib().pp_last_error().decode()))


def init(device=0):
    check(lib().pp_init(device))


def sync():
    check(lib().pp_sync())


def peek_hip_error():
    """(code, message) of the HIP runtime's sticky last error, not cleared; (0, '') = none"""
    msg = C.c_char_p()
    rc = lib().pp_peek_hip_error(C.byref(msg))
    return rc, (msg.value or b"").decode()


# ------------------------------------------------------------------ device arrays
class DevArray:
    """Typed device buffer owned by Python (hipMalloc via the C-ABI)."""

    def __init__(self, n, dtype):
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.ptr = lib().pp_malloc(max(self.n * self.dtype.itemsize, 1))
        if not self.ptr:
            raise PPError("pp_malloc failed: " + lib().pp_last_error().decode())

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.size, a.dtype)
        if a.size:
            check(lib().pp_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes))
        return d

    def to_host(self):
        sync()
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            check(lib().pp_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.size <= self.n
        if a.size:
            check(lib().pp_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes))

    def fill_bytes(self, value):
        check(lib().pp_memset(self.ptr, value, self.n * self.dtype.itemsize))

    def __del__(self):
        try:
            if self.ptr:
                lib().pp_free(self.ptr)
        except Exception:
            pass


class Event:
    def __init__(self):
        self.ev = lib().pp_event_create()

    def record(self):
        check(lib().pp_event_record(self.ev))

    def elapsed_ms(self, stop):
        return lib().pp_event_elapsed_ms(self.ev, stop.ev)

    def __del__(self):
        try:
            lib().pp_event_destroy(self.ev)
        except Exception:
            pass


# ------------------------------------------------------------------ mesh
class Mesh:
    def __init__(self, dim, coords, elem2verts, class_id=None):
        coords = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, dim)
        e2v = np.ascontiguousarray(elem2verts, dtype=np.int32).reshape(-1, dim + 1)
        cid = None if class_id is None else np.ascontiguousarray(class_id, dtype=np.int32)
        self.p = lib().pp_mesh_create(dim, coords.shape[0], coords.ctypes.data, e2v.shape[0],
                                      e2v.ctypes.data, cid.ctypes.data if cid is not None else None)
        if not self.p:
            raise PPError("pp_mesh_create: " + lib().pp_last_error().decode())
        d, nv, ne, ns = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().pp_mesh_info(self.p, C.byref(d), C.byref(nv), C.byref(ne), C.byref(ns)))
        self.dim, self.nverts, self.nelems, self.nsides = d.value, nv.value, ne.value, ns.value

    def tolerance(self):
        return lib().pp_mesh_tolerance(self.p)

    def num_edges(self):
        """Omega_h Mesh::nedges(): sides of a triangle mesh, derived edges of a tet mesh"""
        n = lib().pp_mesh_num_edges(self.p)
        if n < 0:
            check(n)
        return n

    def array(self, which):
        cnt = C.c_size_t()
        lib().pp_mesh_array_dev(self.p, which, C.byref(cnt))
        out = np.empty(cnt.value, dtype=_MESH_DTYPES[which])
        if cnt.value:
            check(lib().pp_mesh_array_to_host(self.p, which, out.ctypes.data))
        return out

    def __del__(self):
        try:
            lib().pp_mesh_destroy(self.p)
        except Exception:
            pass


# ------------------------------------------------------------------ particle structure
PARTICLE_XGCM = [(np.float64, 3), (np.float64, 3), (np.int32, 1), (np.float32, 1), (np.float32, 1)]
PARTICLE_PUSH = [(np.float64, 3), (np.float64, 3), (np.int32, 1)]
PERF160 = [(np.float64, 17), (np.int32, 4), (np.int64, 1)]


def _meta(members):
    mb = np.array([np.dtype(d).itemsize for d, _ in members], dtype=np.int32)
    mc = np.array([n for _, n in members], dtype=np.int32)
    return mb, mc


def _host_info(members, info, n):
    if info is None:
        return None, None
    keep, arr = [], (C.c_void_p * len(members))()
    for i, ((dt, nc), a) in enumerate(zip(members, info)):
        a = np.ascontiguousarray(np.asarray(a, dtype=dt).reshape(nc, n))
        keep.append(a)
        arr[i] = a.ctypes.data
    return arr, keep


class PS:
    def __init__(self, p, members):
        if not p:
            raise PPError("particle structure creation failed: " + lib().pp_last_error().decode())
        self.p = p
        self.members = members

    @classmethod
    def scs(cls, members, ne, ppe, C_=64, sigma=2**31 - 1, V=1024, gids=None, pad_strat=0,
            shuffle_padding=0.1, extra_padding=0.05, particle_elements=None, particle_info=None):
        ppe = np.ascontiguousarray(ppe, dtype=np.int32)
        np_ = int(ppe.sum())
        mb, mc = _meta(members)
        pe = None if particle_elements is None else np.ascontiguousarray(particle_elements, np.int32)
        arr, keep = _host_info(members, particle_info, np_)
        g = None if gids is None else np.ascontiguousarray(gids, dtype=np.int64)
        p = lib().pp_ps_create_scs(C_, sigma, V, ne, np_, ppe.ctypes.data,
                                   g.ctypes.data if g is not None else None, pad_strat,
                                   shuffle_padding, extra_padding, len(members), mb.ctypes.data,
                                   mc.ctypes.data, pe.ctypes.data if pe is not None else None,
                                   C.cast(arr, C.c_void_p) if arr is not None else None)
        return cls(p, members)

    @classmethod
    def csr(cls, members, ne, ppe, gids=None, padding_amount=1.05, particle_elements=None,
            particle_info=None):
        ppe = np.ascontiguousarray(ppe, dtype=np.int32)
        np_ = int(ppe.sum())
        mb, mc = _meta(members)
        pe = None if particle_elements is None else np.ascontiguousarray(particle_elements, np.int32)
        arr, keep = _host_info(members, particle_info, np_)
        g = None if gids is None else np.ascontiguousarray(gids, dtype=np.int64)
        p = lib().pp_ps_create_csr(ne, np_, ppe.ctypes.data,
                                   g.ctypes.data if g is not None else None, padding_amount,
                                   len(members), mb.ctypes.data, mc.ctypes.data,
                                   pe.ctypes.data if pe is not None else None,
                                   C.cast(arr, C.c_void_p) if arr is not None else None)
        return cls(p, members)

    def info(self):
        i = PsInfo()
        check(lib().pp_ps_info(self.p, C.byref(i)))
        return i

    def capacity(self):
        return self.info().capacity

    def nPtcls(self):
        return self.info().num_ptcls

    def nElems(self):
        return self.info().num_elems

    def numRows(self):
        return self.info().num_rows

    def member(self, m):
        """host copy (ncomp, stride) of member m"""
        dt, nc = self.members[m]
        st = self.info().stride
        out = np.empty((nc, st), dtype=dt)
        check(lib().pp_ps_member_to_host(self.p, m, out.ctypes.data))
        return out

    def set_member(self, m, a):
        dt, nc = self.members[m]
        st = self.info().stride
        a = np.ascontiguousarray(np.asarray(a, dtype=dt).reshape(nc, st))
        check(lib().pp_ps_member_from_host(self.p, m, a.ctypes.data))

    def member_ptr(self, m):
        return lib().pp_ps_member_ptr(self.p, m)

    def swap_members(self, a, b):
        check(lib().pp_ps_swap_members(self.p, a, b))

    def layout(self):
        i = self.info()
        d = dict(kind=i.kind, C=i.C, V=i.V, num_chunks=i.num_chunks, num_slices=i.num_slices,
                 capacity=i.capacity, num_rows=i.num_rows)
        noff = (i.num_slices + 1) if i.kind == SCS else (i.num_elems + 1)
        off = np.zeros(noff, dtype=np.int32)
        s2c = np.zeros(max(i.num_slices, 1), dtype=np.int32)
        r2e = np.zeros(max(i.num_rows, 1), dtype=np.int32)
        e2r = np.zeros(max(i.num_rows, 1), dtype=np.int32)
        mask = np.zeros(max(i.capacity, 1), dtype=np.uint8)
        se = np.zeros(max(i.capacity, 1), dtype=np.int32)
        check(lib().pp_ps_layout_to_host(self.p, off.ctypes.data, s2c.ctypes.data, r2e.ctypes.data,
                                         e2r.ctypes.data, mask.ctypes.data, se.ctypes.data))
        d["offsets"] = off
        d["mask"] = mask[:i.capacity]
        d["slot_elem"] = se[:i.capacity]
        if i.kind == SCS:
            d["slice_to_chunk"] = s2c[:i.num_slices]
            d["row_to_element"] = r2e[:i.num_rows]
            d["element_to_row"] = e2r[:i.num_rows]
        return d

    def slot_info(self):
        L = self.layout()
        return L["slot_elem"], L["mask"]

    def rebuild(self, new_element, new_particle_elements=None, new_particle_info=None):
        """new_element: DevArray(int32, capacity) or host array"""
        ne = new_element if isinstance(new_element, DevArray) else DevArray.from_host(
            np.ascontiguousarray(new_element, dtype=np.int32))
        n_new = 0 if new_particle_elements is None else len(new_particle_elements)
        keep = []
        arr = None
        npe = None
        if n_new:
            npe = DevArray.from_host(np.ascontiguousarray(new_particle_elements, dtype=np.int32))
            arr = (C.c_void_p * len(self.members))()
            for i, ((dt, nc), a) in enumerate(zip(self.members, new_particle_info)):
                d = DevArray.from_host(np.ascontiguousarray(np.asarray(a, dtype=dt).reshape(nc, n_new)))
                keep.append(d)
                arr[i] = d.ptr
        check(lib().pp_ps_rebuild(self.p, ne.ptr, n_new, npe.ptr if npe else None,
                                  C.cast(arr, C.c_void_p) if arr is not None else None))
        # (the kernels of the call may still read its inputs when it returns -- include/pumipic_hip.h, "lifetime of
        #  the inputs": wait only when this wrapper made the device copies that die with it; a caller that passes
        #  its own DevArray and no new particles keeps the step free of the wait, as the C++ callers are)
        if ne is not new_element or n_new:
            sync()

    def rebuild_commit(self, new_element, m_x=0, m_xtgt=1):
        """updatePtclPositions + rebuild in one pass (no new particles)"""
        ne = new_element if isinstance(new_element, DevArray) else DevArray.from_host(
            np.ascontiguousarray(new_element, dtype=np.int32))
        check(lib().pp_ps_rebuild_commit(self.p, m_x, m_xtgt, ne.ptr, 0, None, None))

    def set_try_shuffling(self, v):
        """False / 0: never in place; True / 1 (default): the reference's reshuffle decision"""
        check(lib().pp_ps_set_shuffling(self.p, int(v)))

    def rebuild_stats(self):
        """(rebuilds that kept the layout, full re-layouts, full re-layouts fed by the previous one's records)"""
        a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
        check(lib().pp_ps_rebuild_stats(self.p, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_origin_trust(self, on):
        check(lib().pp_ps_set_origin_trust(self.p, int(bool(on))))

    def deferred_state(self):
        """dict(lazy_rec, zero_pending, zero_z_pending, elem_count_valid, slot_elem_valid, hot_row)"""
        a = (C.c_int * 6)()
        check(lib().pp_ps_deferred_state(self.p, C.byref(a)))
        return dict(zip(("lazy_rec", "zero_pending", "zero_z_pending", "elem_count_valid", "slot_elem_valid",
                         "hot_row"), list(a)))

    def materialize(self):
        check(lib().pp_ps_materialize(self.p))

    def clone(self):
        """a deep, independent copy (pp_ps_clone: SellCSigma::copy / CSR::copy on the device)"""
        return PS(lib().pp_ps_clone(self.p), self.members)

    def get_pids(self):
        i = self.info()
        off = DevArray(i.num_elems + 1, np.int32)
        pids = DevArray(max(i.num_ptcls, 1), np.int32)
        check(lib().pp_ps_get_pids(self.p, off.ptr, pids.ptr))
        return off.to_host(), pids.to_host()[:i.num_ptcls]

    def metrics(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(lib().pp_ps_metrics(self.p, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def __del__(self):
        try:
            lib().pp_ps_destroy(self.p)
        except Exception:
            pass


# ------------------------------------------------------------------ operators
def elliptical_setup(ps, h, k, d, m_x=0, m_b=3, m_phi=4):
    check(lib().pp_elliptical_setup(ps.p, m_x, m_b, m_phi, h, k, d))


def elliptical_push(ps, mesh, h, k, d, deg, m_xtgt=1, m_b=3, m_phi=4):
    check(lib().pp_elliptical_push(ps.p, mesh.p, m_xtgt, m_b, m_phi, h, k, d, deg))


def toroidal_push(ps, mesh, h, k, d, deg, m_x=0, m_xtgt=1, m_b=3, m_phi=4):
    check(lib().pp_toroidal_push(ps.p, mesh.p, m_x, m_xtgt, m_b, m_phi, h, k, d, deg))


def linear_push(ps, distance, dx, dy, dz, m_x=0, m_xtgt=1):
    check(lib().pp_linear_push(ps.p, m_x, m_xtgt, distance, dx, dy, dz))


def update_positions(ps, m_x=0, m_xtgt=1):
    check(lib().pp_update_positions(ps.p, m_x, m_xtgt))


def pseudo_push160(ps, parent_elm_data_dev):
    check(lib().pp_pseudo_push160(ps.p, parent_elm_data_dev.ptr))


def push_boris(dev_arrays, dt):
    check(lib().pp_push_boris(dev_arrays[0].n, *[a.ptr for a in dev_arrays], dt))


def search_mesh_2d(mesh, ps, elem_ids=None, looplimit=0, m_x=0, m_xtgt=1, m_pid=2):
    cap = ps.capacity()
    if elem_ids is None:
        elem_ids = DevArray.from_host(np.full(max(cap, 1), -1, dtype=np.int32))
    found = C.c_int()
    check(lib().pp_search_mesh_2d(mesh.p, ps.p, m_x, m_xtgt, m_pid, elem_ids.ptr, looplimit,
                                  C.byref(found)))
    return bool(found.value), elem_ids


def search_mesh(mesh, ps, elem_ids=None, require_intersection=False, looplimit=0, m_x=0, m_xtgt=1,
                m_pid=2):
    cap = max(ps.capacity(), 1)
    seeded = elem_ids is not None
    if elem_ids is None:
        elem_ids = DevArray(cap, np.int32)
    # the reference allocates these itself when they come in empty: -1 / 0 everywhere (tpp:538-540)
    inter_faces = DevArray.from_host(np.full(cap, -1, dtype=np.int32))
    inter_points = DevArray.from_host(np.zeros(cap * mesh.dim))
    found, notin = C.c_int(), C.c_int()
    check(lib().pp_search_mesh(mesh.p, ps.p, m_x, m_xtgt, m_pid, elem_ids.ptr, int(seeded),
                               int(require_intersection), inter_faces.ptr, inter_points.ptr,
                               looplimit, C.byref(found), C.byref(notin)))
    return dict(found=bool(found.value), elem_ids=elem_ids, inter_faces=inter_faces,
                inter_points=inter_points, not_in_elem=notin.value)


def search_mesh_legacy3d(mesh, ps, elem_ids=None, looplimit=0, m_x=0, m_xtgt=1, m_pid=2):
    cap = max(ps.capacity(), 1)
    seeded = elem_ids is not None
    if elem_ids is None:
        elem_ids = DevArray(cap, np.int32)
    xface = DevArray.from_host(np.full(cap, -1, dtype=np.int32))
    xpoints = DevArray.from_host(np.zeros(cap * 3))
    found = C.c_int()
    check(lib().pp_search_mesh_legacy3d(mesh.p, ps.p, m_x, m_xtgt, m_pid, elem_ids.ptr,
                                        int(seeded), xpoints.ptr, xface.ptr, looplimit,
                                        C.byref(found)))
    return dict(found=found.value, elem_ids=elem_ids, xface=xface, xpoints=xpoints)


def search_mesh_3d(mesh, ps, elem_ids=None, looplimit=0, m_x=0, m_xtgt=1, m_pid=2):
    """search_mesh_3d (adjacency.hpp:314-555)."""
    cap = max(ps.capacity(), 1)
    seeded = elem_ids is not None
    if elem_ids is None:
        elem_ids = DevArray(cap, np.int32)
    xface = DevArray.from_host(np.full(cap, -1, dtype=np.int32))
    xpoints = DevArray.from_host(np.zeros(cap * 3))
    found = C.c_int()
    check(lib().pp_search_mesh_3d(mesh.p, ps.p, m_x, m_xtgt, m_pid, elem_ids.ptr, int(seeded),
                                  xpoints.ptr, xface.ptr, looplimit, C.byref(found)))
    return dict(found=found.value, elem_ids=elem_ids, xface=xface, xpoints=xpoints)


def push_search(mesh, ps, h, k, d, deg, elem_ids, seeded=True, looplimit=0, want_found=True,
                m_x=0, m_xtgt=1, m_b=3, m_phi=4):
    found = C.c_int(1)
    check(lib().pp_push_search(mesh.p, ps.p, m_x, m_xtgt, m_b, m_phi, h, k, d, deg, elem_ids.ptr,
                               int(seeded), looplimit, C.byref(found) if want_found else None))
    return bool(found.value)


def create_gyro_ring_mappings(mesh, rmax=0.038, gnr=3, gppr=8, theta=0.0):
    n = max(mesh.nverts * gnr * gppr * (mesh.dim + 1), 1)
    f, b = DevArray(n, np.int32), DevArray(n, np.int32)
    check(lib().pp_create_gyro_ring_mappings(mesh.p, rmax, gnr, gppr, theta, f.ptr, b.ptr))
    return f, b


def gyro_scatter(mesh, ps, v2v_dev, rmax=0.038, gnr=3, gppr=8, out=None):
    if out is None:
        out = DevArray(max(mesh.nverts, 1), np.float64)
    check(lib().pp_gyro_scatter(mesh.p, ps.p, v2v_dev.ptr, rmax, gnr, gppr, out.ptr))
    return out


def gyro_scatter_radius(mesh, ps, radius_dev, v2v_dev, weight_dev=None, rmax=0.038, gnr=3, gppr=8, out=None,
                        want_clipped=True):
    """per-particle radius / weight form of gyroScatter -> (field DevArray, clipped count or None)"""
    if out is None:
        out = DevArray(max(mesh.nverts, 1), np.float64)
    clipped = C.c_int(0)
    check(lib().pp_gyro_scatter_radius(mesh.p, ps.p, radius_dev.ptr if isinstance(radius_dev, DevArray) else radius_dev,
                                       weight_dev.ptr if isinstance(weight_dev, DevArray) else weight_dev,
                                       v2v_dev.ptr, rmax, gnr, gppr, out.ptr,
                                       C.byref(clipped) if want_clipped else None))
    return out, (clipped.value if want_clipped else None)


def gyro_sync_pack(nverts, fwd, bkwd, out=None):
    if out is None:
        out = DevArray(max(2 * nverts, 1), np.float64)
    check(lib().pp_gyro_sync_pack(nverts, fwd.ptr, bkwd.ptr, out.ptr))
    return out


def gather_tet_vtx(mesh, ps, field, dof=1, elem_ids=None, m_x=0):
    """host field [nverts*dof] -> host array (dof, capacity)"""
    cap = max(ps.capacity(), 1)
    f = DevArray.from_host(np.ascontiguousarray(field, dtype=np.float64))
    out = DevArray(cap * dof, np.float64)
    bad = C.c_int(0)
    check(lib().pp_gather_tet_vtx(mesh.p, ps.p, m_x, elem_ids.ptr if elem_ids is not None else None,
                                  f.ptr, dof, out.ptr, C.byref(bad)))
    return out.to_host().reshape(dof, cap), bad.value


def interp2d_field(ps, data, gridx0, gridz0, dx, dz, nx, nz, cyl_symm=True, ncomp=1, comp=0, m_x=0):
    cap = max(ps.capacity(), 1)
    d = DevArray.from_host(np.ascontiguousarray(data, dtype=np.float64))
    out = DevArray(cap, np.float64)
    check(lib().pp_interp2d_field(ps.p, m_x, d.ptr, gridx0, gridz0, dx, dz, nx, nz, int(cyl_symm), ncomp,
                                  comp, out.ptr))
    return out.to_host()


def interp2d_vector(ps, data3, gridx0, gridz0, dx, dz, nx, nz, cyl_symm=False, m_x=0):
    cap = max(ps.capacity(), 1)
    d = DevArray.from_host(np.ascontiguousarray(data3, dtype=np.float64))
    out = DevArray(cap * 3, np.float64)
    check(lib().pp_interp2d_vector(ps.p, m_x, d.ptr, gridx0, gridz0, dx, dz, nx, nz, int(cyl_symm), out.ptr))
    return out.to_host().reshape(3, cap)


def interp3d_field(ps, gridx, gridy, gridz, data, m_x=0):
    cap = max(ps.capacity(), 1)
    gx, gy, gz = (DevArray.from_host(np.ascontiguousarray(g, dtype=np.float64)) for g in (gridx, gridy, gridz))
    d = DevArray.from_host(np.ascontiguousarray(data, dtype=np.float64))
    out = DevArray(cap, np.float64)
    check(lib().pp_interp3d_field(ps.p, m_x, len(gridx), len(gridy), len(gridz), gx.ptr, gy.ptr, gz.ptr,
                                  d.ptr, out.ptr))
    return out.to_host()


def avg_ptcl_density(mesh, ps):
    ec = DevArray(max(mesh.nelems, 1), np.float64)
    vd = DevArray(max(mesh.nverts, 1), np.float64)
    check(lib().pp_avg_ptcl_density(mesh.p, ps.p, ec.ptr, vd.ptr))
    return ec, vd


def set_unsafe_procs(ps, elems_dev, safe_dev, owners_dev, rank, out=None):
    """out = (new_elems, new_procs) DevArrays to reuse (a device allocation per step costs a
    hipMalloc / hipFree pair, and hipFree drains the GPU)"""
    cap = max(ps.capacity(), 1)
    if out is not None and out[0].n >= cap and out[1].n >= cap:
        ne, npr = out
    else:
        ne, npr = DevArray(cap + cap // 10, np.int32), DevArray(cap + cap // 10, np.int32)
    check(lib().pp_set_unsafe_procs(ps.p, elems_dev.ptr, safe_dev.ptr, owners_dev.ptr, rank, ne.ptr,
                                    npr.ptr))
    return ne, npr


def migrate_count(ps, new_element_dev, new_process_dev, rank, nranks):
    counts = np.zeros(nranks, dtype=np.int32)
    check(lib().pp_ps_migrate_count(ps.p, new_element_dev.ptr, new_process_dev.ptr, rank, nranks,
                                    counts.ctypes.data))
    return counts


def migrate_pack(ps, new_element_dev, new_process_dev, rank, nranks, counts):
    total = int(counts.sum())
    gid = DevArray(max(total, 1), np.int64)
    bufs = [DevArray(max(total * nc, 1), dt) for dt, nc in ps.members]
    arr = (C.c_void_p * len(bufs))(*[b.ptr for b in bufs])
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    check(lib().pp_ps_migrate_pack(ps.p, new_element_dev.ptr, new_process_dev.ptr, rank, nranks,
                                   counts.ctypes.data, gid.ptr, C.cast(arr, C.c_void_p)))
    return gid, bufs


def migrate_record_bytes(ps):
    n = lib().pp_ps_migrate_record_bytes(ps.p)
    if n < 0:
        check(n)
    return n


def migrate_pack_records(ps, new_element_dev, new_process_dev, rank, nranks, counts, out_ptr):
    """out_ptr: raw device pointer of a buffer holding counts.sum() records"""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    check(lib().pp_ps_migrate_pack_records(ps.p, new_element_dev.ptr, new_process_dev.ptr, rank,
                                           nranks, counts.ctypes.data, out_ptr))


def migrate_pack_records_commit(ps, new_element_dev, new_process_dev, rank, nranks, counts, out_ptr,
                                m_x=0, m_xtgt=1):
    """records carry the particles after updatePtclPositions (no separate pass)"""
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    check(lib().pp_ps_migrate_pack_records_commit(ps.p, m_x, m_xtgt, new_element_dev.ptr,
                                                  new_process_dev.ptr, rank, nranks,
                                                  counts.ctypes.data_as(c_int_p), out_ptr))


def rebuild_records_scatter(ps, new_element_dev, n_recv, recv_ptr, mesh=None, maps=(), outs=(),
                            commit=True, rmax=0.038, gnr=3, gppr=8, m_x=0, m_xtgt=1,
                            gid2lid_dev=None, ngids=0):
    """pp_ps_rebuild_records with the position commit and the step's gyroScatter calls folded in"""
    n = len(maps)
    v2v = (C.c_void_p * max(n, 1))(*[m.ptr for m in maps])
    out = (C.c_void_p * max(n, 1))(*[o.ptr for o in outs])
    check(lib().pp_ps_rebuild_records_scatter(
        ps.p, m_x if commit else -1, m_xtgt if commit else -1, new_element_dev.ptr, n_recv, recv_ptr,
        gid2lid_dev.ptr if gid2lid_dev is not None else None, ngids, mesh.p if mesh is not None else None,
        n, v2v, out, rmax, gnr, gppr))


def rebuild_records(ps, new_element_dev, n_recv, recv_ptr, gid2lid_dev=None, ngids=0):
    check(lib().pp_ps_rebuild_records(ps.p, new_element_dev.ptr, n_recv, recv_ptr,
                                      gid2lid_dev.ptr if gid2lid_dev is not None else None, ngids))


def closest_point_on_triangle(tris, pts, wnormal=False, reg0=-1):
    """closest_point_on_triangle[_wnormal] (adjacency.hpp:824-1009) for n points; tris is (9,) for
    one shared triangle or (n, 9).  Returns (q[n,3], region[n])."""
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    n = len(pts)
    tris = np.ascontiguousarray(tris, dtype=np.float64)
    stride = 0 if tris.size == 9 else 9
    d_t, d_p = DevArray.from_host(tris.ravel()), DevArray.from_host(pts.ravel())
    d_q = DevArray(max(3 * n, 1), np.float64)
    d_r = DevArray.from_host(np.full(max(n, 1), reg0, dtype=np.int32))
    check(lib().pp_closest_point_on_triangle(n, d_t.ptr, stride, d_p.ptr, int(wnormal), d_q.ptr, d_r.ptr))
    return d_q.to_host()[:3 * n].reshape(n, 3), d_r.to_host()[:n]


def trace_particle_through_mesh(mesh, ps, func=None, elem_ids=None, require_intersection=False,
                                looplimit=0, m_x=0, m_xtgt=1):
    """trace_particle_through_mesh (adjacency.tpp:460-615) through the stepwise entry points.
    func(state) runs where the reference calls its functor; state holds the device arrays
    (elem_ids, inter_faces, last_exit, inter_points, ptcl_done).  None = the default functor."""
    cap = max(ps.capacity(), 1)
    dim = mesh.dim
    seeded = elem_ids is not None
    if elem_ids is None:
        elem_ids = DevArray(cap, np.int32)
    st = dict(elem_ids=elem_ids, inter_faces=DevArray.from_host(np.full(cap, -1, dtype=np.int32)),
              inter_points=DevArray.from_host(np.zeros(cap * dim)), ptcl_done=DevArray(cap, np.int32),
              last_exit=DevArray(cap, np.int32), mesh=mesh, ps=ps,
              require_intersection=bool(require_intersection))
    notin = C.c_int()
    L = lib()
    check(L.pp_trace_begin(mesh.p, ps.p, m_x, m_xtgt, elem_ids.ptr, int(seeded),
                           int(require_intersection), st["inter_faces"].ptr, st["inter_points"].ptr,
                           st["ptcl_done"].ptr, st["last_exit"].ptr, C.byref(notin)))
    found, loops = False, 0
    while not found:
        check(L.pp_trace_find_exit_face(mesh.p, ps.p, m_x, m_xtgt, elem_ids.ptr, st["ptcl_done"].ptr,
                                        st["last_exit"].ptr, st["inter_points"].ptr,
                                        int(not require_intersection)))
        if func is None:
            check(L.pp_trace_check_model_intersection(mesh.p, ps.p, elem_ids.ptr, st["ptcl_done"].ptr,
                                                      st["last_exit"].ptr, int(require_intersection),
                                                      st["inter_faces"].ptr))
        else:
            func(st)
        left = C.c_int()
        check(L.pp_trace_set_new_element(mesh.p, ps.p, elem_ids.ptr, st["ptcl_done"].ptr,
                                         st["last_exit"].ptr, C.byref(left)))
        found = left.value == 0
        loops += 1
        if looplimit and loops >= looplimit:
            nf = C.c_int()
            check(L.pp_trace_not_found(ps.p, elem_ids.ptr, st["ptcl_done"].ptr, C.byref(nf)))
            break
    st.update(found=found, loops=loops, not_in_elem=notin.value)
    return st


def redistribute_particles(ps, percent_moved, seed=0, out=None, strat=1):
    """redistribute_particles (Distribute.h:28-89): strategy 1 uniform (default), 2 gaussian, 3 exponential,
    4 GITRm approximation -> DevArray new_elems."""
    cap = max(ps.capacity(), 1)
    if out is None or out.n < cap:
        out = DevArray(cap, np.int32)
    check(lib().pp_redistribute_particles_dist(ps.p, int(strat), float(percent_moved), int(seed), out.ptr))
    return out


def boris_push_fields(mesh, ps, efield_vtx, bgrid, gridx0, gridz0, dx, dz, nx, nz, dt, cyl=True,
                      elem_ids=None, m_x=0, m_xprev=1, m_v=2):
    """gather (vertex E, grid B) + pushBoris in one kernel; returns the number of degenerate tets"""
    bad = C.c_int()
    check(lib().pp_boris_push_fields(mesh.p, ps.p, m_x, m_xprev, m_v,
                                     elem_ids.ptr if elem_ids is not None else None, efield_vtx.ptr,
                                     bgrid.ptr, gridx0, gridz0, dx, dz, nx, nz, int(cyl), dt, C.byref(bad)))
    return bad.value


def bfs_buffer_layers(mesh, owner_dev, rank, comm_size, safe_layers, ghost_layers, bridge_dim=0):
    """bfsBufferLayers (pumipic_part_construct.cpp:407-437) -> (is_safe DevArray u8, has_part int32[comm_size])"""
    safe = DevArray(max(mesh.nelems, 1), np.uint8)
    part = np.zeros(comm_size, dtype=np.int32)
    check(lib().pp_bfs_buffer_layers(mesh.p, bridge_dim, rank, comm_size, safe_layers, ghost_layers,
                                     owner_dev.ptr, safe.ptr, part.ctypes.data_as(c_int_p)))
    return safe, part


def bfs_safe_inward(mesh, owner_dev, rank, comm_size, safe_layers, has_part, bridge_dim=0):
    """bfsSafeInward (pumipic_part_construct.cpp:439-468) -> safe DevArray u8"""
    part = np.ascontiguousarray(has_part, dtype=np.int32)
    safe = DevArray(max(mesh.nelems, 1), np.uint8)
    check(lib().pp_bfs_safe_inward(mesh.p, bridge_dim, rank, comm_size, safe_layers, owner_dev.ptr,
                                   part.ctypes.data_as(c_int_p), safe.ptr))
    return safe


def rebuild_scatter(ps, mesh, new_element, maps, outs=None, commit=True, rmax=0.038, gnr=3, gppr=8,
                    m_x=0, m_xtgt=1):
    """pp_ps_rebuild[_commit] + one pp_gyro_scatter per map in one call (the scatter is enqueued
    before the rebuild's host sync).  Returns the list of output DevArrays."""
    ne = new_element if isinstance(new_element, DevArray) else DevArray.from_host(
        np.ascontiguousarray(new_element, dtype=np.int32))
    if outs is None:
        outs = [DevArray(max(mesh.nverts, 1), np.float64) for _ in maps]
    n = len(maps)
    v2v = (C.c_void_p * max(n, 1))(*[m.ptr for m in maps])
    out = (C.c_void_p * max(n, 1))(*[o.ptr for o in outs])
    check(lib().pp_ps_rebuild_scatter(ps.p, m_x if commit else -1, m_xtgt if commit else -1, ne.ptr, 0,
                                      None, None, mesh.p if mesh is not None else None, n, v2v, out,
                                      rmax, gnr, gppr))
    return outs


# ------------------------------------------------------------------ communicators (pp_comm.hip)
_ALLTOALL_INT = C.CFUNCTYPE(C.c_int, C.c_void_p, c_int_p, c_int_p)
_ALLTOALLV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                         C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))
_ALLRED_F64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
_ALLRED_I64 = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64)


class CommHostOps(C.Structure):
    _fields_ = [("alltoall_int", _ALLTOALL_INT), ("alltoallv_bytes", _ALLTOALLV),
                ("allreduce_sum_f64", _ALLRED_F64), ("allreduce_sum_i64", _ALLRED_I64)]


class Comm:
    """pp_comm handle.  Constructors: Comm.rccl(id, rank, n), Comm.tcp(addr, port, rank, n),
    Comm.host(ops...), Comm.local(n) -> list, Comm.env(), Comm.torch(group) (host transport whose
    collectives are torch.distributed calls on CPU tensors: gloo in the tests)."""

    def __init__(self, p, keep=None):
        if not p:
            raise PPError("communicator creation failed: " + lib().pp_last_error().decode())
        self.p = p
        self._keep = keep

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        check(lib().pp_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def rccl(cls, id128, rank, nranks):
        buf = (C.c_ubyte * 128).from_buffer_copy(id128)
        return cls(lib().pp_comm_create_rccl(buf, rank, nranks))

    @classmethod
    def tcp(cls, addr, port, rank, nranks):
        return cls(lib().pp_comm_create_tcp(addr.encode(), port, rank, nranks))

    @classmethod
    def env(cls):
        return cls(lib().pp_comm_create_env())

    @classmethod
    def local(cls, nranks):
        arr = (C.c_void_p * nranks)()
        check(lib().pp_comm_create_local(nranks, arr))
        return [cls(arr[r]) for r in range(nranks)]

    @classmethod
    def host(cls, rank, nranks, alltoall_int, alltoallv_bytes, allreduce_f64, allreduce_i64):
        """the four collectives as Python callables on numpy views of the library's host buffers"""
        def _a2a(user, s, r):
            try:
                sv = np.ctypeslib.as_array(s, shape=(nranks,))
                np.ctypeslib.as_array(r, shape=(nranks,))[:] = alltoall_int(sv.copy())
                return 0
            except Exception:  # noqa: BLE001 -- a Python exception must not unwind through C
                import traceback
                traceback.print_exc()
                return -2

        def _a2av(user, s, sb, sd, r, rb, rd):
            try:
                sb_ = np.ctypeslib.as_array(sb, shape=(nranks,)).copy()
                sd_ = np.ctypeslib.as_array(sd, shape=(nranks,)).copy()
                rb_ = np.ctypeslib.as_array(rb, shape=(nranks,)).copy()
                rd_ = np.ctypeslib.as_array(rd, shape=(nranks,)).copy()
                ns, nr = int((sd_ + sb_).max(initial=0)), int((rd_ + rb_).max(initial=0))
                sbuf = np.ctypeslib.as_array(C.cast(s, C.POINTER(C.c_ubyte)), shape=(max(ns, 1),))[:ns]
                rbuf = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_ubyte)), shape=(max(nr, 1),))[:nr]
                alltoallv_bytes(sbuf, sb_, sd_, rbuf, rb_, rd_)
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return -2

        def _arf(user, b, n):
            try:
                v = np.ctypeslib.as_array(b, shape=(int(n),))
                v[:] = allreduce_f64(v.copy())
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return -2

        def _ari(user, b, n):
            try:
                v = np.ctypeslib.as_array(b, shape=(int(n),))
                v[:] = allreduce_i64(v.copy())
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return -2

        ops = CommHostOps(_ALLTOALL_INT(_a2a), _ALLTOALLV(_a2av), _ALLRED_F64(_arf), _ALLRED_I64(_ari))
        return cls(lib().pp_comm_create_host(C.byref(ops), None, rank, nranks), keep=ops)

    @classmethod
    def torch(cls, group=None):
        """host transport over torch.distributed CPU collectives (gloo)"""
        import torch
        import torch.distributed as dist
        rank, n = dist.get_rank(group), dist.get_world_size(group)

        def a2a(send):
            s = torch.from_numpy(np.asarray(send, dtype=np.int32).copy())
            r = torch.empty_like(s)
            _gloo_a2a(r, s, group)
            return r.numpy()

        def a2av(sbuf, sb, sd, rbuf, rb, rd):
            outs = [torch.empty(int(rb[q]), dtype=torch.uint8) for q in range(n)]
            ins = [torch.from_numpy(sbuf[int(sd[q]):int(sd[q] + sb[q])].copy()) for q in range(n)]
            _p2p_exchange(ins, outs, rank, n, group)
            for q in range(n):
                if rb[q]:
                    rbuf[int(rd[q]):int(rd[q] + rb[q])] = outs[q].numpy()

        def arf(v):
            t = torch.from_numpy(np.asarray(v, dtype=np.float64).copy())
            dist.all_reduce(t, group=group)
            return t.numpy()

        def ari(v):
            t = torch.from_numpy(np.asarray(v, dtype=np.int64).copy())
            dist.all_reduce(t, group=group)
            return t.numpy()

        return cls.host(rank, n, a2a, a2av, arf, ari)

    def rank(self):
        return lib().pp_comm_rank(self.p)

    def size(self):
        return lib().pp_comm_size(self.p)

    def kind(self):
        return lib().pp_comm_kind(self.p).decode()

    def exchange_counts(self, send_counts):
        s = np.ascontiguousarray(send_counts, dtype=np.int32)
        r = np.zeros_like(s)
        check(lib().pp_comm_exchange_counts(self.p, s.ctypes.data_as(c_int_p), r.ctypes.data_as(c_int_p)))
        return r

    def allreduce_sum(self, dev_array, n=None):
        check(lib().pp_allreduce_sum(self.p, dev_array.ptr if isinstance(dev_array, DevArray) else dev_array,
                                     dev_array.n if n is None else n))

    def allreduce_sum_host(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.int64).copy()
        check(lib().pp_allreduce_sum_host_i64(self.p, v.ctypes.data_as(C.POINTER(C.c_int64)), len(v)))
        return v

    def barrier(self):
        check(lib().pp_comm_barrier(self.p))

    def selftest(self, nrec=5):
        """checked exchange + all-reduce with known contents (pp_comm_selftest); raises PPError"""
        check(lib().pp_comm_selftest(self.p, int(nrec)))

    def allgather_host(self, values):
        v = np.ascontiguousarray(values)
        out = np.empty((self.size(),) + v.shape, dtype=v.dtype)
        check(lib().pp_comm_allgather_host(self.p, v.ctypes.data, out.ctypes.data, v.nbytes))
        return out

    def destroy(self):
        if self.p:
            lib().pp_comm_destroy(self.p)
            self.p = None


def _gloo_a2a(r, s, group):
    """gloo has no all_to_all_single on every build: gather everything, pick the column"""
    import torch
    import torch.distributed as dist
    n, me = dist.get_world_size(group), dist.get_rank(group)
    rows = [torch.empty_like(s) for _ in range(n)]
    dist.all_gather(rows, s, group=group)
    for q in range(n):
        r[q] = rows[q][me]


def _p2p_exchange(ins, outs, rank, n, group):
    import torch.distributed as dist
    reqs = []
    for q in range(n):
        if q == rank:
            continue
        if outs[q].numel():
            reqs.append(dist.irecv(outs[q], src=q, group=group))
        if ins[q].numel():
            reqs.append(dist.isend(ins[q], dst=q, group=group))
    for rq in reqs:
        rq.wait()


def migrate_plan(nranks, rank, send_counts, recv_counts):
    s = np.ascontiguousarray(send_counts, dtype=np.int32)
    r = np.ascontiguousarray(recv_counts, dtype=np.int32)
    sd, rd = np.zeros(nranks, dtype=np.int64), np.zeros(nranks, dtype=np.int64)
    ns, nr = C.c_int64(), C.c_int64()
    i64p = C.POINTER(C.c_int64)
    check(lib().pp_migrate_plan(nranks, rank, s.ctypes.data_as(c_int_p), r.ctypes.data_as(c_int_p),
                                sd.ctypes.data_as(i64p), rd.ctypes.data_as(i64p), C.byref(ns), C.byref(nr)))
    return sd, rd, ns.value, nr.value


def _migrate_args(ps, new_elems, new_procs, comm, commit, scatter, new_particles, gid2lid, rmax, gnr, gppr,
                  m_x, m_xtgt):
    mesh, maps, outs = scatter if scatter is not None else (None, (), ())
    n = len(maps)
    v2v = (C.c_void_p * max(n, 1))(*[m.ptr for m in maps])
    out = (C.c_void_p * max(n, 1))(*[o.ptr for o in outs])
    n_new, npe, arr, keep = 0, None, None, [v2v, out]
    if new_particles is not None:
        pe, info = new_particles
        n_new = len(pe)
        if n_new:
            npe = DevArray.from_host(np.ascontiguousarray(pe, dtype=np.int32))
            arr = (C.c_void_p * len(ps.members))()
            for i, ((dt, nc), a) in enumerate(zip(ps.members, info)):
                d = DevArray.from_host(np.ascontiguousarray(np.asarray(a, dtype=dt).reshape(nc, n_new)))
                keep.append(d)
                arr[i] = d.ptr
            keep += [npe, arr]
    args = (ps.p, m_x if commit else -1, m_xtgt if commit else -1, new_elems.ptr, new_procs.ptr, comm.p, n_new,
            npe.ptr if npe is not None else None, C.cast(arr, C.c_void_p) if arr is not None else None,
            gid2lid.ptr if gid2lid is not None else None, gid2lid.n if gid2lid is not None else 0,
            mesh.p if mesh is not None else None, n, v2v, out, rmax, gnr, gppr)
    return args, keep


def migrate(ps, new_elems, new_procs, comm, commit=False, scatter=None, new_particles=None, gid2lid=None,
            rmax=0.038, gnr=3, gppr=8, m_x=0, m_xtgt=1):
    """pp_ps_migrate_scatter: SellCSigma::migrate behind one call (any non-local communicator)"""
    args, keep = _migrate_args(ps, new_elems, new_procs, comm, commit, scatter, new_particles, gid2lid, rmax,
                               gnr, gppr, m_x, m_xtgt)
    check(lib().pp_ps_migrate_scatter(*args))
    del keep


def migrate_begin(ps, new_elems, new_procs, comm, commit=False, scatter=None, new_particles=None,
                  gid2lid=None, rmax=0.038, gnr=3, gppr=8, m_x=0, m_xtgt=1):
    args, keep = _migrate_args(ps, new_elems, new_procs, comm, commit, scatter, new_particles, gid2lid, rmax,
                               gnr, gppr, m_x, m_xtgt)
    check(lib().pp_ps_migrate_begin(*args))
    comm._pending_keep = (keep, new_elems, new_procs)  # buffers the end phase still reads


def migrate_end(ps, comm):
    ns, nr = C.c_int(), C.c_int()
    check(lib().pp_ps_migrate_end(ps.p, comm.p, C.byref(ns), C.byref(nr)))
    comm._pending_keep = None
    return ns.value, nr.value


def migrate_ptcls_begin(ps, elem_ids, safe_dev, owners_dev, comm, commit=False, scatter=None, rmax=0.038, gnr=3,
                        gppr=8, m_x=0, m_xtgt=1):
    """pp_migrate_ptcls_begin: setUnsafeProcs + migrate without the new_elems / new_procs arrays"""
    mesh, maps, outs = scatter if scatter is not None else (None, (), ())
    n = len(maps)
    v2v = (C.c_void_p * max(n, 1))(*[m.ptr for m in maps])
    out = (C.c_void_p * max(n, 1))(*[o.ptr for o in outs])
    check(lib().pp_migrate_ptcls_begin(ps.p, m_x if commit else -1, m_xtgt if commit else -1, elem_ids.ptr,
                                       safe_dev.ptr, owners_dev.ptr, comm.p, mesh.p if mesh is not None else None,
                                       n, v2v, out, rmax, gnr, gppr))
    comm._pending_keep = (v2v, out, elem_ids, safe_dev, owners_dev)


def migrate_ptcls(ps, elem_ids, safe_dev, owners_dev, comm, **kw):
    migrate_ptcls_begin(ps, elem_ids, safe_dev, owners_dev, comm, **kw)
    return migrate_end(ps, comm)


def push_search_counters():
    """(not_found, not_in_elem, unmoved_trusted) of the last pp_push_search"""
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    check(lib().pp_push_search_counters(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def last_search_found(ps):
    """the `found` of the most recent pp_push_search, delivered with the totals of the structure's last rebuild
    (no host sync of its own when a full re-layout carried it)"""
    f = C.c_int()
    check(lib().pp_ps_last_search_found(ps.p, C.byref(f)))
    return bool(f.value)


def search_walk_steps():
    """elements visited by the last intersection-mode pp_search_mesh on tets (all particles)"""
    n = C.c_ulonglong()
    check(lib().pp_search_walk_steps(C.byref(n)))
    return int(n.value)


def ray_intersects_triangle(tris, orig, dest, tol, flip=0, segment=False):
    """batch ray / segment vs triangle -> (hit[n], xpoint[n,3], (dproj, closeness, param)[n,3])"""
    orig = np.ascontiguousarray(orig, dtype=np.float64).reshape(-1, 3)
    dest = np.ascontiguousarray(dest, dtype=np.float64).reshape(-1, 3)
    n = len(orig)
    tris = np.ascontiguousarray(tris, dtype=np.float64)
    stride = 0 if tris.size == 9 else 9
    d_t, d_o, d_d = DevArray.from_host(tris.ravel()), DevArray.from_host(orig.ravel()), DevArray.from_host(dest.ravel())
    fl = np.asarray(flip)
    d_f = DevArray.from_host(fl.astype(np.int32)) if fl.ndim else None
    hit, xp, o3 = DevArray(max(n, 1), np.int32), DevArray(max(3 * n, 1), np.float64), DevArray(max(3 * n, 1), np.float64)
    check(lib().pp_ray_intersects_triangle(n, d_t.ptr, stride, d_o.ptr, d_d.ptr, tol, d_f.ptr if d_f is not None else None,
                                           0 if fl.ndim else int(fl), int(segment), hit.ptr, xp.ptr, o3.ptr))
    return hit.to_host()[:n].astype(bool), xp.to_host()[:3 * n].reshape(n, 3), o3.to_host()[:3 * n].reshape(n, 3)


# ------------------------------------------------------------------ PICparts and comm arrays
PART_FULL, PART_BFS, PART_MINIMUM, PART_NONE = 0, 1, 2, 3
OP_SUM, OP_MAX, OP_MIN, OP_BCAST = 0, 1, 2, 3
PART_GIDS, PART_OWNERS, PART_RANK_LIDS, PART_COMM_INDEX, PART_FULL_IDS, PART_ENT_IDS, PART_SAFE = range(7)
_PART_DTYPES = {PART_GIDS: np.int64, PART_OWNERS: np.int32, PART_RANK_LIDS: np.int32, PART_COMM_INDEX: np.int32,
                PART_FULL_IDS: np.int32, PART_ENT_IDS: np.int32, PART_SAFE: np.uint8}


def owner_by_classification(mesh, class_owners, rank):
    co = np.ascontiguousarray(class_owners, dtype=np.int32)
    out = np.empty(mesh.nelems, dtype=np.int32)
    check(lib().pp_owner_by_classification(mesh.p, co.ctypes.data_as(c_int_p), len(co), rank,
                                           out.ctypes.data_as(c_int_p)))
    return out


class _PartMesh:
    """the part's pp_mesh, owned by the PICpart (or the full mesh itself for a FULL buffer)"""

    def __init__(self, p, owner):
        self.p, self._owner = p, owner
        d, nv, ne, ns = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().pp_mesh_info(self.p, C.byref(d), C.byref(nv), C.byref(ne), C.byref(ns)))
        self.dim, self.nverts, self.nelems, self.nsides = d.value, nv.value, ne.value, ns.value

    tolerance = Mesh.tolerance
    array = Mesh.array


class PicPart:
    """pumipic::Mesh (pp_picpart): PicPart(full_mesh, elem_owner, comm, buffer_method, safe_method, ...)"""

    def __init__(self, mesh, elem_owner, comm=None, buffer_method=PART_FULL, safe_method=PART_FULL, bridge_dim=0,
                 buffer_layers=3, safe_layers=1):
        own = np.ascontiguousarray(elem_owner, dtype=np.int32)
        assert len(own) == mesh.nelems
        self._keep = (mesh, comm)
        self.dim = mesh.dim
        self.p = lib().pp_picpart_create(mesh.p, own.ctypes.data_as(c_int_p), buffer_method, safe_method, bridge_dim,
                                         buffer_layers, safe_layers, comm.p if comm is not None else None)
        if not self.p:
            raise PPError("pp_picpart_create: " + lib().pp_last_error().decode())
        f, nb, nv, ne = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib().pp_picpart_info(self.p, C.byref(f), C.byref(nb), C.byref(nv), C.byref(ne)))
        self.is_full_mesh, self.num_buffers = bool(f.value), nb.value
        self.nents = {0: nv.value, mesh.dim: ne.value}
        self.nranks = comm.size() if comm is not None else 1
        self.mesh = _PartMesh(lib().pp_picpart_mesh(self.p), self)
        self.nents[mesh.dim - 1] = self.mesh.nsides  # sides: numbered by the part's own mesh
        if mesh.dim == 3:                             # edges of tets likewise
            self.nents[1] = lib().pp_mesh_num_edges(self.mesh.p)

    def array(self, which, edim=0):
        cnt = C.c_size_t()
        lib().pp_picpart_array_dev(self.p, which, edim, C.byref(cnt))
        out = np.empty(cnt.value, dtype=_PART_DTYPES[which])
        if cnt.value:
            check(lib().pp_picpart_array_to_host(self.p, which, edim, out.ctypes.data))
        return out

    def array_dev(self, which, edim=0):
        cnt = C.c_size_t()
        ptr = lib().pp_picpart_array_dev(self.p, which, edim, C.byref(cnt))
        return ptr, cnt.value

    def nents_offsets(self, edim):
        out = np.empty(self.nranks + 1, dtype=np.int32)
        check(lib().pp_picpart_nents_offsets(self.p, edim, out.ctypes.data_as(c_int_p)))
        return out

    def buffered_ranks(self, edim):
        out = np.empty(max(self.nranks, 1), dtype=np.int32)
        n = C.c_int()
        check(lib().pp_picpart_buffered_ranks(self.p, edim, out.ctypes.data_as(c_int_p), C.byref(n)))
        return out[:n.value].copy()

    def bounded(self, edim):
        """(boundary_parts, offset_bounded, bounded_ent_ids) of src/pumipic_mesh.hpp:131-136"""
        nb, nid = C.c_int(), C.c_int()
        check(lib().pp_picpart_bounded(self.p, edim, C.byref(nb), None, None, C.byref(nid), None))
        parts = np.empty(max(nb.value, 1), dtype=np.int32)
        off = np.empty(nb.value + 1, dtype=np.int32)
        ids = np.empty(max(nid.value, 1), dtype=np.int32)
        check(lib().pp_picpart_bounded(self.p, edim, C.byref(nb), parts.ctypes.data_as(c_int_p),
                                       off.ctypes.data_as(c_int_p), C.byref(nid), ids.ctypes.data_as(c_int_p)))
        return parts[:nb.value].copy(), off, ids[:nid.value].copy()

    def num_global(self, edim):
        return int(lib().pp_picpart_num_global(self.p, edim))

    def complete_parts(self, edim):
        out = np.empty(self.nranks, dtype=np.int32)
        check(lib().pp_picpart_complete_parts(self.p, edim, out.ctypes.data_as(c_int_p)))
        return out

    @staticmethod
    def _dtype(arr):
        if arr.dtype == np.int32:
            return 0
        if arr.dtype == np.float64:
            return 1
        raise TypeError("comm arrays are int32 or float64")

    def create_comm_array(self, edim, nvals, default, dtype=np.float64):
        """createCommArray: nvals values per entity, entity-major"""
        return DevArray.from_host(np.full(self.nents[edim] * nvals, default, dtype=dtype))

    def reduce(self, edim, op, arr):
        nvals = arr.n // max(self.nents.get(edim, 0), 1)
        check(lib().pp_picpart_reduce(self.p, edim, op, self._dtype(arr), nvals, arr.ptr))

    def reduce_begin(self, edim, op, arr):
        nvals = arr.n // max(self.nents.get(edim, 0), 1)
        self._arr = arr
        check(lib().pp_picpart_reduce_begin(self.p, edim, op, self._dtype(arr), nvals, arr.ptr))

    def reduce_mid(self):
        check(lib().pp_picpart_reduce_mid(self.p))

    def reduce_end(self):
        check(lib().pp_picpart_reduce_end(self.p))
        self._arr = None

    def __del__(self):
        try:
            lib().pp_picpart_destroy(self.p)
        except Exception:
            pass


def picpart_reduce_all(parts, edim, op, arrays):
    """reduceCommArray on the virtual ranks of one process (Comm.local): the three phases on every rank in turn"""
    for p, a in zip(parts, arrays):
        p.reduce_begin(edim, op, a)
    for p in parts:
        p.reduce_mid()
    for p in parts:
        p.reduce_end()


class Balancer:
    """pumipic::ParticleBalancer (pp_balancer) of a PicPart"""

    def __init__(self, part):
        self.part = part
        self.p = lib().pp_balancer_create(part.p)
        if not self.p:
            raise PPError("pp_balancer_create: " + lib().pp_last_error().decode())

    def sbars(self):
        n = lib().pp_balancer_num_sbars(self.p)
        out = np.zeros(n, dtype=np.uint64)
        if n:
            check(lib().pp_balancer_sbars(self.p, out.ctypes.data))
        return out

    def sbar_ids(self):
        cnt = C.c_size_t()
        ptr = lib().pp_balancer_sbar_ids_dev(self.p, C.byref(cnt))
        out = np.empty(cnt.value, dtype=np.int32)
        if cnt.value:
            sync()
            check(lib().pp_memcpy_d2h(out.ctypes.data, ptr, out.nbytes))
        return out

    def repartition(self, ps, new_elems, new_procs, tol=1.05, step_factor=0.3):
        check(lib().pp_balancer_repartition(self.p, ps.p, tol, new_elems.ptr, new_procs.ptr, step_factor))

    def repartition_begin(self, ps, new_elems, new_procs):
        self._keep = (ps, new_elems, new_procs)
        check(lib().pp_balancer_repartition_begin(self.p, ps.p, new_elems.ptr, new_procs.ptr))

    def repartition_end(self, tol=1.05, step_factor=0.3):
        check(lib().pp_balancer_repartition_end(self.p, tol, step_factor))
        self._keep = None

    def partition(self, ptcls_per_elem, tol=1.05, step_factor=0.3):
        ppe = np.ascontiguousarray(ptcls_per_elem, dtype=np.int32)
        out = np.empty(int(ppe.sum()), dtype=np.int32)
        check(lib().pp_balancer_partition(self.p, ppe.ctypes.data_as(c_int_p), tol, step_factor,
                                          out.ctypes.data_as(c_int_p)))
        return out

    def partition_begin(self, ptcls_per_elem):
        self._ppe = np.ascontiguousarray(ptcls_per_elem, dtype=np.int32)
        check(lib().pp_balancer_partition_begin(self.p, self._ppe.ctypes.data_as(c_int_p)))

    def partition_end(self, tol=1.05, step_factor=0.3):
        out = np.empty(int(self._ppe.sum()), dtype=np.int32)
        check(lib().pp_balancer_partition_end(self.p, tol, step_factor, out.ctypes.data_as(c_int_p)))
        return out

    def last_plan(self):
        n = C.c_int()
        check(lib().pp_balancer_last_plan(self.p, C.byref(n), None, None, None, None))
        sb, tg = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
        am, w = np.empty(n.value, np.int64), np.zeros(self.part.nranks, np.int64)
        check(lib().pp_balancer_last_plan(self.p, C.byref(n), sb.ctypes.data_as(c_int_p), tg.ctypes.data_as(c_int_p),
                                          am.ctypes.data, w.ctypes.data))
        return [(int(a), int(b), int(c)) for a, b, c in zip(sb, tg, am)], w

    def __del__(self):
        try:
            lib().pp_balancer_destroy(self.p)
        except Exception:
            pass
