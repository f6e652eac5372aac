"""ctypes wrapper around oracle/_build/libppo.so -- ORACLE (test infrastructure, NOT product).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (pumi-pic_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PPO_LIB=<path>: another build of the same sources, e.g. the AddressSanitizer / UBSan one of `make asan`)
_LIB_PATH = os.environ.get("PPO_LIB") or os.path.join(_HERE, "_build", "libppo.so")

c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)
c_long_p = C.POINTER(C.c_long)
c_ubyte_p = C.POINTER(C.c_ubyte)


def build(force=False):
    """Compile the C restatement with gcc (see oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class _MeshS(C.Structure):
    _fields_ = [("dim", C.c_int), ("nverts", C.c_int), ("nelems", C.c_int), ("nsides", C.c_int),
                ("coords", c_double_p), ("elem2verts", c_int_p), ("class_id", c_int_p),
                ("elem2sides", c_int_p), ("side2verts", c_int_p), ("side2elems_off", c_int_p),
                ("side2elems", c_int_p), ("side_exposed", C.POINTER(C.c_byte)),
                ("elem_measure", c_double_p), ("dual_off", c_int_p), ("dual_elems", c_int_p),
                ("vert2elems_off", c_int_p), ("vert2elems", c_int_p)]


class _PsS(C.Structure):
    _fields_ = [("kind", C.c_int), ("num_elems", C.c_int), ("num_ptcls", C.c_int),
                ("capacity", C.c_int), ("num_rows", C.c_int),
                ("C", C.c_int), ("C_max", C.c_int), ("V", C.c_int), ("sigma", C.c_int),
                ("num_chunks", C.c_int), ("num_slices", C.c_int),
                ("offsets", c_int_p), ("slice_to_chunk", c_int_p), ("row_to_element", c_int_p),
                ("element_to_row", c_int_p), ("mask", c_ubyte_p), ("element_to_gid", c_long_p),
                ("pad_strat", C.c_int), ("shuffle_padding", C.c_double),
                ("extra_padding", C.c_double), ("minimize_size", C.c_double),
                ("padding_amount", C.c_double), ("always_realloc", C.c_int),
                ("try_shuffling", C.c_int), ("num_empty_elements", C.c_int),
                ("nmembers", C.c_int), ("member_bytes", c_int_p), ("member_ncomp", c_int_p),
                ("alloc", C.c_long), ("data", C.POINTER(C.c_void_p)), ("swap_alloc", C.c_long),
                ("last_rebuild_was_shuffle", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # idle threads must not spin on shared hosts
        L = C.CDLL(_LIB_PATH)
        L.ppo_set_threads.argtypes = [C.c_int]
        L.ppo_max_threads.restype = C.c_int
        L.ppo_set_threads(1)  # Kokkos::Serial semantics unless a caller asks for more
        L.ppo_mesh_create.restype = C.POINTER(_MeshS)
        L.ppo_mesh_create.argtypes = [C.c_int, C.c_int, c_double_p, C.c_int, c_int_p, c_int_p]
        L.ppo_mesh_destroy.argtypes = [C.POINTER(_MeshS)]
        L.ppo_compute_tolerance_from_area.restype = C.c_double
        L.ppo_compute_tolerance_from_area.argtypes = [C.POINTER(_MeshS)]
        L.ppo_scs_create.restype = C.POINTER(_PsS)
        L.ppo_scs_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_long_p,
                                     C.c_int, C.c_double, C.c_double, C.c_int, c_int_p, c_int_p,
                                     c_int_p, C.POINTER(C.c_void_p)]
        L.ppo_csr_create.restype = C.POINTER(_PsS)
        L.ppo_csr_create.argtypes = [C.c_int, C.c_int, c_int_p, c_long_p, C.c_double, C.c_int,
                                     c_int_p, c_int_p, c_int_p, C.POINTER(C.c_void_p)]
        L.ppo_ps_destroy.argtypes = [C.POINTER(_PsS)]
        L.ppo_ps_slot_info.argtypes = [C.POINTER(_PsS), c_int_p, c_ubyte_p]
        L.ppo_ps_rebuild.argtypes = [C.POINTER(_PsS), c_int_p, C.c_int, c_int_p,
                                     C.POINTER(C.c_void_p)]
        L.ppo_ps_get_pids.argtypes = [C.POINTER(_PsS), c_int_p, c_int_p]
        L.ppo_scs_metrics.argtypes = [C.POINTER(_PsS), c_int_p, c_int_p, c_int_p]
        L.ppo_sincos.argtypes = [C.c_double, c_double_p, c_double_p]
        L.ppo_elliptical_setup.argtypes = [C.POINTER(_PsS), C.c_int, C.c_int, C.c_int, C.c_double,
                                           C.c_double, C.c_double]
        L.ppo_elliptical_push.argtypes = [C.POINTER(_PsS), C.POINTER(_MeshS), C.c_int, C.c_int,
                                          C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_int]
        L.ppo_toroidal_push.argtypes = [C.POINTER(_PsS), C.POINTER(_MeshS), C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_int]
        L.ppo_linear_push.argtypes = [C.POINTER(_PsS), C.c_int, C.c_int, C.c_double, C.c_double,
                                      C.c_double, C.c_double]
        L.ppo_push_boris.argtypes = [C.c_int] + [c_double_p] * 15 + [C.c_double]
        L.ppo_update_positions.argtypes = [C.POINTER(_PsS), C.c_int, C.c_int]
        L.ppo_pseudo_push160.argtypes = [C.POINTER(_PsS), c_double_p]
        L.ppo_search_mesh_2d.restype = C.c_int
        L.ppo_search_mesh_2d.argtypes = [C.POINTER(_MeshS), C.POINTER(_PsS), C.c_int, C.c_int,
                                         C.c_int, c_int_p, C.c_int, c_int_p]
        L.ppo_search_mesh.restype = C.c_int
        L.ppo_search_mesh.argtypes = [C.POINTER(_MeshS), C.POINTER(_PsS), C.c_int, C.c_int, C.c_int,
                                      c_int_p, C.c_int, C.c_int, c_int_p, c_double_p, C.c_int,
                                      c_int_p, c_int_p]
        L.ppo_search_mesh_legacy3d.restype = C.c_int
        L.ppo_search_mesh_legacy3d.argtypes = [C.POINTER(_MeshS), C.POINTER(_PsS), C.c_int, C.c_int,
                                               C.c_int, c_int_p, C.c_int, c_double_p, c_int_p,
                                               C.c_int, c_int_p]
        L.ppo_search_mesh_3d.restype = C.c_int
        L.ppo_search_mesh_3d.argtypes = L.ppo_search_mesh_legacy3d.argtypes
        L.ppo_search_mesh_2d_pt.restype = C.c_int
        L.ppo_search_mesh_2d_pt.argtypes = [C.POINTER(_MeshS), c_double_p, c_double_p, C.c_int,
                                            C.c_int, c_int_p, C.c_int]
        L.ppo_create_gyro_ring_mappings.argtypes = [C.POINTER(_MeshS), C.c_double, C.c_int, C.c_int,
                                                    C.c_double, C.c_int, c_int_p, c_int_p]
        L.ppo_gyro_scatter.argtypes = [C.POINTER(_MeshS), C.POINTER(_PsS), c_int_p, C.c_double,
                                       C.c_int, C.c_int, c_double_p]
        L.ppo_gyro_scatter_radius.argtypes = [C.POINTER(_MeshS), C.POINTER(_PsS), c_double_p, c_double_p, c_int_p,
                                              C.c_double, C.c_int, C.c_int, c_double_p, c_int_p]
        L.ppo_avg_ptcl_density.argtypes = [C.POINTER(_MeshS), C.POINTER(_PsS), c_double_p,
                                           c_double_p]
        L.ppo_gather_tet_vtx.argtypes = [C.POINTER(_MeshS), C.POINTER(_PsS), C.c_int, c_int_p,
                                         c_double_p, C.c_int, c_double_p, c_int_p]
        L.ppo_interp2d_field.argtypes = [C.POINTER(_PsS), C.c_int, c_double_p] + [C.c_double] * 4 + \
            [C.c_int] * 5 + [c_double_p]
        L.ppo_interp2d_vector.argtypes = [C.POINTER(_PsS), C.c_int, c_double_p] + [C.c_double] * 4 + \
            [C.c_int] * 3 + [c_double_p]
        L.ppo_interp3d_field.argtypes = [C.POINTER(_PsS), C.c_int, C.c_int, C.c_int, C.c_int] + \
            [c_double_p] * 5
        L.ppo_closest_point_on_triangle.argtypes = [c_double_p, c_double_p, C.c_int, c_double_p, c_int_p]
        L.ppo_interpolate_tet_vtx.restype = C.c_double
        L.ppo_interpolate_tet_vtx.argtypes = [C.POINTER(_MeshS), c_double_p, C.c_int, c_double_p,
                                              C.c_int, C.c_int]
        L.ppo_set_unsafe_procs.argtypes = [C.POINTER(_PsS), c_int_p, c_ubyte_p, c_int_p, C.c_int,
                                           c_int_p, c_int_p]
        L.ppo_kat_barycentric_tet.argtypes = [c_double_p, c_double_p, C.c_double, c_double_p,
                                              c_double_p, c_double_p]
        L.ppo_kat_barycentric_tri.argtypes = [c_double_p, c_double_p, C.c_double, c_double_p]
        L.ppo_kat_ray_triangle.restype = C.c_int
        L.ppo_kat_ray_triangle.argtypes = [c_double_p, c_double_p, c_double_p, C.c_double, C.c_int,
                                           C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
        L.ppo_kat_line_edge_2d.restype = C.c_int
        L.ppo_kat_line_edge_2d.argtypes = [c_double_p, c_double_p, c_double_p, C.c_double, C.c_int,
                                           c_double_p]
        L.ppo_kat_line_triangle_simple.restype = C.c_int
        L.ppo_kat_line_triangle_simple.argtypes = [c_double_p, c_double_p, c_double_p, C.c_int,
                                                   C.c_double, c_double_p, c_double_p]
        L.ppo_kat_all_positive.restype = C.c_int
        L.ppo_kat_all_positive.argtypes = [c_double_p, C.c_int, C.c_double]
        L.ppo_kat_min3.restype = C.c_int
        L.ppo_kat_min3.argtypes = [c_double_p]
        L.ppo_kat_min_index.restype = C.c_int
        L.ppo_kat_min_index.argtypes = [c_double_p, C.c_int]
        L.ppo_kat_max_index.restype = C.c_int
        L.ppo_kat_max_index.argtypes = [c_double_p, C.c_int]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _view(ptr, n, dtype):
    """numpy view of n items behind a ctypes pointer (no copy)."""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    addr = C.addressof(ptr.contents)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype)


class Mesh:
    """Oracle mesh (derives Omega_h-style adjacency; see oracle/ppo_mesh.c)."""

    def __init__(self, dim, coords, elem2verts, class_id=None):
        coords = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, dim)
        e2v = np.ascontiguousarray(elem2verts, dtype=np.int32).reshape(-1, dim + 1)
        cid = None if class_id is None else np.ascontiguousarray(class_id, dtype=np.int32)
        self._keep = (coords, e2v, cid)
        self.p = lib().ppo_mesh_create(dim, coords.shape[0], _dp(coords), e2v.shape[0], _ip(e2v),
                                       _ip(cid) if cid is not None else None)
        s = self.p.contents
        self.dim, self.nverts, self.nelems, self.nsides = s.dim, s.nverts, s.nelems, s.nsides
        nv = dim + 1
        self.coords = _view(s.coords, self.nverts * dim, np.float64).reshape(-1, dim)
        self.elem2verts = _view(s.elem2verts, self.nelems * nv, np.int32).reshape(-1, nv)
        self.class_id = _view(s.class_id, self.nelems, np.int32)
        self.elem2sides = _view(s.elem2sides, self.nelems * nv, np.int32).reshape(-1, nv)
        self.side2verts = _view(s.side2verts, self.nsides * dim, np.int32).reshape(-1, dim)
        self.side2elems_off = _view(s.side2elems_off, self.nsides + 1, np.int32)
        self.side2elems = _view(s.side2elems, int(self.side2elems_off[-1]), np.int32)
        self.side_exposed = _view(s.side_exposed, self.nsides, np.int8)
        self.elem_measure = _view(s.elem_measure, self.nelems, np.float64)
        self.dual_off = _view(s.dual_off, self.nelems + 1, np.int32)
        self.dual_elems = _view(s.dual_elems, int(self.dual_off[-1]), np.int32)
        self.vert2elems_off = _view(s.vert2elems_off, self.nverts + 1, np.int32)
        self.vert2elems = _view(s.vert2elems, int(self.vert2elems_off[-1]), np.int32)

    def tolerance(self):
        return lib().ppo_compute_tolerance_from_area(self.p)

    def __del__(self):
        try:
            lib().ppo_mesh_destroy(self.p)
        except Exception:
            pass


# member type descriptors: (numpy dtype, ncomp)
PARTICLE_XGCM = [(np.float64, 3), (np.float64, 3), (np.int32, 1), (np.float32, 1), (np.float32, 1)]
PARTICLE_PUSH = [(np.float64, 3), (np.float64, 3), (np.int32, 1)]
PERF160 = [(np.float64, 17), (np.int32, 4), (np.int64, 1)]


def _member_meta(members):
    mb = np.array([np.dtype(d).itemsize for d, _ in members], dtype=np.int32)
    mc = np.array([n for _, n in members], dtype=np.int32)
    return mb, mc


def _info_ptrs(members, info, n):
    """info: list of arrays shaped (ncomp, n) (component-major) or (n,) -> void* array"""
    if info is None:
        return None, None
    keep = []
    arr = (C.c_void_p * len(members))()
    for i, ((dt, nc), a) in enumerate(zip(members, info)):
        a = np.ascontiguousarray(np.asarray(a, dtype=dt).reshape(nc, n))
        keep.append(a)
        arr[i] = a.ctypes.data
    return arr, keep


class PS:
    """Oracle particle structure (SCS or CSR)."""

    def __init__(self, p, members):
        self.p = p
        self.members = members

    @classmethod
    def scs(cls, members, ne, ppe, C_max=1, sigma=2**31 - 1, V=1024, gids=None,
            pad_strat=0, shuffle_padding=0.1, extra_padding=0.05, particle_elements=None,
            particle_info=None):
        ppe = np.ascontiguousarray(ppe, dtype=np.int32)
        np_ = int(ppe.sum())
        mb, mc = _member_meta(members)
        pe = None if particle_elements is None else np.ascontiguousarray(particle_elements,
                                                                          dtype=np.int32)
        arr, keep = _info_ptrs(members, particle_info, np_)
        g = None if gids is None else np.ascontiguousarray(gids, dtype=np.int64)
        p = lib().ppo_scs_create(C_max, sigma, V, ne, np_, _ip(ppe),
                                 g.ctypes.data_as(c_long_p) if g is not None else None, pad_strat,
                                 shuffle_padding, extra_padding, len(members), _ip(mb), _ip(mc),
                                 _ip(pe) if pe is not None else None, arr)
        return cls(p, members)

    @classmethod
    def csr(cls, members, ne, ppe, gids=None, padding_amount=1.05, particle_elements=None,
            particle_info=None):
        ppe = np.ascontiguousarray(ppe, dtype=np.int32)
        np_ = int(ppe.sum())
        mb, mc = _member_meta(members)
        pe = None if particle_elements is None else np.ascontiguousarray(particle_elements,
                                                                          dtype=np.int32)
        arr, keep = _info_ptrs(members, particle_info, np_)
        g = None if gids is None else np.ascontiguousarray(gids, dtype=np.int64)
        p = lib().ppo_csr_create(ne, np_, _ip(ppe),
                                 g.ctypes.data_as(c_long_p) if g is not None else None,
                                 padding_amount, len(members), _ip(mb), _ip(mc),
                                 _ip(pe) if pe is not None else None, arr)
        return cls(p, members)

    # --- scalar accessors
    @property
    def s(self):
        return self.p.contents

    def capacity(self):
        return self.s.capacity

    def nPtcls(self):
        return self.s.num_ptcls

    def nElems(self):
        return self.s.num_elems

    def numRows(self):
        return self.s.num_rows

    def alloc(self):
        return self.s.alloc

    def set_try_shuffling(self, v):
        self.s.try_shuffling = int(v)

    def layout(self):
        s = self.s
        d = dict(kind=s.kind, C=s.C, V=s.V, num_chunks=s.num_chunks, num_slices=s.num_slices,
                 capacity=s.capacity, num_rows=s.num_rows)
        if s.kind == 0:
            d["offsets"] = _view(s.offsets, s.num_slices + 1, np.int32).copy()
            d["slice_to_chunk"] = _view(s.slice_to_chunk, s.num_slices, np.int32).copy()
            d["row_to_element"] = _view(s.row_to_element, s.num_rows, np.int32).copy()
            d["element_to_row"] = _view(s.element_to_row, s.num_rows, np.int32).copy()
            d["mask"] = _view(s.mask, s.capacity, np.uint8).copy()
        else:
            d["offsets"] = _view(s.offsets, s.num_elems + 1, np.int32).copy()
        return d

    def member(self, m):
        """numpy view (ncomp, alloc) of member m (component-major SoA)."""
        dt, nc = self.members[m]
        s = self.s
        n = s.alloc * nc
        addr = s.data[m]
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(addr)
        return np.frombuffer(buf, dtype=dt).reshape(nc, s.alloc)

    def slot_info(self):
        cap = self.capacity()
        e = np.empty(cap, dtype=np.int32)
        m = np.empty(cap, dtype=np.uint8)
        lib().ppo_ps_slot_info(self.p, _ip(e), m.ctypes.data_as(c_ubyte_p))
        return e, m

    def rebuild(self, new_element, new_particle_elements=None, new_particle_info=None):
        ne = np.ascontiguousarray(new_element, dtype=np.int32)
        n_new = 0 if new_particle_elements is None else len(new_particle_elements)
        npe = None if n_new == 0 else np.ascontiguousarray(new_particle_elements, dtype=np.int32)
        arr, keep = _info_ptrs(self.members, new_particle_info, n_new) if n_new else (None, None)
        lib().ppo_ps_rebuild(self.p, _ip(ne), n_new, _ip(npe) if npe is not None else None, arr)

    def get_pids(self):
        off = np.zeros(self.nElems() + 1, dtype=np.int32)
        pids = np.zeros(max(self.nPtcls(), 1), dtype=np.int32)
        lib().ppo_ps_get_pids(self.p, _ip(off), _ip(pids))
        return off, pids[:self.nPtcls()]

    def metrics(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        lib().ppo_scs_metrics(self.p, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def __del__(self):
        try:
            lib().ppo_ps_destroy(self.p)
        except Exception:
            pass


# ---------------------------------------------------------------- operators
def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().ppo_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def elliptical_setup(ps, h, k, d, m_x=0, m_b=3, m_phi=4):
    lib().ppo_elliptical_setup(ps.p, m_x, m_b, m_phi, h, k, d)


def elliptical_push(ps, mesh, h, k, d, deg, trig=1, m_xtgt=1, m_b=3, m_phi=4):
    lib().ppo_elliptical_push(ps.p, mesh.p, m_xtgt, m_b, m_phi, h, k, d, deg, trig)


def toroidal_push(ps, mesh, h, k, d, deg, trig=1, m_x=0, m_xtgt=1, m_b=3, m_phi=4):
    lib().ppo_toroidal_push(ps.p, mesh.p, m_x, m_xtgt, m_b, m_phi, h, k, d, deg, trig)


def linear_push(ps, distance, dx, dy, dz, m_x=0, m_xtgt=1):
    lib().ppo_linear_push(ps.p, m_x, m_xtgt, distance, dx, dy, dz)


def update_positions(ps, m_x=0, m_xtgt=1):
    lib().ppo_update_positions(ps.p, m_x, m_xtgt)


def pseudo_push160(ps, parent_elm_data):
    a = np.ascontiguousarray(parent_elm_data, dtype=np.float64)
    lib().ppo_pseudo_push160(ps.p, _dp(a))


def push_boris(x, y, z, xp, yp, zp, vx, vy, vz, ex, ey, ez, br, bt, bz, dt):
    arrs = [x, y, z, xp, yp, zp, vx, vy, vz, ex, ey, ez, br, bt, bz]
    for a in arrs:
        assert a.dtype == np.float64 and a.flags.c_contiguous
    lib().ppo_push_boris(len(x), *[_dp(a) for a in arrs], dt)


def search_mesh_2d(mesh, ps, elem_ids=None, looplimit=0, m_x=0, m_xtgt=1, m_pid=2):
    cap = ps.capacity()
    if elem_ids is None:
        elem_ids = np.full(cap, -1, dtype=np.int32)
    loops = C.c_int()
    found = lib().ppo_search_mesh_2d(mesh.p, ps.p, m_x, m_xtgt, m_pid, _ip(elem_ids), looplimit,
                                     C.byref(loops))
    return bool(found), elem_ids, loops.value


def search_mesh(mesh, ps, elem_ids=None, require_intersection=False, looplimit=0, m_x=0, m_xtgt=1,
                m_pid=2):
    cap = ps.capacity()
    seeded = elem_ids is not None
    if elem_ids is None:
        elem_ids = np.full(cap, -1, dtype=np.int32)
    inter_faces = np.full(cap, -1, dtype=np.int32)
    inter_points = np.zeros(cap * mesh.dim, dtype=np.float64)
    loops, notin = C.c_int(), C.c_int()
    found = lib().ppo_search_mesh(mesh.p, ps.p, m_x, m_xtgt, m_pid, _ip(elem_ids), int(seeded),
                                  int(require_intersection), _ip(inter_faces), _dp(inter_points),
                                  looplimit, C.byref(loops), C.byref(notin))
    return dict(found=bool(found), elem_ids=elem_ids, inter_faces=inter_faces,
                inter_points=inter_points.reshape(cap, mesh.dim), loops=loops.value,
                not_in_elem=notin.value)


_TRACE_FUNCTOR = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, c_int_p, c_int_p,
                             c_double_p, c_int_p, C.c_int, C.c_int)


def trace_particle_through_mesh(mesh, ps, func, elem_ids=None, require_intersection=False,
                                looplimit=0, m_x=0, m_xtgt=1, m_pid=2):
    """trace_particle_through_mesh (adjacency.tpp:460-615) with a caller-supplied functor:
    func(state) is called once per walk iteration where the reference calls `func` (tpp:563);
    state = dict of numpy views (elem_ids, inter_faces, last_exit, inter_points, ptcl_done)."""
    cap = ps.capacity()
    dim = mesh.dim
    seeded = elem_ids is not None
    if elem_ids is None:
        elem_ids = np.full(cap, -1, dtype=np.int32)
    inter_faces = np.full(cap, -1, dtype=np.int32)
    inter_points = np.zeros(cap * dim, dtype=np.float64)

    def _cb(ctx, mesh_p, ps_p, e, f, le, ipt, done, mx, mxt):
        st = dict(elem_ids=np.ctypeslib.as_array(e, (cap,)), inter_faces=np.ctypeslib.as_array(f, (cap,)),
                  last_exit=np.ctypeslib.as_array(le, (cap,)),
                  inter_points=np.ctypeslib.as_array(ipt, (cap * dim,)),
                  ptcl_done=np.ctypeslib.as_array(done, (cap,)), mesh=mesh, ps=ps,
                  require_intersection=bool(require_intersection))
        func(st)

    cb = _TRACE_FUNCTOR(_cb)
    L = lib()
    L.ppo_trace_particle_through_mesh.restype = C.c_int
    L.ppo_trace_particle_through_mesh.argtypes = [
        C.POINTER(_MeshS), C.POINTER(_PsS), C.c_int, C.c_int, C.c_int, c_int_p, C.c_int, C.c_int,
        c_int_p, c_double_p, C.c_int, c_int_p, c_int_p, _TRACE_FUNCTOR, C.c_void_p]
    loops, notin = C.c_int(), C.c_int()
    found = L.ppo_trace_particle_through_mesh(mesh.p, ps.p, m_x, m_xtgt, m_pid, _ip(elem_ids),
                                              int(seeded), int(require_intersection),
                                              _ip(inter_faces), _dp(inter_points), looplimit,
                                              C.byref(loops), C.byref(notin), cb, None)
    return dict(found=bool(found), elem_ids=elem_ids, inter_faces=inter_faces,
                inter_points=inter_points, loops=loops.value, not_in_elem=notin.value)


def search_mesh_legacy3d(mesh, ps, elem_ids=None, looplimit=0, m_x=0, m_xtgt=1, m_pid=2):
    cap = ps.capacity()
    seeded = elem_ids is not None
    if elem_ids is None:
        elem_ids = np.full(cap, -1, dtype=np.int32)
    xface = np.full(cap, -1, dtype=np.int32)
    xpoints = np.zeros(cap * 3, dtype=np.float64)
    loops = C.c_int()
    found = lib().ppo_search_mesh_legacy3d(mesh.p, ps.p, m_x, m_xtgt, m_pid, _ip(elem_ids),
                                           int(seeded), _dp(xpoints), _ip(xface), looplimit,
                                           C.byref(loops))
    return dict(found=found, elem_ids=elem_ids, xface=xface, xpoints=xpoints.reshape(cap, 3),
                loops=loops.value)


def search_mesh_3d(mesh, ps, elem_ids=None, looplimit=0, m_x=0, m_xtgt=1, m_pid=2):
    """search_mesh_3d (adjacency.hpp:314-555)."""
    cap = ps.capacity()
    seeded = elem_ids is not None
    if elem_ids is None:
        elem_ids = np.full(cap, -1, dtype=np.int32)
    xface = np.full(cap, -1, dtype=np.int32)
    xpoints = np.zeros(cap * 3, dtype=np.float64)
    loops = C.c_int()
    found = lib().ppo_search_mesh_3d(mesh.p, ps.p, m_x, m_xtgt, m_pid, _ip(elem_ids), int(seeded),
                                     _dp(xpoints), _ip(xface), looplimit, C.byref(loops))
    return dict(found=found, elem_ids=elem_ids, xface=xface, xpoints=xpoints.reshape(cap, 3),
                loops=loops.value)


def search_mesh_2d_pt(mesh, orig, dest, initial_elem, looplimit=0, pid=0):
    o = np.ascontiguousarray(orig, dtype=np.float64)
    d = np.ascontiguousarray(dest, dtype=np.float64)
    loops = C.c_int()
    e = lib().ppo_search_mesh_2d_pt(mesh.p, _dp(o), _dp(d), pid, initial_elem, C.byref(loops),
                                    looplimit)
    return e, loops.value


def create_gyro_ring_mappings(mesh, rmax=0.038, gnr=3, gppr=8, theta=0.0, trig=0):
    n = mesh.nverts * gnr * gppr * (mesh.dim + 1)
    f = np.empty(n, dtype=np.int32)
    b = np.empty(n, dtype=np.int32)
    lib().ppo_create_gyro_ring_mappings(mesh.p, rmax, gnr, gppr, theta, trig, _ip(f), _ip(b))
    return f, b


def gyro_scatter(mesh, ps, v2v, rmax=0.038, gnr=3, gppr=8):
    v2v = np.ascontiguousarray(v2v, dtype=np.int32)
    w = np.zeros(mesh.nverts, dtype=np.float64)
    lib().ppo_gyro_scatter(mesh.p, ps.p, _ip(v2v), rmax, gnr, gppr, _dp(w))
    return w


def gyro_scatter_radius(mesh, ps, radius, v2v, weight=None, rmax=0.038, gnr=3, gppr=8):
    """gyroScatter with a per-particle radius (slot-indexed) and optional weight -> (field, clipped)"""
    v2v = np.ascontiguousarray(v2v, dtype=np.int32)
    radius = np.ascontiguousarray(radius, dtype=np.float64)
    wt = None if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
    w = np.zeros(mesh.nverts, dtype=np.float64)
    clipped = C.c_int(0)
    lib().ppo_gyro_scatter_radius(mesh.p, ps.p, _dp(radius), _dp(wt) if wt is not None else None, _ip(v2v), rmax,
                                  gnr, gppr, _dp(w), C.byref(clipped))
    return w, clipped.value


def gather_tet_vtx(mesh, ps, field, dof=1, elem_ids=None, m_x=0):
    cap = max(ps.capacity(), 1)
    field = np.ascontiguousarray(field, dtype=np.float64)
    out = np.zeros((dof, cap), dtype=np.float64)
    bad = C.c_int(0)
    ids = None if elem_ids is None else _ip(np.ascontiguousarray(elem_ids, dtype=np.int32))
    lib().ppo_gather_tet_vtx(mesh.p, ps.p, m_x, ids, _dp(field), dof, _dp(out), C.byref(bad))
    return out, bad.value


def interp2d_field(ps, data, gridx0, gridz0, dx, dz, nx, nz, cyl_symm=True, ncomp=1, comp=0, m_x=0):
    data = np.ascontiguousarray(data, dtype=np.float64)
    out = np.zeros(max(ps.capacity(), 1), dtype=np.float64)
    lib().ppo_interp2d_field(ps.p, m_x, _dp(data), gridx0, gridz0, dx, dz, nx, nz, int(cyl_symm), ncomp,
                             comp, _dp(out))
    return out


def interp2d_vector(ps, data3, gridx0, gridz0, dx, dz, nx, nz, cyl_symm=False, m_x=0):
    data3 = np.ascontiguousarray(data3, dtype=np.float64)
    out = np.zeros((3, max(ps.capacity(), 1)), dtype=np.float64)
    lib().ppo_interp2d_vector(ps.p, m_x, _dp(data3), gridx0, gridz0, dx, dz, nx, nz, int(cyl_symm), _dp(out))
    return out


def interp3d_field(ps, gridx, gridy, gridz, data, m_x=0):
    gx, gy, gz, d = (np.ascontiguousarray(a, dtype=np.float64) for a in (gridx, gridy, gridz, data))
    out = np.zeros(max(ps.capacity(), 1), dtype=np.float64)
    lib().ppo_interp3d_field(ps.p, m_x, len(gx), len(gy), len(gz), _dp(gx), _dp(gy), _dp(gz), _dp(d), _dp(out))
    return out


def avg_ptcl_density(mesh, ps):
    ec = np.zeros(mesh.nelems, dtype=np.float64)
    vd = np.zeros(mesh.nverts, dtype=np.float64)
    lib().ppo_avg_ptcl_density(mesh.p, ps.p, _dp(ec), _dp(vd))
    return ec, vd


def set_unsafe_procs(ps, elems, safe, owners, rank):
    cap = ps.capacity()
    elems = np.ascontiguousarray(elems, dtype=np.int32)
    safe = np.ascontiguousarray(safe, dtype=np.uint8)
    owners = np.ascontiguousarray(owners, dtype=np.int32)
    ne = np.zeros(cap, dtype=np.int32)
    npr = np.zeros(cap, dtype=np.int32)
    lib().ppo_set_unsafe_procs(ps.p, _ip(elems), safe.ctypes.data_as(c_ubyte_p), _ip(owners), rank,
                               _ip(ne), _ip(npr))
    return ne, npr


# ---------------------------------------------------------------- geometry KATs
def barycentric_tet(M, p, vol):
    M = np.ascontiguousarray(M, dtype=np.float64).reshape(12)
    p = np.ascontiguousarray(p, dtype=np.float64)
    a, b, c = (np.zeros(4) for _ in range(3))
    lib().ppo_kat_barycentric_tet(_dp(M), _dp(p), float(vol), _dp(a), _dp(b), _dp(c))
    return a, b, c


def barycentric_tri(fc, p, area):
    fc = np.ascontiguousarray(fc, dtype=np.float64).reshape(6)
    p = np.ascontiguousarray(p, dtype=np.float64)
    b = np.zeros(3)
    lib().ppo_kat_barycentric_tri(_dp(fc), _dp(p), float(area), _dp(b))
    return b


def ray_triangle(fv, o, d, tol, flip, segment=False):
    fv = np.ascontiguousarray(fv, dtype=np.float64).reshape(9)
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    xp = np.zeros(3)
    dproj, close, par = C.c_double(), C.c_double(), C.c_double()
    hit = lib().ppo_kat_ray_triangle(_dp(fv), _dp(o), _dp(d), tol, int(flip), int(segment), _dp(xp),
                                     C.byref(dproj), C.byref(close), C.byref(par))
    return bool(hit), xp, dproj.value, close.value, par.value


def line_edge_2d(ev, o, d, tol, flip):
    ev = np.ascontiguousarray(ev, dtype=np.float64).reshape(4)
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    xp = np.zeros(2)
    hit = lib().ppo_kat_line_edge_2d(_dp(ev), _dp(o), _dp(d), tol, int(flip), _dp(xp))
    return bool(hit), xp


def line_triangle_simple(abc, o, d, reverse=False, tol=0.0):
    abc = np.ascontiguousarray(abc, dtype=np.float64).reshape(9)
    o = np.ascontiguousarray(o, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    xp = np.zeros(3)
    dproj = C.c_double()
    hit = lib().ppo_kat_line_triangle_simple(_dp(abc), _dp(o), _dp(d), int(reverse), tol, _dp(xp),
                                             C.byref(dproj))
    return bool(hit), xp, dproj.value


def all_positive(a, tol=1e-10):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return bool(lib().ppo_kat_all_positive(_dp(a), len(a), tol))


def min3(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().ppo_kat_min3(_dp(a))


def min_index(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().ppo_kat_min_index(_dp(a), len(a))


def max_index(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().ppo_kat_max_index(_dp(a), len(a))


def closest_point_on_triangle(abc, p, wnormal=False, reg0=-1):
    """closest_point_on_triangle[_wnormal] (adjacency.hpp:824-1009) -> (q[3], region)."""
    abc = np.ascontiguousarray(abc, dtype=np.float64).reshape(9)
    p = np.ascontiguousarray(p, dtype=np.float64)
    q = np.zeros(3)
    reg = C.c_int(reg0)
    lib().ppo_closest_point_on_triangle(_dp(abc), _dp(p), int(wnormal), _dp(q), C.byref(reg))
    return q, reg.value


def redistribute_particles(ps, percent_moved, seed=0):
    """redistribute_particles (Distribute.h:28-89), uniform strategy -> new_elems[capacity]."""
    out = np.empty(max(ps.capacity(), 1), dtype=np.int32)
    L = lib()
    L.ppo_redistribute_particles.argtypes = [C.POINTER(_PsS), C.c_double, C.c_ulonglong, c_int_p]
    L.ppo_redistribute_particles.restype = None
    L.ppo_redistribute_particles(ps.p, float(percent_moved), int(seed), _ip(out))
    return out[:ps.capacity()]


def redistribute_particles_dist(ps, strat, percent_moved, seed=0):
    """redistribute_particles with distribution strategy 1-4 (Distribute.h:28-89 + Distribute.cpp) -> new_elems"""
    out = np.full(max(ps.capacity(), 1), -1, dtype=np.int32)
    L = lib()
    L.ppo_redistribute_particles_dist.argtypes = [C.POINTER(_PsS), C.c_int, C.c_double, C.c_ulonglong, c_int_p]
    L.ppo_redistribute_particles_dist.restype = None
    L.ppo_redistribute_particles_dist(ps.p, int(strat), float(percent_moved), int(seed), _ip(out))
    return out[:ps.capacity()]


def set_threads(n):
    """OpenMP threads of the per-particle loops (results do not depend on it)."""
    lib().ppo_set_threads(int(n))


def max_threads():
    """host cores this process may really use: processors, affinity mask and cgroup CPU quota"""
    n = int(lib().ppo_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def bfs_buffer_layers(mesh, owner, rank, comm_size, safe_layers, ghost_layers, bridge_dim=0):
    """bfsBufferLayers (pumipic_part_construct.cpp:407-437) -> (is_safe[nelems] u8, has_part[comm_size])"""
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    safe = np.zeros(max(mesh.nelems, 1), dtype=np.uint8)
    part = np.zeros(comm_size, dtype=np.int32)
    L = lib()
    L.ppo_bfs_buffer_layers.restype = None
    L.ppo_bfs_buffer_layers.argtypes = [C.POINTER(_MeshS), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        c_int_p, c_ubyte_p, c_int_p]
    L.ppo_bfs_buffer_layers(mesh.p, bridge_dim, rank, comm_size, safe_layers, ghost_layers, _ip(owner),
                            safe.ctypes.data_as(c_ubyte_p), _ip(part))
    return safe[:mesh.nelems], part


def bfs_safe_inward(mesh, owner, rank, safe_layers, has_part, bridge_dim=0):
    """bfsSafeInward (pumipic_part_construct.cpp:439-468) -> safe[nelems] u8"""
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    part = np.ascontiguousarray(has_part, dtype=np.int32)
    safe = np.zeros(max(mesh.nelems, 1), dtype=np.uint8)
    L = lib()
    L.ppo_bfs_safe_inward.restype = None
    L.ppo_bfs_safe_inward.argtypes = [C.POINTER(_MeshS), C.c_int, C.c_int, C.c_int, c_int_p, c_int_p,
                                      c_ubyte_p]
    L.ppo_bfs_safe_inward(mesh.p, bridge_dim, rank, safe_layers, _ip(owner), _ip(part),
                          safe.ctypes.data_as(c_ubyte_p))
    return safe[:mesh.nelems]
