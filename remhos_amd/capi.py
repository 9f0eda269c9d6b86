"""ctypes binding of the C ABI in include/rmh.h (the drop-in boundary).

Only plumbing lives here: every numeric operation is a HIP kernel inside librmh.so.
The loader fails loudly when the extension is missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librmh.so")

# every symbol include/rmh.h declares (tests/test_capi_symbols.py checks the header against this)
SYMBOLS = [
    "rmh_create", "rmh_destroy", "rmh_last_error", "rmh_version", "rmh_set_stream", "rmh_stream_create_reserving", "rmh_stream_destroy", "rmh_batch_order", "rmh_setup",
    "rmh_set_ghost_u", "rmh_set_ghost_minmax", "rmh_halo_pack", "rmh_ho_apply", "rmh_lumped_mass",
    "rmh_compute_lumped_mass", "rmh_lo_massavg", "rmh_lo_rdsubcell", "rmh_lo_rd", "rmh_elem_minmax", "rmh_bounds",
    "rmh_fct_clipscale", "rmh_limit_fused", "rmh_limit_fused_lo", "rmh_stage_fused", "rmh_stage_fused_range", "rmh_stage_fused_chain",
    "rmh_halo_pack_records", "rmh_set_ghost_records", "rmh_timers", "rmh_reset_timers", "rmh_enable_timers",
    "rmh_last_cg_iters", "rmh_set_mass_tol", "rmh_get_mass_tol", "rmh_set_mass_completion", "rmh_set_lo_type", "rmh_set_bounds_type", "rmh_set_dt_control",
    "rmh_dt_estimate_reset", "rmh_dt_estimate_update", "rmh_dt_estimate_get", "rmh_invalidate_extrema",
    "rmh_exchange_setup", "rmh_comm_unique_id", "rmh_comm_init", "rmh_comm_attach", "rmh_comm_connect_local",
    "rmh_exchange_begin", "rmh_exchange_end", "rmh_exchange_minmax_begin", "rmh_exchange_minmax_end", "rmh_exchange_buffers", "rmh_exchange_peer", "rmh_allreduce", "rmh_comm_count",
    "rmh_build_tables", "rmh_build_tables_2d", "rmh_product_ratio", "rmh_elem_minmax_masked", "rmh_fct_product",
    "rmh_check_violation",
]


class RmhLayout(C.Structure):
    _fields_ = [
        ("dim", C.c_int),
        ("order", C.c_int),
        ("mesh_order", C.c_int),
        ("exec_mode", C.c_int),
        ("ne_owned", C.c_int),
        ("ne_ghost", C.c_int),
        ("x0", C.c_void_p),
        ("vel", C.c_void_p),
        ("face_nbr", C.c_void_p),
        ("stencil27", C.c_void_p),
        ("subcell_vel", C.c_void_p),
        ("device", C.c_int),
    ]


class RmhExchangeDesc(C.Structure):
    _fields_ = [
        ("n_peers", C.c_int),
        ("peer_rank", C.POINTER(C.c_int)),
        ("send_count", C.POINTER(C.c_int)),
        ("send_elems", C.POINTER(C.POINTER(C.c_int))),
        ("recv_first", C.POINTER(C.c_int)),
        ("recv_count", C.POINTER(C.c_int)),
    ]


class RmhViolation(C.Structure):
    """rmh_violation (include/rmh.h): verdict of rmh_check_violation"""
    _fields_ = [
        ("count", C.c_longlong),
        ("first", C.c_longlong),
        ("over", C.c_double),
        ("under", C.c_double),
        ("first_min", C.c_double),
        ("first_value", C.c_double),
        ("first_max", C.c_double),
    ]


class RmhError(RuntimeError):
    pass


def load_library(path: str | None = None) -> C.CDLL:
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RmhError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  remhos_amd has no CPU fallback."
        )
    if os.path.basename(path) != "librmh_emu.so":
        # PyTorch ships its own HIP runtime: it must be in the process before this library pulls one in, otherwise the
        # two runtimes fight over the device (torch then reports "No HIP GPUs are available")
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    p, d, i = C.c_void_p, C.c_double, C.c_int
    lib.rmh_create.argtypes = [C.POINTER(RmhLayout), C.POINTER(p)]
    lib.rmh_destroy.argtypes = [p]
    lib.rmh_destroy.restype = None
    lib.rmh_last_error.restype = C.c_char_p
    lib.rmh_version.restype = C.c_char_p
    lib.rmh_set_stream.argtypes = [p, p]
    lib.rmh_stream_create_reserving.argtypes = [i, i, C.POINTER(p)]
    if hasattr(lib, "rmh_batch_order"):  # (older development builds of the library lack it: tools/kbench.py A/B runs)
        lib.rmh_batch_order.argtypes = [p, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.rmh_stream_destroy.argtypes = [p]
    lib.rmh_setup.argtypes = [p, d]
    lib.rmh_set_ghost_u.argtypes = [p, p]
    lib.rmh_set_ghost_minmax.argtypes = [p, p, p]
    lib.rmh_halo_pack.argtypes = [p, p, p, i, p, p, p]
    lib.rmh_ho_apply.argtypes = [p, p, p]
    lib.rmh_lumped_mass.argtypes = [p]
    lib.rmh_lumped_mass.restype = p
    lib.rmh_compute_lumped_mass.argtypes = [p, d, p]
    lib.rmh_lo_massavg.argtypes = [p, p, p, d, p]
    lib.rmh_lo_rdsubcell.argtypes = [p, p, p]
    lib.rmh_lo_rd.argtypes = [p, p, p]
    lib.rmh_elem_minmax.argtypes = [p, p, p, p]
    lib.rmh_bounds.argtypes = [p, p, p, p, p]
    lib.rmh_fct_clipscale.argtypes = [p, p, p, p, p, p, p, d, p]
    lib.rmh_limit_fused.argtypes = [p, p, p, d, p, p, d, d, d, p]
    lib.rmh_limit_fused_lo.argtypes = [p, p, p, p, d, p, p, d, d, d, p]
    lib.rmh_stage_fused.argtypes = [p, p, d, p, d, d, d, p, p]
    lib.rmh_halo_pack_records.argtypes = [p, p, p, C.c_int, p]
    lib.rmh_set_ghost_records.argtypes = [p, p]
    lib.rmh_stage_fused_range.argtypes = [p, p, d, p, d, d, d, p, p, C.c_int, C.c_int, C.c_int]
    lib.rmh_stage_fused_chain.argtypes = [p, p, d, p, d, d, d, p, p, C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.POINTER(C.c_ulonglong)]
    lib.rmh_timers.argtypes = [p, C.POINTER(d * 4)]
    lib.rmh_reset_timers.argtypes = [p]
    lib.rmh_enable_timers.argtypes = [p, i]
    lib.rmh_last_cg_iters.argtypes = [p, C.POINTER(i)]
    lib.rmh_set_mass_tol.argtypes = [p, d, d, i]
    lib.rmh_get_mass_tol.argtypes = [p, C.POINTER(d), C.POINTER(d), C.POINTER(i)]
    lib.rmh_set_mass_completion.argtypes = [p, i, i]
    lib.rmh_set_lo_type.argtypes = [p, i]
    lib.rmh_set_bounds_type.argtypes = [p, i]
    lib.rmh_invalidate_extrema.argtypes = [p]
    lib.rmh_set_dt_control.argtypes = [p, i]
    lib.rmh_dt_estimate_reset.argtypes = [p]
    lib.rmh_dt_estimate_update.argtypes = [p, p, p, p, p]
    lib.rmh_dt_estimate_get.argtypes = [p, C.POINTER(d)]
    ll = C.c_longlong
    lib.rmh_exchange_setup.argtypes = [p, C.POINTER(RmhExchangeDesc), i]
    lib.rmh_comm_unique_id.argtypes = [C.c_char_p]
    lib.rmh_comm_init.argtypes = [p, C.c_char_p, i, i]
    lib.rmh_comm_attach.argtypes = [p, p]
    lib.rmh_comm_connect_local.argtypes = [p, i, p, i]
    lib.rmh_exchange_begin.argtypes = [p, p]
    lib.rmh_exchange_end.argtypes = [p]
    lib.rmh_exchange_minmax_begin.argtypes = [p, p, p]
    lib.rmh_exchange_minmax_end.argtypes = [p]
    lib.rmh_exchange_buffers.argtypes = [p, C.POINTER(p), C.POINTER(ll), C.POINTER(p), C.POINTER(ll)]
    lib.rmh_exchange_peer.argtypes = [p, i, C.POINTER(i), C.POINTER(ll), C.POINTER(ll), C.POINTER(ll), C.POINTER(ll)]
    lib.rmh_allreduce.argtypes = [p, C.POINTER(d), i, i]
    lib.rmh_build_tables.argtypes = [i, i, p, p, p]
    lib.rmh_build_tables_2d.argtypes = [i, i, p, p, p]
    lib.rmh_product_ratio.argtypes = [p, p, p, p, p, p]
    lib.rmh_elem_minmax_masked.argtypes = [p, p, p, p, p, p]
    lib.rmh_fct_product.argtypes = [p, p, p, p, p, p, p, p, p, d, p]
    lib.rmh_check_violation.argtypes = [p, p, d, p, p, p, p, d, p, C.POINTER(RmhViolation)]
    return lib


def _ptr(x):
    """device pointer of a torch tensor / host pointer of a numpy array / raw int / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x.ctypes.data


class Context:
    """One rmh_ctx: thin, argument-checking wrapper.  Vector arguments are torch CUDA tensors
    (or host numpy arrays when `lib` is the g++ emulation build used by the CPU tests)."""

    def __init__(self, lib, *, order, exec_mode, x0, vel, face_nbr, stencil27, ne_ghost=0,
                 subcell_vel=None, device=0, mesh_order=2):
        import numpy as np

        self.lib = lib
        self._keep = []
        ne = face_nbr.shape[0]
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        vel = np.ascontiguousarray(vel, dtype=np.float64)
        face_nbr = np.ascontiguousarray(face_nbr, dtype=np.int32)
        stencil27 = np.ascontiguousarray(stencil27, dtype=np.int32)
        dim = x0.shape[1]  # 3: hexahedra; 2: quadrilaterals (nodes [ne][2][9], 4 faces, 3 x 3 stencil -- include/rmh.h)
        assert dim in (2, 3) and x0.shape == (ne, dim, 3**dim) and vel.shape == (ne, dim, 3**dim)
        assert face_nbr.shape == (ne, 2 * dim) and stencil27.shape == (ne, 3**dim)
        L = RmhLayout()
        L.dim, L.order, L.mesh_order, L.exec_mode = dim, order, mesh_order, exec_mode
        L.ne_owned, L.ne_ghost = ne, ne_ghost
        L.x0, L.vel = x0.ctypes.data, vel.ctypes.data
        L.face_nbr, L.stencil27 = face_nbr.ctypes.data, stencil27.ctypes.data
        if subcell_vel is not None:
            subcell_vel = np.ascontiguousarray(subcell_vel, dtype=np.float64)
            assert subcell_vel.shape == (ne, dim, (order + 1) ** dim)
            L.subcell_vel = subcell_vel.ctypes.data
        L.device = device
        self.ne, self.order, self.ndof, self.dim = ne, order, (order + 1) ** dim, dim
        h = C.c_void_p()
        self._check(lib.rmh_create(C.byref(L), C.byref(h)))
        self.h = h

    def _check(self, rc):
        if rc != 0:
            raise RmhError(f"rmh error {rc}: {self.lib.rmh_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.rmh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        self._check(self.lib.rmh_set_stream(self.h, stream_handle))

    def setup(self, t):
        self._check(self.lib.rmh_setup(self.h, float(t)))

    def set_ghost_u(self, ug):
        self._keep_ref("ug", ug)
        self._check(self.lib.rmh_set_ghost_u(self.h, _ptr(ug)))

    def set_ghost_minmax(self, gmin, gmax):
        self._keep_ref("gmm", (gmin, gmax))
        self._check(self.lib.rmh_set_ghost_minmax(self.h, _ptr(gmin), _ptr(gmax)))

    def set_ghost_records(self, rec):
        self._keep_ref("rec", rec)
        self._check(self.lib.rmh_set_ghost_records(self.h, _ptr(rec)))

    def halo_pack_records(self, u, send_elems, nsend, rec):
        self._check(self.lib.rmh_halo_pack_records(self.h, _ptr(u), _ptr(send_elems), int(nsend), _ptr(rec)))

    def _keep_ref(self, key, obj):
        self.__dict__["_ref_" + key] = obj

    def halo_pack(self, u, send_elems, nsend, rows, out_min, out_max):
        self._check(self.lib.rmh_halo_pack(self.h, _ptr(u), _ptr(send_elems), int(nsend), _ptr(rows), _ptr(out_min),
                                           _ptr(out_max)))

    def ho_apply(self, u, du):
        self._check(self.lib.rmh_ho_apply(self.h, _ptr(u), _ptr(du)))

    def lumped_mass_ptr(self):
        return self.lib.rmh_lumped_mass(self.h)

    def compute_lumped_mass(self, t, m):
        self._check(self.lib.rmh_compute_lumped_mass(self.h, float(t), _ptr(m)))

    def lo_massavg(self, u, du_ho, dt, du_lo):
        self._check(self.lib.rmh_lo_massavg(self.h, _ptr(u), _ptr(du_ho), float(dt), _ptr(du_lo)))

    def lo_rdsubcell(self, u, du_lo):
        self._check(self.lib.rmh_lo_rdsubcell(self.h, _ptr(u), _ptr(du_lo)))

    def lo_rd(self, u, du_lo):
        self._check(self.lib.rmh_lo_rd(self.h, _ptr(u), _ptr(du_lo)))

    def elem_minmax(self, u, xe_min, xe_max):
        self._check(self.lib.rmh_elem_minmax(self.h, _ptr(u), _ptr(xe_min), _ptr(xe_max)))

    def bounds(self, xe_min, xe_max, u_min, u_max):
        self._check(self.lib.rmh_bounds(self.h, _ptr(xe_min), _ptr(xe_max), _ptr(u_min), _ptr(u_max)))

    def fct_clipscale(self, u, m, du_ho, du_lo, u_min, u_max, dt, du):
        self._check(self.lib.rmh_fct_clipscale(self.h, _ptr(u), _ptr(m), _ptr(du_ho), _ptr(du_lo), _ptr(u_min),
                                               _ptr(u_max), float(dt), _ptr(du)))

    def limit_fused(self, u, du_ho, dt, du=None, x_base=None, a=0.0, b=1.0, dt_rk=0.0, y_out=None):
        self._check(self.lib.rmh_limit_fused(self.h, _ptr(u), _ptr(du_ho), float(dt), _ptr(du), _ptr(x_base),
                                             float(a), float(b), float(dt_rk), _ptr(y_out)))

    def limit_fused_lo(self, u, du_ho, du_lo, dt, du=None, x_base=None, a=0.0, b=1.0, dt_rk=0.0, y_out=None):
        self._check(self.lib.rmh_limit_fused_lo(self.h, _ptr(u), _ptr(du_ho), _ptr(du_lo), float(dt), _ptr(du),
                                                _ptr(x_base), float(a), float(b), float(dt_rk), _ptr(y_out)))

    def stage_fused(self, u, dt, y_out, x_base=None, a=0.0, b=1.0, dt_rk=0.0, du=None, token=0):
        """token: what the stage that wrote u returned (0: the extrema of u are recomputed); returns the token of y_out"""
        return self.stage_fused_range(u, dt, y_out, 0, self.ne, True, x_base=x_base, a=a, b=b, dt_rk=dt_rk, du=du, token=token)

    def stage_fused_range(self, u, dt, y_out, e_begin, e_end, finish, x_base=None, a=0.0, b=1.0, dt_rk=0.0, du=None, token=0):
        out = C.c_ulonglong(0)
        self._check(self.lib.rmh_stage_fused_chain(self.h, _ptr(u), float(dt), _ptr(x_base), float(a), float(b),
                                                   float(dt_rk), _ptr(y_out), _ptr(du), int(e_begin), int(e_end),
                                                   1 if finish else 0, int(token), C.byref(out)))
        return out.value

    def enable_timers(self, on=True):
        self._check(self.lib.rmh_enable_timers(self.h, 1 if on else 0))

    def reset_timers(self):
        self._check(self.lib.rmh_reset_timers(self.h))

    def timers(self):
        t = (C.c_double * 4)()
        self._check(self.lib.rmh_timers(self.h, C.byref(t)))
        return list(t)

    def last_cg_iters(self):
        n = C.c_int(0)
        self._check(self.lib.rmh_last_cg_iters(self.h, C.byref(n)))
        return n.value

    def batch_order(self, n_elements):
        """(layer_elements, batch_elements, chunk, weave) of a stage launch over n_elements elements: rmh_batch_order"""
        v = [C.c_int(0) for _ in range(4)]
        self._check(self.lib.rmh_batch_order(self.h, int(n_elements), *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def set_lo_type(self, lo_type):
        self._check(self.lib.rmh_set_lo_type(self.h, int(lo_type)))

    def invalidate_extrema(self):
        self._check(self.lib.rmh_invalidate_extrema(self.h))

    def set_bounds_type(self, bt):
        self._check(self.lib.rmh_set_bounds_type(self.h, int(bt)))

    def set_dt_control(self, on=True):
        self._check(self.lib.rmh_set_dt_control(self.h, 1 if on else 0))

    def dt_estimate_reset(self):
        self._check(self.lib.rmh_dt_estimate_reset(self.h))

    def dt_estimate_update(self, x, dx, x_min, x_max):
        self._check(self.lib.rmh_dt_estimate_update(self.h, _ptr(x), _ptr(dx), _ptr(x_min), _ptr(x_max)))

    def dt_estimate_get(self):
        v = C.c_double()
        self._check(self.lib.rmh_dt_estimate_get(self.h, C.byref(v)))
        return v.value

    # -- neighbour exchange inside the library (include/rmh.h) ---------------------------------------------------
    def exchange_setup(self, peers, compact=True):
        """peers: [(rank, send_elems, recv_slots)] as the case builder lists them (recv slots contiguous)."""
        import numpy as np

        n = len(peers)
        ranks = (C.c_int * max(n, 1))(*[int(r) for r, _, _ in peers])
        scount = (C.c_int * max(n, 1))(*[len(s) for _, s, _ in peers])
        rfirst = (C.c_int * max(n, 1))(*[int(r[0]) if len(r) else 0 for _, _, r in peers])
        rcount = (C.c_int * max(n, 1))(*[len(r) for _, _, r in peers])
        for _, _, r in peers:
            assert len(r) == 0 or (np.asarray(r) == np.arange(int(r[0]), int(r[0]) + len(r))).all(), "ghost slots of a peer must be contiguous"
        arrs = [np.ascontiguousarray(s, dtype=np.int32) for _, s, _ in peers]
        sptr = (C.POINTER(C.c_int) * max(n, 1))(*[a.ctypes.data_as(C.POINTER(C.c_int)) for a in arrs])
        desc = RmhExchangeDesc(n, ranks, scount, sptr, rfirst, rcount)
        self._check(self.lib.rmh_exchange_setup(self.h, C.byref(desc), 1 if compact else 0))
        self.n_peers = n

    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self._check(self.lib.rmh_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid, nranks, rank):
        self._check(self.lib.rmh_comm_init(self.h, uid, int(nranks), int(rank)))

    def comm_connect_local(self, k, other, other_k):
        self._check(self.lib.rmh_comm_connect_local(self.h, int(k), other.h, int(other_k)))

    def exchange_begin(self, u):
        self._check(self.lib.rmh_exchange_begin(self.h, _ptr(u)))

    def exchange_end(self):
        self._check(self.lib.rmh_exchange_end(self.h))

    def exchange_minmax(self, xe_min, xe_max):
        """ghost extrema := the neighbours' values of the given element extrema (RCCL or self / same-process peers)"""
        self._check(self.lib.rmh_exchange_minmax_begin(self.h, _ptr(xe_min), _ptr(xe_max)))
        self._check(self.lib.rmh_exchange_minmax_end(self.h))

    def exchange_buffers(self):
        """(send_ptr, send_doubles, ghost_ptr, ghost_doubles) of the library-owned buffers"""
        sp, gp, sn, gn = C.c_void_p(), C.c_void_p(), C.c_longlong(), C.c_longlong()
        self._check(self.lib.rmh_exchange_buffers(self.h, C.byref(sp), C.byref(sn), C.byref(gp), C.byref(gn)))
        return sp.value, sn.value, gp.value, gn.value

    def exchange_peer(self, k):
        """(rank, send_offset, send_doubles, recv_offset, recv_doubles) of neighbour k"""
        r = C.c_int()
        v = [C.c_longlong() for _ in range(4)]
        self._check(self.lib.rmh_exchange_peer(self.h, int(k), C.byref(r), *[C.byref(x) for x in v]))
        return (r.value,) + tuple(x.value for x in v)

    def comm_count(self):
        """ranks of the context's RCCL communicator as the communicator reports them (0: none)"""
        n = C.c_int(-1)
        self._check(self.lib.rmh_comm_count(self.h, C.byref(n)))
        return n.value

    def allreduce(self, vals, op="sum"):
        arr = (C.c_double * len(vals))(*[float(v) for v in vals])
        self._check(self.lib.rmh_allreduce(self.h, arr, len(vals), {"sum": 0, "min": 1, "max": 2}[op]))
        return list(arr)

    # -- product-field remap (-ps) ---------------------------------------------------------------------------------
    def product_ratio(self, us, u, s, active_el, active_dofs):
        self._check(self.lib.rmh_product_ratio(self.h, _ptr(us), _ptr(u), _ptr(s), _ptr(active_el), _ptr(active_dofs)))

    def elem_minmax_masked(self, u, active_el, active_dofs, xe_min, xe_max):
        self._check(self.lib.rmh_elem_minmax_masked(self.h, _ptr(u), _ptr(active_el), _ptr(active_dofs), _ptr(xe_min), _ptr(xe_max)))

    def fct_product(self, us, m, d_us_ho, s_min, s_max, u_new, active_el, active_dofs, dt, d_us):
        self._check(self.lib.rmh_fct_product(self.h, _ptr(us), _ptr(m), _ptr(d_us_ho), _ptr(s_min), _ptr(s_max), _ptr(u_new),
                                             _ptr(active_el), _ptr(active_dofs), float(dt), _ptr(d_us)))

    def check_violation(self, u, u_min, u_max, dt=0.0, du=None, tol=1e-12, bound_scale=None, active_dofs=None):
        """check_violation of remhos.cpp:1557-1594 (-vb) on u + dt * du (du None: on u itself); returns the verdict as a dict"""
        v = RmhViolation()
        self._check(self.lib.rmh_check_violation(self.h, _ptr(u), float(dt), _ptr(du), _ptr(u_min), _ptr(u_max), _ptr(bound_scale),
                                                 float(tol), _ptr(active_dofs), C.byref(v)))
        return {k: getattr(v, k) for k, _ in RmhViolation._fields_}

    def set_mass_tol(self, rel_tol, abs_tol=0.0, max_iter=100):
        self._check(self.lib.rmh_set_mass_tol(self.h, float(rel_tol), float(abs_tol), int(max_iter)))

    def set_mass_completion(self, jacobi_step=True, constant_mode=True):
        self._check(self.lib.rmh_set_mass_completion(self.h, int(bool(jacobi_step)), int(bool(constant_mode))))
