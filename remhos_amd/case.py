"""ctypes view of the host-side case builder (include/rmh_driver.h): mesh lattice, box
partition, halo lists, initial condition -- all computed by the C++ host code
(remhos_amd/csrc/rmh_host.cpp); this file only exposes the arrays as numpy views."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "librmh_host.so")

DRIVER_SYMBOLS = [
    "rmhd_case_create", "rmhd_case_destroy", "rmhd_last_error", "rmhd_case_get_info", "rmhd_case_x0",
    "rmhd_case_vel", "rmhd_case_u0", "rmhd_case_s0", "rmhd_case_subcell_vel", "rmhd_case_face_nbr", "rmhd_case_stencil27",
    "rmhd_case_owned_gid", "rmhd_case_ghost_gid", "rmhd_case_peer", "rmhd_case_save", "rmhd_run", "rmhd_run_state", "rmhd_run_rank", "rmhd_run_partitioned", "rmhd_id_file_exchange", "rmhd_axpby",
]


class RmhdConfig(C.Structure):
    _fields_ = [
        ("mesh", C.c_char * 32), ("rs", C.c_int), ("order", C.c_int), ("problem", C.c_int),
        ("dt", C.c_double), ("t_final", C.c_double), ("max_steps", C.c_int), ("lo_type", C.c_int),
        ("fused", C.c_int), ("px", C.c_int), ("py", C.c_int), ("pz", C.c_int), ("rank", C.c_int),
        ("bounds_type", C.c_int), ("dt_control", C.c_int), ("ho_type", C.c_int), ("save", C.c_int),
        ("rs_extra", C.c_int * 3), ("pa", C.c_int), ("self_wrap", C.c_int), ("warmup_steps", C.c_int),
        ("ps", C.c_int), ("ode_solver", C.c_int), ("tile_rows", C.c_int), ("verify_bounds", C.c_int),
    ]


class RmhdCaseInfo(C.Structure):
    _fields_ = [
        ("order", C.c_int), ("exec_mode", C.c_int), ("ndof", C.c_int), ("ne_owned", C.c_int),
        ("ne_ghost", C.c_int), ("n_peers", C.c_int), ("ne_halo", C.c_int), ("dim", C.c_int), ("ne_global", C.c_longlong),
        ("n", C.c_int * 3), ("lo", C.c_int * 3), ("nl", C.c_int * 3), ("dt", C.c_double),
        ("bb_min", C.c_double * 3), ("bb_max", C.c_double * 3),
    ]


class RmhdResult(C.Structure):
    _fields_ = [
        ("final_mass", C.c_double), ("max_value", C.c_double), ("mass0", C.c_double), ("mass_loss", C.c_double),
        ("dt", C.c_double), ("t_end", C.c_double), ("steps", C.c_int), ("stages", C.c_int),
        ("global_dofs", C.c_longlong),
        ("t_rhs", C.c_double), ("t_inv", C.c_double), ("t_lo", C.c_double), ("t_fct", C.c_double), ("t_total", C.c_double),
        ("fom_rhs", C.c_double), ("fom_inv", C.c_double), ("fom_lo", C.c_double), ("fom_fct", C.c_double), ("fom", C.c_double),
        ("wall", C.c_double), ("fom_wall", C.c_double), ("cg_iters_max", C.c_int), ("repeats", C.c_int),
        ("timed_stages", C.c_int), ("n_peers", C.c_int), ("transport", C.c_int), ("comm_ranks", C.c_int),
        ("send_bytes_per_stage", C.c_longlong), ("recv_bytes_per_stage", C.c_longlong),
        ("final_mass_us", C.c_double), ("mass0_us", C.c_double), ("mass_loss_us", C.c_double), ("s_max", C.c_double),
        ("has_errors", C.c_int), ("pad2_", C.c_int), ("err_l1", C.c_double), ("err_l2", C.c_double), ("err_linf", C.c_double),
        ("timer_every", C.c_int), ("timer_steps", C.c_int),
    ]


def make_config(mesh="periodic-cube", rs=1, order=3, problem=10, dt=-1.0, t_final=0.5, max_steps=-1, lo_type=5,
                fused=1, part=(1, 1, 1), rank=0, bounds_type=0, dt_control=0, ho_type=3, save=0,
                rs_extra=(0, 0, 0), pa=0, self_wrap=0, warmup_steps=0, ps=0, ode_solver=3, tile_rows=0, verify_bounds=0) -> RmhdConfig:
    c = RmhdConfig()
    c.mesh = mesh.encode()
    c.rs, c.order, c.problem = rs, order, problem
    c.dt, c.t_final, c.max_steps, c.lo_type, c.fused = dt, t_final, max_steps, lo_type, fused
    c.px, c.py, c.pz = part
    c.rank = rank
    c.bounds_type, c.dt_control, c.ho_type, c.save = bounds_type, dt_control, ho_type, save
    c.rs_extra[0], c.rs_extra[1], c.rs_extra[2] = (int(k) for k in rs_extra)
    c.pa = int(pa)
    c.self_wrap = int(self_wrap)
    c.warmup_steps = int(warmup_steps)
    c.ps, c.ode_solver = int(ps), int(ode_solver)
    c.tile_rows = int(tile_rows)
    c.verify_bounds = int(verify_bounds)
    return c


def bind_driver(lib: C.CDLL) -> C.CDLL:
    p = C.c_void_p
    lib.rmhd_case_create.argtypes = [C.POINTER(RmhdConfig)]
    lib.rmhd_case_create.restype = p
    lib.rmhd_case_destroy.argtypes = [p]
    lib.rmhd_case_destroy.restype = None
    lib.rmhd_last_error.restype = C.c_char_p
    lib.rmhd_case_get_info.argtypes = [p, C.POINTER(RmhdCaseInfo)]
    lib.rmhd_case_save.argtypes = [p, C.c_double, p, C.c_char_p, C.c_char_p]
    if hasattr(lib, "rmhd_axpby"):  # (device library only: librmh_host.so has no kernels)
        lib.rmhd_axpby.argtypes = [C.c_double, p, C.c_double, p, p, C.c_longlong, p]
        lib.rmhd_axpby.restype = C.c_int
    for name in ("x0", "vel", "u0", "s0", "subcell_vel", "face_nbr", "stencil27", "owned_gid", "ghost_gid"):
        f = getattr(lib, "rmhd_case_" + name)
        f.argtypes = [p]
        f.restype = p
    lib.rmhd_case_peer.argtypes = [p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(p),
                                   C.POINTER(C.c_int), C.POINTER(p)]
    if hasattr(lib, "rmhd_run"):
        lib.rmhd_run.argtypes = [C.POINTER(RmhdConfig), C.POINTER(RmhdResult)]
        lib.rmhd_run_state.argtypes = [C.POINTER(RmhdConfig), C.POINTER(RmhdResult), p, p]
        lib.rmhd_run_rank.argtypes = [C.POINTER(RmhdConfig), C.c_char_p, C.c_int, C.POINTER(RmhdResult), p, p]
        lib.rmhd_run_partitioned.argtypes = [C.POINTER(RmhdConfig), C.c_char_p, C.c_int, C.POINTER(RmhdResult)]
        if hasattr(lib, "rmhd_id_file_exchange"):  # (older development builds of the library lack it: tools/kbench.py A/B runs)
            lib.rmhd_id_file_exchange.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    return lib


def load_host_library(path: str | None = None) -> C.CDLL:
    path = path or HOST_LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python __graft_entry__.py` to build it")
    return bind_driver(C.CDLL(path))


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    if not ptr or n == 0:
        return np.zeros(shape, dtype=dtype)
    ct = {np.float64: C.c_double, np.int32: C.c_int, np.int64: C.c_longlong}[dtype]
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))
    return arr.reshape(shape).copy()


class Case:
    """Host arrays of one rank's block of a Remhos case."""

    def __init__(self, lib: C.CDLL, cfg: RmhdConfig):
        h = lib.rmhd_case_create(C.byref(cfg))
        if not h:
            raise RuntimeError("rmhd_case_create: " + lib.rmhd_last_error().decode())
        info = RmhdCaseInfo()
        lib.rmhd_case_get_info(h, C.byref(info))
        self.cfg = cfg
        self.order, self.exec_mode, self.ndof = info.order, info.exec_mode, info.ndof
        self.ne_owned, self.ne_ghost, self.ne_global = info.ne_owned, info.ne_ghost, info.ne_global
        self.ne_halo = info.ne_halo
        self.n, self.lo, self.nl = list(info.n), list(info.lo), list(info.nl)
        self.dt = info.dt
        self.bb_min, self.bb_max = list(info.bb_min), list(info.bb_max)
        ne, nd = self.ne_owned, self.ndof
        dim = self.dim = info.dim  # 3, or 2 for the quadrilateral lattices (inline-quad, periodic-square)
        self.x0 = _view(lib.rmhd_case_x0(h), (ne, dim, 3**dim), np.float64)
        self.vel = _view(lib.rmhd_case_vel(h), (ne, dim, 3**dim), np.float64)
        self.u0 = _view(lib.rmhd_case_u0(h), (ne, nd), np.float64)
        self.s0 = _view(lib.rmhd_case_s0(h), (ne, nd), np.float64)
        sv = lib.rmhd_case_subcell_vel(h)
        self.subcell_vel = _view(sv, (ne, dim, nd), np.float64) if sv else None
        self.face_nbr = _view(lib.rmhd_case_face_nbr(h), (ne, 2 * dim), np.int32)
        self.stencil27 = _view(lib.rmhd_case_stencil27(h), (ne, 3**dim), np.int32)
        self.owned_gid = _view(lib.rmhd_case_owned_gid(h), (ne,), np.int64)
        self.ghost_gid = _view(lib.rmhd_case_ghost_gid(h), (self.ne_ghost,), np.int64)
        self.peers = []
        for k in range(info.n_peers):
            rank, ns, nr = C.c_int(), C.c_int(), C.c_int()
            ps, pr = C.c_void_p(), C.c_void_p()
            lib.rmhd_case_peer(h, k, C.byref(rank), C.byref(ns), C.byref(ps), C.byref(nr), C.byref(pr))
            self.peers.append((rank.value, _view(ps.value, (ns.value,), np.int32), _view(pr.value, (nr.value,), np.int32)))
        lib.rmhd_case_destroy(h)
        self._lib = lib

    def save(self, t, u, mesh_path, gf_path=None):
        """-save (remhos.cpp:1015-1030, 1365-1380): MFEM mesh at pseudo-time t (+ GridFunction of the host array u)."""
        lib = self._lib
        h = lib.rmhd_case_create(C.byref(self.cfg))
        try:
            uu = None if u is None else np.ascontiguousarray(u, dtype=np.float64)
            rc = lib.rmhd_case_save(h, float(t), None if uu is None else uu.ctypes.data, str(mesh_path).encode(),
                                    None if gf_path is None else str(gf_path).encode())
            if rc != 0:
                raise RuntimeError("rmhd_case_save: " + lib.rmhd_last_error().decode())
        finally:
            lib.rmhd_case_destroy(h)
