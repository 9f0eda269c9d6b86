"""ctypes wrapper of oracle/libcpu_port.so (TEST INFRASTRUCTURE: cpu_baseline + large-mesh checker)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota (the GPU boxes show 256
    logical CPUs but run their containers with cpu.max = 16 CPUs: 128 OpenMP threads there time-share 16 CPUs' worth and
    are four times SLOWER than 16 threads -- what rounds 1 and 2 reported as the 128-thread baseline)"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            quota = float(txt[0]) if txt[0] != "max" else -1.0
            period = float(txt[1]) if len(txt) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = max(1, min(n, int(quota / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def load():
    # one thread per usable core, pinned (reported with the baseline): set before the OpenMP runtime starts
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    name = "libcpu_port.so"
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
        if "avx512f" in flags and "avx512dq" in flags and os.environ.get("RMH_CPU_PORT_AVX2", "0") != "1":
            name = "libcpu_port_v4.so"  # the element batches of the HO loop are 8 doubles wide
    except (OSError, StopIteration):
        pass
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        import subprocess

        subprocess.check_call(["make", "-C", _HERE])
    lib = C.CDLL(path)
    p, d, i = C.c_void_p, C.c_double, C.c_int
    lib.cpu_stage.argtypes = [i, i, i, p, p, p, p, d, d, p, p, p, p, p, d, d, i]
    lib.cpu_rk3_step.argtypes = [i, i, i, p, p, p, p, d, d, p, p, d, d, i]
    lib.cpu_lumped_mass.argtypes = [i, i, i, p, p, d, p]
    lib.cpu_lumped_mass.restype = None
    lib.cpu_buckets.argtypes = [C.POINTER(d * 4), i]
    lib.cpu_buckets.restype = None
    lib.cpu_set_threads.argtypes = [i]
    lib.cpu_set_threads.restype = None
    # (torch's import has usually started the OpenMP runtime already: the environment above is then too late)
    lib.cpu_set_threads(int(os.environ.get("OMP_NUM_THREADS", usable_cpus())))
    return lib


class CpuPort:
    """RK3 stepping of a case (arrays in the C-ABI layouts of include/rmh.h) on the host cores."""

    def __init__(self, order, exec_mode, x0, vel, face_nbr, stencil27, u0, rel_tol=1e-14, abs_tol=0.0, completion=False):
        """rel_tol 1e-14: converged local solve (default); (0, 1e-8, True) = the product's -pa rule: DGMassInverse's abs 1e-8
        (remhos_ho.cpp:79-80) + Jacobi step + constant mode"""
        self.lib = load()
        self.p, self.mode = order, exec_mode
        self.x0 = np.ascontiguousarray(x0, dtype=np.float64)
        self.vel = np.ascontiguousarray(vel, dtype=np.float64)
        self.nbr = np.ascontiguousarray(face_nbr, dtype=np.int32)
        self.st = np.ascontiguousarray(stencil27, dtype=np.int32)
        self.u = np.ascontiguousarray(u0, dtype=np.float64).copy()
        self.ne = self.nbr.shape[0]
        self.work = np.zeros(4 * self.u.size + 2 * self.ne)
        self.t = 0.0
        self.rel_tol, self.abs_tol, self.completion = rel_tol, abs_tol, int(bool(completion))
        self.threads = self.lib.cpu_num_threads()
        self.simd_width = self.lib.cpu_simd_width()

    def step(self, dt):
        it = self.lib.cpu_rk3_step(self.p, self.ne, self.mode, self.x0.ctypes.data, self.vel.ctypes.data,
                                   self.nbr.ctypes.data, self.st.ctypes.data, self.t, dt, self.u.ctypes.data,
                                   self.work.ctypes.data, self.rel_tol, self.abs_tol, self.completion)
        self.t += dt
        return it

    def stage(self, u, t, dt):
        n = u.size
        du, m, dh = np.zeros_like(u), np.zeros_like(u), np.zeros_like(u)
        xe = np.zeros(2 * self.ne)
        u = np.ascontiguousarray(u)
        self.lib.cpu_stage(self.p, self.ne, self.mode, self.x0.ctypes.data, self.vel.ctypes.data, self.nbr.ctypes.data,
                           self.st.ctypes.data, t, dt, u.ctypes.data, du.ctypes.data, m.ctypes.data, dh.ctypes.data,
                           xe.ctypes.data, self.rel_tol, self.abs_tol, self.completion)
        return du, m, dh

    def buckets(self, reset=False):
        """TimingData-style buckets (seconds: RHS, INV, LO, FCT) accumulated since the last reset"""
        t = (C.c_double * 4)()
        self.lib.cpu_buckets(C.byref(t), 1 if reset else 0)
        return list(t)

    def mass(self, t):
        m = np.zeros_like(self.u)
        self.lib.cpu_lumped_mass(self.p, self.ne, self.mode, self.x0.ctypes.data, self.vel.ctypes.data, t, m.ctypes.data)
        return float((m * self.u).sum())
