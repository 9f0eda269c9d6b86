"""RK3-SSP stage loop of one rank through the C ABI, with the halo exchange of multi-rank runs.

Everything numeric is a HIP kernel behind include/rmh.h; torch supplies device buffers, the
stream and torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  Stage sequence = AdvectionOperator::Mult (remhos.cpp:1596-1916) inside
RK3SSPSolver::Step [MFEM]; per stage ONE neighbour exchange (rmh_exchange_begin / rmh_exchange_end: plan, pack
kernels, ghost records and the RCCL transport are inside the library) carries, for every element a neighbour rank
needs, its face layer -- or all its values -- of u (ParGridFunction::ExchangeFaceNbrData, remhos_ho.cpp:122) plus
its min/max (the GroupCommunicator min/max of remhos_tools.cpp:461-466).
"""
from __future__ import annotations

import torch

from .capi import Context


class Stepper:
    def __init__(self, lib, case, device="cuda:0", dist=None, fused=True, one_kernel=True, overlap=True):
        self.case = case
        self.lib = lib
        if hasattr(lib, "rmhd_axpby") and not lib.rmhd_axpby.argtypes:  # (a library handle that did not go through case.bind_driver)
            import ctypes as _C
            lib.rmhd_axpby.argtypes = [_C.c_double, _C.c_void_p, _C.c_double, _C.c_void_p, _C.c_void_p, _C.c_longlong, _C.c_void_p]
            lib.rmhd_axpby.restype = _C.c_int
        self.overlap = overlap  # one-kernel stage: interior elements overlap the halo exchange
        self.dev = torch.device(device)
        # validation aid (tools/two_ranks_one_gpu.py): a backend that is not stream-aware needs the pack kernel finished
        # before it reads the send buffer, and the kernels that read the previous ghosts finished before it overwrites them
        import os as _os
        self.sync_exchange = _os.environ.get("RMH_SYNC_EXCHANGE", "0") == "1" and self.dev.type == "cuda"
        self.dist = dist if (dist is not None and case.peers) else None
        self.defer_exchange = False
        self.lo = int(case.cfg.lo_type)
        self.fused_lo4 = fused and self.lo in (3, 4)  # lo 3 / 4: HO kernel + RD kernel + fused limiter/RK kernel
        self.fused = fused and self.lo == 5
        fused = self.fused
        self.one_kernel = one_kernel and (self.fused or self.fused_lo4)  # whole stage in one kernel (rmh_stage_fused)
        dev_index = self.dev.index or 0
        self.ctx = Context(lib, order=case.order, exec_mode=case.exec_mode, x0=case.x0, vel=case.vel,
                           face_nbr=case.face_nbr, stencil27=case.stencil27, ne_ghost=case.ne_ghost,
                           subcell_vel=case.subcell_vel, device=dev_index if self.dev.type == "cuda" else 0)
        if self.dev.type == "cuda":
            self.ctx.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        if self.lo in (3, 4):
            self.ctx.set_lo_type(self.lo)
        if getattr(case.cfg, "ho_type", 3) == 2:
            self.ctx.set_mass_tol(1e-12, 0.0, 500)  # CGHOSolver (-ho 2, remhos_ho.cpp:60-63): see solvers.hpp
        elif getattr(case.cfg, "pa", 0):
            # LocalInverseHOSolver on a partially assembled M: DGMassInverse's stopping rule (remhos_ho.cpp:79-80),
            # completed by one Jacobi step and the constant mode (include/remhos_amd/solvers.hpp)
            self.ctx.set_mass_tol(0.0, 1e-8, 100)
            self.ctx.set_mass_completion(True, True)
        self.dtc = bool(getattr(case.cfg, "dt_control", 0))
        if getattr(case.cfg, "bounds_type", 0):
            self.ctx.set_bounds_type(case.cfg.bounds_type)
        if self.dtc:
            self.ctx.set_dt_control(True)
        self.repeats = 0
        ne, nd, ng = case.ne_owned, case.ndof, case.ne_ghost
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.x = torch.from_numpy(case.u0).to(self.dev).contiguous()
        self.y = torch.empty_like(self.x)
        self.k = torch.empty_like(self.x)
        self.y2 = torch.empty_like(self.x) if self.one_kernel else None
        if self.fused_lo4:
            self.du_lo = torch.empty_like(self.x)
        elif not fused:
            self.du_lo = torch.empty_like(self.x)
            self.umin = torch.empty_like(self.x)
            self.umax = torch.empty_like(self.x)
            self.xe_min = torch.empty(ne, **f64)
            self.xe_max = torch.empty(ne, **f64)
            self.m = torch.empty_like(self.x)
            self.du = torch.empty_like(self.x)
        self.t = 0.0
        self.dt = case.dt
        self.tok_x = 0  # token of the element extrema of self.x (rmh_stage_fused_chain); 0: none
        # Neighbour exchange: plan, pack kernels, ghost records and (with RCCL) the transport live in the library
        # (rmh_exchange_*, include/rmh.h).  Transports: "rccl" = grouped ncclSend/ncclRecv inside the library on a
        # communicator made from a unique id broadcast over the process group; "external" = this class moves the
        # library's send segments into the neighbours' ghost segments with torch.distributed (gloo in the CPU tests,
        # or the fallback when the in-library communicator cannot be created); "local" = the peers are contexts of
        # this process (connect_local_peers).
        self.transport = None
        # compact records (face layer + extrema instead of whole elements) need every block at least two elements
        # thick in the partitioned directions; decided from the partition alone, so that all ranks agree
        part = (case.cfg.px, case.cfg.py, case.cfg.pz)
        thin = any(part[d] > 1 and case.n[d] // part[d] < 2 for d in range(3))
        self.compact = _os.environ.get("RMH_COMPACT", "1") != "0" and not thin
        self.ops = []
        if case.peers:
            self.ctx.exchange_setup(case.peers, compact=self.compact)
            if self.dist is not None:
                self._connect(self.dist)
            elif getattr(case.cfg, "self_wrap", 0):
                # self-loop (rmhd_config.self_wrap): the only neighbour is this rank itself.  On the GPU the exchange
                # goes through a one-rank RCCL communicator -- grouped ncclSend / ncclRecv to the own rank, the real
                # transport of multi-GPU runs -- unless RMH_EXCHANGE=local asks for device copies (host emulation: always)
                if self.dev.type == "cuda" and _os.environ.get("RMH_EXCHANGE", "rccl") == "rccl":
                    self.ctx.comm_init(self.ctx.comm_unique_id(), 1, 0)
                    self.transport = "rccl"
                else:
                    self.ctx.comm_connect_local(0, self.ctx, 0)
                    self.transport = "local"

    def _buffer_tensor(self, ptr, n):
        """torch view of n doubles of a library-owned buffer (no copy)"""
        if self.dev.type == "cuda":
            class _Cai:
                __cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}

            return torch.as_tensor(_Cai(), device=self.dev)
        import ctypes as C

        import numpy as np

        return torch.from_numpy(np.ctypeslib.as_array((C.c_double * n).from_address(ptr)))

    def _connect(self, d):
        import os as _os

        want_rccl = d.get_backend() == "nccl" and _os.environ.get("RMH_EXCHANGE", "rccl") == "rccl"
        ok = 0
        if want_rccl:
            # Every collective below is entered by every rank whatever happened locally (a rank that skipped one while
            # the others wait in it would hang the job).  1. can this rank reach RCCL at all?  rmh_comm_unique_id goes
            # through the same dlopen / symbol table as rmh_comm_init; the id of ranks other than 0 is discarded.
            my_id = None
            try:
                my_id = self.ctx.comm_unique_id()
            except Exception as e:  # noqa: BLE001
                self.rccl_error = str(e)
            flag = torch.tensor([1 if my_id is not None else 0], device=self.dev)
            d.all_reduce(flag, op=d.ReduceOp.MIN)  # (backend nccl here: device tensor)
            if int(flag[0]):
                # 2. all ranks can: rank 0's id to everybody, then the communicator (ncclCommInitRank is itself collective)
                box = [my_id if d.get_rank() == 0 else None]
                d.broadcast_object_list(box, src=0)
                try:
                    self.ctx.comm_init(box[0], d.get_world_size(), d.get_rank())
                    ok = 1
                except Exception as e:  # noqa: BLE001
                    self.rccl_error = str(e)
                flag = torch.tensor([ok], device=self.dev)
                d.all_reduce(flag, op=d.ReduceOp.MIN)
                ok = int(flag[0])
        if ok:
            self.transport = "rccl"
            return
        self.transport = "external"
        sp, sn, gp, gn = self.ctx.exchange_buffers()
        send = self._buffer_tensor(sp, max(sn, 1))
        ghost = self._buffer_tensor(gp, max(gn, 1))
        for k in range(len(self.case.peers)):
            rank, so, scount, ro, rcount = self.ctx.exchange_peer(k)
            if scount:
                self.ops.append(d.P2POp(d.isend, send[so:so + scount], rank))
            if rcount:
                self.ops.append(d.P2POp(d.irecv, ghost[ro:ro + rcount], rank))

    def connect_local_peers(self, steppers):
        """All blocks in this process: steppers[r] is the Stepper of rank r (device copies instead of RCCL)."""
        for k, (rank, _, _) in enumerate(self.case.peers):
            other = steppers[rank]
            ok = [j for j, (r2, _, _) in enumerate(other.case.peers) if r2 == self.case.cfg.rank]
            self.ctx.comm_connect_local(k, other.ctx, ok[0])
        self.transport = "local"

    # -- halo exchange: neighbour all-to-all of ghost records ----------------------------------------
    def exchange_begin(self, u):
        """Pack and post the exchange; returns work handles of the external transport."""
        if not self.case.peers:
            return []
        if self.transport is None:
            raise RuntimeError("this rank has neighbour ranks but no transport (torch.distributed group or local peers)")
        self.ctx.exchange_begin(u)
        if self.transport != "external":
            return []
        if self.sync_exchange:
            torch.cuda.synchronize(self.dev)  # backends that are not stream-aware (gloo on device tensors) read the send buffer now
        return self.dist.batch_isend_irecv(self.ops)

    def exchange_end(self, works):
        for w in works:
            w.wait()
        if self.case.peers:
            self.ctx.exchange_end()

    # -- one RK stage: out = a*x + b*(u + dt*F(u, t)) ----------------------------------------------
    def stage(self, u, t, dt, x_base, a, b, out, token=0):
        """token: what the stage that wrote u returned, if u is still that output (one-kernel stage: saves the pass that
        recomputes the element extrema of u, rmh_stage_fused_chain); returns the token of `out` (0 for the other paths)"""
        c = self.ctx
        if self.defer_exchange:
            works = []  # (lockstep driver of several same-process blocks: it has begun and will end the exchange itself)
        else:
            works = self.exchange_begin(u)
        c.setup(t)
        if self.one_kernel and self.case.peers and self.overlap and not self.defer_exchange:
            # elements that reach no ghost run while the exchange is in flight (the RCCL send/recv
            # kernels live on the library's exchange stream), the halo-dependent shell after it
            nh, ne = self.case.ne_halo, self.case.ne_owned
            c.stage_fused_range(u, dt, out, nh, ne, False, x_base=x_base, a=a, b=b, dt_rk=dt, token=token)
            self.exchange_end(works)
            return c.stage_fused_range(u, dt, out, 0, nh, True, x_base=x_base, a=a, b=b, dt_rk=dt, token=token)
        if not self.defer_exchange:
            self.exchange_end(works)
        if self.one_kernel:
            return c.stage_fused(u, dt, out, x_base=x_base, a=a, b=b, dt_rk=dt, token=token)
        c.ho_apply(u, self.k)
        if self.fused_lo4:
            (c.lo_rd if self.lo == 3 else c.lo_rdsubcell)(u, self.du_lo)
            c.limit_fused_lo(u, self.k, self.du_lo, dt, du=None, x_base=x_base, a=a, b=b, dt_rk=dt, y_out=out)
            return 0
        if self.fused:
            c.limit_fused(u, self.k, dt, du=None, x_base=x_base, a=a, b=b, dt_rk=dt, y_out=out)
            return 0
        # the reference's call sequence (remhos.cpp:1815-1831)
        m = c.lumped_mass_ptr()  # refreshed by rmh_ho_apply at this stage's mesh position (remhos.cpp:1632)
        if self.lo == 4:
            c.lo_rdsubcell(u, self.du_lo)
        elif self.lo == 3:
            c.lo_rd(u, self.du_lo)
        else:
            c.lo_massavg(u, self.k, dt, self.du_lo)
        c.elem_minmax(u, self.xe_min, self.xe_max)
        c.bounds(self.xe_min, self.xe_max, self.umin, self.umax)
        du = self.du
        c.fct_clipscale(u, m, self.k, self.du_lo, self.umin, self.umax, dt, du)
        c.dt_estimate_update(u, self.du_lo, self.umin, self.umax)  # remhos.cpp:1839-1842 (no-op with a fixed dt)
        # the RK vector update is the caller's (MFEM's RK3SSPSolver: add / Add on Vectors): out = a x_base + b (u + dt du),
        # in place in du -- no temporaries
        if x_base is None and b == 1.0:
            torch.add(u, du, alpha=dt, out=out)
        else:
            # RK3SSPSolver::Step: y.Add(dt, k); add(a, x, b, y, z) -- two vector operations, as in the reference's integrator
            torch.add(u, du, alpha=dt, out=du)
            if x_base is not None and hasattr(self.lib, "rmhd_axpby"):
                stream = torch.cuda.current_stream(self.dev).cuda_stream if self.dev.type == "cuda" else None
                if self.lib.rmhd_axpby(a, x_base.data_ptr(), b, du.data_ptr(), out.data_ptr(), du.numel(), stream) != 0:
                    raise RuntimeError("rmhd_axpby: " + self.lib.rmhd_last_error().decode())
            else:
                du.mul_(b)
                if x_base is None:
                    out.copy_(du)
                else:
                    torch.add(du, x_base, alpha=a, out=out)
        return 0

    def step(self, dt):
        """RK3SSPSolver::Step: stage times t, t+dt, t+dt/2."""
        x, y, t = self.x, self.y, self.t
        if self.one_kernel:
            # the one-kernel stage reads neighbour traces of its input while other workgroups already
            # write the output: input and output must be different vectors (x_base may be the output)
            # (tok_x: the token of self.x, valid as long as nothing but this chain writes it -- run() drops it when it
            # restores a saved state)
            tok = self.stage(x, t, dt, None, 0.0, 1.0, y, token=self.tok_x)
            tok = self.stage(y, t + dt, dt, x, 0.75, 0.25, self.y2, token=tok)
            self.tok_x = self.stage(self.y2, t + dt / 2, dt, x, 1.0 / 3.0, 2.0 / 3.0, x, token=tok)
            self.t = t + dt
            return
        self.stage(x, t, dt, None, 0.0, 1.0, y)
        self.stage(y, t + dt, dt, x, 0.75, 0.25, y)
        self.stage(y, t + dt / 2, dt, x, 1.0 / 3.0, 2.0 / 3.0, x)
        self.t = t + dt

    def run(self, max_steps=-1, t_final=None):
        """time loop of remhos.cpp:1146-1296"""
        if t_final is None:
            t_final = 1.0 if self.case.exec_mode == 1 else self.case.cfg.t_final
        ti, ti_total, done = 0, 0, False  # accepted steps / all steps incl. repeated ones (remhos.cpp:1142)
        while not done:
            dt_real = min(self.dt, t_final - self.t)
            if self.dtc:
                # AdvectionOperator::SetDt + ResetTimeStepRatio (remhos.cpp:1150-1152); Sold = S (:1171)
                self.ctx.dt_estimate_reset()
                if not hasattr(self, "x_old"):
                    self.x_old = torch.empty_like(self.x)
                self.x_old.copy_(self.x)
                t_old = self.t
            self.step(dt_real)
            ti += 1
            ti_total += 1
            if self.dtc:
                est = self.ctx.dt_estimate_get()
                if self.dist is not None:
                    red = torch.tensor([est], dtype=torch.float64, device=self.dev)
                    self.dist.all_reduce(red, op=self.dist.ReduceOp.MIN)  # MPI_Allreduce MIN, remhos.cpp:1993
                    est = float(red[0])
                ratio = est / dt_real if dt_real != 0.0 else 0.0
                if ratio < 1.0:
                    # repeat with the proper time step (remhos.cpp:1181-1193)
                    ti -= 1
                    self.t = t_old
                    self.x.copy_(self.x_old)
                    self.tok_x = 0  # (x is no longer the output of the last stage: its extrema are recomputed)
                    self.dt = 0.85 * self.dt
                    self.repeats += 1
                    if self.dt < 1e-12:
                        raise RuntimeError("The time step crashed!")
                    continue
                if ratio > 1.25:
                    self.dt *= 1.02
            done = self.t >= t_final - 1e-8 * self.dt
            if ti_total == max_steps:  # -ms counts repeated steps too (remhos.cpp:1296)
                done = True
        return ti

    def local_mass_and_max(self, t=None):
        """sum_i m_i u_i with the lumped mass at pseudo-time t, and max u (remhos.cpp:1382-1414)."""
        m = torch.empty_like(self.x)
        tt = self.t if t is None else t
        self.ctx.compute_lumped_mass(tt if self.case.exec_mode == 1 else 0.0, m)
        return float((m * self.x).sum()), float(self.x.max())

    def close(self):
        self.ctx.close()


def lockstep_step(steppers, dt):
    """One RK3-SSP step of ALL blocks of a partition held by this process (Stepper.connect_local_peers): per stage
    every block posts its exchange, every block completes it, then every block runs its stage."""
    def stage(get_u, t_off, get_base, a, b, get_out):
        for s in steppers:
            s.exchange_begin(get_u(s))
        for s in steppers:
            s.exchange_end([])
        for s in steppers:
            s.defer_exchange = True
            try:
                s.tok_x = s.stage(get_u(s), s.t + t_off, dt, get_base(s), a, b, get_out(s), token=s.tok_x)
            finally:
                s.defer_exchange = False

    if steppers[0].one_kernel:
        stage(lambda s: s.x, 0.0, lambda s: None, 0.0, 1.0, lambda s: s.y)
        stage(lambda s: s.y, dt, lambda s: s.x, 0.75, 0.25, lambda s: s.y2)
        stage(lambda s: s.y2, dt / 2, lambda s: s.x, 1.0 / 3.0, 2.0 / 3.0, lambda s: s.x)
    else:
        stage(lambda s: s.x, 0.0, lambda s: None, 0.0, 1.0, lambda s: s.y)
        stage(lambda s: s.y, dt, lambda s: s.x, 0.75, 0.25, lambda s: s.y)
        stage(lambda s: s.y, dt / 2, lambda s: s.x, 1.0 / 3.0, 2.0 / 3.0, lambda s: s.x)
    for s in steppers:
        s.t += dt
