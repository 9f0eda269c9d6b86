"""RK3-SSP stage loop of one rank through the C ABI, with the halo exchange of multi-rank runs.

Everything numeric is a HIP kernel behind include/rmh.h; torch supplies device buffers, the
stream and torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  Stage sequence = AdvectionOperator::Mult (remhos.cpp:1596-1916) inside
RK3SSPSolver::Step [MFEM]; per stage ONE neighbour exchange carries, for every element a
neighbour rank needs, its ndof values of u (ParGridFunction::ExchangeFaceNbrData,
remhos_ho.cpp:122) plus its min/max (the GroupCommunicator min/max of remhos_tools.cpp:461-466).
"""
from __future__ import annotations

import torch

from .capi import Context


class Stepper:
    def __init__(self, lib, case, device="cuda:0", dist=None, fused=True, one_kernel=True, overlap=True):
        self.case = case
        self.overlap = overlap  # one-kernel stage: interior elements overlap the halo exchange
        self.dev = torch.device(device)
        # validation aid (tools/two_ranks_one_gpu.py): a backend that is not stream-aware needs the pack kernel finished
        # before it reads the send buffer, and the kernels that read the previous ghosts finished before it overwrites them
        import os as _os
        self.sync_exchange = _os.environ.get("RMH_SYNC_EXCHANGE", "0") == "1" and self.dev.type == "cuda"
        self.dist = dist if (dist is not None and case.peers) else None
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
        self.t = 0.0
        self.dt = case.dt
        # ghost storage and exchange plan.  A ghost record is [ndof values of u | min | max]; the ghost slots of
        # a peer are a contiguous range (ghosts are ordered by owner rank, then global id), so ONE message per
        # peer lands directly in the ghost block; one device pack kernel gathers the records of all peers' send lists
        self.ghost = torch.zeros(max(ng, 1), nd + 2, **f64)
        if ng:
            self.ctx.set_ghost_records(self.ghost)
        self.plan = []
        send_all, off = [], 0
        for rank, send, recv in case.peers:
            r0, r1 = int(recv[0]), int(recv[-1]) + 1
            assert r1 - r0 == len(recv) and (recv == range(r0, r1)).all(), "ghost slots of a peer must be contiguous"
            self.plan.append((rank, off, off + len(send), r0, r1))
            send_all.append(send)
            off += len(send)
        self.nsend = off
        self.ops = []
        if self.plan:
            import numpy as np

            self.send_elems = torch.from_numpy(np.concatenate(send_all).astype("int32")).to(self.dev)
            self.srec = torch.empty(self.nsend, nd + 2, **f64)
            if self.dist is not None:
                d = self.dist
                for rank, s0, s1, r0, r1 in self.plan:
                    self.ops += [d.P2POp(d.isend, self.srec[s0:s1], rank), d.P2POp(d.irecv, self.ghost[r0:r1], rank)]

    # -- halo exchange: neighbour all-to-all of ghost records ----------------------------------------
    def exchange(self, u):
        """Post the exchange; returns the work handles (wait on them before anything reads the ghosts)."""
        if not self.plan:
            return []
        if self.dist is None:
            raise RuntimeError("this rank has neighbour ranks but no torch.distributed group was given")
        self.ctx.halo_pack_records(u, self.send_elems, self.nsend, self.srec)
        if self.sync_exchange:
            torch.cuda.synchronize(self.dev)  # backends that are not stream-aware (gloo on device tensors) read srec now
        return self.dist.batch_isend_irecv(self.ops)

    # -- one RK stage: out = a*x + b*(u + dt*F(u, t)) ----------------------------------------------
    def stage(self, u, t, dt, x_base, a, b, out):
        c = self.ctx
        works = self.exchange(u)
        c.setup(t)
        if self.one_kernel and works and self.overlap:
            # elements that reach no ghost run while the exchange is in flight (the RCCL send/recv
            # kernels live on the process group's own stream), the halo-dependent shell after it
            nh, ne = self.case.ne_halo, self.case.ne_owned
            c.stage_fused_range(u, dt, out, nh, ne, False, x_base=x_base, a=a, b=b, dt_rk=dt)
            for w in works:
                w.wait()
            c.stage_fused_range(u, dt, out, 0, nh, True, x_base=x_base, a=a, b=b, dt_rk=dt)
            return
        for w in works:
            w.wait()
        if self.one_kernel:
            c.stage_fused(u, dt, out, x_base=x_base, a=a, b=b, dt_rk=dt)
            return
        c.ho_apply(u, self.k)
        if self.fused_lo4:
            (c.lo_rd if self.lo == 3 else c.lo_rdsubcell)(u, self.du_lo)
            c.limit_fused_lo(u, self.k, self.du_lo, dt, du=None, x_base=x_base, a=a, b=b, dt_rk=dt, y_out=out)
            return
        if self.fused:
            c.limit_fused(u, self.k, dt, du=None, x_base=x_base, a=a, b=b, dt_rk=dt, y_out=out)
            return
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
        du = torch.empty_like(u)
        c.fct_clipscale(u, m, self.k, self.du_lo, self.umin, self.umax, dt, du)
        c.dt_estimate_update(u, self.du_lo, self.umin, self.umax)  # remhos.cpp:1839-1842 (no-op with a fixed dt)
        y = u + dt * du
        if x_base is None:
            out.copy_(b * y)
        else:
            out.copy_(a * x_base + b * y)

    def step(self, dt):
        """RK3SSPSolver::Step: stage times t, t+dt, t+dt/2."""
        x, y, t = self.x, self.y, self.t
        if self.one_kernel:
            # the one-kernel stage reads neighbour traces of its input while other workgroups already
            # write the output: input and output must be different vectors (x_base may be the output)
            self.stage(x, t, dt, None, 0.0, 1.0, y)
            self.stage(y, t + dt, dt, x, 0.75, 0.25, self.y2)
            self.stage(self.y2, t + dt / 2, dt, x, 1.0 / 3.0, 2.0 / 3.0, x)
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
                    self.ctx.invalidate_extrema()
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
