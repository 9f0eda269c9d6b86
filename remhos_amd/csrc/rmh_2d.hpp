// dim = 2: the HO solver of the RK stage on quadrilateral tensor lattices (Q2 mesh nodes, Bernstein DG space of order p,
// Q = p + 2 Gauss-Legendre points per direction -- SURVEY A.2), behind the same C ABI as the 3-D path (rmh_layout.dim = 2).
// What the reference's own test table runs on a device (remhos_tests.cpp:38-107: inline-quad, -pa -ho 3 -lo 5 -fct 2; its
// CUDA entry #10 is this case) and BASELINE.json configs[0]'s mesh family.  The limiter side of a 2-D stage is the
// dimension-generic streaming kernels of rmh_stream.hpp / rmh_kernels.hpp (DIM = 2: D^2 dofs per element, the 3 x 3 element
// stencil embedded in the middle layer of the 27-entry table).
//
// Reference semantics: LocalInverseHOSolver::CalcHOSolution, PA branch (remhos_ho.cpp:119-128): K_HO = ConvectionIntegrator +
// transposed DGTraceIntegrator (remhos.cpp:646-678), DGMassInverse (remhos_ho.cpp:79-80: Jacobi-PCG in the Gauss-Legendre
// nodal basis, stopping rule (D^-1 r, r) <= max(rel^2 nom0, abs^2)), lumped mass M_HO 1 (remhos.cpp:1632), element extrema
// (remhos_tools.cpp:497-523).  Oracle: oracle/remhos_oracle.py (dim-generic), pinned in 2-D by the reference's ctest masses and
// autotest values (tests/test_oracle_kat.py).
//
// Shape: ONE element per one-wavefront workgroup -- Q^2 <= 64 quadrature points and D^2 <= 49 dofs fit the 64 lanes at every
// order p <= 6, so every role (quadrature point, face point, dof) is one lane and every barrier is wavefront-local.  A 2-D
// element is 9 + 9 node pairs and D^2 values: the contractions are done densely from LDS (D^2 or Q^2 terms per lane), the
// geometry (Jacobian, adj(J) v, det J, face normals) is recomputed from the 9 nodes each stage like in 3-D.  2-D runs are the
// reference's regression and plumbing cases (10^3 ... 10^5 dofs): the kernel is written for parity, not tuned.
#pragma once
#include "rmh_stream.hpp"

namespace rmh
{

struct Ho2Args
{
   const double *u;      // [ne][D2]
   const double *x0;     // [ne][2][9] mesh nodes, node a = ax + 3 ay
   const double *vel;    // [ne][2][9] remap: displacement; transport: nodal velocity
   const int *face_nbr;  // [ne][4], face f = 2 c + side; neighbour element or -1
   const double *tab;    // TabLayout2<P>
   double *du;           // [ne][D2]
   double *m;            // [ne][D2] lumped mass
   double *xe_min, *xe_max; // [ne] extrema of u (may be null)
   int *cg_iters;
   double t;
   int move;
   double alpha, upw;
   double rel2, abs2;
   int max_iter, jacobi_step, mass_fix;
   const double *subvel; // [ne][2][D2] sub-mesh node velocity (lo 4; rmh_layout.subcell_vel) or null
   int rd_subcell;       // MODE 2: 1 subcell residual distribution (lo 4), 0 plain residual distribution (lo 3)
};

// MODE 0: HO solution, lumped mass, element extrema; MODE 4: lumped mass only (rmh_compute_lumped_mass);
// MODE 2: the residual-distribution LO solvers (PAResidualDistribution[Subcell]::CalcLOSolution, remhos_lo.cpp:965-1034,
// 1620-1802; host form :111-245): du = du_LO, lumped mass
template <int P, int MODE>
__global__ void __launch_bounds__(64) ho2d_kernel(Ho2Args a)
{
   using T = TabLayout2<P>;
   constexpr int D = T::D, Q = T::Q, D2 = D * D, Q2 = Q * Q;
   static_assert(Q2 <= 64 && D2 <= 64 && 4 * Q <= 64, "one lane per quadrature point, dof and face point");
   __shared__ double sT[T::N2];
   __shared__ double sX[2][9], sV[2][9], sU[D2], sN[4][D], sG[Q2], sW[Q2], sF[4][Q], sP[D2], sPQ[Q2];
   constexpr int NS = P * P; // subcells
   __shared__ double sX0[2][9], sS[4][Q], sSv[2][D2], sCP[MODE == 2 ? NS : 1][4], sCN[MODE == 2 ? NS : 1][4];
   const int tid = threadIdx.x;
   const size_t e = blockIdx.x;
   for (int i = tid; i < T::N2; i += 64) { sT[i] = a.tab[i]; }
   if (tid < 18)
   {
      const double x = a.x0[e * 18 + tid], v = a.vel[e * 18 + tid];
      sV[tid / 9][tid % 9] = v;
      sX[tid / 9][tid % 9] = a.move ? x + a.t * v : x;
      sX0[tid / 9][tid % 9] = x;
   }
   if (MODE == 2 && a.rd_subcell && tid < D2)
   {
      sSv[0][tid] = a.subvel[e * 2 * D2 + tid];
      sSv[1][tid] = a.subvel[e * 2 * D2 + D2 + tid];
   }
   if (MODE == 0 || MODE == 2)
   {
      if (tid < D2) { sU[tid] = a.u[e * D2 + tid]; }
      if (tid < 4 * D)
      {
         // the neighbour sees face (c, side) as (c, 1 - side) with the same tangential orientation: its face layer
         const int f = tid / D, it = tid % D, c = f / 2, side = f % 2;
         const int nb = a.face_nbr[e * 4 + f];
         const int layer = side == 0 ? P : 0;
         const int dof = c == 0 ? layer + D * it : it + D * layer;
         sN[f][it] = nb >= 0 ? a.u[(size_t)nb * D2 + dof] : 0.0; // boundary: u_nbr := 0
      }
   }
   __syncthreads();
   const double *B = sT + T::oB, *G = sT + T::oG, *L = sT + T::oL, *dL = sT + T::odL, *W = sT + T::oW;
   const double *Bg = sT + T::oBg, *Bg2 = sT + T::oBg2, *Ci = sT + T::oCi, *BgE = sT + T::oBgE;
   // ---- quadrature points: geometry, D . grad u, w det J ------------------------------------------------
   if (tid < Q2)
   {
      const int qx = tid % Q, qy = tid / Q;
      double J[2][2] = {{0, 0}, {0, 0}}, v[2] = {0, 0};
#pragma unroll
      for (int ay = 0; ay < 3; ay++)
      {
#pragma unroll
         for (int ax = 0; ax < 3; ax++)
         {
            const double lx = L[qx * 3 + ax], ly = L[qy * 3 + ay], dx = dL[qx * 3 + ax], dy = dL[qy * 3 + ay];
#pragma unroll
            for (int c = 0; c < 2; c++)
            {
               const double xn = sX[c][ax + 3 * ay];
               J[c][0] += dx * ly * xn; // d x_c / d xi
               J[c][1] += lx * dy * xn; // d x_c / d eta
               v[c] += lx * ly * sV[c][ax + 3 * ay];
            }
         }
      }
      const double w = W[qx] * W[qy];
      const double detJ = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      sW[tid] = w * detJ;
      if (MODE == 0 || MODE == 2)
      {
         // D = alpha w adj(J) v (remhos_lo.cpp:1113-1132), adj(J) = [[J11, -J01], [-J10, J00]]
         const double d0 = a.alpha * w * (J[1][1] * v[0] - J[0][1] * v[1]);
         const double d1 = a.alpha * w * (-J[1][0] * v[0] + J[0][0] * v[1]);
         double ux = 0, uy = 0;
#pragma unroll
         for (int iy = 0; iy < D; iy++)
         {
#pragma unroll
            for (int ix = 0; ix < D; ix++)
            {
               const double uu = sU[ix + D * iy];
               ux += G[qx * D + ix] * B[qy * D + iy] * uu;
               uy += B[qx * D + ix] * G[qy * D + iy] * uu;
            }
         }
         sG[tid] = d0 * ux + d1 * uy;
      }
   }
   // ---- face points: upwind speed w max(0, +-v.n_out) and the jump of the traces ---------------------------
   if ((MODE == 0 || MODE == 2) && tid < 4 * Q)
   {
      const int f = tid / Q, q1 = tid % Q, c = f / 2, side = f % 2;
      // tangent dX/dxi_t at (xi_c = side, xi_t = x_q1): nodes of the face are ac = 2 side, at = 0..2
      double Tg[2] = {0, 0}, v[2] = {0, 0};
#pragma unroll
      for (int at = 0; at < 3; at++)
      {
         const int node = c == 0 ? 2 * side + 3 * at : at + 3 * (2 * side);
#pragma unroll
         for (int k = 0; k < 2; k++)
         {
            Tg[k] += dL[q1 * 3 + at] * sX[k][node];
            v[k] += L[q1 * 3 + at] * sV[k][node];
         }
      }
      // area-weighted outward normal: the tangent turned by -90 degrees (xi = 1, eta = 0) or +90 degrees (xi = 0, eta = 1)
      const double sg = ((c == 0) == (side == 1)) ? 1.0 : -1.0;
      const double un = sg * (v[0] * Tg[1] - v[1] * Tg[0]);
      const double sp = W[q1] * fmax(0.0, a.upw * un);
      const int own_layer = side == 0 ? 0 : P;
      double uo = 0, un_ = 0;
#pragma unroll
      for (int it = 0; it < D; it++)
      {
         const int dof = c == 0 ? own_layer + D * it : it + D * own_layer;
         uo += B[q1 * D + it] * sU[dof];
         un_ += B[q1 * D + it] * sN[f][it];
      }
      sF[f][q1] = sp * (un_ - uo);
      sS[f][q1] = sp;
   }
   __syncthreads();
   // ---- dofs: tests (Gauss-Legendre nodal basis for the solve, Bernstein for the lumped mass) ---------------
   const bool dof = tid < D2;
   const int ix = dof ? tid % D : 0, iy = dof ? tid / D : 0;
   double mi = 0, b = 0, diag = 1;
   {
      double dg = 0;
#pragma unroll
      for (int qy = 0; qy < Q; qy++)
      {
#pragma unroll
         for (int qx = 0; qx < Q; qx++)
         {
            const double wd = sW[qx + Q * qy];
            mi += B[qx * D + ix] * B[qy * D + iy] * wd;
            if (MODE == 0)
            {
               b += Bg[qx * D + ix] * Bg[qy * D + iy] * sG[qx + Q * qy];
               dg += Bg2[qx * D + ix] * Bg2[qy * D + iy] * wd;
            }
         }
      }
      if (MODE == 0)
      {
         // faces: the GL nodal basis does not vanish at the element boundary -- every dof tests every face (BgE)
#pragma unroll
         for (int f = 0; f < 4; f++)
         {
            const int c = f / 2, side = f % 2;
            const int ic = c == 0 ? ix : iy, it = c == 0 ? iy : ix;
            double s = 0;
#pragma unroll
            for (int q1 = 0; q1 < Q; q1++) { s += Bg[q1 * D + it] * sF[f][q1]; }
            b += BgE[side * D + ic] * s;
         }
         diag = dg;
      }
   }
   if (dof) { a.m[e * D2 + tid] = mi; }
   if (MODE == 2)
   {
      constexpr double eps = 1e-15, gamma = 1.0;
      const double ui = dof ? sU[tid] : 0.0;
      // z = K_vol u tested with the Bernstein basis; lumped upwind face fluxes: B^T D B 1 = B^T D (remhos_lo.cpp:854-868)
      double zi = 0, fl = 0;
      if (dof)
      {
#pragma unroll
         for (int qy = 0; qy < Q; qy++)
         {
#pragma unroll
            for (int qx = 0; qx < Q; qx++) { zi += B[qx * D + ix] * B[qy * D + iy] * sG[qx + Q * qy]; }
         }
#pragma unroll
         for (int f = 0; f < 4; f++)
         {
            const int c = f / 2, side = f % 2;
            const int ic = c == 0 ? ix : iy, it = c == 0 ? iy : ix;
            if (ic == (side == 0 ? 0 : P))
            {
               double coef = 0;
#pragma unroll
               for (int q1 = 0; q1 < Q; q1++) { coef += B[q1 * D + it] * sS[f][q1]; }
               fl += coef * (sN[f][it] - ui);
            }
         }
      }
      const double xmax = wave_bcast<63>(wave_minmax<false>(dof ? ui : -INFINITY));
      const double xmin = wave_bcast<63>(wave_minmax<true>(dof ? ui : INFINITY));
      const double xsum = wave_bcast<63>(wave_sum(ui));
      const double rhoP = wave_bcast<63>(wave_sum(dof ? fmax(0.0, zi) : 0.0));
      const double rhoN = wave_bcast<63>(wave_sum(dof ? fmin(0.0, zi) : 0.0));
      double wP = (xmax - ui) / (D2 * xmax - xsum + eps);
      double wN = (xmin - ui) / (D2 * xmin - xsum - eps);
      if (a.rd_subcell)
      {
         // subcell fluctuations with the 1-point rule on the closed-uniform sub-mesh (remhos_lo.cpp:1051-1192, 1473-1612):
         // lane = subcell (mx, my), corners j = jx + 2 jy
         const bool sub = tid < NS;
         const int mx = sub ? tid % P : 0, my = sub ? tid / P : 0;
         const double *Lcu = sT + T::oLcu;
         double fP = 0, fN = 0;
         if (sub)
         {
            double xc[4][2], uc[4], vm[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
               const int cx = mx + (j & 1), cy = my + (j >> 1), dj = cx + D * cy;
               uc[j] = sU[dj];
#pragma unroll
               for (int k = 0; k < 2; k++)
               {
                  // start position = Q2 map of the START mesh at (cx / p, cy / p) (MakeRefined, remhos.cpp:796-826), moved
                  // with the sub-mesh velocity (remhos.cpp:837-853)
                  double x = 0;
#pragma unroll
                  for (int ay = 0; ay < 3; ay++)
                  {
#pragma unroll
                     for (int ax = 0; ax < 3; ax++) { x += Lcu[cx * 3 + ax] * Lcu[cy * 3 + ay] * sX0[k][ax + 3 * ay]; }
                  }
                  xc[j][k] = a.move ? x + a.t * sSv[k][dj] : x;
                  vm[k] += 0.25 * sSv[k][dj];
               }
            }
            // J_sub = sum_j corner_j (x) dpsi_j, dpsi_j = (+-1/2, +-1/2); weights W_j = dpsi_j . (alpha adj(J_sub) v_mid)
            double Js[2][2] = {{0, 0}, {0, 0}};
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
#pragma unroll
               for (int k = 0; k < 2; k++)
               {
                  Js[k][0] += ((j & 1) ? 0.5 : -0.5) * xc[j][k];
                  Js[k][1] += ((j >> 1) ? 0.5 : -0.5) * xc[j][k];
               }
            }
            const double v0 = a.alpha * (Js[1][1] * vm[0] - Js[0][1] * vm[1]);
            const double v1 = a.alpha * (-Js[1][0] * vm[0] + Js[0][0] * vm[1]);
            double fluct = 0, smax = uc[0], smin = uc[0], ssum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
               fluct += (((j & 1) ? 0.5 : -0.5) * v0 + ((j >> 1) ? 0.5 : -0.5) * v1) * uc[j];
               smax = fmax(smax, uc[j]);
               smin = fmin(smin, uc[j]);
               ssum += uc[j];
            }
            fP = fmax(0.0, fluct);
            fN = fmin(0.0, fluct);
            const double swP = 4 * smax - ssum + eps, swN = 4 * smin - ssum - eps;
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
               sCP[tid][j] = fP * ((smax - uc[j]) / swP);
               sCN[tid][j] = fN * ((smin - uc[j]) / swN);
            }
         }
         const double sumFP = wave_bcast<63>(wave_sum(fP)), sumFN = wave_bcast<63>(wave_sum(fN));
         __syncthreads();
         double nwP = 0, nwN = 0;
         if (dof)
         {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
               const int sx = ix - (j & 1), sy = iy - (j >> 1); // the subcell that has this dof as corner j
               if (sx >= 0 && sx < P && sy >= 0 && sy < P)
               {
                  nwP += sCP[sx + P * sy][j];
                  nwN += sCN[sx + P * sy][j];
               }
            }
         }
         double aux = gamma / (rhoP + eps);
         wP = wP * (1.0 - fmin(aux * sumFP, 1.0)) + fmin(aux, 1.0 / (sumFP + eps)) * nwP;
         aux = gamma / (rhoN - eps);
         wN = wN * (1.0 - fmin(aux * sumFN, 1.0)) + fmax(aux, 1.0 / (sumFN - eps)) * nwN;
      }
      if (dof) { a.du[e * D2 + tid] = (fl + wP * rhoP + wN * rhoN) / mi; }
      if (a.xe_min && tid == 0) { a.xe_min[e] = xmin; a.xe_max[e] = xmax; }
      return;
   }
   if (MODE != 0) { return; }
   if (!dof) { b = 0; mi = 0; }
   if (a.xe_min)
   {
      double lo = dof ? sU[tid] : INFINITY, hi = dof ? sU[tid] : -INFINITY;
      lo = wave_minmax<true>(lo);
      hi = wave_minmax<false>(hi);
      if (tid == 63) { a.xe_min[e] = lo; a.xe_max[e] = hi; }
   }
   // ---- element-local Jacobi-PCG in the GL nodal basis (DGMassInverse) --------------------------------------
   auto dot = [&](double x, double y) { return wave_bcast<63>(wave_sum(dof ? x * y : 0.0)); };
   auto mass_apply = [&](double p) {
      __syncthreads();
      if (dof) { sP[tid] = p; }
      __syncthreads();
      if (tid < Q2)
      {
         const int qx = tid % Q, qy = tid / Q;
         double s = 0;
#pragma unroll
         for (int jy = 0; jy < D; jy++)
         {
#pragma unroll
            for (int jx = 0; jx < D; jx++) { s += Bg[qx * D + jx] * Bg[qy * D + jy] * sP[jx + D * jy]; }
         }
         sPQ[tid] = s * sW[tid];
      }
      __syncthreads();
      double z = 0;
#pragma unroll
      for (int qy = 0; qy < Q; qy++)
      {
#pragma unroll
         for (int qx = 0; qx < Q; qx++) { z += Bg[qx * D + ix] * Bg[qy * D + iy] * sPQ[qx + Q * qy]; }
      }
      return z;
   };
   double x = 0, r = b, z = r / diag, d = z;
   double nom = dot(z, r);
   const double tol = fmax(a.rel2 * nom, a.abs2);
   int it = 0;
   while (nom > tol && it < a.max_iter) // (uniform over the wavefront: the sums are broadcast)
   {
      const double Ad = mass_apply(d);
      const double den = dot(d, Ad);
      const double al = nom / (den != 0.0 ? den : 1.0);
      x += al * d;
      r -= al * Ad;
      z = r / diag;
      const double betanom = dot(z, r);
      const double beta = betanom / (nom != 0.0 ? nom : 1.0);
      d = z + beta * d;
      nom = betanom;
      it++;
   }
   if (tid == 0 && it > *a.cg_iters) { atomicMax(a.cg_iters, it); }
   if (a.jacobi_step) { x += r / diag; }
   // ---- back to Bernstein coefficients: x_b = (Ci (x) Ci) x_g -------------------------------------------------
   __syncthreads();
   if (dof) { sP[tid] = x; }
   __syncthreads();
   double xb = 0;
#pragma unroll
   for (int ky = 0; ky < D; ky++)
   {
#pragma unroll
      for (int kx = 0; kx < D; kx++) { xb += Ci[ix * D + kx] * Ci[iy * D + ky] * sP[kx + D * ky]; }
   }
   if (a.mass_fix)
   {
      // constant mode: the element's mass rate sum m x_b equals 1^T b (rmh_set_mass_completion)
      const double sb = wave_bcast<63>(wave_sum(b)), smx = wave_bcast<63>(wave_sum(dof ? mi * xb : 0.0));
      const double sm = wave_bcast<63>(wave_sum(mi));
      xb += (sb - smx) / sm;
   }
   if (dof) { a.du[e * D2 + tid] = xb; }
}

} // namespace rmh
