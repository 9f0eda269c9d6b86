// CPU port of the Remhos RK stage (HO -> mass-based LO -> overlap bounds -> ClipScale) --
// TEST INFRASTRUCTURE ONLY: the host-core baseline that bench.py times beside the GPU
// (cpu_baseline.kind = "port") and a fast checker for meshes too large for the numpy oracle.
// Never linked into or called by the product (remhos_amd/).
//
// Parity status: pinned through tests/test_cpu_port.py -- the port is compared with
// oracle/remhos_oracle.py (itself pinned by the reference's known answers) and reproduces the
// reference's ctest #3 / #7 final masses (remhos_tests.cpp:63-68, 81-86).
//
// Restated from the reference (same algorithm as its CPU partial-assembly path; sum-factorised
// like MFEM's PA kernels, OpenMP over elements):
//   HO   : K_vol (ConvectionIntegrator PA apply, remhos.cpp:646-657; qdata remhos_lo.cpp:1155-1190,
//          apply remhos_lo.cpp:1473-1612) + upwind DG trace (remhos.cpp:659-678) + element-local
//          Jacobi-PCG mass solve in the Gauss-Legendre basis (DGMassInverse, remhos_ho.cpp:79-80,126)
//   LO   : MassBasedAvg (remhos_lo.cpp:247-324)
//   bounds: ComputeElementsMinMax + ComputeOverlapBounds (remhos_tools.cpp:432-523)
//   FCT  : ClipScaleSolver (remhos_fct.cpp:449-541)
//   RK3-SSP step and the remap re-setup per stage (remhos.cpp:1598-1637).
// 1-D tables come from remhos_amd/csrc/rmh_tables.hpp (pure host math, shared on purpose: the
// tables themselves are pinned by every parity test).
#include "../remhos_amd/csrc/rmh_tables.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{

struct Tab
{
   int P, D, Q;
   std::vector<double> t;
   int oB, oG, oL, odL, oW, oBg, oCi;
};

template <int P>
Tab make_tab()
{
   using T = rmh::TabLayout<P>;
   Tab b;
   b.P = P; b.D = T::D; b.Q = T::Q;
   b.t = rmh::make_tables<P>();
   b.oB = T::oB; b.oG = T::oG; b.oL = T::oL; b.odL = T::odL; b.oW = T::oW; b.oBg = T::oBg; b.oCi = T::oCi;
   return b;
}

Tab get_tab(int p)
{
   switch (p)
   {
      case 1: return make_tab<1>();
      case 2: return make_tab<2>();
      case 3: return make_tab<3>();
      case 4: return make_tab<4>();
      case 5: return make_tab<5>();
      default: return make_tab<6>();
   }
}

// out[.., q, ..] = sum_i M[q*ni + i] in[.., i, ..] along direction dir of an (n0,n1,n2) tensor (x fastest)
void contract(const double *M, int nq, int ni, int dir, const int n[3], const double *in, double *out)
{
   int no[3] = {n[0], n[1], n[2]};
   no[dir] = nq;
   const int si = (dir == 0) ? 1 : (dir == 1 ? n[0] : n[0] * n[1]);
   const int so = (dir == 0) ? 1 : (dir == 1 ? no[0] : no[0] * no[1]);
   for (int k2 = 0; k2 < no[2]; k2++)
   {
      for (int k1 = 0; k1 < no[1]; k1++)
      {
         for (int k0 = 0; k0 < no[0]; k0++)
         {
            const int k[3] = {k0, k1, k2};
            int ib = 0, st = 1;
            for (int d = 0; d < 3; d++)
            {
               if (d != dir) { ib += k[d] * st; }
               st *= n[d];
            }
            double acc = 0.0;
            for (int i = 0; i < ni; i++) { acc += M[k[dir] * ni + i] * in[ib + i * si]; }
            out[k0 + no[0] * (k1 + no[1] * k2)] = acc;
            (void)so;
         }
      }
   }
}

// transpose version: out[.., i, ..] = sum_q M[q*ni + i] in[.., q, ..]
void contract_t(const double *M, int nq, int ni, int dir, const int n[3], const double *in, double *out)
{
   int no[3] = {n[0], n[1], n[2]};
   no[dir] = ni;
   const int si = (dir == 0) ? 1 : (dir == 1 ? n[0] : n[0] * n[1]);
   for (int k2 = 0; k2 < no[2]; k2++)
   {
      for (int k1 = 0; k1 < no[1]; k1++)
      {
         for (int k0 = 0; k0 < no[0]; k0++)
         {
            const int k[3] = {k0, k1, k2};
            int ib = 0, st = 1;
            for (int d = 0; d < 3; d++)
            {
               if (d != dir) { ib += k[d] * st; }
               st *= n[d];
            }
            double acc = 0.0;
            for (int q = 0; q < nq; q++) { acc += M[q * ni + k[dir]] * in[ib + q * si]; }
            out[k0 + no[0] * (k1 + no[1] * k2)] = acc;
         }
      }
   }
}

struct Work
{
   std::vector<double> a, b, c, X, V, J, g, wd, Dq;
};

// tensor apply of three 1-D matrices (values or derivative per direction), D^3 / 3^3 -> Q^3
void interp3(const double *Mx, const double *My, const double *Mz, int nq, int ni, const double *in, double *out,
             Work &w)
{
   const int n0[3] = {ni, ni, ni};
   const int n1[3] = {nq, ni, ni};
   const int n2[3] = {nq, nq, ni};
   w.a.resize((size_t)nq * ni * ni);
   w.b.resize((size_t)nq * nq * ni);
   contract(Mx, nq, ni, 0, n0, in, w.a.data());
   contract(My, nq, ni, 1, n1, w.a.data(), w.b.data());
   contract(Mz, nq, ni, 2, n2, w.b.data(), out);
}

void test3(const double *Mx, const double *My, const double *Mz, int nq, int ni, const double *in, double *out,
           Work &w)
{
   const int n0[3] = {nq, nq, nq};
   const int n1[3] = {nq, nq, ni};
   const int n2[3] = {nq, ni, ni};
   w.a.resize((size_t)nq * nq * ni);
   w.b.resize((size_t)nq * ni * ni);
   contract_t(Mz, nq, ni, 2, n0, in, w.a.data());
   contract_t(My, nq, ni, 1, n1, w.a.data(), w.b.data());
   contract_t(Mx, nq, ni, 0, n2, w.b.data(), out);
}

} // namespace

extern "C" {

// One stage: du = F(u, t) with dt the full step (LO/FCT).  Arrays as in include/rmh.h (host memory).
// ws_m receives the lumped mass.  Returns the max PCG iteration count.
int cpu_stage(int p, int ne, int exec_mode, const double *x0, const double *vel, const int *face_nbr,
              const int *stencil27, double t, double dt, const double *u, double *du, double *ws_m,
              double *ws_duho, double *ws_xe /* [2*ne] */, double rel_tol)
{
   const Tab T = get_tab(p);
   const int D = T.D, Q = T.Q, D2 = D * D, D3 = D2 * D, Q2 = Q * Q, Q3 = Q2 * Q;
   const double *B = &T.t[T.oB], *G = &T.t[T.oG], *L = &T.t[T.oL], *dL = &T.t[T.odL], *W = &T.t[T.oW];
   const double *Bg = &T.t[T.oBg], *Ci = &T.t[T.oCi];
   const double alpha = exec_mode == 1 ? 1.0 : -1.0, upw = exec_mode == 1 ? 1.0 : -1.0;
   int itmax = 0;

#pragma omp parallel reduction(max : itmax)
   {
   // per-thread scratch (allocated once per thread, not per element)
   Work w;
   std::vector<double> X(81), V(81), Jc(9 * Q3), vq(3 * Q3), tmp(Q3), Dq(3 * Q3), wd(Q3), g(Q3), gr(3 * Q3);
   std::vector<double> rhs(D3), rg(D3), xg(D3), dd(D3), Ad(D3), dg(D3), zz(D3), t1(D3), t2(D3), val(Q2), Bg2(Q * D);
   for (int i = 0; i < Q * D; i++) { Bg2[i] = Bg[i] * Bg[i]; }
#pragma omp for schedule(static)
   for (int e = 0; e < ne; e++)
   {
      const double *ue = u + (size_t)e * D3;
      for (int i = 0; i < 81; i++)
      {
         V[i] = vel[(size_t)e * 81 + i];
         X[i] = x0[(size_t)e * 81 + i] + (exec_mode == 1 ? t * V[i] : 0.0);
      }
      // geometry at the quadrature points (remap: moved mesh, remhos.cpp:1598-1608)
      for (int c = 0; c < 3; c++)
      {
         interp3(dL, L, L, Q, 3, &X[c * 27], &Jc[(c * 3 + 0) * Q3], w);
         interp3(L, dL, L, Q, 3, &X[c * 27], &Jc[(c * 3 + 1) * Q3], w);
         interp3(L, L, dL, Q, 3, &X[c * 27], &Jc[(c * 3 + 2) * Q3], w);
         interp3(L, L, L, Q, 3, &V[c * 27], &vq[c * Q3], w);
      }
      for (int q = 0; q < Q3; q++)
      {
         const int qx = q % Q, qy = (q / Q) % Q, qz = q / Q2;
         double J[3][3];
         for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) { J[a][b] = Jc[(a * 3 + b) * Q3 + q]; } }
         const double A11 = J[1][1] * J[2][2] - J[1][2] * J[2][1], A12 = J[2][1] * J[0][2] - J[0][1] * J[2][2];
         const double A13 = J[0][1] * J[1][2] - J[1][1] * J[0][2], A21 = J[2][0] * J[1][2] - J[1][0] * J[2][2];
         const double A22 = J[0][0] * J[2][2] - J[0][2] * J[2][0], A23 = J[1][0] * J[0][2] - J[0][0] * J[1][2];
         const double A31 = J[1][0] * J[2][1] - J[2][0] * J[1][1], A32 = J[2][0] * J[0][1] - J[0][0] * J[2][1];
         const double A33 = J[0][0] * J[1][1] - J[0][1] * J[1][0];
         const double detJ = J[0][0] * A11 + J[0][1] * A21 + J[0][2] * A31;
         const double w3 = W[qx] * W[qy] * W[qz];
         const double v0 = vq[q], v1 = vq[Q3 + q], v2 = vq[2 * Q3 + q];
         Dq[q] = alpha * w3 * (A11 * v0 + A12 * v1 + A13 * v2);
         Dq[Q3 + q] = alpha * w3 * (A21 * v0 + A22 * v1 + A23 * v2);
         Dq[2 * Q3 + q] = alpha * w3 * (A31 * v0 + A32 * v1 + A33 * v2);
         wd[q] = w3 * detJ;
      }
      // K_vol u
      interp3(G, B, B, Q, D, ue, &gr[0], w);
      interp3(B, G, B, Q, D, ue, &gr[Q3], w);
      interp3(B, B, G, Q, D, ue, &gr[2 * Q3], w);
      for (int q = 0; q < Q3; q++) { g[q] = Dq[q] * gr[q] + Dq[Q3 + q] * gr[Q3 + q] + Dq[2 * Q3 + q] * gr[2 * Q3 + q]; }
      test3(B, B, B, Q, D, g.data(), rhs.data(), w);
      // lumped mass M 1
      test3(B, B, B, Q, D, wd.data(), &ws_m[(size_t)e * D3], w);
      // faces (own outward normal)
      for (int f = 0; f < 6; f++)
      {
         const int c = f >> 1, side = f & 1, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
         const int pw3[3] = {1, 3, 9}, pwD[3] = {1, D, D2};
         const int nb = face_nbr[(size_t)e * 6 + f];
         const double *un = nb >= 0 ? u + (size_t)nb * D3 : nullptr;
         for (int q2 = 0; q2 < Q; q2++)
         {
            for (int q1 = 0; q1 < Q; q1++)
            {
               double t1v[3] = {0, 0, 0}, t2v[3] = {0, 0, 0}, vf[3] = {0, 0, 0};
               for (int a2 = 0; a2 < 3; a2++)
               {
                  for (int a1 = 0; a1 < 3; a1++)
                  {
                     const int n = (side ? 2 : 0) * pw3[c] + a1 * pw3[c1] + a2 * pw3[c2];
                     const double w1 = dL[q1 * 3 + a1] * L[q2 * 3 + a2], w2 = L[q1 * 3 + a1] * dL[q2 * 3 + a2];
                     const double w0 = L[q1 * 3 + a1] * L[q2 * 3 + a2];
                     for (int k = 0; k < 3; k++)
                     {
                        t1v[k] += w1 * X[k * 27 + n];
                        t2v[k] += w2 * X[k * 27 + n];
                        vf[k] += w0 * V[k * 27 + n];
                     }
                  }
               }
               const double nx = t1v[1] * t2v[2] - t1v[2] * t2v[1], ny = t1v[2] * t2v[0] - t1v[0] * t2v[2];
               const double nz = t1v[0] * t2v[1] - t1v[1] * t2v[0];
               double vn = vf[0] * nx + vf[1] * ny + vf[2] * nz;
               if (!side) { vn = -vn; }
               const double s = std::fmax(0.0, upw * vn) * W[q1] * W[q2];
               double jump = 0.0;
               for (int i2 = 0; i2 < D; i2++)
               {
                  for (int i1 = 0; i1 < D; i1++)
                  {
                     const double own = ue[(side ? (D - 1) : 0) * pwD[c] + i1 * pwD[c1] + i2 * pwD[c2]];
                     const double nbv = un ? un[(side ? 0 : (D - 1)) * pwD[c] + i1 * pwD[c1] + i2 * pwD[c2]] : 0.0;
                     jump += B[q1 * D + i1] * B[q2 * D + i2] * (nbv - own);
                  }
               }
               val[q1 + Q * q2] = s * jump;
            }
         }
         for (int i2 = 0; i2 < D; i2++)
         {
            for (int i1 = 0; i1 < D; i1++)
            {
               double acc = 0.0;
               for (int q2 = 0; q2 < Q; q2++)
               {
                  for (int q1 = 0; q1 < Q; q1++) { acc += B[q1 * D + i1] * B[q2 * D + i2] * val[q1 + Q * q2]; }
               }
               rhs[(side ? (D - 1) : 0) * pwD[c] + i1 * pwD[c1] + i2 * pwD[c2]] += acc;
            }
         }
      }
      // local mass solve: Jacobi-PCG in the GL nodal basis
      const int nD[3] = {D, D, D};
      contract_t(Ci, D, D, 0, nD, rhs.data(), t1.data()); // b_g = Ci^T (x)3 b : out[k] = sum_i Ci[i*D+k] in[i]
      contract_t(Ci, D, D, 1, nD, t1.data(), t2.data());
      contract_t(Ci, D, D, 2, nD, t2.data(), rg.data());
      test3(Bg2.data(), Bg2.data(), Bg2.data(), Q, D, wd.data(), dg.data(), w);
      double nom = 0.0;
      for (int i = 0; i < D3; i++)
      {
         xg[i] = 0.0;
         dd[i] = rg[i] / dg[i];
         nom += rg[i] * dd[i];
      }
      const double tol = rel_tol * rel_tol * nom;
      int it = 0;
      while (nom > tol && it < 100)
      {
         interp3(Bg, Bg, Bg, Q, D, dd.data(), tmp.data(), w);
         for (int q = 0; q < Q3; q++) { tmp[q] *= wd[q]; }
         test3(Bg, Bg, Bg, Q, D, tmp.data(), Ad.data(), w);
         double den = 0.0;
         for (int i = 0; i < D3; i++) { den += dd[i] * Ad[i]; }
         if (!(den > 0.0)) { break; }
         const double al = nom / den;
         double bn = 0.0;
         for (int i = 0; i < D3; i++)
         {
            xg[i] += al * dd[i];
            rg[i] -= al * Ad[i];
            bn += rg[i] * (rg[i] / dg[i]);
         }
         const double beta = bn / nom;
         for (int i = 0; i < D3; i++) { dd[i] = rg[i] / dg[i] + beta * dd[i]; }
         nom = bn;
         it++;
      }
      itmax = std::max(itmax, it);
      contract(Ci, D, D, 0, nD, xg.data(), t1.data()); // x_b = Ci (x)3 x_g : out[i] = sum_k Ci[i*D+k] in[k]
      contract(Ci, D, D, 1, nD, t1.data(), t2.data());
      contract(Ci, D, D, 2, nD, t2.data(), &ws_duho[(size_t)e * D3]);
      double lo = INFINITY, hi = -INFINITY;
      for (int i = 0; i < D3; i++)
      {
         lo = std::fmin(lo, ue[i]);
         hi = std::fmax(hi, ue[i]);
      }
      ws_xe[e] = lo;
      ws_xe[ne + e] = hi;
   }
   } // omp parallel

   // LimitMult: MassBasedAvg + overlap bounds + ClipScale
#pragma omp parallel
   {
   std::vector<double> fc(D3), dl(D3);
#pragma omp for schedule(static)
   for (int e = 0; e < ne; e++)
   {
      const double *ue = u + (size_t)e * D3, *m = ws_m + (size_t)e * D3, *dh = ws_duho + (size_t)e * D3;
      double mass = 0.0, vol = 0.0;
      for (int i = 0; i < D3; i++)
      {
         mass += m[i] * (ue[i] + dt * dh[i]);
         vol += m[i];
      }
      const double ubar = mass / vol;
      double smin[27], smax[27];
      for (int s = 0; s < 27; s++)
      {
         const int nb = stencil27[(size_t)e * 27 + s];
         smin[s] = nb >= 0 ? ws_xe[nb] : INFINITY;
         smax[s] = nb >= 0 ? ws_xe[ne + nb] : -INFINITY;
      }
      double sumPos = 0.0, sumNeg = 0.0;
      const double eps = 1.0e-15;
      for (int i = 0; i < D3; i++)
      {
         const int idx[3] = {i % D, (i / D) % D, i / D2};
         double umin = INFINITY, umax = -INFINITY;
         for (int oz = (idx[2] == 0 ? -1 : 0); oz <= (idx[2] == D - 1 ? 1 : 0); oz++)
         {
            for (int oy = (idx[1] == 0 ? -1 : 0); oy <= (idx[1] == D - 1 ? 1 : 0); oy++)
            {
               for (int ox = (idx[0] == 0 ? -1 : 0); ox <= (idx[0] == D - 1 ? 1 : 0); ox++)
               {
                  const int s = (ox + 1) + 3 * (oy + 1) + 9 * (oz + 1);
                  umin = std::fmin(umin, smin[s]);
                  umax = std::fmax(umax, smax[s]);
               }
            }
         }
         dl[i] = (ubar - ue[i]) / dt;
         const double u_new_lo = ue[i] + dt * dl[i];
         const double f_clip_min = m[i] / dt * (umin - u_new_lo);
         const double f_clip_max = m[i] / dt * (umax - u_new_lo);
         double f = m[i] * (dh[i] - dl[i]);
         f = std::fmin(f_clip_max, std::fmax(f_clip_min, f));
         fc[i] = f;
         sumNeg += std::fmin(f, 0.0);
         sumPos += std::fmax(f, 0.0);
      }
      const double new_mass = sumNeg + sumPos;
      for (int i = 0; i < D3; i++)
      {
         double f = fc[i];
         if (new_mass > eps) { f = std::fmin(0.0, f) - std::fmax(0.0, f) * sumNeg / sumPos; }
         if (new_mass < -eps) { f = std::fmax(0.0, f) - std::fmin(0.0, f) * sumPos / sumNeg; }
         du[(size_t)e * D3 + i] = dl[i] + f / m[i];
      }
   }
   } // omp parallel
   return itmax;
}

// RK3-SSP step in place on u (size ne*ndof); work arrays are allocated by the caller (5 * ne*ndof + 2*ne).
int cpu_rk3_step(int p, int ne, int exec_mode, const double *x0, const double *vel, const int *face_nbr,
                 const int *stencil27, double t, double dt, double *u, double *work, double rel_tol)
{
   const size_t n = (size_t)ne * (p + 1) * (p + 1) * (p + 1);
   double *y = work, *k = work + n, *m = work + 2 * n, *dh = work + 3 * n, *xe = work + 4 * n;
   int it = cpu_stage(p, ne, exec_mode, x0, vel, face_nbr, stencil27, t, dt, u, k, m, dh, xe, rel_tol);
#pragma omp parallel for
   for (long long i = 0; i < (long long)n; i++) { y[i] = u[i] + dt * k[i]; }
   it = std::max(it, cpu_stage(p, ne, exec_mode, x0, vel, face_nbr, stencil27, t + dt, dt, y, k, m, dh, xe, rel_tol));
#pragma omp parallel for
   for (long long i = 0; i < (long long)n; i++) { y[i] = 0.75 * u[i] + 0.25 * (y[i] + dt * k[i]); }
   it = std::max(it, cpu_stage(p, ne, exec_mode, x0, vel, face_nbr, stencil27, t + dt / 2, dt, y, k, m, dh, xe, rel_tol));
#pragma omp parallel for
   for (long long i = 0; i < (long long)n; i++) { u[i] = (1.0 / 3.0) * u[i] + (2.0 / 3.0) * (y[i] + dt * k[i]); }
   return it;
}

// lumped mass at pseudo-time t (initial / final mass)
void cpu_lumped_mass(int p, int ne, int exec_mode, const double *x0, const double *vel, double t, double *m)
{
   const Tab T = get_tab(p);
   const int D = T.D, Q = T.Q, D3 = D * D * D, Q2 = Q * Q, Q3 = Q2 * Q;
   const double *B = &T.t[T.oB], *L = &T.t[T.oL], *dL = &T.t[T.odL], *W = &T.t[T.oW];
#pragma omp parallel for schedule(static)
   for (int e = 0; e < ne; e++)
   {
      Work w;
      std::vector<double> X(81), Jc(9 * Q3), wd(Q3);
      for (int i = 0; i < 81; i++) { X[i] = x0[(size_t)e * 81 + i] + (exec_mode == 1 ? t * vel[(size_t)e * 81 + i] : 0.0); }
      for (int c = 0; c < 3; c++)
      {
         interp3(dL, L, L, Q, 3, &X[c * 27], &Jc[(c * 3 + 0) * Q3], w);
         interp3(L, dL, L, Q, 3, &X[c * 27], &Jc[(c * 3 + 1) * Q3], w);
         interp3(L, L, dL, Q, 3, &X[c * 27], &Jc[(c * 3 + 2) * Q3], w);
      }
      for (int q = 0; q < Q3; q++)
      {
         double J[3][3];
         for (int a = 0; a < 3; a++) { for (int b = 0; b < 3; b++) { J[a][b] = Jc[(a * 3 + b) * Q3 + q]; } }
         const double detJ = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[2][0] * J[1][2] - J[1][0] * J[2][2]) +
                             J[0][2] * (J[1][0] * J[2][1] - J[2][0] * J[1][1]);
         wd[q] = W[q % Q] * W[(q / Q) % Q] * W[q / Q2] * detJ;
      }
      test3(B, B, B, Q, D, wd.data(), &m[(size_t)e * D3], w);
   }
}

int cpu_num_threads(void)
{
#ifdef _OPENMP
   return omp_get_max_threads();
#else
   return 1;
#endif
}

} // extern "C"
