// CPU port of the Remhos RK stage (HO -> mass-based LO -> overlap bounds -> ClipScale) --
// TEST INFRASTRUCTURE ONLY: the host-core baseline that bench.py times beside the GPU
// (cpu_baseline.kind = "port") and a fast checker for meshes too large for the numpy oracle.
// Never linked into or called by the product (remhos_amd/), and it shares no source with it: the 1-D tables are
// generated here (Tables<P>), independently of remhos_amd/csrc/rmh_tables.hpp.
//
// Parity status: pinned through tests/test_cpu_port.py and tests/test_golden.py -- the port is compared with
// oracle/remhos_oracle.py (itself pinned by the reference's known answers) and reproduces the
// reference's ctest #3 / #7 final masses (remhos_tests.cpp:63-68, 81-86).
//
// Restated from the reference (same algorithm as its CPU partial-assembly path: sum-factorised like MFEM's PA kernels,
// every contraction specialised on (p + 1, p + 3) at compile time, OpenMP over elements; the HO loop -- geometry, K u,
// local mass solve -- is vectorised ACROSS elements, batches of 8 (AVX-512) or 4 (AVX2) elements, one per SIMD lane):
//   HO   : K_vol (ConvectionIntegrator PA apply, remhos.cpp:646-657; qdata remhos_lo.cpp:1155-1190,
//          apply remhos_lo.cpp:1473-1612) + upwind DG trace (remhos.cpp:659-678) + element-local
//          Jacobi-PCG mass solve in the Gauss-Legendre basis (DGMassInverse, remhos_ho.cpp:79-80,126)
//   LO   : MassBasedAvg (remhos_lo.cpp:247-324)
//   bounds: ComputeElementsMinMax + ComputeOverlapBounds (remhos_tools.cpp:432-523)
//   FCT  : ClipScaleSolver (remhos_fct.cpp:449-541)
//   RK3-SSP step and the remap re-setup per stage (remhos.cpp:1598-1637).
// Stopwatch buckets like TimingData (remhos_tools.hpp:52-64): RHS = geometry + K u (the reference charges the remap
// re-assembly of K to it, remhos.cpp:1634-1637), INV = mass-solve set-up + PCG (:1620-1623), LO, FCT (incl. bounds).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{

// ---- 1-D tables ---------------------------------------------------------------------------------------------------
// Gauss-Legendre points / weights on [0, 1]: roots of the Legendre polynomial by bisection on the interlacing
// intervals of the previous degree, polished by Newton
void gauss_legendre(int n, double *x, double *w)
{
   auto legendre = [](int m, double z, double &dp)
   {
      double p0 = 1.0, p1 = z;
      if (m == 0) { dp = 0.0; return 1.0; }
      for (int k = 2; k <= m; k++)
      {
         const double pk = ((2 * k - 1) * z * p1 - (k - 1) * p0) / k;
         p0 = p1;
         p1 = pk;
      }
      dp = m * (z * p1 - p0) / (z * z - 1.0);
      return p1;
   };
   std::vector<double> prev, cur;
   for (int m = 1; m <= n; m++)
   {
      cur.assign(m, 0.0);
      for (int i = 0; i < m; i++)
      {
         double a = i == 0 ? -1.0 : prev[i - 1], b = i == m - 1 ? 1.0 : prev[i], dp;
         const double fa = legendre(m, a, dp);
         for (int it = 0; it < 200; it++)
         {
            const double mid = 0.5 * (a + b);
            if ((legendre(m, mid, dp) > 0) == (fa > 0)) { a = mid; } else { b = mid; }
         }
         double z = 0.5 * (a + b);
         for (int it = 0; it < 3; it++) { const double f = legendre(m, z, dp); z -= f / dp; }
         cur[i] = z;
      }
      prev = cur;
   }
   for (int i = 0; i < n; i++)
   {
      double dp;
      legendre(n, cur[i], dp);
      x[i] = 0.5 * (1.0 + cur[i]);
      w[i] = 1.0 / ((1.0 - cur[i] * cur[i]) * dp * dp); // (2 / ((1-z^2) P'^2)) / 2 for [0, 1]
   }
}

template <int P>
struct Tables
{
   static constexpr int D = P + 1, Q = P + 3; // Q: rule of order 2p + 2*3 - 1
   double W[Q];
   double B[Q * D], G[Q * D];     // Bernstein values / derivatives at the quadrature points, [q][i]
   double Bt[D * Q], Gt[D * Q];   // transposed, [i][q]
   double L[Q * 3], dL[Q * 3];    // quadratic Lagrange basis on {0, 1/2, 1} (mesh nodes), [q][a]
   double Bg[Q * D], Bgt[D * Q];  // Gauss-Legendre nodal basis (D nodes) at the quadrature points
   double Bg2t[D * Q];            // its square, transposed (Jacobi diagonal)
   double Ci[D * D], Cit[D * D];  // x_b = Ci x_g: inverse of V[k][i] = Bernstein_i(GL node k)

   Tables()
   {
      double xq[Q], xg[D], wg[D];
      gauss_legendre(Q, xq, W);
      gauss_legendre(D, xg, wg);
      auto binom = [](int n, int k) { double r = 1.0; for (int i = 1; i <= k; i++) { r = r * (n - k + i) / i; } return r; };
      auto bern = [&](int p, int i, double t) { return (i < 0 || i > p) ? 0.0 : binom(p, i) * std::pow(t, i) * std::pow(1.0 - t, p - i); };
      for (int q = 0; q < Q; q++)
      {
         const double t = xq[q];
         for (int i = 0; i < D; i++)
         {
            B[q * D + i] = bern(P, i, t);
            G[q * D + i] = P * (bern(P - 1, i - 1, t) - bern(P - 1, i, t));
            Bt[i * Q + q] = B[q * D + i];
            Gt[i * Q + q] = G[q * D + i];
            double l = 1.0;
            for (int k = 0; k < D; k++) { if (k != i) { l *= (t - xg[k]) / (xg[i] - xg[k]); } }
            Bg[q * D + i] = l;
            Bgt[i * Q + q] = l;
            Bg2t[i * Q + q] = l * l;
         }
         L[q * 3 + 0] = 2.0 * (t - 0.5) * (t - 1.0);
         L[q * 3 + 1] = -4.0 * t * (t - 1.0);
         L[q * 3 + 2] = 2.0 * t * (t - 0.5);
         dL[q * 3 + 0] = 4.0 * t - 3.0;
         dL[q * 3 + 1] = -8.0 * t + 4.0;
         dL[q * 3 + 2] = 4.0 * t - 1.0;
      }
      // Ci = V^-1 by Gauss-Jordan with partial pivoting
      double A[D][2 * D];
      for (int k = 0; k < D; k++)
      {
         for (int i = 0; i < D; i++) { A[k][i] = bern(P, i, xg[k]); A[k][D + i] = k == i ? 1.0 : 0.0; }
      }
      for (int c = 0; c < D; c++)
      {
         int piv = c;
         for (int r = c + 1; r < D; r++) { if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) { piv = r; } }
         for (int j = 0; j < 2 * D; j++) { std::swap(A[c][j], A[piv][j]); }
         const double inv = 1.0 / A[c][c];
         for (int j = 0; j < 2 * D; j++) { A[c][j] *= inv; }
         for (int r = 0; r < D; r++)
         {
            if (r == c) { continue; }
            const double f = A[r][c];
            for (int j = 0; j < 2 * D; j++) { A[r][j] -= f * A[c][j]; }
         }
      }
      for (int i = 0; i < D; i++) { for (int k = 0; k < D; k++) { Ci[i * D + k] = A[i][D + k]; Cit[k * D + i] = A[i][D + k]; } }
   }
};

// ---- contractions, all extents known at compile time ------------------------------------------------------------
// out[(k2*N1 + k1)*NQ + q] = sum_i M[q*NI + i] in[(k2*N1 + k1)*NI + i]
// (T = double: one element; T = vd: a batch of VL elements, one per SIMD lane -- the tables stay scalar)
template <int NQ, int NI, int N12, class T>
inline void cx(const double *M, const T *in, T *out)
{
   for (int k = 0; k < N12; k++)
   {
      for (int q = 0; q < NQ; q++)
      {
         T acc = T() * 0.0;
         for (int i = 0; i < NI; i++) { acc += M[q * NI + i] * in[k * NI + i]; }
         out[k * NQ + q] = acc;
      }
   }
}
// out[(k2*NQ + q)*N0 + k0] = sum_i M[q*NI + i] in[(k2*NI + i)*N0 + k0]
template <int NQ, int NI, int N0, int N2, class T>
inline void cy(const double *M, const T *in, T *out)
{
   for (int k2 = 0; k2 < N2; k2++)
   {
      for (int q = 0; q < NQ; q++)
      {
         T acc[N0];
         for (int k0 = 0; k0 < N0; k0++) { acc[k0] = T() * 0.0; }
         for (int i = 0; i < NI; i++)
         {
            const double m = M[q * NI + i];
            for (int k0 = 0; k0 < N0; k0++) { acc[k0] += m * in[(k2 * NI + i) * N0 + k0]; }
         }
         for (int k0 = 0; k0 < N0; k0++) { out[(k2 * NQ + q) * N0 + k0] = acc[k0]; }
      }
   }
}
// out[q*N01 + k] = sum_i M[q*NI + i] in[i*N01 + k]
template <int NQ, int NI, int N01, class T>
inline void cz(const double *M, const T *in, T *out)
{
   for (int q = 0; q < NQ; q++)
   {
      for (int k = 0; k < N01; k++) { out[q * N01 + k] = T() * 0.0; }
      for (int i = 0; i < NI; i++)
      {
         const double m = M[q * NI + i];
         for (int k = 0; k < N01; k++) { out[q * N01 + k] += m * in[i * N01 + k]; }
      }
   }
}
// NI^3 -> NQ^3 with one matrix per direction ([q][i] layout); and the transposed use NQ^3 -> NI^3 with [i][q] matrices
template <int NQ, int NI, class T>
inline void interp3(const double *Mx, const double *My, const double *Mz, const T *in, T *out, T *wa, T *wb)
{
   cx<NQ, NI, NI * NI>(Mx, in, wa);        // [NI][NI][NQ]
   cy<NQ, NI, NQ, NI>(My, wa, wb);         // [NI][NQ][NQ]
   cz<NQ, NI, NQ * NQ>(Mz, wb, out);       // [NQ][NQ][NQ]
}
template <int NI, int NQ, class T>
inline void test3(const double *Mxt, const double *Myt, const double *Mzt, const T *in, T *out, T *wa, T *wb)
{
   cz<NI, NQ, NQ * NQ>(Mzt, in, wa);       // [NI][NQ][NQ]
   cy<NI, NQ, NQ, NI>(Myt, wa, wb);        // [NI][NI][NQ]
   cx<NI, NQ, NI * NI>(Mxt, wb, out);      // [NI][NI][NI]
}

// A batch of VL elements, one per SIMD lane (GCC vector extension: lowered to AVX-512 or to pairs of AVX2 operations,
// whatever the build targets): the element loop of the HO part is vectorised ACROSS elements -- inside one element the
// contraction extents (3 ... 9) are too short to fill a vector.
#ifdef __AVX512F__
constexpr int VL = 8;
#else
constexpr int VL = 4;
#endif
typedef double vd __attribute__((vector_size(VL * sizeof(double))));
typedef long long vm __attribute__((vector_size(VL * sizeof(long long))));
inline vd vsel(vm mask, vd a, vd b) { return mask ? a : b; }
inline vd vmax0(vd x) { return vsel(x > 0.0, x, x * 0.0); }
inline bool vany(vm m)
{
   bool any = false;
   for (int l = 0; l < VL; l++) { any = any || m[l] != 0; }
   return any;
}

double g_bucket[4] = {0, 0, 0, 0}; // RHS, INV, LO, FCT: max over threads, accumulated over calls

inline double now()
{
#ifdef _OPENMP
   return omp_get_wtime();
#else
   return 0.0;
#endif
}

template <int P>
int stage_t(int ne, int exec_mode, const double *x0, const double *vel, const int *face_nbr, const int *stencil27, double t,
            double dt, const double *u, double *du, double *ws_m, double *ws_duho, double *ws_xe, double rel_tol, double abs_tol,
            int completion)
{
   static const Tables<P> T;
   constexpr int D = P + 1, Q = P + 3, D2 = D * D, D3 = D2 * D, Q2 = Q * Q, Q3 = Q2 * Q;
   const double alpha = exec_mode == 1 ? 1.0 : -1.0, upw = exec_mode == 1 ? 1.0 : -1.0;
   int itmax = 0;
   double b_rhs = 0.0, b_inv = 0.0, b_lo = 0.0, b_fct = 0.0;
   const int nbatch = (ne + VL - 1) / VL;

   // HO part: batches of VL elements, lane l = element min(e0 + l, ne - 1) (lanes past the end repeat the last element
   // and are not stored)
#pragma omp parallel reduction(max : itmax, b_rhs, b_inv)
   {
   std::vector<vd> buf((size_t)81 * 2 + 9 * Q3 + 3 * Q3 + Q3 + 3 * Q3 + Q3 + Q3 + 3 * Q3 + Q3 + Q3 + 9 * D3 + Q2 + Q * D + D3);
   vd *X = buf.data(), *V = X + 81, *Jc = V + 81, *vq = Jc + 9 * Q3, *tmp = vq + 3 * Q3, *Dq = tmp + Q3, *wd = Dq + 3 * Q3,
      *g = wd + Q3, *gr = g + Q3, *wa = gr + 3 * Q3, *wb = wa + Q3, *rhs = wb + Q3, *rg = rhs + D3, *xg = rg + D3, *dd = xg + D3,
      *Ad = dd + D3, *dg = Ad + D3, *t1 = dg + D3, *t2 = t1 + D3, *mm = t2 + D3, *val = mm + D3, *row = val + Q2, *uev = row + Q * D;
   double my_rhs = 0.0, my_inv = 0.0;
#pragma omp for schedule(static)
   for (int b = 0; b < nbatch; b++)
   {
      const double c0 = now();
      int el[VL];
      for (int l = 0; l < VL; l++) { el[l] = std::min(b * VL + l, ne - 1); }
      for (int i = 0; i < 81; i++)
      {
         for (int l = 0; l < VL; l++)
         {
            V[i][l] = vel[(size_t)el[l] * 81 + i];
            X[i][l] = x0[(size_t)el[l] * 81 + i] + (exec_mode == 1 ? t * V[i][l] : 0.0);
         }
      }
      for (int i = 0; i < D3; i++) { for (int l = 0; l < VL; l++) { uev[i][l] = u[(size_t)el[l] * D3 + i]; } }
      // geometry at the quadrature points (remap: moved mesh, remhos.cpp:1598-1608)
      for (int c = 0; c < 3; c++)
      {
         interp3<Q, 3>(T.dL, T.L, T.L, &X[c * 27], &Jc[(c * 3 + 0) * Q3], wa, wb);
         interp3<Q, 3>(T.L, T.dL, T.L, &X[c * 27], &Jc[(c * 3 + 1) * Q3], wa, wb);
         interp3<Q, 3>(T.L, T.L, T.dL, &X[c * 27], &Jc[(c * 3 + 2) * Q3], wa, wb);
         interp3<Q, 3>(T.L, T.L, T.L, &V[c * 27], &vq[c * Q3], wa, wb);
      }
      for (int qz = 0; qz < Q; qz++)
      {
         for (int qy = 0; qy < Q; qy++)
         {
            const double wyz = T.W[qy] * T.W[qz];
            for (int qx = 0; qx < Q; qx++)
            {
               const int q = qx + Q * (qy + Q * qz);
               const vd J00 = Jc[0 * Q3 + q], J01 = Jc[1 * Q3 + q], J02 = Jc[2 * Q3 + q];
               const vd J10 = Jc[3 * Q3 + q], J11 = Jc[4 * Q3 + q], J12 = Jc[5 * Q3 + q];
               const vd J20 = Jc[6 * Q3 + q], J21 = Jc[7 * Q3 + q], J22 = Jc[8 * Q3 + q];
               const vd A11 = J11 * J22 - J12 * J21, A12 = J21 * J02 - J01 * J22, A13 = J01 * J12 - J11 * J02;
               const vd A21 = J20 * J12 - J10 * J22, A22 = J00 * J22 - J02 * J20, A23 = J10 * J02 - J00 * J12;
               const vd A31 = J10 * J21 - J20 * J11, A32 = J20 * J01 - J00 * J21, A33 = J00 * J11 - J01 * J10;
               const vd detJ = J00 * A11 + J01 * A21 + J02 * A31;
               const double w3 = T.W[qx] * wyz;
               const vd v0 = vq[q], v1 = vq[Q3 + q], v2 = vq[2 * Q3 + q];
               Dq[q] = (alpha * w3) * (A11 * v0 + A12 * v1 + A13 * v2);
               Dq[Q3 + q] = (alpha * w3) * (A21 * v0 + A22 * v1 + A23 * v2);
               Dq[2 * Q3 + q] = (alpha * w3) * (A31 * v0 + A32 * v1 + A33 * v2);
               wd[q] = w3 * detJ;
            }
         }
      }
      // K_vol u
      interp3<Q, D>(T.G, T.B, T.B, uev, &gr[0], wa, wb);
      interp3<Q, D>(T.B, T.G, T.B, uev, &gr[Q3], wa, wb);
      interp3<Q, D>(T.B, T.B, T.G, uev, &gr[2 * Q3], wa, wb);
      for (int q = 0; q < Q3; q++) { g[q] = Dq[q] * gr[q] + Dq[Q3 + q] * gr[Q3 + q] + Dq[2 * Q3 + q] * gr[2 * Q3 + q]; }
      test3<D, Q>(T.Bt, T.Bt, T.Bt, g, rhs, wa, wb);
      // faces (own outward normal), sum-factorised per face
      for (int f = 0; f < 6; f++)
      {
         const int c = f >> 1, side = f & 1, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
         const int pw3[3] = {1, 3, 9}, pwD[3] = {1, D, D2};
         const double *un[VL];
         for (int l = 0; l < VL; l++)
         {
            const int nb = face_nbr[(size_t)el[l] * 6 + f];
            un[l] = nb >= 0 ? u + (size_t)nb * D3 : nullptr;
         }
         // jump of the traces, contracted along i1: row[q1][i2]
         for (int i2 = 0; i2 < D; i2++)
         {
            vd jmp[D];
            for (int i1 = 0; i1 < D; i1++)
            {
               const int io = (side ? (D - 1) : 0) * pwD[c] + i1 * pwD[c1] + i2 * pwD[c2];
               const int in = (side ? 0 : (D - 1)) * pwD[c] + i1 * pwD[c1] + i2 * pwD[c2];
               vd nbv;
               for (int l = 0; l < VL; l++) { nbv[l] = un[l] ? un[l][in] : 0.0; }
               jmp[i1] = nbv - uev[io];
            }
            for (int q1 = 0; q1 < Q; q1++)
            {
               vd acc = jmp[0] * 0.0;
               for (int i1 = 0; i1 < D; i1++) { acc += T.B[q1 * D + i1] * jmp[i1]; }
               row[q1 * D + i2] = acc;
            }
         }
         for (int q2 = 0; q2 < Q; q2++)
         {
            for (int q1 = 0; q1 < Q; q1++)
            {
               vd z = X[0] * 0.0;
               vd t1v[3] = {z, z, z}, t2v[3] = {z, z, z}, vf[3] = {z, z, z};
               for (int a2 = 0; a2 < 3; a2++)
               {
                  for (int a1 = 0; a1 < 3; a1++)
                  {
                     const int n = (side ? 2 : 0) * pw3[c] + a1 * pw3[c1] + a2 * pw3[c2];
                     const double w1 = T.dL[q1 * 3 + a1] * T.L[q2 * 3 + a2], w2 = T.L[q1 * 3 + a1] * T.dL[q2 * 3 + a2];
                     const double w0 = T.L[q1 * 3 + a1] * T.L[q2 * 3 + a2];
                     for (int k = 0; k < 3; k++)
                     {
                        t1v[k] += w1 * X[k * 27 + n];
                        t2v[k] += w2 * X[k * 27 + n];
                        vf[k] += w0 * V[k * 27 + n];
                     }
                  }
               }
               const vd nx = t1v[1] * t2v[2] - t1v[2] * t2v[1], ny = t1v[2] * t2v[0] - t1v[0] * t2v[2];
               const vd nz = t1v[0] * t2v[1] - t1v[1] * t2v[0];
               vd vn = vf[0] * nx + vf[1] * ny + vf[2] * nz;
               if (!side) { vn = -vn; }
               const vd sp = vmax0(upw * vn) * (T.W[q1] * T.W[q2]);
               vd jump = z;
               for (int i2 = 0; i2 < D; i2++) { jump += T.B[q2 * D + i2] * row[q1 * D + i2]; }
               val[q1 + Q * q2] = sp * jump;
            }
         }
         // test with the face's Bernstein functions: along q1 first
         for (int q2 = 0; q2 < Q; q2++)
         {
            for (int i1 = 0; i1 < D; i1++)
            {
               vd acc = X[0] * 0.0;
               for (int q1 = 0; q1 < Q; q1++) { acc += T.B[q1 * D + i1] * val[q1 + Q * q2]; }
               row[q2 * D + i1] = acc;
            }
         }
         for (int i2 = 0; i2 < D; i2++)
         {
            for (int i1 = 0; i1 < D; i1++)
            {
               vd acc = X[0] * 0.0;
               for (int q2 = 0; q2 < Q; q2++) { acc += T.B[q2 * D + i2] * row[q2 * D + i1]; }
               rhs[(side ? (D - 1) : 0) * pwD[c] + i1 * pwD[c1] + i2 * pwD[c2]] += acc;
            }
         }
      }
      const double c1 = now();
      my_rhs += c1 - c0;
      // lumped mass M 1 (remhos.cpp:1632) and the local mass solve: Jacobi-PCG in the GL nodal basis
      test3<D, Q>(T.Bt, T.Bt, T.Bt, wd, mm, wa, wb);
      // b_g = Ci^T (x)3 b : out[k] = sum_i Ci[i*D+k] in[i]
      cx<D, D, D2>(T.Cit, rhs, t1);
      cy<D, D, D, D>(T.Cit, t1, t2);
      cz<D, D, D2>(T.Cit, t2, rg);
      test3<D, Q>(T.Bg2t, T.Bg2t, T.Bg2t, wd, dg, wa, wb);
      vd nom = X[0] * 0.0, sum_b = nom, vol = nom;
      for (int i = 0; i < D3; i++)
      {
         xg[i] = nom * 0.0;
         dd[i] = rg[i] / dg[i];
         nom += rg[i] * dd[i];
         sum_b += rg[i];
         vol += mm[i];
      }
      // DGMassInverse's stopping rule (remhos_ho.cpp:79-80): (D^-1 r, r) <= max(rel^2 nom0, abs^2); lanes that are done
      // are frozen (step length 0) while the others of the batch go on
      vd tol = rel_tol * rel_tol * nom;
      tol = vsel(tol > abs_tol * abs_tol, tol, tol * 0.0 + abs_tol * abs_tol);
      int it = 0;
      vm act = nom > tol;
      while (vany(act) && it < 100)
      {
         interp3<Q, D>(T.Bg, T.Bg, T.Bg, dd, tmp, wa, wb);
         for (int q = 0; q < Q3; q++) { tmp[q] *= wd[q]; }
         test3<D, Q>(T.Bgt, T.Bgt, T.Bgt, tmp, Ad, wa, wb);
         vd den = nom * 0.0;
         for (int i = 0; i < D3; i++) { den += dd[i] * Ad[i]; }
         act = act & (den > 0.0);
         const vd al = vsel(act, nom / den, nom * 0.0);
         vd bn = nom * 0.0;
         for (int i = 0; i < D3; i++)
         {
            xg[i] += al * dd[i];
            rg[i] -= al * Ad[i];
            bn += rg[i] * (rg[i] / dg[i]);
         }
         const vd beta = vsel(act, bn / nom, nom * 0.0);
         for (int i = 0; i < D3; i++) { dd[i] = vsel(act, rg[i] / dg[i] + beta * dd[i], dd[i]); }
         nom = vsel(act, bn, nom);
         act = act & (nom > tol);
         it++;
      }
      itmax = std::max(itmax, it);
      if (completion)
      {
         // the completion of the product path (rmh_set_mass_completion): one Jacobi step on the left-over residual ...
         for (int i = 0; i < D3; i++) { xg[i] += rg[i] / dg[i]; }
      }
      // x_b = Ci (x)3 x_g : out[i] = sum_k Ci[i*D+k] in[k]
      cx<D, D, D2>(T.Ci, xg, t1);
      cy<D, D, D, D>(T.Ci, t1, t2);
      cz<D, D, D2>(T.Ci, t2, xg);
      if (completion)
      {
         // ... and the constant mode: the element's mass rate becomes exactly 1^T b
         vd mx = nom * 0.0;
         for (int i = 0; i < D3; i++) { mx += mm[i] * xg[i]; }
         const vd cst = (sum_b - mx) / vol;
         for (int i = 0; i < D3; i++) { xg[i] += cst; }
      }
      vd lo = uev[0], hi = uev[0];
      for (int i = 1; i < D3; i++)
      {
         lo = vsel(uev[i] < lo, uev[i], lo);
         hi = vsel(uev[i] > hi, uev[i], hi);
      }
      for (int l = 0; l < VL && b * VL + l < ne; l++)
      {
         const int e = b * VL + l;
         for (int i = 0; i < D3; i++)
         {
            ws_m[(size_t)e * D3 + i] = mm[i][l];
            ws_duho[(size_t)e * D3 + i] = xg[i][l];
         }
         ws_xe[e] = lo[l];
         ws_xe[ne + e] = hi[l];
      }
      my_inv += now() - c1;
   }
   b_rhs = my_rhs;
   b_inv = my_inv;
   } // omp parallel

   // LimitMult: MassBasedAvg + overlap bounds + ClipScale
#pragma omp parallel reduction(max : b_lo, b_fct)
   {
   double fc[D3], dl[D3];
   double my_lo = 0.0, my_fct = 0.0;
#pragma omp for schedule(static)
   for (int e = 0; e < ne; e++)
   {
      const double c0 = now();
      const double *ue = u + (size_t)e * D3, *m = ws_m + (size_t)e * D3, *dh = ws_duho + (size_t)e * D3;
      double mass = 0.0, vol = 0.0;
      for (int i = 0; i < D3; i++)
      {
         mass += m[i] * (ue[i] + dt * dh[i]);
         vol += m[i];
      }
      const double ubar = mass / vol;
      for (int i = 0; i < D3; i++) { dl[i] = (ubar - ue[i]) / dt; }
      const double c1 = now();
      my_lo += c1 - c0;
      double smin[27], smax[27];
      for (int s = 0; s < 27; s++)
      {
         const int nb = stencil27[(size_t)e * 27 + s];
         smin[s] = nb >= 0 ? ws_xe[nb] : INFINITY;
         smax[s] = nb >= 0 ? ws_xe[ne + nb] : -INFINITY;
      }
      // per-dof bounds = extrema over the elements that share the dof's CG node: a dof on the low / high layer of a
      // direction also sees the neighbour on that side.  27 boxes per element (low layer, interior, high layer per
      // direction), then one look-up per dof.
      double bmin[27], bmax[27];
      for (int b3 = 0; b3 < 27; b3++)
      {
         const int bx = b3 % 3, by = (b3 / 3) % 3, bz = b3 / 9;
         double lo = INFINITY, hi = -INFINITY;
         for (int oz = (bz == 0 ? -1 : 0); oz <= (bz == 2 ? 1 : 0); oz++)
         {
            for (int oy = (by == 0 ? -1 : 0); oy <= (by == 2 ? 1 : 0); oy++)
            {
               for (int ox = (bx == 0 ? -1 : 0); ox <= (bx == 2 ? 1 : 0); ox++)
               {
                  const int s = (ox + 1) + 3 * (oy + 1) + 9 * (oz + 1);
                  lo = std::fmin(lo, smin[s]);
                  hi = std::fmax(hi, smax[s]);
               }
            }
         }
         bmin[b3] = lo;
         bmax[b3] = hi;
      }
      double sumPos = 0.0, sumNeg = 0.0;
      const double eps = 1.0e-15;
      for (int i = 0; i < D3; i++)
      {
         const int ix = i % D, iy = (i / D) % D, iz = i / D2;
         const int b3 = (ix == 0 ? 0 : (ix == D - 1 ? 2 : 1)) + 3 * (iy == 0 ? 0 : (iy == D - 1 ? 2 : 1)) + 9 * (iz == 0 ? 0 : (iz == D - 1 ? 2 : 1));
         const double umin = bmin[b3], umax = bmax[b3];
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
      my_fct += now() - c1;
   }
   b_lo = my_lo;
   b_fct = my_fct;
   } // omp parallel
   g_bucket[0] += b_rhs;
   g_bucket[1] += b_inv;
   g_bucket[2] += b_lo;
   g_bucket[3] += b_fct;
   return itmax;
}

template <int P>
void lumped_mass_t(int ne, int exec_mode, const double *x0, const double *vel, double t, double *m)
{
   static const Tables<P> T;
   constexpr int D = P + 1, Q = P + 3, D3 = D * D * D, Q2 = Q * Q, Q3 = Q2 * Q;
#pragma omp parallel
   {
   alignas(64) double X[81], Jc[9 * Q3], wd[Q3], wa[Q3], wb[Q3];
#pragma omp for schedule(static)
   for (int e = 0; e < ne; e++)
   {
      for (int i = 0; i < 81; i++) { X[i] = x0[(size_t)e * 81 + i] + (exec_mode == 1 ? t * vel[(size_t)e * 81 + i] : 0.0); }
      for (int c = 0; c < 3; c++)
      {
         interp3<Q, 3>(T.dL, T.L, T.L, &X[c * 27], &Jc[(c * 3 + 0) * Q3], wa, wb);
         interp3<Q, 3>(T.L, T.dL, T.L, &X[c * 27], &Jc[(c * 3 + 1) * Q3], wa, wb);
         interp3<Q, 3>(T.L, T.L, T.dL, &X[c * 27], &Jc[(c * 3 + 2) * Q3], wa, wb);
      }
      for (int q = 0; q < Q3; q++)
      {
         const double J00 = Jc[0 * Q3 + q], J01 = Jc[1 * Q3 + q], J02 = Jc[2 * Q3 + q];
         const double J10 = Jc[3 * Q3 + q], J11 = Jc[4 * Q3 + q], J12 = Jc[5 * Q3 + q];
         const double J20 = Jc[6 * Q3 + q], J21 = Jc[7 * Q3 + q], J22 = Jc[8 * Q3 + q];
         const double detJ = J00 * (J11 * J22 - J12 * J21) + J01 * (J20 * J12 - J10 * J22) + J02 * (J10 * J21 - J20 * J11);
         wd[q] = T.W[q % Q] * T.W[(q / Q) % Q] * T.W[q / Q2] * detJ;
      }
      test3<D, Q>(T.Bt, T.Bt, T.Bt, wd, &m[(size_t)e * D3], wa, wb);
   }
   }
}

#define CPU_DISPATCH(p, expr)                   \
   switch (p)                                   \
   {                                            \
      case 1: { constexpr int P = 1; expr; break; } \
      case 2: { constexpr int P = 2; expr; break; } \
      case 3: { constexpr int P = 3; expr; break; } \
      case 4: { constexpr int P = 4; expr; break; } \
      case 5: { constexpr int P = 5; expr; break; } \
      default: { constexpr int P = 6; expr; break; } \
   }

// NUMA placement: the caller's arrays (numpy) were touched first by one thread, i.e. they live on one memory node, and the
// 128 threads of a two-socket host would all read the mesh from there.  cpu_place() copies an input array once into a
// buffer whose pages are first touched by the threads that will read them (the same static partition of the elements
// as the stage loops); the copy is reused as long as the caller passes the same pointer and size.
struct Placed
{
   const void *src = nullptr;
   size_t bytes = 0;
   void *buf = nullptr;
};
template <class T>
const T *cpu_place(Placed &pl, const T *src, size_t per_elem, int ne)
{
   const size_t bytes = sizeof(T) * per_elem * (size_t)ne;
   if (pl.src == src && pl.bytes == bytes) { return (const T *)pl.buf; }
   std::free(pl.buf);
   pl.buf = std::malloc(bytes);
   pl.src = src;
   pl.bytes = bytes;
   T *dst = (T *)pl.buf;
#pragma omp parallel for schedule(static)
   for (int e = 0; e < ne; e++) { std::memcpy(dst + per_elem * (size_t)e, src + per_elem * (size_t)e, sizeof(T) * per_elem); }
   return dst;
}
Placed g_px0, g_pvel, g_pnbr, g_pst;

} // namespace

extern "C" {

// One stage: du = F(u, t) with dt the full step (LO/FCT).  Arrays as in include/rmh.h (host memory).
// ws_m receives the lumped mass.  Returns the max PCG iteration count.
// Local mass solve: PCG stopped at (D^-1 r, r) <= max(rel_tol^2 nom0, abs_tol^2) like DGMassInverse (remhos_ho.cpp:79-80);
// completion != 0 adds the product path's Jacobi step + constant mode (rmh_set_mass_completion).
int cpu_stage(int p, int ne, int exec_mode, const double *x0, const double *vel, const int *face_nbr,
              const int *stencil27, double t, double dt, const double *u, double *du, double *ws_m,
              double *ws_duho, double *ws_xe /* [2*ne] */, double rel_tol, double abs_tol, int completion)
{
   int it = 0;
   CPU_DISPATCH(p, it = stage_t<P>(ne, exec_mode, x0, vel, face_nbr, stencil27, t, dt, u, du, ws_m, ws_duho, ws_xe, rel_tol, abs_tol,
                                   completion));
   return it;
}

// RK3-SSP step in place on u (size ne*ndof); work arrays are allocated by the caller (4 * ne*ndof + 2*ne).
int cpu_rk3_step(int p, int ne, int exec_mode, const double *x0, const double *vel, const int *face_nbr,
                 const int *stencil27, double t, double dt, double *u_caller, double *work, double rel_tol, double abs_tol, int completion)
{
   const size_t n = (size_t)ne * (p + 1) * (p + 1) * (p + 1);
   x0 = cpu_place(g_px0, x0, 81, ne);
   vel = cpu_place(g_pvel, vel, 81, ne);
   face_nbr = cpu_place(g_pnbr, face_nbr, 6, ne);
   stencil27 = cpu_place(g_pst, stencil27, 27, ne);
   double *u = u_caller; // (the state stays where the caller has it: a placed copy per step cost more than it gained)
   double *y = work, *k = work + n, *m = work + 2 * n, *dh = work + 3 * n, *xe = work + 4 * n;
   int it = cpu_stage(p, ne, exec_mode, x0, vel, face_nbr, stencil27, t, dt, u, k, m, dh, xe, rel_tol, abs_tol, completion);
#pragma omp parallel for
   for (long long i = 0; i < (long long)n; i++) { y[i] = u[i] + dt * k[i]; }
   it = std::max(it, cpu_stage(p, ne, exec_mode, x0, vel, face_nbr, stencil27, t + dt, dt, y, k, m, dh, xe, rel_tol, abs_tol, completion));
#pragma omp parallel for
   for (long long i = 0; i < (long long)n; i++) { y[i] = 0.75 * u[i] + 0.25 * (y[i] + dt * k[i]); }
   it = std::max(it, cpu_stage(p, ne, exec_mode, x0, vel, face_nbr, stencil27, t + dt / 2, dt, y, k, m, dh, xe, rel_tol, abs_tol, completion));
#pragma omp parallel for
   for (long long i = 0; i < (long long)n; i++) { u_caller[i] = (1.0 / 3.0) * u[i] + (2.0 / 3.0) * (y[i] + dt * k[i]); }
   return it;
}

// lumped mass at pseudo-time t (initial / final mass)
void cpu_lumped_mass(int p, int ne, int exec_mode, const double *x0, const double *vel, double t, double *m)
{
   CPU_DISPATCH(p, lumped_mass_t<P>(ne, exec_mode, x0, vel, t, m));
}

// TimingData-style buckets accumulated since the last reset: seconds in RHS, INV, LO, FCT (max over threads per stage)
void cpu_buckets(double t[4], int reset)
{
   for (int k = 0; k < 4; k++)
   {
      t[k] = g_bucket[k];
      if (reset) { g_bucket[k] = 0.0; }
   }
}

int cpu_simd_width(void) { return VL; } // elements per batch of the HO loop

// (the OpenMP runtime may have been started by another library of the process before OMP_NUM_THREADS could be set)
void cpu_set_threads(int n)
{
#ifdef _OPENMP
   if (n > 0) { omp_set_num_threads(n); }
#else
   (void)n;
#endif
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
