// Host-side 1-D basis / quadrature tables for the hot path.
//
// MFEM supplies these to the reference through DofToQuad / IntRules;
// here they are generated directly:
//   * Gauss-Legendre rule with Q = p + 3 points on [0,1]  (order 2p + mo*dim - 1 with mo = 2,
//     dim = 3: MassIntegrator / ConvectionIntegrator / DGTraceIntegrator rules);
//   * Bernstein basis of degree p (DG_FECollection(order, dim, BasisType::Positive),
//     remhos.cpp:588-590) and its derivative at those points;
//   * quadratic Lagrange basis on {0, 1/2, 1} (Gauss-Lobatto mesh nodes of order 2,
//     remhos.cpp:513-527);
//   * the Gauss-Legendre nodal basis used by the element-local mass solve
//     (DGMassInverse(fes, BasisType::GaussLegendre), remhos_ho.cpp:79) and the 1-D change of
//     basis Bernstein <-> Gauss-Legendre nodal.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace rmh
{

// table layout shared by host and device code (offsets in doubles); NQ Gauss-Legendre points per direction
template <int P, int NQ>
struct TabLayoutQ
{
   static constexpr int D = P + 1, Q = NQ;
   static constexpr int oB = 0;             // B[q*D+i]   Bernstein values
   static constexpr int oG = oB + Q * D;    // G[q*D+i]   Bernstein derivatives
   static constexpr int oL = oG + Q * D;    // L[q*3+a]   mesh Lagrange values
   static constexpr int odL = oL + Q * 3;   // dL[q*3+a]  mesh Lagrange derivatives
   static constexpr int oW = odL + Q * 3;   // w[q]       quadrature weights
   static constexpr int oBg = oW + Q;       // Bg[q*D+k]  GL nodal basis at quadrature points
   static constexpr int oBg2 = oBg + Q * D; // Bg2 = Bg^2 (Jacobi diagonal)
   static constexpr int oBgE = oBg2 + Q * D; // BgE[side*D+k] GL nodal basis at xi = 0 / 1
   // Integer tables of the fused limiter (dim = 3), packed into doubles so that they travel with the LDS copy -- index arithmetic a
   // dof thread otherwise repeats per round (divisions by D, compare / select chains: ~15-25 VALU instructions each) becomes
   // one LDS byte read:
   //   BoxQ[s3]  the 8 corners of dof class s3's box in the 3 x 3 x 3 stencil (remhos_tools.cpp:432-495: offsets {-1, 0} on a low
   //             layer, {0} inside, {0, +1} on a high layer, per direction), as byte offsets 8 q of the stencil entries -- 8 bytes
   //   Cls[i]    the class s3 = cx + 3 cy + 9 cz of dof i (c = 0 low face layer, 1 interior, 2 high face layer) -- one byte per dof
   // (p = 3 +0.4 %, p = 4 +1.1 %, p = 6 +1.2 %; RMH_ITAB5: see N2S)
#ifndef RMH_ITAB5
#define RMH_ITAB5 1
#endif
   static constexpr bool ITAB = (P != 5) || RMH_ITAB5;
   static constexpr int oBoxQ = oBgE + 2 * D;
   static constexpr int oCls = oBoxQ + (ITAB ? 27 : 0);
   // N2S: what the stage kernels (ho_kernel2) copy to LDS -- the entries they index by LANE.  The change-of-basis tables and the
   // closed-uniform Lagrange rows behind it are only read through scalar loads there (round 6: out of the copy -- at p = 5 the
   // 720 bytes are the difference between 10 and 11 resident workgroups per CU).
   static constexpr int N2S = oCls + (ITAB ? (D * D * D + 7) / 8 : 0);
   static constexpr int oCi = N2S;            // Ci[i*D+k]  inverse of C[k][i] = Bernstein_i(gl node k)
   static constexpr int oLcu = oCi + D * D;   // Lcu[i*3+a]  mesh Lagrange basis at the closed-uniform points i/p
   static constexpr int oCf = oLcu + 3 * D;   // Cf[i*D+k] = C[k][i]: GL-tested -> Bernstein-tested moments (phi^B_i = sum_k C[k][i] l_k)
   static constexpr int N2 = oCf + D * D;     // extended table (what the other kernels copy to LDS)
   // Transposed copies for the contractions that form ONE output per table COLUMN (y-leg of the test tensors, y-back leg of the
   // mass apply, q1-contraction of the face rows: sum over q of T[q*D+k] x[q] for fixed k).  With the q-major table each output
   // gathers Q separate entries through scalar loads -- 2 x 4 s_load_dwordx2 and two exposed waits per output at p = 5; from
   // the transposed rows it is one or two wide loads.  Constant-memory only: not part of the LDS copy.
   static constexpr int oBgT = N2;            // BgT[k*Q+q]  = Bg[q*D+k]
   static constexpr int oBT = oBgT + D * Q;   // BT[i*Q+q]   = B[q*D+i]
   static constexpr int oBg2T = oBT + D * Q;  // Bg2T[k*Q+q] = Bg2[q*D+k]
   // Trace table (dim = 3; global copy only, read per LANE in the load phase): for entry r6 = f D^2 + i1 + D i2 of an element's six
   // face layers, packed into 32 bits: the own face dof's offset in the element | << 10 the offset of the same dof in the face
   // neighbour's OPPOSITE layer | << 20 the dof's index in the layer.  What the load phase otherwise takes apart per entry -- face,
   // side, axis strides, two mixed-radix offsets: ~25 integer instructions, three entries per lane.
   static constexpr int oTr = oBg2T + D * Q;   // 6 D^2 ints = 3 D^2 doubles
   static constexpr int N3 = oTr + 3 * D * D;  // everything in constant memory / the global copy
};
// dim = 3: Q = p + 3 (order 2p + 2*3 - 1); dim = 2: Q = p + 2 (order 2p + 2*2 - 1) -- SURVEY A.2
template <int P>
struct TabLayout : TabLayoutQ<P, P + 3>
{
};
template <int P>
struct TabLayout2 : TabLayoutQ<P, P + 2>
{
};

inline void gauss_legendre_01(int n, std::vector<double> &x, std::vector<double> &w)
{
   x.assign(n, 0.0);
   w.assign(n, 0.0);
   for (int i = 0; i < n; i++)
   {
      // Newton iteration on P_n, start from the Chebyshev guess
      double z = std::cos(M_PI * (i + 0.75) / (n + 0.5));
      double pp = 0.0;
      for (int it = 0; it < 100; it++)
      {
         double p1 = 1.0, p2 = 0.0;
         for (int j = 0; j < n; j++)
         {
            const double p3 = p2;
            p2 = p1;
            p1 = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
         }
         pp = n * (z * p1 - p2) / (z * z - 1.0);
         const double dz = p1 / pp;
         z -= dz;
         if (std::fabs(dz) < 1e-16) { break; }
      }
      // ascending order on [0,1]
      x[n - 1 - i] = 0.5 * (1.0 + z);
      w[n - 1 - i] = 1.0 / ((1.0 - z * z) * pp * pp);
   }
}

inline double binom(int n, int k)
{
   double r = 1.0;
   for (int i = 1; i <= k; i++) { r = r * (n - k + i) / i; }
   return r;
}

inline double bernstein(int p, int i, double t)
{
   if (i < 0 || i > p) { return 0.0; }
   return binom(p, i) * std::pow(t, i) * std::pow(1.0 - t, p - i);
}

inline double dbernstein(int p, int i, double t)
{
   if (p == 0) { return 0.0; }
   return p * (bernstein(p - 1, i - 1, t) - bernstein(p - 1, i, t));
}

inline void lagrange(const std::vector<double> &nodes, double t, std::vector<double> &L,
                     std::vector<double> &dL)
{
   const int n = (int)nodes.size();
   L.assign(n, 1.0);
   dL.assign(n, 0.0);
   for (int j = 0; j < n; j++)
   {
      for (int m = 0; m < n; m++)
      {
         if (m != j) { L[j] *= (t - nodes[m]) / (nodes[j] - nodes[m]); }
      }
      for (int k = 0; k < n; k++)
      {
         if (k == j) { continue; }
         double term = 1.0 / (nodes[j] - nodes[k]);
         for (int m = 0; m < n; m++)
         {
            if (m != j && m != k) { term *= (t - nodes[m]) / (nodes[j] - nodes[m]); }
         }
         dL[j] += term;
      }
   }
}

// dense inverse by Gauss-Jordan with partial pivoting (tiny D x D matrices)
inline void invert(int n, std::vector<double> &a)
{
   std::vector<double> inv(n * n, 0.0);
   for (int i = 0; i < n; i++) { inv[i * n + i] = 1.0; }
   for (int c = 0; c < n; c++)
   {
      int piv = c;
      for (int r = c + 1; r < n; r++)
      {
         if (std::fabs(a[r * n + c]) > std::fabs(a[piv * n + c])) { piv = r; }
      }
      for (int k = 0; k < n; k++)
      {
         std::swap(a[c * n + k], a[piv * n + k]);
         std::swap(inv[c * n + k], inv[piv * n + k]);
      }
      const double d = 1.0 / a[c * n + c];
      for (int k = 0; k < n; k++) { a[c * n + k] *= d; inv[c * n + k] *= d; }
      for (int r = 0; r < n; r++)
      {
         if (r == c) { continue; }
         const double f = a[r * n + c];
         for (int k = 0; k < n; k++)
         {
            a[r * n + k] -= f * a[c * n + k];
            inv[r * n + k] -= f * inv[c * n + k];
         }
      }
   }
   a = inv;
}

template <int P, int NQ>
inline std::vector<double> make_tables_q()
{
   using T = TabLayoutQ<P, NQ>;
   constexpr int D = T::D, Q = T::Q;
   std::vector<double> tab(T::N3, 0.0);
   std::vector<double> xq, wq, xg, wg;
   gauss_legendre_01(Q, xq, wq);
   gauss_legendre_01(D, xg, wg);
   const std::vector<double> mesh_nodes = {0.0, 0.5, 1.0};
   std::vector<double> L, dL;
   for (int q = 0; q < Q; q++)
   {
      for (int i = 0; i < D; i++)
      {
         tab[T::oB + q * D + i] = bernstein(P, i, xq[q]);
         tab[T::oG + q * D + i] = dbernstein(P, i, xq[q]);
      }
      lagrange(mesh_nodes, xq[q], L, dL);
      for (int a = 0; a < 3; a++)
      {
         tab[T::oL + q * 3 + a] = L[a];
         tab[T::odL + q * 3 + a] = dL[a];
      }
      tab[T::oW + q] = wq[q];
      lagrange(xg, xq[q], L, dL);
      for (int k = 0; k < D; k++)
      {
         tab[T::oBg + q * D + k] = L[k];
         tab[T::oBg2 + q * D + k] = L[k] * L[k];
      }
   }
   for (int side = 0; side < 2; side++)
   {
      lagrange(xg, (double)side, L, dL);
      for (int k = 0; k < D; k++) { tab[T::oBgE + side * D + k] = L[k]; }
   }
   for (int i = 0; i < D; i++)
   {
      lagrange(mesh_nodes, (double)i / P, L, dL);
      for (int a = 0; a < 3; a++) { tab[T::oLcu + i * 3 + a] = L[a]; }
   }
   // C[k][i] = Bernstein_i(x_gl[k]);  Ci = C^-1 stored as Ci[i*D+k]
   std::vector<double> C(D * D);
   for (int k = 0; k < D; k++)
   {
      for (int i = 0; i < D; i++) { C[k * D + i] = bernstein(P, i, xg[k]); }
   }
   for (int i = 0; i < D; i++)
   {
      for (int k = 0; k < D; k++) { tab[T::oCf + i * D + k] = C[k * D + i]; }
   }
   invert(D, C);
   for (int i = 0; i < D; i++)
   {
      for (int k = 0; k < D; k++) { tab[T::oCi + i * D + k] = C[i * D + k]; }
   }
   if (T::ITAB)
   {
      unsigned char box[27][8], cls[((D * D * D + 7) / 8) * 8] = {};
      for (int s3 = 0; s3 < 27; s3++)
      {
         const int sc[3] = {s3 % 3, (s3 / 3) % 3, s3 / 9};
         int lo[3], hi[3];
         for (int c = 0; c < 3; c++) { lo[c] = (sc[c] == 0) ? 0 : 1; hi[c] = (sc[c] == 2) ? 2 : 1; }
         for (int c = 0; c < 8; c++)
         {
            const int q = ((c & 1) ? hi[0] : lo[0]) + 3 * ((c & 2) ? hi[1] : lo[1]) + 9 * ((c & 4) ? hi[2] : lo[2]);
            box[s3][c] = (unsigned char)(8 * q);
         }
         std::memcpy(&tab[T::oBoxQ + s3], box[s3], 8);
      }
      for (int i = 0; i < D * D * D; i++)
      {
         const int b[3] = {i % D, (i / D) % D, i / (D * D)};
         int s3 = 0, w = 1;
         for (int c = 0; c < 3; c++, w *= 3) { s3 += w * (b[c] == 0 ? 0 : (b[c] == P ? 2 : 1)); }
         cls[i] = (unsigned char)s3;
      }
      std::memcpy(&tab[T::oCls], cls, sizeof(cls));
   }
   {
      std::vector<unsigned> tr(6 * D * D + 1, 0u);
      const int st[3] = {1, D, D * D};
      for (int r6 = 0; r6 < 6 * D * D; r6++)
      {
         const int f = r6 / (D * D), r = r6 % (D * D), i1 = r % D, i2 = r / D, c = f >> 1, side = f & 1;
         const int strc = st[c], str1 = st[(c + 1) % 3], str2 = st[(c + 2) % 3];
         const unsigned own = (unsigned)((side ? P * strc : 0) + i1 * str1 + i2 * str2);
         const unsigned nbr = (unsigned)((side ? 0 : P) * strc + i1 * str1 + i2 * str2);
         tr[r6] = own | nbr << 10 | (unsigned)r << 20;
      }
      std::memcpy(&tab[T::oTr], tr.data(), sizeof(unsigned) * 6 * D * D);
   }
   for (int q = 0; q < Q; q++)
   {
      for (int k = 0; k < D; k++)
      {
         tab[T::oBgT + k * Q + q] = tab[T::oBg + q * D + k];
         tab[T::oBT + k * Q + q] = tab[T::oB + q * D + k];
         tab[T::oBg2T + k * Q + q] = tab[T::oBg2 + q * D + k];
      }
   }
   return tab;
}

template <int P>
inline std::vector<double> make_tables() { return make_tables_q<P, P + 3>(); }

} // namespace rmh
