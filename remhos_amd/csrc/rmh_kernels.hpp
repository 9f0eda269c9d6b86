// HIP kernels of the Remhos RK stage for gfx950 (FP64 throughout).
//
// Work decomposition: ONE ELEMENT PER WORKGROUP of NT = 64 threads (one wavefront) for
// p <= 5 and 128 threads (two wavefronts) for p = 6.  All of an element's tensors live in
// LDS / registers between the coalesced E-vector load of u and the store of du; the
// quadrature data ("PA data" of the reference: mass w*detJ, convection w*adj(J)*v, face
// v.n) is recomputed from the element's 27 Q2 nodes and never touches HBM.
//
// Thread roles inside an element
//   "column" role : thread (qx, qy) owns the quadrature column {(qx, qy, qz)}, qz in registers
//   "dof" role    : thread i owns dof(s) i, i + NT, ...
//   "task" role   : flat loops over small intermediate tensors
//
// Reference semantics per kernel are cited at each kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "rmh_tables.hpp"

namespace rmh
{

template <int P>
struct KCfg : TabLayout<P>
{
   using T = TabLayout<P>;
   static constexpr int D = T::D, Q = T::Q;
   static constexpr int D2 = D * D, D3 = D * D * D, Q2 = Q * Q, Q3 = Q * Q * Q;
   static constexpr int NT = (Q2 <= 64) ? 64 : 128;
   static constexpr int NW = NT / 64;
   static constexpr int DPT = (D3 + NT - 1) / NT; // dofs per thread
};

// 1-D tables of every order in constant memory: compile-time indexed reads become scalar loads
// (s_load) everywhere, including after barriers where the compiler no longer treats loads from
// ordinary global memory as invariant.  Filled by rmh_create (identical for all contexts).
constexpr int RMH_TAB_STRIDE = 448;
__constant__ double c_tab[7][RMH_TAB_STRIDE];

struct HoArgs
{
   const double *u;        // [ne_owned][D3]
   const double *u_ghost;  // [ne_ghost][gh_ustride] or null (gh_ustride = D3, or D3 + 2 for ghost records)
   const double *x0;       // [ne][3][27]
   const double *vel;      // [ne][3][27]
   const int *face_nbr;    // [ne][6]
   const double *tab;      // TabLayout<P>
   const double *subvel;   // [ne][3][D3] sub-mesh node velocity (lo 4) or null
   const double *subx0;    // [ne][3][D3] sub-mesh start positions (set up once by subcell_setup_kernel)
   const double *subvmid;  // [ne][3][P^3] subcell midpoint velocity = mean of the 8 corner values
   double *du;             // [ne][D3]
   double *m;              // [ne][D3] lumped mass
   double *xe_min, *xe_max; // [ne]
   int *cg_iters;          // [1] max over elements (atomicMax)
   int ne_owned;
   int e_begin, e_end;     // elements this launch works on (ho_kernel2; the whole rank: 0, ne_owned)
   double t;               // pseudo-time (remap) -- X = x0 + t*vel
   int move;               // 1: remap (mesh moves), 0: transport (static mesh)
   double alpha;           // ConvectionIntegrator alpha: -1 transport, +1 remap (remhos.cpp:648-657)
   double upw;             // upwind sign: s = max(0, upw * v.n_out): -1 transport, +1 remap
   double rel2, abs2;      // squared tolerances of the local PCG
   int max_iter;
   // fused stage (ho_kernel2<P, true>): LimitMult for -lo 5 -fct 2 and the RK update in the same kernel
   const int *stencil27;            // [ne][27]
   const double *gh_min, *gh_max;   // ghost element extrema, element g at [g * gh_mstride]
   int gh_ustride, gh_mstride;
   int gh_compact;                  // 1: a ghost's trace record holds only the face layer facing this rank, [i1 + D*i2]
   double dt;                       // full time step (LO / FCT)
   const double *x_base;            // y_out = rk_a * x_base + rk_b * (u + dt_rk * du)
   double rk_a, rk_b, dt_rk;
   double *y_out;
   double *xe_min_out, *xe_max_out; // extrema of y_out (input extrema of the next stage)
   int bounds_type;                 // 0 overlap bounds, 1 face-neighbour bounds
   int rd_subcell;                  // RD solver: 1 subcell fluctuations (lo 4), 0 plain PAResidualDistribution (lo 3)
   double *dt_est;                  // null, or the running minimum of UpdateTimeStepEstimate
};

// ---------------------------------------------------------------------------------------
// block-wide reductions (one or two wavefronts)
// ---------------------------------------------------------------------------------------
template <int NW>
__device__ inline double block_sum(double v, double *s_red)
{
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off); }
   if (NW == 1) { return v; }
   __syncthreads();
   if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = v; }
   __syncthreads();
   double r = 0.0;
#pragma unroll
   for (int k = 0; k < NW; k++) { r += s_red[k]; }
   return r;
}

template <int NW>
__device__ inline double block_min(double v, double *s_red)
{
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) { v = fmin(v, __shfl_xor(v, off)); }
   if (NW == 1) { return v; }
   __syncthreads();
   if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = v; }
   __syncthreads();
   double r = s_red[0];
#pragma unroll
   for (int k = 1; k < NW; k++) { r = fmin(r, s_red[k]); }
   return r;
}

template <int NW>
__device__ inline double block_max(double v, double *s_red)
{
#pragma unroll
   for (int off = 32; off > 0; off >>= 1) { v = fmax(v, __shfl_xor(v, off)); }
   if (NW == 1) { return v; }
   __syncthreads();
   if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = v; }
   __syncthreads();
   double r = s_red[0];
#pragma unroll
   for (int k = 1; k < NW; k++) { r = fmax(r, s_red[k]); }
   return r;
}

// out[.., k_dir, ..] = sum_i M(i,k) in[.., i_dir, ..]  over a D x D x D tensor, direction `dir`.
// TRANS = false : M(i,k) = mat[k*D + i]   (out_k = sum_i mat[k][i] in_i)
// TRANS = true  : M(i,k) = mat[i*D + k]   (out_k = sum_i mat[i][k] in_i)
template <int D, int NT, bool TRANS>
__device__ inline void dof_tensor_apply(const double *mat, int dir, const double *in, double *out)
{
   constexpr int D3 = D * D * D;
   const int stride = (dir == 0) ? 1 : (dir == 1 ? D : D * D);
   for (int o = threadIdx.x; o < D3; o += NT)
   {
      const int k = (o / stride) % D;
      const int base = o - k * stride;
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < D; i++)
      {
         const double mv = TRANS ? mat[i * D + k] : mat[k * D + i];
         acc += mv * in[base + i * stride];
      }
      out[o] = acc;
   }
}

// ---------------------------------------------------------------------------------------
// HO kernel: du = M^-1 (K_vol + K_face) u, plus lumped mass and element extrema.
//
//   K_vol  : ConvectionIntegrator PA apply      (remhos.cpp:646-657; qdata as in
//            remhos_lo.cpp:1155-1190, apply as in remhos_lo.cpp:1473-1612)
//   K_face : transposed DGTraceIntegrator, upwind (remhos.cpp:659-678)
//   M^-1   : element-local PCG in the Gauss-Legendre nodal basis with Jacobi preconditioner
//            (DGMassInverse, remhos_ho.cpp:79-80,126)
//   lumped mass M_HO * 1 (remhos.cpp:1632); element min/max (remhos_tools.cpp:497-523)
//
// MODE 0: everything.  MODE 1: lumped mass only (rmh_compute_lumped_mass).
// ---------------------------------------------------------------------------------------
template <int P, int MODE>
__global__ void __launch_bounds__(KCfg<P>::NT) ho_kernel(HoArgs a)
{
   using C = KCfg<P>;
   constexpr int D = C::D, Q = C::Q, D2 = C::D2, D3 = C::D3, Q2 = C::Q2, NT = C::NT, NW = C::NW;

   __shared__ double tab[C::N2];
   __shared__ double sX[81], sV[81];
   __shared__ double su[D3];
   __shared__ double sNb[6 * D2];
   __shared__ double sT1[9 * Q * 9];
   __shared__ double sU1[2 * Q * D2];
   __shared__ double sR3[3 * Q2 * D];
   __shared__ double sR2[3 * Q * D2];
   __shared__ double sF[6 * Q2];
   __shared__ double s_rhs[D3], s_m[D3], s_dg[D3], s_x[D3], s_r[D3], s_d[D3], s_Ad[D3];
   __shared__ double s_red[4];

   const int tid = threadIdx.x;
   const int e = blockIdx.x;
   const double *tB = tab + C::oB, *tG = tab + C::oG, *tL = tab + C::oL, *tdL = tab + C::odL;
   const double *tW = tab + C::oW, *tBg = tab + C::oBg, *tBg2 = tab + C::oBg2, *tCi = tab + C::oCi;

   // ---- phase A: coalesced loads ------------------------------------------------------
   for (int i = tid; i < C::N2; i += NT) { tab[i] = a.tab[i]; }
   for (int i = tid; i < 81; i += NT)
   {
      const double x0 = a.x0[(size_t)e * 81 + i];
      const double v = a.vel[(size_t)e * 81 + i];
      sV[i] = v;
      sX[i] = a.move ? x0 + a.t * v : x0;
   }
   if (MODE != 1)
   {
      for (int i = tid; i < D3; i += NT) { su[i] = a.u[(size_t)e * D3 + i]; }
      // neighbour traces: the D^2 face-layer Bernstein coefficients of the neighbour
      for (int k = tid; k < 6 * D2; k += NT)
      {
         const int f = k / D2, r = k % D2;
         const int i1 = r % D, i2 = r / D;
         const int c = f >> 1, side = f & 1;
         const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
         const int strc = (c == 0) ? 1 : (c == 1 ? D : D2);
         const int str1 = (c1 == 0) ? 1 : (c1 == 1 ? D : D2);
         const int str2 = (c2 == 0) ? 1 : (c2 == 1 ? D : D2);
         const int nb = a.face_nbr[(size_t)e * 6 + f];
         double val = 0.0; // boundary: u_nbr = 0 (no inflow data enters the HO path)
         if (nb >= 0)
         {
            const double *un = (nb < a.ne_owned) ? a.u + (size_t)nb * D3
                               : a.u_ghost + (size_t)(nb - a.ne_owned) * a.gh_ustride;
            const int ic = side ? 0 : P; // the neighbour's opposite face layer
            // (compact ghost records hold exactly that layer, ordered like this face: rmh_exchange_setup)
            val = (a.gh_compact && nb >= a.ne_owned) ? un[r] : un[ic * strc + i1 * str1 + i2 * str2];
         }
         sNb[k] = val;
      }
   }
   __syncthreads();

   // ---- phase B: geometry, contraction along x ------------------------------------------
   // sT1[(arr*Q + qx)*9 + ay + 3*az], arr = 3*comp + kind, kind 0: L.X, 1: dL.X, 2: L.V
   for (int k = tid; k < 9 * Q * 9; k += NT)
   {
      const int arr = k / (Q * 9), r = k % (Q * 9);
      const int qx = r / 9, n2 = r % 9;
      const int comp = arr / 3, kind = arr % 3;
      const double *src = (kind == 2 ? sV : sX) + comp * 27 + 3 * n2;
      const double *w = (kind == 1 ? tdL : tL) + qx * 3;
      sT1[k] = w[0] * src[0] + w[1] * src[1] + w[2] * src[2];
   }
   __syncthreads();

   // ---- phase C: column threads finish the geometry in registers --------------------------
   const bool col = tid < Q2;
   const int qx = tid % Q, qy = (tid / Q) % Q;
   double Dq[3][Q], wd[Q];
   if (col)
   {
      double A[3][4][3];
#pragma unroll
      for (int comp = 0; comp < 3; comp++)
      {
#pragma unroll
         for (int az = 0; az < 3; az++)
         {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int ay = 0; ay < 3; ay++)
            {
               const double Ly = tL[qy * 3 + ay], dLy = tdL[qy * 3 + ay];
               const int n2 = ay + 3 * az;
               const double xl = sT1[((comp * 3 + 0) * Q + qx) * 9 + n2];
               const double xd = sT1[((comp * 3 + 1) * Q + qx) * 9 + n2];
               const double vl = sT1[((comp * 3 + 2) * Q + qx) * 9 + n2];
               a0 += Ly * xd;  // d/dxi
               a1 += dLy * xl; // d/deta
               a2 += Ly * xl;  // d/dzeta (with dL_z)
               a3 += Ly * vl;  // velocity value
            }
            A[comp][0][az] = a0; A[comp][1][az] = a1; A[comp][2][az] = a2; A[comp][3][az] = a3;
         }
      }
      const double wxy = tW[qx] * tW[qy];
#pragma unroll
      for (int qz = 0; qz < Q; qz++)
      {
         double J[3][3], v[3];
#pragma unroll
         for (int comp = 0; comp < 3; comp++)
         {
            double j0 = 0, j1 = 0, j2 = 0, vv = 0;
#pragma unroll
            for (int az = 0; az < 3; az++)
            {
               const double Lz = tL[qz * 3 + az], dLz = tdL[qz * 3 + az];
               j0 += Lz * A[comp][0][az];
               j1 += Lz * A[comp][1][az];
               j2 += dLz * A[comp][2][az];
               vv += Lz * A[comp][3][az];
            }
            J[comp][0] = j0; J[comp][1] = j1; J[comp][2] = j2; v[comp] = vv;
         }
         // adj(J), rows as in remhos_lo.cpp:1168-1180 (A_cr = cofactor so that adj*J = det*I)
         const double A11 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
         const double A12 = J[2][1] * J[0][2] - J[0][1] * J[2][2];
         const double A13 = J[0][1] * J[1][2] - J[1][1] * J[0][2];
         const double A21 = J[2][0] * J[1][2] - J[1][0] * J[2][2];
         const double A22 = J[0][0] * J[2][2] - J[0][2] * J[2][0];
         const double A23 = J[1][0] * J[0][2] - J[0][0] * J[1][2];
         const double A31 = J[1][0] * J[2][1] - J[2][0] * J[1][1];
         const double A32 = J[2][0] * J[0][1] - J[0][0] * J[2][1];
         const double A33 = J[0][0] * J[1][1] - J[0][1] * J[1][0];
         const double detJ = J[0][0] * A11 + J[0][1] * A21 + J[0][2] * A31;
         const double w3 = wxy * tW[qz];
         const double aw = a.alpha * w3;
         Dq[0][qz] = aw * (A11 * v[0] + A12 * v[1] + A13 * v[2]);
         Dq[1][qz] = aw * (A21 * v[0] + A22 * v[1] + A23 * v[2]);
         Dq[2][qz] = aw * (A31 * v[0] + A32 * v[1] + A33 * v[2]);
         wd[qz] = w3 * detJ;
      }
   }

   // ---- phase D: u contracted along x -------------------------------------------------
   if (MODE != 1)
   {
      for (int k = tid; k < 2 * Q * D2; k += NT)
      {
         const int kind = k / (Q * D2), r = k % (Q * D2);
         const int q = r / D2, i2 = r % D2;
         const double *w = (kind ? tG : tB) + q * D;
         double acc = 0.0;
#pragma unroll
         for (int ix = 0; ix < D; ix++) { acc += w[ix] * su[ix + D * i2]; }
         sU1[k] = acc; // [kind][qx][iy + D*iz]
      }
   }
   __syncthreads();

   // ---- phase E: column threads: grad u, D.grad u, and the z-leg of the test contractions ---
   if (col)
   {
      double g[Q];
      if (MODE != 1)
      {
         double UB[D], UG[D], UU[D];
#pragma unroll
         for (int iz = 0; iz < D; iz++)
         {
            double b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
            for (int iy = 0; iy < D; iy++)
            {
               const double By = tB[qy * D + iy], Gy = tG[qy * D + iy];
               const double ub = sU1[(0 * Q + qx) * D2 + iy + D * iz];
               const double ug = sU1[(1 * Q + qx) * D2 + iy + D * iz];
               b0 += By * ug; // d/dxi
               b1 += Gy * ub; // d/deta
               b2 += By * ub; // value along x,y
            }
            UB[iz] = b0; UG[iz] = b1; UU[iz] = b2;
         }
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            double gx = 0, gy = 0, gz = 0;
#pragma unroll
            for (int iz = 0; iz < D; iz++)
            {
               const double Bz = tB[qz * D + iz], Gz = tG[qz * D + iz];
               gx += Bz * UB[iz];
               gy += Bz * UG[iz];
               gz += Gz * UU[iz];
            }
            g[qz] = Dq[0][qz] * gx + Dq[1][qz] * gy + Dq[2][qz] * gz;
         }
      }
      // test along z: r=0: phi_i * (D.grad u); r=1: phi_i * w detJ (lumped mass);
      //               r=2: psi_k^2 * w detJ (Jacobi diagonal of the GL-basis mass matrix)
#pragma unroll
      for (int iz = 0; iz < D; iz++)
      {
         double r0 = 0, r1 = 0, r2 = 0;
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            const double Bz = tB[qz * D + iz];
            if (MODE != 1) { r0 += Bz * g[qz]; r2 += tBg2[qz * D + iz] * wd[qz]; }
            r1 += Bz * wd[qz];
         }
         sR3[(0 * Q2 + qx + Q * qy) * D + iz] = r0;
         sR3[(1 * Q2 + qx + Q * qy) * D + iz] = r1;
         sR3[(2 * Q2 + qx + Q * qy) * D + iz] = r2;
      }
   }
   __syncthreads();

   // ---- phase F/G: remaining legs of the test contractions -----------------------------------
   for (int k = tid; k < 3 * Q * D2; k += NT)
   {
      const int r = k / (Q * D2), rem = k % (Q * D2);
      const int q = rem / D2, i2 = rem % D2;
      const int iy = i2 % D, iz = i2 / D;
      const double *w = (r == 2) ? tBg2 : tB;
      double acc = 0.0;
#pragma unroll
      for (int jy = 0; jy < Q; jy++) { acc += w[jy * D + iy] * sR3[(r * Q2 + q + Q * jy) * D + iz]; }
      sR2[k] = acc; // [r][qx][iy + D*iz]
   }
   __syncthreads();
   for (int k = tid; k < 3 * D3; k += NT)
   {
      const int r = k / D3, i = k % D3;
      const int ix = i % D, i2 = i / D;
      const double *w = (r == 2) ? tBg2 : tB;
      double acc = 0.0;
#pragma unroll
      for (int jx = 0; jx < Q; jx++) { acc += w[jx * D + ix] * sR2[(r * Q + jx) * D2 + i2]; }
      if (r == 0) { s_rhs[i] = acc; }
      else if (r == 1) { s_m[i] = acc; }
      else { s_dg[i] = acc; }
   }
   __syncthreads();

   if (MODE == 1)
   {
      for (int i = tid; i < D3; i += NT) { a.m[(size_t)e * D3 + i] = s_m[i]; }
      return;
   }

   // ---- phase H: faces ------------------------------------------------------------------------
   // every element integrates its own six faces with its own outward normal:
   //   y_i += sum_q w_q max(0, upw * v.n_out) (u_nbr - u_own)(q) phi_i(q)     
   for (int k = tid; k < 6 * Q2; k += NT)
   {
      const int f = k / Q2, r = k % Q2;
      const int q1 = r % Q, q2 = r / Q;
      const int c = f >> 1, side = f & 1;
      const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
      const int nc = (c == 0) ? 1 : (c == 1 ? 3 : 9);
      const int n1 = (c1 == 0) ? 1 : (c1 == 1 ? 3 : 9);
      const int n2 = (c2 == 0) ? 1 : (c2 == 1 ? 3 : 9);
      const int nbase = side ? 2 * nc : 0;
      double t1[3] = {0, 0, 0}, t2[3] = {0, 0, 0}, vf[3] = {0, 0, 0};
#pragma unroll
      for (int a2 = 0; a2 < 3; a2++)
      {
         const double L2 = tL[q2 * 3 + a2], dL2 = tdL[q2 * 3 + a2];
#pragma unroll
         for (int a1 = 0; a1 < 3; a1++)
         {
            const double L1 = tL[q1 * 3 + a1], dL1 = tdL[q1 * 3 + a1];
            const int n = nbase + a1 * n1 + a2 * n2;
            const double w1 = dL1 * L2, w2 = L1 * dL2, w0 = L1 * L2;
#pragma unroll
            for (int comp = 0; comp < 3; comp++)
            {
               const double x = sX[comp * 27 + n];
               t1[comp] += w1 * x;
               t2[comp] += w2 * x;
               vf[comp] += w0 * sV[comp * 27 + n];
            }
         }
      }
      // t1 x t2 = row c of adj(J): area-weighted normal pointing towards +xi_c
      const double nx = t1[1] * t2[2] - t1[2] * t2[1];
      const double ny = t1[2] * t2[0] - t1[0] * t2[2];
      const double nz = t1[0] * t2[1] - t1[1] * t2[0];
      double vn = vf[0] * nx + vf[1] * ny + vf[2] * nz;
      if (!side) { vn = -vn; }
      const double s = fmax(0.0, a.upw * vn) * tW[q1] * tW[q2];
      // jump of the traces at the face point
      const int dc = (c == 0) ? 1 : (c == 1 ? D : D2);
      const int d1 = (c1 == 0) ? 1 : (c1 == 1 ? D : D2);
      const int d2 = (c2 == 0) ? 1 : (c2 == 1 ? D : D2);
      const int dbase = side ? P * dc : 0;
      double jump = 0.0;
#pragma unroll
      for (int i2 = 0; i2 < D; i2++)
      {
         double rowacc = 0.0;
#pragma unroll
         for (int i1 = 0; i1 < D; i1++)
         {
            rowacc += tB[q1 * D + i1] * (sNb[f * D2 + i1 + D * i2] - su[dbase + i1 * d1 + i2 * d2]);
         }
         jump += tB[q2 * D + i2] * rowacc;
      }
      sF[k] = (MODE == 2) ? s : s * jump;
   }
   __syncthreads();
   if (MODE == 2)
   {
      // ============ subcell residual distribution (PAResidualDistributionSubcell::CalcLOSolution,
      // remhos_lo.cpp:1620-1802; host form remhos_lo.cpp:111-245) ============================
      // here s_rhs = z = K_vol u (Bernstein test), s_m = lumped mass, sF = w_q max(0, upw v.n_out)
      constexpr int NS = P * P * P; // subcells (remhos_tools.cpp:678-734)
      // [3][D3] sub-mesh node positions at pseudo-time t live in the (unused) PCG vectors
      auto s_xs3 = [&](int comp) -> double * { return comp == 0 ? s_x : (comp == 1 ? s_r : s_d); };
      double *s_fl = sR3;            // [6][NS]: fluctP, fluctN, xMax, xMin, swP, swN per subcell
      static_assert(6 * NS <= 3 * Q2 * D, "s_fl fits in sR3");
      const double eps = 1.E-15, gamma = 1.0;
      // (1) sub-mesh nodes: x_sub(t) = x0_sub + t * v_sub, x0_sub = Q2 map of the START mesh at the
      //     closed-uniform points i/p (remhos.cpp:796-867, 1262-1274)
      for (int k = tid; k < 3 * D3; k += NT)
      {
         const int comp = k / D3, i = k % D3;
         const int ix = i % D, iy = (i / D) % D, iz = i / D2;
         const double *x0 = a.x0 + (size_t)e * 81 + comp * 27;
         const double *Lx = tab + C::oLcu + ix * 3, *Ly = tab + C::oLcu + iy * 3, *Lz = tab + C::oLcu + iz * 3;
         double acc = 0.0;
#pragma unroll
         for (int az = 0; az < 3; az++)
         {
#pragma unroll
            for (int ay = 0; ay < 3; ay++)
            {
#pragma unroll
               for (int ax = 0; ax < 3; ax++) { acc += Lx[ax] * Ly[ay] * Lz[az] * x0[ax + 3 * (ay + 3 * az)]; }
            }
         }
         const double vs = a.subvel[((size_t)e * 3 + comp) * D3 + i];
         s_xs3(comp)[i] = a.move ? acc + a.t * vs : acc;
      }
      // (2) lumped upwind face fluxes (ApplyFaceTerms, remhos_lo.cpp:795-871): gathered per dof
      double duf[C::DPT];
#pragma unroll
      for (int kk = 0; kk < C::DPT; kk++)
      {
         const int i = tid + kk * NT;
         duf[kk] = 0.0;
         if (i < D3)
         {
            const int idx[3] = {i % D, (i / D) % D, i / D2};
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 3; c++)
            {
               const int ic = idx[c];
               if (ic == 0 || ic == P)
               {
                  const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                  const int i1 = idx[c1], i2 = idx[c2];
                  const int side = (ic == P) ? 1 : 0;
                  const double *F = sF + (2 * c + side) * Q2;
                  double coef = 0.0;
#pragma unroll
                  for (int q2 = 0; q2 < Q; q2++)
                  {
                     double rowacc = 0.0;
#pragma unroll
                     for (int q1 = 0; q1 < Q; q1++) { rowacc += tB[q1 * D + i1] * F[q1 + Q * q2]; }
                     coef += tB[q2 * D + i2] * rowacc;
                  }
                  acc += coef * (sNb[(2 * c + side) * D2 + i1 + D * i2] - su[i]);
               }
            }
            duf[kk] = acc;
         }
      }
      __syncthreads();
      // (3) subcell fluctuations F_m = sum_j W_mj u_j with the 1-point rule on the trilinear subcell
      //     (SetupSubCellPA3D / ApplySubCellWeights, remhos_lo.cpp:1137-1192, 1313-1618) and the
      //     subcell extrema (remhos_lo.cpp:1733-1757)
      for (int m = tid; m < NS; m += NT)
      {
         const int mx = m % P, my = (m / P) % P, mz = m / (P * P);
         const int base = mx + D * my + D2 * mz;
         double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, vm[3] = {0, 0, 0};
         double umax = -INFINITY, umin = INFINITY, usum = 0.0;
#pragma unroll
         for (int j = 0; j < 8; j++)
         {
            const int i = base + (j & 1) + D * ((j >> 1) & 1) + D2 * (j >> 2);
            const double uj = su[i];
            umax = fmax(umax, uj);
            umin = fmin(umin, uj);
            usum += uj;
#pragma unroll
            for (int comp = 0; comp < 3; comp++)
            {
               const double x = s_xs3(comp)[i];
               J[comp][0] += ((j & 1) ? 0.25 : -0.25) * x;
               J[comp][1] += ((j & 2) ? 0.25 : -0.25) * x;
               J[comp][2] += ((j & 4) ? 0.25 : -0.25) * x;
               vm[comp] += 0.125 * a.subvel[((size_t)e * 3 + comp) * D3 + i];
            }
         }
         const double A11 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
         const double A12 = J[2][1] * J[0][2] - J[0][1] * J[2][2];
         const double A13 = J[0][1] * J[1][2] - J[1][1] * J[0][2];
         const double A21 = J[2][0] * J[1][2] - J[1][0] * J[2][2];
         const double A22 = J[0][0] * J[2][2] - J[0][2] * J[2][0];
         const double A23 = J[1][0] * J[0][2] - J[0][0] * J[1][2];
         const double A31 = J[1][0] * J[2][1] - J[2][0] * J[1][1];
         const double A32 = J[2][0] * J[0][1] - J[0][0] * J[2][1];
         const double A33 = J[0][0] * J[1][1] - J[0][1] * J[1][0];
         const double q0 = a.alpha * (A11 * vm[0] + A12 * vm[1] + A13 * vm[2]);
         const double q1 = a.alpha * (A21 * vm[0] + A22 * vm[1] + A23 * vm[2]);
         const double q2 = a.alpha * (A31 * vm[0] + A32 * vm[1] + A33 * vm[2]);
         double fluct = 0.0;
#pragma unroll
         for (int j = 0; j < 8; j++)
         {
            const int i = base + (j & 1) + D * ((j >> 1) & 1) + D2 * (j >> 2);
            const double w = ((j & 1) ? 0.25 : -0.25) * q0 + ((j & 2) ? 0.25 : -0.25) * q1 + ((j & 4) ? 0.25 : -0.25) * q2;
            fluct += w * su[i];
         }
         s_fl[0 * NS + m] = fmax(0., fluct);
         s_fl[1 * NS + m] = fmin(0., fluct);
         s_fl[2 * NS + m] = umax;
         s_fl[3 * NS + m] = umin;
         s_fl[4 * NS + m] = 8 * umax - usum + eps;
         s_fl[5 * NS + m] = 8 * umin - usum - eps;
      }
      __syncthreads();
      // (4) element sums (remhos_lo.cpp:1702-1716)
      double lmax = -INFINITY, lmin = INFINITY, lsum = 0.0, lrp = 0.0, lrn = 0.0, lfp = 0.0, lfn = 0.0;
      for (int i = tid; i < D3; i += NT)
      {
         const double ui = su[i], zi = s_rhs[i];
         lmax = fmax(lmax, ui);
         lmin = fmin(lmin, ui);
         lsum += ui;
         lrp += fmax(0., zi);
         lrn += fmin(0., zi);
      }
      for (int m = tid; m < NS; m += NT) { lfp += s_fl[m]; lfn += s_fl[NS + m]; }
      const double xe_max = block_max<NW>(lmax, s_red);
      const double xe_min = block_min<NW>(lmin, s_red);
      const double xSum = block_sum<NW>(lsum, s_red);
      const double rhoP = block_sum<NW>(lrp, s_red);
      const double rhoN = block_sum<NW>(lrn, s_red);
      const double sumFluctP = block_sum<NW>(lfp, s_red);
      const double sumFluctN = block_sum<NW>(lfn, s_red);
      const double sumWeightsP = D3 * xe_max - xSum + eps;
      const double sumWeightsN = D3 * xe_min - xSum - eps;
      // (5) nodal weights (eqs. 58-59) gathered from the <= 8 subcells around each dof, final formula
#pragma unroll
      for (int kk = 0; kk < C::DPT; kk++)
      {
         const int i = tid + kk * NT;
         if (i < D3)
         {
            const int ix = i % D, iy = (i / D) % D, iz = i / D2;
            const double ui = su[i];
            double nwP = 0.0, nwN = 0.0;
            // the reference accumulates over subcells m ascending; same order here
            for (int dz = 1; dz >= 0; dz--)
            {
               for (int dy = 1; dy >= 0; dy--)
               {
                  for (int dx = 1; dx >= 0; dx--)
                  {
                     const int mx = ix - dx, my = iy - dy, mz = iz - dz;
                     if (mx >= 0 && mx < P && my >= 0 && my < P && mz >= 0 && mz < P)
                     {
                        const int m = mx + P * (my + P * mz);
                        nwP += s_fl[0 * NS + m] * ((s_fl[2 * NS + m] - ui) / s_fl[4 * NS + m]);
                        nwN += s_fl[1 * NS + m] * ((s_fl[3 * NS + m] - ui) / s_fl[5 * NS + m]);
                     }
                  }
               }
            }
            double weightP = (xe_max - ui) / sumWeightsP;
            double weightN = (xe_min - ui) / sumWeightsN;
            double aux = gamma / (rhoP + eps);
            weightP *= 1. - fmin(aux * sumFluctP, 1.);
            weightP += fmin(aux, 1. / (sumFluctP + eps)) * nwP;
            aux = gamma / (rhoN - eps);
            weightN *= 1. - fmin(aux * sumFluctN, 1.);
            weightN += fmax(aux, 1. / (sumFluctN - eps)) * nwN;
            a.du[(size_t)e * D3 + i] = (duf[kk] + weightP * rhoP + weightN * rhoN) / s_m[i];
            a.m[(size_t)e * D3 + i] = s_m[i];
         }
      }
      if (tid == 0)
      {
         a.xe_min[e] = xe_min;
         a.xe_max[e] = xe_max;
      }
      return;
   }
   // test with the face-layer basis functions, gathered per dof (no atomics)
   for (int i = tid; i < D3; i += NT)
   {
      const int idx[3] = {i % D, (i / D) % D, i / D2};
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < 3; c++)
      {
         const int ic = idx[c];
         if (ic == 0 || ic == P)
         {
            const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
            const int i1 = idx[c1], i2 = idx[c2];
            // for P == 0 both faces touch the single dof
            for (int side = (ic == 0 ? 0 : 1); side <= (ic == P ? 1 : 0); side++)
            {
               const double *F = sF + (2 * c + side) * Q2;
               double fa = 0.0;
#pragma unroll
               for (int q2 = 0; q2 < Q; q2++)
               {
                  double rowacc = 0.0;
#pragma unroll
                  for (int q1 = 0; q1 < Q; q1++) { rowacc += tB[q1 * D + i1] * F[q1 + Q * q2]; }
                  fa += tB[q2 * D + i2] * rowacc;
               }
               acc += fa;
            }
         }
      }
      s_rhs[i] += acc;
   }
   __syncthreads();

   // ---- phase I: element-local mass solve ---------------------------------------------------------
   // b_g = Ci^T (x) Ci^T (x) Ci^T  b   (rhs in the Gauss-Legendre nodal basis)
   dof_tensor_apply<D, NT, true>(tCi, 0, s_rhs, s_x);
   __syncthreads();
   dof_tensor_apply<D, NT, true>(tCi, 1, s_x, s_d);
   __syncthreads();
   dof_tensor_apply<D, NT, true>(tCi, 2, s_d, s_r);
   __syncthreads();

   double part = 0.0;
   for (int i = tid; i < D3; i += NT)
   {
      const double r = s_r[i];
      const double z = r / s_dg[i];
      s_x[i] = 0.0;
      s_d[i] = z;
      part += r * z;
   }
   double nom = block_sum<NW>(part, s_red);
   const double tol = fmax(a.rel2 * nom, a.abs2);
   int it = 0;
   while (nom > tol && it < a.max_iter)
   {
      __syncthreads();
      // Ad = M_g d : x leg
      for (int k = tid; k < Q * D2; k += NT)
      {
         const int q = k / D2, i2 = k % D2;
         double acc = 0.0;
#pragma unroll
         for (int ix = 0; ix < D; ix++) { acc += tBg[q * D + ix] * s_d[ix + D * i2]; }
         sU1[k] = acc;
      }
      __syncthreads();
      if (col)
      {
         double Y[D];
#pragma unroll
         for (int iz = 0; iz < D; iz++)
         {
            double acc = 0.0;
#pragma unroll
            for (int iy = 0; iy < D; iy++) { acc += tBg[qy * D + iy] * sU1[qx * D2 + iy + D * iz]; }
            Y[iz] = acc;
         }
         double val[Q];
#pragma unroll
         for (int qz = 0; qz < Q; qz++)
         {
            double acc = 0.0;
#pragma unroll
            for (int iz = 0; iz < D; iz++) { acc += tBg[qz * D + iz] * Y[iz]; }
            val[qz] = acc * wd[qz];
         }
#pragma unroll
         for (int iz = 0; iz < D; iz++)
         {
            double acc = 0.0;
#pragma unroll
            for (int qz = 0; qz < Q; qz++) { acc += tBg[qz * D + iz] * val[qz]; }
            sR3[(qx + Q * qy) * D + iz] = acc;
         }
      }
      __syncthreads();
      for (int k = tid; k < Q * D2; k += NT)
      {
         const int q = k / D2, i2 = k % D2;
         const int iy = i2 % D, iz = i2 / D;
         double acc = 0.0;
#pragma unroll
         for (int jy = 0; jy < Q; jy++) { acc += tBg[jy * D + iy] * sR3[(q + Q * jy) * D + iz]; }
         sR2[k] = acc;
      }
      __syncthreads();
      part = 0.0;
      for (int i = tid; i < D3; i += NT)
      {
         const int ix = i % D, i2 = i / D;
         double acc = 0.0;
#pragma unroll
         for (int jx = 0; jx < Q; jx++) { acc += tBg[jx * D + ix] * sR2[jx * D2 + i2]; }
         s_Ad[i] = acc;
         part += s_d[i] * acc;
      }
      const double den = block_sum<NW>(part, s_red);
      if (!(den > 0.0)) { break; }
      const double al = nom / den;
      part = 0.0;
      for (int i = tid; i < D3; i += NT)
      {
         s_x[i] += al * s_d[i];
         const double r = s_r[i] - al * s_Ad[i];
         s_r[i] = r;
         part += r * (r / s_dg[i]);
      }
      const double betanom = block_sum<NW>(part, s_red);
      const double beta = betanom / nom;
      for (int i = tid; i < D3; i += NT) { s_d[i] = s_r[i] / s_dg[i] + beta * s_d[i]; }
      nom = betanom;
      it++;
   }
   __syncthreads();
   // x_b = Ci (x) Ci (x) Ci  x_g
   dof_tensor_apply<D, NT, false>(tCi, 0, s_x, s_d);
   __syncthreads();
   dof_tensor_apply<D, NT, false>(tCi, 1, s_d, s_r);
   __syncthreads();
   dof_tensor_apply<D, NT, false>(tCi, 2, s_r, s_x);
   __syncthreads();

   // ---- phase J: stores ------------------------------------------------------------------------
   double lmin = INFINITY, lmax = -INFINITY;
   for (int i = tid; i < D3; i += NT)
   {
      a.du[(size_t)e * D3 + i] = s_x[i];
      a.m[(size_t)e * D3 + i] = s_m[i];
      lmin = fmin(lmin, su[i]);
      lmax = fmax(lmax, su[i]);
   }
   lmin = block_min<NW>(lmin, s_red);
   lmax = block_max<NW>(lmax, s_red);
   if (tid == 0)
   {
      a.xe_min[e] = lmin;
      a.xe_max[e] = lmax;
      // only when it can raise the maximum: atomics on one address serialise at the memory side
      if (it > 0 && it > __builtin_nontemporal_load(a.cg_iters)) { atomicMax(a.cg_iters, it); }
   }
}

// ---------------------------------------------------------------------------------------
// Element extrema: DofInfo::ComputeElementsMinMax (remhos_tools.cpp:497-523)
// ---------------------------------------------------------------------------------------
template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) elem_minmax_kernel(const double *u, double *xe_min, double *xe_max)
{
   using C = KCfg<P>;
   __shared__ double s_red[4];
   const int e = blockIdx.x;
   double lmin = INFINITY, lmax = -INFINITY;
   for (int i = threadIdx.x; i < C::D3; i += C::NT)
   {
      const double v = u[(size_t)e * C::D3 + i];
      lmin = fmin(lmin, v);
      lmax = fmax(lmax, v);
   }
   lmin = block_min<C::NW>(lmin, s_red);
   lmax = block_max<C::NW>(lmax, s_red);
   if (threadIdx.x == 0) { xe_min[e] = lmin; xe_max[e] = lmax; }
}

// per-dof overlap bounds from the 27-element stencil held in LDS (smin/smax[27]):
// a dof on the low/high face layer of direction c also sees the -1/+1 neighbours in c
template <int P>
__device__ inline void dof_bounds(int i, const double *smin, const double *smax, double &lo, double &hi)
{
   constexpr int D = P + 1;
   const int idx[3] = {i % D, (i / D) % D, i / (D * D)};
   int olo[3], ohi[3];
#pragma unroll
   for (int c = 0; c < 3; c++)
   {
      olo[c] = (idx[c] == 0) ? -1 : 0;
      ohi[c] = (idx[c] == P) ? 1 : 0;
   }
   lo = INFINITY;
   hi = -INFINITY;
   for (int oz = olo[2]; oz <= ohi[2]; oz++)
   {
      for (int oy = olo[1]; oy <= ohi[1]; oy++)
      {
         for (int ox = olo[0]; ox <= ohi[0]; ox++)
         {
            const int s = (ox + 1) + 3 * (oy + 1) + 9 * (oz + 1);
            lo = fmin(lo, smin[s]);
            hi = fmax(hi, smax[s]);
         }
      }
   }
}

// DofInfo::ComputeBounds (remhos_tools.hpp:168-182): bounds type 0 = overlap bounds above, 1 = the element
// and its face neighbours (ComputeMatrixSparsityBounds, remhos_tools.cpp:381-430), one interval per element
template <int P>
__device__ inline void dof_bounds_bt(int bt, int i, const double *smin, const double *smax, double &lo, double &hi)
{
   if (bt == 0) { dof_bounds<P>(i, smin, smax, lo, hi); return; }
   constexpr int fs[7] = {13, 12, 14, 10, 16, 4, 22};
   lo = INFINITY;
   hi = -INFINITY;
#pragma unroll
   for (int k = 0; k < 7; k++)
   {
      lo = fmin(lo, smin[fs[k]]);
      hi = fmax(hi, smax[fs[k]]);
   }
}

// AdvectionOperator::UpdateTimeStepEstimate (remhos.cpp:1968-1998) for one dof: the largest dt with
// x_min <= x + dt*dx <= x_max (+inf when |dx| <= 1e-12)
__device__ inline double dt_candidate(double x, double dx, double xmin, double xmax)
{
   constexpr double eps = 1e-12;
   double c = INFINITY;
   if (dx > eps) { c = (xmax - x) / dx; }
   else if (dx < -eps) { c = (xmin - x) / dx; }
   return c + 0.0; // -0.0 -> +0.0: the minimum below orders the bit patterns
}

// min into a device scalar that only ever decreases; candidates are >= 0, whose bit patterns order like
// the values.  Guarded like the PCG iteration counter: the atomic is issued only when it can lower the value.
__device__ inline void atomic_min_nonneg(double *p, double v)
{
   if (v < __builtin_nontemporal_load(p))
   {
      atomicMin((unsigned long long *)p, (unsigned long long)__double_as_longlong(v));
   }
}

__device__ inline void load_stencil(int e, int ne_owned, const int *stencil27, const double *xe_min,
                                    const double *xe_max, const double *gh_min, const double *gh_max, int gh_mstride,
                                    double *smin, double *smax)
{
   const int tid = threadIdx.x;
   if (tid < 27)
   {
      const int nb = stencil27[(size_t)e * 27 + tid];
      double lo = INFINITY, hi = -INFINITY;
      if (nb >= 0)
      {
         if (nb < ne_owned) { lo = xe_min[nb]; hi = xe_max[nb]; }
         else { lo = gh_min[(size_t)(nb - ne_owned) * gh_mstride]; hi = gh_max[(size_t)(nb - ne_owned) * gh_mstride]; }
      }
      smin[tid] = lo;
      smax[tid] = hi;
   }
}

// DofInfo::ComputeBounds -> ComputeOverlapBounds (remhos_tools.cpp:432-495)
template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) bounds_kernel(int bt, const int *stencil27, int ne_owned, const double *xe_min,
                                                             const double *xe_max, const double *gh_min,
                                                             const double *gh_max, int gh_mstride, double *u_min, double *u_max)
{
   using C = KCfg<P>;
   __shared__ double smin[27], smax[27];
   const int e = blockIdx.x;
   load_stencil(e, ne_owned, stencil27, xe_min, xe_max, gh_min, gh_max, gh_mstride, smin, smax);
   __syncthreads();
   for (int i = threadIdx.x; i < C::D3; i += C::NT)
   {
      double lo, hi;
      dof_bounds_bt<P>(bt, i, smin, smax, lo, hi);
      u_min[(size_t)e * C::D3 + i] = lo;
      u_max[(size_t)e * C::D3 + i] = hi;
   }
}

// lo 4 set-up (once per context): sub-mesh start positions = Q2 map of the start mesh at the closed-uniform
// points i/p (ParMesh::MakeRefined(pmesh, order, ClosedUniform), remhos.cpp:796-826) and the subcell midpoint
// velocities (mean of the corner values of v_sub_gf; lom.subcellCoeff->Eval at the 1-point rule,
// remhos_lo.cpp:1051-1082).
template <int P>
__global__ void __launch_bounds__(64) subcell_setup_kernel(const double *x0, const double *subvel, const double *tab,
                                                           double *subx0, double *subvmid)
{
   constexpr int D = P + 1, D2 = D * D, D3 = D2 * D, NS = P * P * P;
   using T = TabLayout<P>;
   const int e = blockIdx.x;
   for (int k = threadIdx.x; k < 3 * D3; k += 64)
   {
      const int comp = k / D3, i = k % D3;
      const int ix = i % D, iy = (i / D) % D, iz = i / D2;
      const double *xn = x0 + (size_t)e * 81 + comp * 27;
      const double *Lx = tab + T::oLcu + ix * 3, *Ly = tab + T::oLcu + iy * 3, *Lz = tab + T::oLcu + iz * 3;
      double acc = 0.0;
      for (int az = 0; az < 3; az++)
      {
         for (int ay = 0; ay < 3; ay++)
         {
            for (int ax = 0; ax < 3; ax++) { acc += Lx[ax] * Ly[ay] * Lz[az] * xn[ax + 3 * (ay + 3 * az)]; }
         }
      }
      subx0[(size_t)e * 3 * D3 + k] = acc;
   }
   for (int k = threadIdx.x; k < 3 * NS; k += 64)
   {
      const int comp = k / NS, m = k % NS;
      const int base = (m % P) + D * ((m / P) % P) + D2 * (m / (P * P));
      double acc = 0.0;
      for (int j = 0; j < 8; j++)
      {
         acc += 0.125 * subvel[((size_t)e * 3 + comp) * D3 + base + (j & 1) + D * ((j >> 1) & 1) + D2 * (j >> 2)];
      }
      subvmid[(size_t)e * 3 * NS + k] = acc;
   }
}

// Halo pack: for every element a neighbour rank needs (send_elems, all peers concatenated) copy its
// ndof values of u and its min / max into contiguous send buffers -- the payload of
// ParGridFunction::ExchangeFaceNbrData (remhos_ho.cpp:122) and of the GroupCommunicator min/max
// reduction (remhos_tools.cpp:461-466) in one pass.
template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) halo_pack_kernel(const double *u, const int *send_elems, double *rows,
                                                                double *out_min, double *out_max, int row_stride,
                                                                int mm_stride)
{
   using C = KCfg<P>;
   __shared__ double s_red[4];
   const int k = blockIdx.x;
   const int e = send_elems[k];
   double lmin = INFINITY, lmax = -INFINITY;
   for (int i = threadIdx.x; i < C::D3; i += C::NT)
   {
      const double v = u[(size_t)e * C::D3 + i];
      rows[(size_t)k * row_stride + i] = v;
      lmin = fmin(lmin, v);
      lmax = fmax(lmax, v);
   }
   lmin = block_min<C::NW>(lmin, s_red);
   lmax = block_max<C::NW>(lmax, s_red);
   if (threadIdx.x == 0) { out_min[(size_t)k * mm_stride] = lmin; out_max[(size_t)k * mm_stride] = lmax; }
}

// MassBasedAvg::CalcLOSolution (remhos_lo.cpp:247-324): du_lo = (ubar - u)/dt with
// ubar = int (u + dt du_ho) / int 1 over the element at the current mesh position.
// int u_h = sum_i m_i u_i with the lumped mass (Bernstein partition of unity).
template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) lo_massavg_kernel(const double *u, const double *du_ho, const double *m,
                                                                 double dt, double *du_lo)
{
   using C = KCfg<P>;
   __shared__ double s_red[4];
   const int e = blockIdx.x;
   double uu[C::DPT];
   double mass = 0.0, vol = 0.0;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      uu[k] = 0.0;
      if (i < C::D3)
      {
         const size_t g = (size_t)e * C::D3 + i;
         uu[k] = u[g];
         const double mi = m[g];
         mass += mi * (uu[k] + dt * du_ho[g]);
         vol += mi;
      }
   }
   mass = block_sum<C::NW>(mass, s_red);
   vol = block_sum<C::NW>(vol, s_red);
   const double ubar = mass / vol;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      if (i < C::D3) { du_lo[(size_t)e * C::D3 + i] = (ubar - uu[k]) / dt; }
   }
}

// ClipScaleSolver::CalcFCTSolution (remhos_fct.cpp:449-541), one element per workgroup,
// wave-reduced sumPos / sumNeg, the f_clip scratch stays in registers.
template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) fct_clipscale_kernel(const double *u, const double *m, const double *du_ho,
                                                                    const double *du_lo, const double *u_min,
                                                                    const double *u_max, double dt, double *du)
{
   using C = KCfg<P>;
   __shared__ double s_red[4];
   const int e = blockIdx.x;
   constexpr double eps = 1.0e-15;
   double f[C::DPT], mm[C::DPT], dl[C::DPT];
   double sumPos = 0.0, sumNeg = 0.0;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      f[k] = 0.0; mm[k] = 1.0; dl[k] = 0.0;
      if (i < C::D3)
      {
         const size_t g = (size_t)e * C::D3 + i;
         const double ui = u[g];
         mm[k] = m[g];
         dl[k] = du_lo[g];
         const double u_new_lo = ui + dt * dl[k];
         const double f_clip_min = mm[k] / dt * (u_min[g] - u_new_lo);
         const double f_clip_max = mm[k] / dt * (u_max[g] - u_new_lo);
         double fc = mm[k] * (du_ho[g] - dl[k]);
         fc = fmin(f_clip_max, fmax(f_clip_min, fc));
         f[k] = fc;
         sumNeg += fmin(fc, 0.0);
         sumPos += fmax(fc, 0.0);
      }
   }
   sumNeg = block_sum<C::NW>(sumNeg, s_red);
   sumPos = block_sum<C::NW>(sumPos, s_red);
   const double new_mass = sumNeg + sumPos;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      if (i < C::D3)
      {
         double fc = f[k];
         if (new_mass > eps) { fc = fmin(0.0, fc) - fmax(0.0, fc) * sumNeg / sumPos; }
         if (new_mass < -eps) { fc = fmax(0.0, fc) - fmin(0.0, fc) * sumPos / sumNeg; }
         du[(size_t)e * C::D3 + i] = dl[k] + fc / mm[k];
      }
   }
}

// ---------------------------------------------------------------------------------------
// Product-field remap (-ps): the per-element pieces of AdvectionOperator::LimitMult's second block
// (remhos.cpp:1848-1915).  One element per workgroup, flags are bytes (mfem::Array<bool>).
// ---------------------------------------------------------------------------------------
constexpr double RMH_EMPTY_ZONE_TOL = 1e-12; // remhos_sync.hpp:20

// ComputeBoolIndicators (remhos_sync.cpp:23-47) and, with us != null, ComputeRatio (:50-96): s = us / u on the active
// dofs, the mean of the active ratios elsewhere in an active element, 0 in empty elements.
template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) product_ratio_kernel(const double *us, const double *u, double *s,
                                                                    unsigned char *active_el, unsigned char *active_dofs)
{
   using C = KCfg<P>;
   __shared__ double s_red[4];
   const int e = blockIdx.x;
   double r[C::DPT];
   bool act[C::DPT];
   double sum = 0.0, cnt = 0.0;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      r[k] = 0.0; act[k] = false;
      if (i < C::D3)
      {
         const size_t g = (size_t)e * C::D3 + i;
         const double ui = u[g];
         act[k] = ui > RMH_EMPTY_ZONE_TOL;
         active_dofs[g] = act[k] ? 1 : 0;
         if (us && act[k]) { r[k] = us[g] / ui; sum += r[k]; }
         if (act[k]) { cnt += 1.0; }
      }
   }
   cnt = block_sum<C::NW>(cnt, s_red);
   if (threadIdx.x == 0) { active_el[e] = cnt > 0.0 ? 1 : 0; }
   if (!us) { return; }
   sum = block_sum<C::NW>(sum, s_red);
   const double s_avg = cnt > 0.0 ? sum / cnt : 0.0;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      if (i < C::D3) { s[(size_t)e * C::D3 + i] = act[k] ? r[k] : s_avg; }
   }
}

// DofInfo::ComputeElementsMinMax with the active-element / active-dof masks (remhos_tools.cpp:497-523): inactive
// elements and dofs do not contribute; an inactive element gets (+inf, -inf), the identities of the bounds stencil.
template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) elem_minmax_masked_kernel(const double *u, const unsigned char *active_el,
                                                                         const unsigned char *active_dofs, double *xe_min,
                                                                         double *xe_max)
{
   using C = KCfg<P>;
   __shared__ double s_red[4];
   const int e = blockIdx.x;
   double lmin = INFINITY, lmax = -INFINITY;
   if (active_el[e])
   {
      for (int i = threadIdx.x; i < C::D3; i += C::NT)
      {
         const size_t g = (size_t)e * C::D3 + i;
         if (active_dofs[g])
         {
            lmin = fmin(lmin, u[g]);
            lmax = fmax(lmax, u[g]);
         }
      }
   }
   lmin = block_min<C::NW>(lmin, s_red);
   lmax = block_max<C::NW>(lmax, s_red);
   if (threadIdx.x == 0) { xe_min[e] = lmin; xe_max[e] = lmax; }
}

// ClipScaleSolver::CalcFCTProduct (remhos_fct.cpp:543-566) in one pass over the element:
//   FCTSolver::CalcCompatibleLOProduct (remhos_fct.cpp:26-115): s_avg = mass_us / mass_u, pulled onto [smin, smax] of the
//     active dofs when it misses them by round-off only, local bounds widened to s_avg where they exclude it (s_min / s_max
//     are updated in place like the reference's), d_us_LO = (u_new s_avg - us) / dt;
//   FCTSolver::ScaleProductBounds (:117-153): us_min/max = s_min/max * u_new on active dofs, 0 elsewhere;
//   ClipScaleSolver::CalcFCTSolution (:449-541) on (us, m, d_us_HO, d_us_LO, us_min, us_max);
//   ZeroOutEmptyDofs (remhos_sync.cpp:98-116): empty elements get d_us = 0.
template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) fct_product_kernel(const double *us, const double *m, const double *d_us_ho,
                                                                  double *s_min, double *s_max, const double *u_new,
                                                                  const unsigned char *active_el,
                                                                  const unsigned char *active_dofs, double dt, double *d_us)
{
   using C = KCfg<P>;
   __shared__ double s_red[4];
   const int e = blockIdx.x;
   constexpr double eps12 = 1e-12, eps = 1.0e-15;
   const bool el_on = active_el[e] != 0;
   double usv[C::DPT], mm[C::DPT], dho[C::DPT], un[C::DPT], lo[C::DPT], hi[C::DPT];
   bool act[C::DPT];
   double mass_us = 0.0, mass_u = 0.0, smin = INFINITY, smax = -INFINITY;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      usv[k] = 0.0; mm[k] = 1.0; dho[k] = 0.0; un[k] = 0.0; lo[k] = 0.0; hi[k] = 0.0; act[k] = false;
      if (i < C::D3)
      {
         const size_t g = (size_t)e * C::D3 + i;
         usv[k] = us[g]; mm[k] = m[g]; dho[k] = d_us_ho[g]; un[k] = u_new[g];
         lo[k] = s_min[g]; hi[k] = s_max[g];
         act[k] = active_dofs[g] != 0;
         mass_us += (usv[k] + dt * dho[k]) * mm[k];
         mass_u += un[k] * mm[k];
         if (act[k]) { smin = fmin(smin, lo[k]); smax = fmax(smax, hi[k]); }
      }
   }
   mass_us = block_sum<C::NW>(mass_us, s_red);
   mass_u = block_sum<C::NW>(mass_u, s_red);
   smin = block_min<C::NW>(smin, s_red);
   smax = block_max<C::NW>(smax, s_red);
   double s_avg = el_on ? mass_us / mass_u : 0.0;
   // (the reference repeats these two tests for every active dof; they do not depend on the dof)
   if (el_on && s_avg < smin && mass_us + eps12 > smin * mass_u) { s_avg = smin; }
   if (el_on && s_avg > smax && mass_us - eps12 < smax * mass_u) { s_avg = smax; }
   double f[C::DPT], dl[C::DPT];
   double sumPos = 0.0, sumNeg = 0.0;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      f[k] = 0.0; dl[k] = 0.0;
      if (i < C::D3)
      {
         const size_t g = (size_t)e * C::D3 + i;
         double us_lo = 0.0, us_hi = 0.0;
         if (el_on)
         {
            if (act[k])
            {
               if (s_avg + eps12 < lo[k]) { lo[k] = s_avg; s_min[g] = s_avg; }
               if (s_avg - eps12 > hi[k]) { hi[k] = s_avg; s_max[g] = s_avg; }
               us_lo = lo[k] * un[k];
               us_hi = hi[k] * un[k];
            }
            dl[k] = (un[k] * s_avg - usv[k]) / dt;
         }
         const double us_new_lo = usv[k] + dt * dl[k];
         const double f_clip_min = mm[k] / dt * (us_lo - us_new_lo);
         const double f_clip_max = mm[k] / dt * (us_hi - us_new_lo);
         double fc = mm[k] * (dho[k] - dl[k]);
         fc = fmin(f_clip_max, fmax(f_clip_min, fc));
         f[k] = fc;
         sumNeg += fmin(fc, 0.0);
         sumPos += fmax(fc, 0.0);
      }
   }
   sumNeg = block_sum<C::NW>(sumNeg, s_red);
   sumPos = block_sum<C::NW>(sumPos, s_red);
   const double new_mass = sumNeg + sumPos;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      if (i < C::D3)
      {
         double fc = f[k];
         if (new_mass > eps) { fc = fmin(0.0, fc) - fmax(0.0, fc) * sumNeg / sumPos; }
         if (new_mass < -eps) { fc = fmax(0.0, fc) - fmin(0.0, fc) * sumPos / sumNeg; }
         d_us[(size_t)e * C::D3 + i] = el_on ? dl[k] + fc / mm[k] : 0.0;
      }
   }
}

// Fused LimitMult for -lo 5 -fct 2 (remhos.cpp:1798-1845): MassBasedAvg + overlap bounds +
// ClipScale (+ optional RK update) in one pass; du_lo, u_min, u_max are never materialised.
struct LimitArgs
{
   const double *u, *du_ho, *m;
   const double *du_lo; // null: mass-based average computed here (lo 5); else the LO rate of another solver (lo 4)
   const int *stencil27;
   const double *xe_min, *xe_max, *gh_min, *gh_max;
   int gh_mstride;
   int ne_owned;
   double dt;
   double *du;          // may be null when y_out is given
   const double *x_base; // RK: y_out = a*x_base + b*(u + dt_rk*du)
   double a, b, dt_rk;
   double *y_out;       // may be null
   int bounds_type;     // 0 overlap, 1 face neighbours
   double *dt_est;      // null, or the running minimum of UpdateTimeStepEstimate(u, du_LO, u_min, u_max)
};

// UpdateTimeStepEstimate for the granular call sequence: min over the dofs into *dt_est
template <int P>
__global__ void __launch_bounds__(256) dt_estimate_kernel(const double *x, const double *dx, const double *xmin,
                                                          const double *xmax, size_t n, double *dt_est)
{
   __shared__ double s_red[4];
   double c = INFINITY;
   for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
   {
      c = fmin(c, dt_candidate(x[i], dx[i], xmin[i], xmax[i]));
   }
   c = block_min<4>(c, s_red);
   if (threadIdx.x == 0) { atomic_min_nonneg(dt_est, c); }
}

template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) limit_fused_kernel(LimitArgs a)
{
   using C = KCfg<P>;
   __shared__ double s_red[4];
   __shared__ double smin[27], smax[27];
   const int e = blockIdx.x;
   constexpr double eps = 1.0e-15;
   load_stencil(e, a.ne_owned, a.stencil27, a.xe_min, a.xe_max, a.gh_min, a.gh_max, a.gh_mstride, smin, smax);
   double uu[C::DPT], mm[C::DPT], dh[C::DPT];
   double mass = 0.0, vol = 0.0;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      uu[k] = 0.0; mm[k] = 1.0; dh[k] = 0.0;
      if (i < C::D3)
      {
         const size_t g = (size_t)e * C::D3 + i;
         uu[k] = a.u[g];
         mm[k] = a.m[g];
         dh[k] = a.du_ho[g];
         mass += mm[k] * (uu[k] + a.dt * dh[k]);
         vol += mm[k];
      }
   }
   __syncthreads(); // stencil in LDS
   mass = block_sum<C::NW>(mass, s_red);
   vol = block_sum<C::NW>(vol, s_red);
   const double ubar = mass / vol;
   double f[C::DPT], dl[C::DPT];
   double sumPos = 0.0, sumNeg = 0.0, dtc = INFINITY;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      f[k] = 0.0; dl[k] = 0.0;
      if (i < C::D3)
      {
         double lo, hi;
         dof_bounds_bt<P>(a.bounds_type, i, smin, smax, lo, hi);
         dl[k] = a.du_lo ? a.du_lo[(size_t)e * C::D3 + i] : (ubar - uu[k]) / a.dt;
         dtc = fmin(dtc, dt_candidate(uu[k], dl[k], lo, hi));
         const double u_new_lo = uu[k] + a.dt * dl[k];
         const double f_clip_min = mm[k] / a.dt * (lo - u_new_lo);
         const double f_clip_max = mm[k] / a.dt * (hi - u_new_lo);
         double fc = mm[k] * (dh[k] - dl[k]);
         fc = fmin(f_clip_max, fmax(f_clip_min, fc));
         f[k] = fc;
         sumNeg += fmin(fc, 0.0);
         sumPos += fmax(fc, 0.0);
      }
   }
   sumNeg = block_sum<C::NW>(sumNeg, s_red);
   sumPos = block_sum<C::NW>(sumPos, s_red);
   if (a.dt_est)
   {
      dtc = block_min<C::NW>(dtc, s_red);
      if (threadIdx.x == 0) { atomic_min_nonneg(a.dt_est, dtc); }
   }
   const double new_mass = sumNeg + sumPos;
#pragma unroll
   for (int k = 0; k < C::DPT; k++)
   {
      const int i = threadIdx.x + k * C::NT;
      if (i < C::D3)
      {
         const size_t g = (size_t)e * C::D3 + i;
         double fc = f[k];
         if (new_mass > eps) { fc = fmin(0.0, fc) - fmax(0.0, fc) * sumNeg / sumPos; }
         if (new_mass < -eps) { fc = fmax(0.0, fc) - fmin(0.0, fc) * sumPos / sumNeg; }
         const double dui = dl[k] + fc / mm[k];
         if (a.du) { a.du[g] = dui; }
         if (a.y_out)
         {
            const double y = uu[k] + a.dt_rk * dui;
            a.y_out[g] = (a.x_base ? a.a * a.x_base[g] : 0.0) + a.b * y;
         }
      }
   }
}

} // namespace rmh
