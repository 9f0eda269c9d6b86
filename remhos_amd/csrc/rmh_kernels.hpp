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

template <int P, int DIM = 3> // (DIM = 2: D3 is the dof count D^2 of a quadrilateral element -- the streaming kernels only)
struct KCfg : TabLayout<P>
{
   using T = TabLayout<P>;
   static constexpr int D = T::D, Q = T::Q;
   static constexpr int D2 = D * D, D3 = DIM == 3 ? D * D * D : D * D, Q2 = Q * Q, Q3 = Q * Q * Q;
   static constexpr int NT = (Q2 <= 64) ? 64 : 128;
   static constexpr int NW = NT / 64;
   static constexpr int DPT = (D3 + NT - 1) / NT; // dofs per thread
};

// 1-D tables of every order in constant memory: compile-time indexed reads become scalar loads
// (s_load) everywhere, including after barriers where the compiler no longer treats loads from
// ordinary global memory as invariant.  Filled by rmh_create (identical for all contexts).
constexpr int RMH_TAB_STRIDE = 864; // (>= TabLayout<6>::N3 = 854)
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
   const double *fgeo;     // [slots][Q][3][Q] face speed coefficients, one block per face (face_geom_kernel, rmh_ho2.hpp)
   const int *face_rows;   // [ne][6] block of every element face in fgeo; bit 31: read with the sign flipped
   double *du;             // [ne][D3]
   double *m;              // [ne][D3] lumped mass
   double *xe_min, *xe_max; // [ne]
   int *cg_iters;          // [1] max over elements (atomicMax)
   int ne_owned;
   int e_begin, e_end;     // elements this launch works on (ho_kernel2; the whole rank: 0, ne_owned)
   int xcd_chunk;          // batches per chunk of the XCD-aware batch order (ho_kernel2): chunks are dealt round-robin to the 8 XCDs; 0: one contiguous eighth each
   int xcd_weave;          // log2 of the lattice layers woven into one chunk (batch j of a chunk: layer j % 2^w, position j >> w); 0: none
   int xcd_rounds;         // whole rounds of 8 chunks in this launch = (n_batches / 8) / xcd_chunk, and floor(2^32 / xcd_chunk): the kernel divides by multiplication
   unsigned xcd_inv;
   int reverse;            // 1: the batches of this launch are taken from the last to the first (alternating stages: rmh_ctx::alt_order)
   double t;               // pseudo-time (remap) -- X = x0 + t*vel
   int move;               // 1: remap (mesh moves), 0: transport (static mesh)
   double alpha;           // ConvectionIntegrator alpha: -1 transport, +1 remap (remhos.cpp:648-657)
   double upw;             // upwind sign: s = max(0, upw * v.n_out): -1 transport, +1 remap
   double rel2, abs2;      // squared tolerances of the local PCG
   int max_iter;
   int jacobi_step;        // completion of the local solve (rmh_set_mass_completion): 1: x += D^-1 r behind the PCG loop
   int mass_fix;           //   1: constant mode -- du_HO += (1^T b - sum m du_HO) / |element| behind the back-transform
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

// Stores of E-vectors that are not read again before they have left the caches -- every output of the streaming kernels:
// nontemporal.  Measured (tools/gbench.py, -rs 5, p = 3): the write-only bounds kernel 275 -> 163 us (3.7 -> 6.2 TB/s),
// ClipScale 554 -> 527 us, the fused limiter 392 -> 385 us; p = 4, 5 alike, the mass-based average +-3 %.
#ifndef RMH_NT_STORES
#define RMH_NT_STORES 1
#endif
__device__ inline void store_stream(double *p, double x) // an output that is not read again soon
{
#if defined(__HIP_DEVICE_COMPILE__) && RMH_NT_STORES
   __builtin_nontemporal_store(x, p);
#else
   *p = x;
#endif
}

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

// ---------------------------------------------------------------------------------------
// Lumped mass M_HO * 1 at pseudo-time t (remhos.cpp:719-727, 1625-1632; the initial / final mass of the report,
// remhos.cpp:1073-1076, 1394-1403): m_i = sum_q w_q detJ(q) B_i(q), sum-factorised.  One element per workgroup;
// geometry from the 27 nodes x0 + t v like the stage kernel (rmh_ho2.hpp), without u.  Called a few times per run.
// ---------------------------------------------------------------------------------------
template <int P>
// The nodes arrive in the form the context keeps on the device -- hierarchical along the directions of the bit mask `hier`
// ((n0, n1 - n0, n2 - n0), see RMH_HIER in rmh_ho2.hpp; 0: nodal) -- and are put back into nodal form in LDS first.
__global__ void __launch_bounds__(KCfg<P>::NT) lumped_mass_kernel(const double *x0, const double *vel, const double *gtab,
                                                                  double t, int move, int hier, double *m)
{
   using C = KCfg<P>;
   constexpr int D = C::D, Q = C::Q, D2 = C::D2, D3 = C::D3, Q2 = C::Q2, NT = C::NT;
   __shared__ double tab[C::N2];
   __shared__ double sX[81];
   __shared__ double sT1[6 * Q * 9];
   __shared__ double sR3[Q2 * D];
   __shared__ double sR2[Q * D2];
   const int tid = threadIdx.x, e = blockIdx.x;
   const double *tB = tab + C::oB, *tL = tab + C::oL, *tdL = tab + C::odL, *tW = tab + C::oW;
   for (int i = tid; i < C::N2; i += NT) { tab[i] = gtab[i]; }
   for (int i = tid; i < 81; i += NT)
   {
      const double x = x0[(size_t)e * 81 + i];
      sX[i] = move ? x + t * vel[(size_t)e * 81 + i] : x;
   }
   __syncthreads();
   for (int dir = 2; dir >= 0; dir--) // (the host took the differences along x, then y, then z: undone in the reverse order)
   {
      if (!((hier >> dir) & 1)) { continue; }
      const int st = dir == 0 ? 1 : (dir == 1 ? 3 : 9);
      for (int i = tid; i < 81; i += NT)
      {
         const int a = ((i % 27) / st) % 3;
         if (a > 0) { sX[i] += sX[i - a * st]; } // (entries with a = 0 are only read in this pass)
      }
      __syncthreads();
   }
   // x-contraction of the nodes: sT1[((comp*2 + kind)*Q + qx)*9 + ay + 3*az], kind 0: L.X, 1: dL.X
   for (int k = tid; k < 6 * Q * 9; k += NT)
   {
      const int arr = k / (Q * 9), r = k % (Q * 9);
      const int qx = r / 9, n2 = r % 9;
      const int comp = arr / 2, kind = arr % 2;
      const double *src = sX + comp * 27 + 3 * n2;
      const double *w = (kind ? tdL : tL) + qx * 3;
      sT1[k] = w[0] * src[0] + w[1] * src[1] + w[2] * src[2];
   }
   __syncthreads();
   // column (qx, qy): Jacobian at the Q points of the column, w detJ tested with the Bernstein basis along z
   if (tid < Q2)
   {
      const int qx = tid % Q, qy = tid / Q;
      double A[3][3][3]; // [comp][d/dxi, d/deta, value for d/dzeta][az]
#pragma unroll
      for (int comp = 0; comp < 3; comp++)
      {
#pragma unroll
         for (int az = 0; az < 3; az++)
         {
            double a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
            for (int ay = 0; ay < 3; ay++)
            {
               const double Ly = tL[qy * 3 + ay], dLy = tdL[qy * 3 + ay];
               const double xl = sT1[((comp * 2 + 0) * Q + qx) * 9 + ay + 3 * az];
               const double xd = sT1[((comp * 2 + 1) * Q + qx) * 9 + ay + 3 * az];
               a0 += Ly * xd;
               a1 += dLy * xl;
               a2 += Ly * xl;
            }
            A[comp][0][az] = a0; A[comp][1][az] = a1; A[comp][2][az] = a2;
         }
      }
      double r1[D];
#pragma unroll
      for (int iz = 0; iz < D; iz++) { r1[iz] = 0.0; }
      const double wxy = tW[qx] * tW[qy];
#pragma unroll
      for (int qz = 0; qz < Q; qz++)
      {
         double J[3][3];
#pragma unroll
         for (int comp = 0; comp < 3; comp++)
         {
            double j0 = 0, j1 = 0, j2 = 0;
#pragma unroll
            for (int az = 0; az < 3; az++)
            {
               const double Lz = tL[qz * 3 + az], dLz = tdL[qz * 3 + az];
               j0 += Lz * A[comp][0][az];
               j1 += Lz * A[comp][1][az];
               j2 += dLz * A[comp][2][az];
            }
            J[comp][0] = j0; J[comp][1] = j1; J[comp][2] = j2;
         }
         const double A11 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
         const double A21 = J[2][0] * J[1][2] - J[1][0] * J[2][2];
         const double A31 = J[1][0] * J[2][1] - J[2][0] * J[1][1];
         const double wdq = wxy * tW[qz] * (J[0][0] * A11 + J[0][1] * A21 + J[0][2] * A31);
#pragma unroll
         for (int iz = 0; iz < D; iz++) { r1[iz] += tB[qz * D + iz] * wdq; }
      }
#pragma unroll
      for (int iz = 0; iz < D; iz++) { sR3[tid * D + iz] = r1[iz]; }
   }
   __syncthreads();
   for (int k = tid; k < Q * D2; k += NT)
   {
      const int q = k / D2, i2 = k % D2;
      const int iy = i2 % D, iz = i2 / D;
      double acc = 0.0;
#pragma unroll
      for (int jy = 0; jy < Q; jy++) { acc += tB[jy * D + iy] * sR3[(q + Q * jy) * D + iz]; }
      sR2[k] = acc;
   }
   __syncthreads();
   for (int i = tid; i < D3; i += NT)
   {
      const int ix = i % D, i2 = i / D;
      double acc = 0.0;
#pragma unroll
      for (int jx = 0; jx < Q; jx++) { acc += tB[jx * D + ix] * sR2[jx * D2 + i2]; }
      m[(size_t)e * D3 + i] = acc;
   }
}

// (ComputeElementsMinMax, ComputeBounds and the fused LimitMult are wavefront-per-element streaming kernels: rmh_stream.hpp)

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
template <int P, int DIM = 3>
__global__ void __launch_bounds__((KCfg<P, DIM>::NT)) lo_massavg_kernel(const double *u, const double *du_ho, const double *m,
                                                                 double dt, double *du_lo)
{
   using C = KCfg<P, DIM>;
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
      if (i < C::D3) { store_stream(du_lo + (size_t)e * C::D3 + i, (ubar - uu[k]) / dt); }
   }
}

// ClipScaleSolver::CalcFCTSolution (remhos_fct.cpp:449-541), one element per workgroup,
// wave-reduced sumPos / sumNeg, the f_clip scratch stays in registers.
template <int P, int DIM = 3>
__global__ void __launch_bounds__((KCfg<P, DIM>::NT)) fct_clipscale_kernel(const double *u, const double *m, const double *du_ho,
                                                                    const double *du_lo, const double *u_min,
                                                                    const double *u_max, double dt, double *du)
{
   using C = KCfg<P, DIM>;
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
         store_stream(du + (size_t)e * C::D3 + i, dl[k] + fc / mm[k]);
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


// check_violation (remhos.cpp:1557-1594; -vb): the reference walks the dofs on the host and aborts at the FIRST one outside
// [u_min - tol, u_max + tol].  Here every dof is tested in one streaming pass and the verdict is four device words:
//   v[0] number of violating dofs, v[1] the smallest violating index (what the reference reports), v[2] / v[3] the largest
//   over- / undershoot u_new - u_max, u_min - u_new among them (bit patterns of non-negative doubles order like the values).
// du == null: u holds u_new itself (first overload); scale != null: the bounds are (u_min * scale, u_max * scale), the
// ScaleProductBounds form of CalcFCTProduct's check (remhos_fct.cpp:117-153, 568-610); active: the dofs to look at, or null.
__global__ void __launch_bounds__(256) check_violation_kernel(const double *u, double dt, const double *du, const double *umin,
                                                              const double *umax, const double *scale, double tol,
                                                              const unsigned char *active, size_t n, unsigned long long *v)
{
   unsigned long long cnt = 0, first = ~0ull;
   double over = 0.0, under = 0.0;
   for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
   {
      if (active && !active[i]) { continue; }
      const double un = du ? u[i] + dt * du[i] : u[i];
      const double s = scale ? scale[i] : 1.0;
      const double lo = scale ? umin[i] * s : umin[i], hi = scale ? umax[i] * s : umax[i];
      if (un + tol < lo || un > hi + tol)
      {
         cnt++;
         first = first < i ? first : (unsigned long long)i;
         over = fmax(over, un - hi);
         under = fmax(under, lo - un);
      }
   }
   if (cnt) // (a violation is the exception: no block-wide reduction on the path every launch takes)
   {
      atomicAdd(&v[0], cnt);
      atomicMin(&v[1], first);
      atomicMax(&v[2], (unsigned long long)__double_as_longlong(over));
      atomicMax(&v[3], (unsigned long long)__double_as_longlong(under));
   }
}

} // namespace rmh
