// Streaming kernels of the GRANULAR call sequence (the reference's own order of calls, remhos.cpp:1815-1831):
// ComputeElementsMinMax, ComputeBounds and the fused LimitMult (MassBasedAvg + bounds + ClipScale + RK update).
// Each reads and writes every E-vector once: they are HBM-bound, and their measure is the fraction of the HBM rate
// (tools/gbench.py).
//
// Shape (round 4).  The first generation gave every element its own 64-thread workgroup: 885 k workgroups per launch
// at -rs 5, each with one load per thread and a dependent chain stencil index -> neighbour extrema -> LDS -> barrier in
// front of its arithmetic -- 1.9 TB/s (extrema), 2.6 TB/s (bounds), 3.4 TB/s (fused limiter).  Here a WAVEFRONT owns
// whole elements and nothing synchronises more than a wavefront:
//   * 256-thread workgroups, a grid of a few thousand, each wavefront walks the elements grid-stride, U elements per
//     pass with all their loads issued before the first use (8-byte loads per lane, 512 B per wave-instruction);
//   * dof i of an element sits in lane i mod 64, round i / 64 -- coalesced, and the same for every batch and partition,
//     so the sums (DPP trees in a fixed order, wave_sum) do not depend on where an element is;
//   * the 27-element stencil of the overlap bounds is reduced ONCE per element to the 27 "classes" a dof can belong to
//     (low face / interior / high face per direction): lanes 0..26 hold the neighbours' extrema, write them to the
//     wavefront's own LDS slot, form the box minima / maxima of their class from at most 8 entries, and every dof then
//     reads ONE 16-byte entry (the first generation: up to 8 divergent pairs of reads per dof).
#pragma once
#include "rmh_ho2.hpp"

namespace rmh
{

// A wavefront's own LDS writes are read by other lanes of the same wavefront: the hardware executes one wavefront's LDS
// instructions in order, the compiler must not reorder them, and the host emulation (every lane an OS thread) has to wait.
__device__ inline void wave_lds_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
   __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
   __builtin_amdgcn_wave_barrier();
   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#elif defined(HIPEMU)
   hipemu_wave_sync();
#endif
}

// Passes (U elements each) are dealt to the wavefronts of the grid in grid-stride order: neighbouring wavefronts work on
// neighbouring memory at the same time.  (Measured: one contiguous chunk of elements per wavefront instead: -3 ... -8 % for
// every kernel here.)
#define RMH_PASS_LOOP(wave) for (int e0 = (wave) * U; e0 < ne; e0 += nwaves * U)

template <int P, int DIM = 3> // (DIM = 2: D3 is the dof count D^2 of a quadrilateral element; its 3 x 3 element stencil sits
struct SCfg                  //  in the middle layer of the 27-entry table, every dof is "interior" along z)
{
   static constexpr int D = P + 1, D2 = D * D, D3 = DIM == 3 ? D * D * D : D * D;
   static constexpr int NT = 256, NW = NT / 64;
   static constexpr int DPT = (D3 + 63) / 64; // dof rounds of a wavefront per element
   // elements a wavefront has in flight per pass: ~8 loads per lane and array for the kernels that read one or two
   // arrays, ~4 for the fused limiter (four to five arrays, six live values per dof)
   static constexpr int U8 = DPT >= 5 ? 1 : (DPT >= 3 ? 2 : (DPT == 2 ? 4 : 8));
   // (fused limiter, RK update form, swept: p = 3: 1 471 us, 2 413, 4 390, 8 440; p = 4: 1 238, 2 214, 4 217; p = 5: 1 153, 2 163;
   //  p = 6: 1 232, 2 252)
   static constexpr int U4 = DPT >= 3 ? 1 : (DPT == 2 ? 2 : 4);
   // bounds (write-only): an even number -- the pass is stored in 16-byte pieces -- measured per order with the streaming
   // stores (p = 3: 8 163 us, 16 201 us; p = 4: 4 138 us, 8 104 us; p = 5: 2 72 us, 4 60 us, 16 68 us; p = 6: 2 175 us, 4 160 us,
   // 8 121 us, 16 142 us)
   static constexpr int UB = (DPT == 3 || DPT == 4) ? 4 : 8;
   // grid: enough wavefronts to fill the chip several times over, few enough that a wavefront amortises its start
   static int grid(int ne, int u)
   {
      const long long passes = ((long long)ne + u - 1) / u;
      const long long wgs = (passes + NW - 1) / NW;
      return (int)(wgs < 1 ? 1 : (wgs > 256 * 16 ? 256 * 16 : wgs));
   }
   // class of a dof: 0 low face layer, 1 interior, 2 high face layer, per direction (c = cx + 3 cy + 9 cz)
   __device__ static int dof_class(int i)
   {
      const int ix = i % D, iy = (i / D) % D, iz = i / D2;
      const int cx = (ix == 0) ? 0 : ((ix == P) ? 2 : 1);
      const int cy = (iy == 0) ? 0 : ((iy == P) ? 2 : 1);
      const int cz = DIM == 2 ? 1 : ((iz == 0) ? 0 : ((iz == P) ? 2 : 1));
      return cx + 3 * cy + 9 * cz;
   }
};

struct alignas(16) MinMax
{
   double lo, hi;
};

// The extrema of stencil neighbour nb (owned element, ghost, or -1: none) -- straight-line: the address is selected, not
// the load, so that the loads of all the elements of a pass are in flight together (with a branch per case the compiler
// drains the memory counter between them)
__device__ inline MinMax stencil_entry(int nb, int ne_owned, const double *xe_min, const double *xe_max, const double *gh_min,
                                       const double *gh_max, int gh_mstride)
{
   const int k = max(nb, 0);
   const bool owned = k < ne_owned;
   const size_t go = (size_t)max(k - ne_owned, 0) * gh_mstride;
   const double *pmin = owned ? xe_min + k : gh_min + go;
   const double *pmax = owned ? xe_max + k : gh_max + go;
   MinMax r;
   r.lo = *pmin;
   r.hi = *pmax;
   if (nb < 0) { r.lo = INFINITY; r.hi = -INFINITY; }
   return r;
}

template <int N, class F>
__device__ inline void static_for(F &&f)
{
   if constexpr (N > 0)
   {
      static_for<N - 1>(f);
      f(std::integral_constant<int, N - 1>());
   }
}

struct OpSum { __device__ static double f(double a, double b) { return a + b; } };
struct OpMin { __device__ static double f(double a, double b) { return fmin(a, b); } };
struct OpMax { __device__ static double f(double a, double b) { return fmax(a, b); } };

template <int CTRL>
__device__ inline double dpp_mov_all(double v) // the value of the lane the control selects (controls that give every lane a source)
{
   const int lo = __double2loint(v), hi = __double2hiint(v);
#if defined(__HIP_DEVICE_COMPILE__)
   return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false), __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false));
#else
   return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false));
#endif
}

template <class Op>
__device__ inline double comb32(double a, double b) { swap32(a, b); return Op::f(a, b); }
template <class Op>
__device__ inline double comb16(double a, double b) { swap16(a, b); return Op::f(a, b); }
template <class Op>
__device__ inline double row_reduce(double v)
{
   v = Op::f(v, dpp_mov_all<0xB1>(v));  // quad_perm [1, 0, 3, 2]
   v = Op::f(v, dpp_mov_all<0x4E>(v));  // quad_perm [2, 3, 0, 1]
   v = Op::f(v, dpp_mov_all<0x141>(v)); // row_half_mirror
   v = Op::f(v, dpp_mov_all<0x140>(v)); // row_mirror
   return v;
}

template <class Op, int U>
__device__ inline void wave_reduce_multi(double (&v)[U])
{
   static_assert(U == 1 || U == 2 || U == 4 || U == 8, "elements per pass");
   if constexpr (U == 8)
   {
      const double m0 = comb32<Op>(v[0], v[1]), m1 = comb32<Op>(v[2], v[3]);
      const double m2 = comb32<Op>(v[4], v[5]), m3 = comb32<Op>(v[6], v[7]);
      v[0] = row_reduce<Op>(comb16<Op>(m0, m1)); // rows: elements 0, 2, 1, 3
      v[1] = row_reduce<Op>(comb16<Op>(m2, m3)); //       elements 4, 6, 5, 7
   }
   else if constexpr (U == 4)
   {
      const double m0 = comb32<Op>(v[0], v[1]), m1 = comb32<Op>(v[2], v[3]);
      v[0] = row_reduce<Op>(comb16<Op>(m0, m1));
   }
   else if constexpr (U == 2)
   {
      const double m = comb32<Op>(v[0], v[1]);
      v[0] = row_reduce<Op>(comb16<Op>(m, m)); // rows: elements 0, 0, 1, 1
   }
   else
   {
      const double m = comb32<Op>(v[0], v[0]);
      v[0] = row_reduce<Op>(comb16<Op>(m, m));
   }
}
template <int U>
constexpr int red_vec(int j) { return U == 8 ? j / 4 : 0; }
template <int U>
constexpr int red_row(int j) { return U >= 4 ? ((j % 4) == 1 ? 2 : ((j % 4) == 2 ? 1 : j % 4)) : (U == 2 ? 2 * j : 0); }
// element J's result in all lanes (a scalar register pair on the device)
template <int U, int J>
__device__ inline double wave_pick(const double (&v)[U]) { return wave_bcast<16 * red_row<U>(J)>(v[red_vec<U>(J)]); }

// Class tables of the U elements of a pass in the wavefront's LDS slot s[U][27]: on entry lane t < 27 holds, per element,
// the extrema of stencil entry t = (ox + 1) + 3 (oy + 1) + 9 (oz + 1); on exit s[j][c] = min / max over the elements that
// share the CG nodes of class c (DofInfo::ComputeOverlapBounds, remhos_tools.cpp:432-495: a dof on the low / high face
// layer of a direction also sees the -1 / +1 neighbours there) -- or, bounds type 1, the element and its face neighbours
// for every class (ComputeMatrixSparsityBounds, remhos_tools.cpp:381-430).  The 27 U (element, class) tasks are spread
// over the 64 lanes; a task reads its box as eight entries (a one-wide direction reads its entry twice): no loop, no
// divergence.
template <int U>
__device__ inline void class_tables(int bt, int lane, const MinMax (&raw)[U], MinMax *s)
{
   if (lane < 27)
   {
#pragma unroll
      for (int j = 0; j < U; j++) { s[27 * j + lane] = raw[j]; }
   }
   wave_lds_fence();
   constexpr int NR = (27 * U + 63) / 64;
   MinMax r[NR];
#pragma unroll
   for (int rr = 0; rr < NR; rr++)
   {
      const int t = min(lane + 64 * rr, 27 * U - 1);
      const int j = t / 27, c = t - 27 * j;
      const MinMax *sj = s + 27 * j;
      MinMax acc = {INFINITY, -INFINITY};
      if (bt == 0)
      {
         const int cx = c % 3, cy = (c / 3) % 3, cz = c / 9;
         const int x[2] = {cx == 0 ? 0 : 1, cx == 2 ? 2 : 1};
         const int y[2] = {cy == 0 ? 0 : 3, cy == 2 ? 6 : 3};
         const int z[2] = {cz == 0 ? 0 : 9, cz == 2 ? 18 : 9};
#pragma unroll
         for (int k = 0; k < 8; k++)
         {
            const MinMax v = sj[x[k & 1] + y[(k >> 1) & 1] + z[k >> 2]];
            acc.lo = fmin(acc.lo, v.lo);
            acc.hi = fmax(acc.hi, v.hi);
         }
      }
      else
      {
         constexpr int fs[7] = {13, 12, 14, 10, 16, 4, 22};
#pragma unroll
         for (int k = 0; k < 7; k++)
         {
            const MinMax v = sj[fs[k]];
            acc.lo = fmin(acc.lo, v.lo);
            acc.hi = fmax(acc.hi, v.hi);
         }
      }
      r[rr] = acc;
   }
   wave_lds_fence(); // every lane has read the raw entries
#pragma unroll
   for (int rr = 0; rr < NR; rr++)
   {
      const int t = lane + 64 * rr;
      if (t < 27 * U) { s[t] = r[rr]; }
   }
   wave_lds_fence();
}

// ---------------------------------------------------------------------------------------
// Element extrema: DofInfo::ComputeElementsMinMax (remhos_tools.cpp:497-523)
// ---------------------------------------------------------------------------------------
template <int P, int DIM = 3>
__global__ void __launch_bounds__((SCfg<P, DIM>::NT)) elem_minmax_kernel(const double *u, double *xe_min, double *xe_max, int ne)
{
   using C = SCfg<P, DIM>;
   constexpr int U = C::U8;
   const int lane = threadIdx.x & 63;
   const int nwaves = gridDim.x * C::NW;
   RMH_PASS_LOOP(blockIdx.x * C::NW + (threadIdx.x >> 6))
   {
      double v[U][C::DPT];
#pragma unroll
      for (int j = 0; j < U; j++)
      {
         const size_t base = (size_t)min(e0 + j, ne - 1) * C::D3;
#pragma unroll
         for (int k = 0; k < C::DPT; k++) { v[j][k] = u[base + min(lane + 64 * k, C::D3 - 1)]; } // (past the end: a duplicate)
      }
      double lo[U], hi[U];
#pragma unroll
      for (int j = 0; j < U; j++)
      {
         lo[j] = v[j][0];
         hi[j] = v[j][0];
#pragma unroll
         for (int k = 1; k < C::DPT; k++)
         {
            lo[j] = fmin(lo[j], v[j][k]);
            hi[j] = fmax(hi[j], v[j][k]);
         }
      }
      wave_reduce_multi<OpMin, U>(lo);
      wave_reduce_multi<OpMax, U>(hi);
      // element j's results sit in row red_row(j) of vector red_vec(j): the first lanes of the rows of a vector store
      // together (one store instruction per vector and array, not one per element)
      constexpr int NV = U == 8 ? 2 : 1;        // vectors
      constexpr int RS = U >= 4 ? 16 : (U == 2 ? 32 : 64); // lanes per element in a vector
      const int grp = lane / RS;
      const int jv = U >= 4 ? ((grp == 1) ? 2 : ((grp == 2) ? 1 : grp)) : grp; // the element of this lane's rows within its vector
#pragma unroll
      for (int v4 = 0; v4 < NV; v4++)
      {
         const int e = e0 + 4 * v4 + jv;
         if ((lane & (RS - 1)) == 0 && e < ne)
         {
            xe_min[e] = lo[v4];
            xe_max[e] = hi[v4];
         }
      }
   }
}

// ---------------------------------------------------------------------------------------
// DofInfo::ComputeBounds -> ComputeOverlapBounds (remhos_tools.cpp:432-495) / bounds type 1
// ---------------------------------------------------------------------------------------
// (Write-only apart from the stencil: the stores decide.  With 8-byte stores per lane -- dof i in lane i mod 64 -- the kernel
// ran at 3.7-4.0 TB/s at every order; the elements of a pass are contiguous in memory, so the pass is stored as one flat
// range of U D3 doubles, 16 bytes per lane and store instruction -- two neighbouring dofs, each with its own class entry.)
__device__ inline void store_pair(double *p, double x, double y) // p 16-byte aligned
{
#if defined(__HIP_DEVICE_COMPILE__)
   typedef double dbl2 __attribute__((ext_vector_type(2)));
   const dbl2 v = {x, y};
#if RMH_NT_STORES
   __builtin_nontemporal_store(v, (dbl2 *)p);
#else
   *(dbl2 *)p = v;
#endif
#else
   p[0] = x;
   p[1] = y;
#endif
}

template <int P, int DIM = 3>
__global__ void __launch_bounds__((SCfg<P, DIM>::NT)) bounds_kernel(int bt, const int *stencil27, int ne_owned, const double *xe_min,
                                                             const double *xe_max, const double *gh_min,
                                                             const double *gh_max, int gh_mstride, double *u_min, double *u_max,
                                                             int wide)
{
   using C = SCfg<P, DIM>;
   constexpr int U = C::UB;
   constexpr int L = U * C::D3, NP = L / 2, RP = (NP + 63) / 64; // doubles, pairs and store rounds of a pass
   static_assert(L % 2 == 0, "a pass starts 16-byte aligned");
   __shared__ MinMax s_cls[C::NW][U * 27];
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
   const int nwaves = gridDim.x * C::NW;
   const int ne = ne_owned;
   // table entries of the two dofs of this lane's pairs: the same in every pass
   int o0[RP], o1[RP];
#pragma unroll
   for (int r = 0; r < RP; r++)
   {
      const int d0 = 2 * min(lane + 64 * r, NP - 1), d1 = d0 + 1;
      o0[r] = (d0 / C::D3) * 27 + C::dof_class(d0 % C::D3);
      o1[r] = (d1 / C::D3) * 27 + C::dof_class(d1 % C::D3);
   }
   const MinMax *tab = s_cls[wv];
   RMH_PASS_LOOP(blockIdx.x * C::NW + wv)
   {
      int nb[U];
#pragma unroll
      for (int j = 0; j < U; j++) { nb[j] = stencil27[(size_t)min(e0 + j, ne - 1) * 27 + min(lane, 26)]; }
      sched_fence(); // (all index loads, then all gathers: left alone the compiler interleaves them element by element with a full wait each)
      MinMax raw[U];
#pragma unroll
      for (int j = 0; j < U; j++) { raw[j] = stencil_entry(nb[j], ne_owned, xe_min, xe_max, gh_min, gh_max, gh_mstride); }
      sched_fence();
      class_tables<U>(bt, lane, raw, s_cls[wv]);
      const int lv = min(U, ne - e0) * C::D3; // doubles of this pass that exist
      double *pmin = u_min + (size_t)e0 * C::D3, *pmax = u_max + (size_t)e0 * C::D3;
#pragma unroll
      for (int r = 0; r < RP; r++)
      {
         const int d0 = 2 * (lane + 64 * r);
         if (d0 < lv)
         {
            const MinMax b0 = tab[o0[r]], b1 = tab[o1[r]];
            if (wide && d0 + 1 < lv)
            {
               store_pair(pmin + d0, b0.lo, b1.lo);
               store_pair(pmax + d0, b0.hi, b1.hi);
            }
            else
            {
               store_stream(pmin + d0, b0.lo);
               store_stream(pmax + d0, b0.hi);
               if (d0 + 1 < lv)
               {
                  store_stream(pmin + d0 + 1, b1.lo);
                  store_stream(pmax + d0 + 1, b1.hi);
               }
            }
         }
      }
      wave_lds_fence(); // the slots are rewritten by the next pass
   }
}

// ---------------------------------------------------------------------------------------
// Fused LimitMult for -lo 5 -fct 2 (remhos.cpp:1798-1845): MassBasedAvg (remhos_lo.cpp:247-324) + overlap bounds +
// ClipScale (remhos_fct.cpp:449-541) (+ optional RK update) in one pass; du_lo, u_min, u_max are never materialised.
// With LimitArgs::du_lo the LO rate of another solver (lo 3 / 4) takes the place of the mass-based average.
// ---------------------------------------------------------------------------------------
template <int P, int DIM = 3>
__global__ void __launch_bounds__((SCfg<P, DIM>::NT)) limit_fused_kernel(LimitArgs a)
{
   using C = SCfg<P, DIM>;
   constexpr int U = C::U4;
   constexpr double eps = 1.0e-15;
   __shared__ MinMax s_cls[C::NW][U][27];
   const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
   const int nwaves = gridDim.x * C::NW;
   const int ne = a.ne_owned;
   int cls[C::DPT];
#pragma unroll
   for (int k = 0; k < C::DPT; k++) { cls[k] = C::dof_class(min(lane + 64 * k, C::D3 - 1)); }
   RMH_PASS_LOOP(blockIdx.x * C::NW + wv)
   {
      int nb[U];
#pragma unroll
      for (int j = 0; j < U; j++) { nb[j] = a.stencil27[(size_t)min(e0 + j, ne - 1) * 27 + min(lane, 26)]; }
      double uu[U][C::DPT], mm[U][C::DPT], dh[U][C::DPT], dl[U][C::DPT], xb[U][C::DPT];
#pragma unroll
      for (int j = 0; j < U; j++)
      {
         const size_t base = (size_t)min(e0 + j, ne - 1) * C::D3;
#pragma unroll
         for (int k = 0; k < C::DPT; k++)
         {
            const size_t g = base + min(lane + 64 * k, C::D3 - 1);
            uu[j][k] = a.u[g];
            mm[j][k] = a.m[g];
            dh[j][k] = a.du_ho[g];
            dl[j][k] = a.du_lo ? a.du_lo[g] : 0.0;
            xb[j][k] = (a.y_out && a.x_base) ? a.x_base[g] : 0.0;
         }
      }
      sched_fence(); // (index and vector loads first, then the gathers, all in flight together)
      MinMax raw[U];
#pragma unroll
      for (int j = 0; j < U; j++) { raw[j] = stencil_entry(nb[j], ne, a.xe_min, a.xe_max, a.gh_min, a.gh_max, a.gh_mstride); }
      sched_fence();
      class_tables<U>(a.bounds_type, lane, raw, &s_cls[wv][0][0]);
      double dtc = INFINITY;
      if (!a.du_lo)
      {
         // MassBasedAvg: ubar = int (u + dt du_HO) / int 1 with the lumped mass
         double mass[U], vol[U];
#pragma unroll
         for (int j = 0; j < U; j++)
         {
            mass[j] = 0.0;
            vol[j] = 0.0;
#pragma unroll
            for (int k = 0; k < C::DPT; k++)
            {
               if (lane + 64 * k < C::D3)
               {
                  mass[j] += mm[j][k] * (uu[j][k] + a.dt * dh[j][k]);
                  vol[j] += mm[j][k];
               }
            }
         }
         wave_reduce_multi<OpSum, U>(mass);
         wave_reduce_multi<OpSum, U>(vol);
         static_for<U>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const double ubar = wave_pick<U, j>(mass) / wave_pick<U, j>(vol);
#pragma unroll
            for (int k = 0; k < C::DPT; k++) { dl[j][k] = (ubar - uu[j][k]) / a.dt; }
         });
      }
      double f[U][C::DPT], sumPos[U], sumNeg[U];
#pragma unroll
      for (int j = 0; j < U; j++)
      {
         sumPos[j] = 0.0;
         sumNeg[j] = 0.0;
#pragma unroll
         for (int k = 0; k < C::DPT; k++)
         {
            f[j][k] = 0.0;
            if (lane + 64 * k < C::D3)
            {
               const MinMax b = s_cls[wv][j][cls[k]];
               if (a.dt_est) { dtc = fmin(dtc, dt_candidate(uu[j][k], dl[j][k], b.lo, b.hi)); }
               const double u_new_lo = uu[j][k] + a.dt * dl[j][k];
               const double f_clip_min = mm[j][k] / a.dt * (b.lo - u_new_lo);
               const double f_clip_max = mm[j][k] / a.dt * (b.hi - u_new_lo);
               double fc = mm[j][k] * (dh[j][k] - dl[j][k]);
               fc = fmin(f_clip_max, fmax(f_clip_min, fc));
               f[j][k] = fc;
               sumNeg[j] += fmin(fc, 0.0);
               sumPos[j] += fmax(fc, 0.0);
            }
         }
      }
      wave_reduce_multi<OpSum, U>(sumNeg);
      wave_reduce_multi<OpSum, U>(sumPos);
      static_for<U>([&](auto J) {
         constexpr int j = decltype(J)::value;
         const double sNeg = wave_pick<U, j>(sumNeg), sPos = wave_pick<U, j>(sumPos);
         const double new_mass = sNeg + sPos;
         if (e0 + j < ne)
         {
            const size_t base = (size_t)(e0 + j) * C::D3;
#pragma unroll
            for (int k = 0; k < C::DPT; k++)
            {
               const int i = lane + 64 * k;
               if (i < C::D3)
               {
                  double fc = f[j][k];
                  if (new_mass > eps) { fc = fmin(0.0, fc) - fmax(0.0, fc) * sNeg / sPos; }
                  if (new_mass < -eps) { fc = fmax(0.0, fc) - fmin(0.0, fc) * sPos / sNeg; }
                  const double dui = dl[j][k] + fc / mm[j][k];
                  if (a.du) { store_stream(a.du + base + i, dui); }
                  if (a.y_out)
                  {
                     const double y = uu[j][k] + a.dt_rk * dui;
                     store_stream(a.y_out + base + i, (a.x_base ? a.a * xb[j][k] : 0.0) + a.b * y);
                  }
               }
            }
         }
      });
      if (a.dt_est)
      {
         // (elements past the end are duplicates of the last one: the minimum does not change)
         dtc = wave_minmax<true>(dtc);
         if (lane == 63) { atomic_min_nonneg(a.dt_est, dtc); }
      }
      wave_lds_fence(); // the slots are rewritten by the next pass
   }
}

} // namespace rmh
