// C ABI of the MI355X-native Remhos hot path: context, dispatch over the polynomial order,
// HIP-event stopwatches.  See include/rmh.h for the contract of every entry point.
#include "rmh_ctx.hpp"
#include "rmh_kernels.hpp"
#include "rmh_ho2.hpp"
#include "rmh_stream.hpp"
#include "rmh_2d.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

using namespace rmh;

namespace rmh
{
thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
   g_last_error = msg;
   return code;
}
} // namespace rmh

namespace
{

// the context's element extrema no longer belong to the output of a finished fused stage: no token is valid, and a
// stage that was left unfinished is forgotten
inline void extrema_dropped(rmh_ctx *c)
{
   c->xe_token = 0;
   c->stage_open = false;
}

int timers_resolve(rmh_ctx *c);

int timer_begin(rmh_ctx *c, int bucket, EventPair &ep)
{
   if (!c->timers_on) { return 0; }
   if (c->pool.empty())
   {
      RMH_HIP(hipEventCreate(&ep.a));
      RMH_HIP(hipEventCreate(&ep.b));
   }
   else
   {
      ep = c->pool.back();
      c->pool.pop_back();
   }
   RMH_HIP(hipEventRecord(ep.a, c->stream));
   (void)bucket;
   return 0;
}

int timer_end(rmh_ctx *c, int bucket, EventPair &ep)
{
   if (!c->timers_on) { return 0; }
   RMH_HIP(hipEventRecord(ep.b, c->stream));
   c->pending[bucket].push_back(ep);
   if (c->pending[bucket].size() >= 4096) { return timers_resolve(c); }
   return 0;
}

int timers_resolve(rmh_ctx *c)
{
   RMH_HIP(hipStreamSynchronize(c->stream));
   for (int b = 0; b < 4; b++)
   {
      for (auto &ep : c->pending[b])
      {
         float ms = 0.f;
         RMH_HIP(hipEventElapsedTime(&ms, ep.a, ep.b));
         c->tacc[b] += 1e-3 * ms;
         c->pool.push_back(ep);
      }
      c->pending[b].clear();
   }
   return 0;
}

template <typename T>
int upload(T **dst, const T *src, size_t n)
{
   RMH_HIP(hipMalloc((void **)dst, n * sizeof(T)));
   RMH_HIP(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
   return 0;
}

// elements per workgroup of the one-kernel stage at this order and LO solver (rmh_batch_order)
template <int P>
static int stage_batch_elements(const rmh_ctx *c)
{
   if (c->lo_type == 4 || c->lo_type == 3) { return K2Cfg<(P >= 2 ? P : 2), true, true>::NB; }
   return K2Cfg<P>::NB;
}

// Chunk of the XCD-aware batch order of ho_kernel2 for a launch of nblk batches of NB elements (HoArgs::xcd_chunk, xcd_weave):
// 2^weave lattice layers, woven batch by batch, so that +-z neighbours run at the same time -- inside one XCD (one L2) between
// the woven layers, on the neighbouring XCD (through the Infinity Cache) at the chunk ends.  The layer is taken a little short
// or long where that makes whole rounds of 8 chunks cover the launch.  0 (contiguous eighths) where the element numbering
// shows no layers, a layer is too small to keep an XCD's workgroups inside it, or the launch holds less than one round.
static int xcd_chunk_for(const rmh_ctx *c, int nblk, int NB, int *weave)
{
   *weave = 0;
   if (c->xcd_chunk_env >= 0) { *weave = c->xcd_weave; return (c->xcd_chunk_env >> c->xcd_weave) << c->xcd_weave; }
   if (c->layer_stride <= 0) { return 0; }
   const double layer = (double)c->layer_stride / NB;
   const int q8 = nblk >> 3;
   if (layer < 256.0 || q8 < 2 * layer) { return 0; }
   for (int w = c->xcd_weave; w >= 0; w--)
   {
      const double chunk = layer * (1 << w);
      const int rounds = (int)(q8 / chunk + 0.5);
      if (rounds < 1) { continue; }
      int sub = (int)((q8 / rounds) >> w);                                          // whole rounds cover the launch ...
      if (std::fabs(sub - layer) > 0.005 * layer + 1.0) { sub = (int)(layer + 0.5); } // ... unless that bends the layer by more than 0.5 %
      if (((long long)sub << w) > q8) { continue; }
      *weave = w;
      return sub << w;
   }
   return 0;
}

template <int P>
int create_tables(rmh_ctx *c)
{
   std::vector<double> tab = make_tables<P>();
   static_assert(TabLayout<P>::N3 <= RMH_TAB_STRIDE, "constant table too small");
   RMH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), tab.data(), tab.size() * sizeof(double),
                             (size_t)P * RMH_TAB_STRIDE * sizeof(double), hipMemcpyHostToDevice));
   return upload(&c->d_tab, tab.data(), tab.size());
}

// face speed table (every ho_kernel2 reads it): made on the context's stream the first time such
// a kernel is launched
template <int P>
int ensure_face_table(rmh_ctx *c)
{
   if (c->d_fgeo) { return 0; }
   RMH_HIP(hipMalloc((void **)&c->d_fgeo, (size_t)c->face_slots * FaceGeo<P>::SLOT * sizeof(double)));
   hipLaunchKernelGGL((face_geom_kernel<P>), dim3(c->ne), dim3(256), 0, c->stream, (const double *)c->d_x0,
                      (const double *)c->d_vel, (const double *)c->d_tab, c->exec_mode == 1 ? 1 : 0,
                      (const int *)c->d_face_rows, c->d_fgeo);
   RMH_HIP(hipGetLastError());
   // One node representation on the device (round 6).  The set-up kernels take the NODAL values -- the face speed table has to be
   // the same bits on both sides of a shared face and for every partition of the mesh, and n0 + (n1 - n0) is not n1.  This was the
   // last of them (the lo 4 sub-mesh is set up in rmh_create): the nodal arrays are released, what stays resident is the
   // hierarchical form the stage kernels read (-1.15 GB at -rs 5); rmh_compute_lumped_mass restores nodal values in LDS.  (Once
   // per context, at its first HO / stage launch: one stream synchronisation.)
   RMH_HIP(hipStreamSynchronize(c->stream));
   RMH_HIP(hipFree(c->d_x0));
   RMH_HIP(hipFree(c->d_vel));
   c->d_x0 = c->d_vel = nullptr;
   return 0;
}

template <int P, int MODE>
int launch_ho(rmh_ctx *c, const double *u, double *du, double *m, double t)
{
   HoArgs a;
   a.u = u;
   a.u_ghost = c->u_ghost;
   a.gh_ustride = c->gh_ustride;
   a.gh_mstride = c->gh_mstride;
   a.gh_compact = c->gh_compact;
   a.x0 = c->d_x0h; // (ho_kernel2 takes the nodes in hierarchical form, see RMH_HIER in rmh_ho2.hpp)
   a.vel = c->d_velh;
   a.face_nbr = c->d_nbr;
   a.tab = c->d_tab;
   a.subvel = c->d_subvel;
   a.subx0 = c->d_subx0;
   a.fgeo = nullptr;
   a.face_rows = nullptr;
   a.subvmid = c->d_subvmid;
   a.du = du;
   a.m = m;
   a.xe_min = c->d_xe_min;
   a.xe_max = c->d_xe_max;
   a.cg_iters = c->d_cg;
   a.ne_owned = c->ne;
   a.e_begin = 0;
   a.e_end = c->ne;
   a.bounds_type = 0;
   a.dt_est = nullptr;
   a.rd_subcell = c->lo_type == 3 ? 0 : 1;
   a.t = t;
   a.move = c->exec_mode == 1;
   a.alpha = c->exec_mode == 1 ? 1.0 : -1.0;
   a.upw = c->exec_mode == 1 ? 1.0 : -1.0;
   a.rel2 = c->rel_tol * c->rel_tol;
   a.abs2 = c->abs_tol * c->abs_tol;
   a.max_iter = c->max_iter;
   a.mass_fix = c->mass_fix;
   a.jacobi_step = c->jacobi_step;
   a.stencil27 = c->d_st27;
   a.gh_min = c->gh_min;
   a.gh_max = c->gh_max;
   a.dt = 0.0;
   a.x_base = nullptr;
   a.rk_a = 0.0;
   a.rk_b = 1.0;
   a.dt_rk = 0.0;
   a.y_out = nullptr;
   a.xe_min_out = nullptr;
   a.xe_max_out = nullptr;
   if (MODE == 2)
   {
      constexpr int P2 = P >= 2 ? P : 2; // subcell schemes need order >= 2 (checked by the callers)
      constexpr int NB = K2Cfg<P2, true>::NB;
      // (the face speed table is made the first time a kernel that reads it is launched)
      {
         if (int rc = ensure_face_table<P2>(c)) { return rc; }
         a.fgeo = c->d_fgeo;
         a.face_rows = c->d_face_rows;
      }
      a.xcd_chunk = xcd_chunk_for(c, (c->ne + NB - 1) / NB, NB, &a.xcd_weave);
      a.xcd_rounds = a.xcd_chunk > 0 ? (((c->ne + NB - 1) / NB) >> 3) / a.xcd_chunk : 0;
      a.xcd_inv = a.xcd_chunk > 1 ? (unsigned)(0x100000000ull / (unsigned)a.xcd_chunk) : 0xffffffffu;
      a.reverse = 0;
      hipLaunchKernelGGL((ho_kernel2<P2, 2>), dim3((c->ne + NB - 1) / NB), dim3(K2Cfg<P2, true>::NT), 0, c->stream, a);
   }
   else
   {
      constexpr int NB = K2Cfg<P>::NB;
      // (the face speed table is made the first time a kernel that reads it is launched)
      {
         if (int rc = ensure_face_table<P>(c)) { return rc; }
         a.fgeo = c->d_fgeo;
         a.face_rows = c->d_face_rows;
      }
      a.xcd_chunk = xcd_chunk_for(c, (c->ne + NB - 1) / NB, NB, &a.xcd_weave);
      a.xcd_rounds = a.xcd_chunk > 0 ? (((c->ne + NB - 1) / NB) >> 3) / a.xcd_chunk : 0;
      a.xcd_inv = a.xcd_chunk > 1 ? (unsigned)(0x100000000ull / (unsigned)a.xcd_chunk) : 0xffffffffu;
      a.reverse = 0;
      hipLaunchKernelGGL((ho_kernel2<P, 0>), dim3((c->ne + NB - 1) / NB), dim3(K2Cfg<P>::NT), 0, c->stream, a);
   }
   RMH_HIP(hipGetLastError());
   return 0;
}

template <int P>
int launch_stage_fused(rmh_ctx *c, const double *u, double dt, const double *x_base, double ra, double rb, double dt_rk,
                       double *y_out, double *du, int e_begin, int e_end)
{
   HoArgs a;
   a.u = u;
   a.u_ghost = c->u_ghost;
   a.gh_ustride = c->gh_ustride;
   a.gh_mstride = c->gh_mstride;
   a.gh_compact = c->gh_compact;
   a.x0 = c->d_x0h; // (ho_kernel2 takes the nodes in hierarchical form, see RMH_HIER in rmh_ho2.hpp)
   a.vel = c->d_velh;
   a.face_nbr = c->d_nbr;
   a.tab = c->d_tab;
   a.subvel = c->d_subvel;
   a.subx0 = c->d_subx0;
   a.fgeo = nullptr;
   a.face_rows = nullptr;
   a.subvmid = c->d_subvmid;
   a.du = du;
   a.m = c->d_m;
   a.xe_min = c->d_xe_min;
   a.xe_max = c->d_xe_max;
   a.cg_iters = c->d_cg;
   a.ne_owned = c->ne;
   a.e_begin = e_begin;
   a.e_end = e_end;
   a.t = c->t;
   a.move = c->exec_mode == 1;
   a.alpha = c->exec_mode == 1 ? 1.0 : -1.0;
   a.upw = c->exec_mode == 1 ? 1.0 : -1.0;
   a.rel2 = c->rel_tol * c->rel_tol;
   a.abs2 = c->abs_tol * c->abs_tol;
   a.max_iter = c->max_iter;
   a.mass_fix = c->mass_fix;
   a.jacobi_step = c->jacobi_step;
   a.stencil27 = c->d_st27;
   a.gh_min = c->gh_min;
   a.gh_max = c->gh_max;
   a.dt = dt;
   a.x_base = x_base;
   a.rk_a = ra;
   a.rk_b = rb;
   a.dt_rk = dt_rk;
   a.y_out = y_out;
   a.xe_min_out = c->d_xe_min2;
   a.xe_max_out = c->d_xe_max2;
   a.bounds_type = c->bounds_type;
   a.dt_est = c->dt_control ? c->d_dt_est : nullptr;
   a.rd_subcell = c->lo_type == 3 ? 0 : 1;
   // one workgroup per batch of NB elements (the kernel maps blockIdx.x to a batch XCD-aware)
   if (c->lo_type == 4 || c->lo_type == 3)
   {
      constexpr int P4 = P >= 2 ? P : 2; // subcell schemes need order >= 2 (checked by the caller)
      using C = K2Cfg<P4, true, true>;
      const int nblk = (e_end - e_begin + C::NB - 1) / C::NB;
      const int grid = nblk;
      a.xcd_chunk = xcd_chunk_for(c, nblk, C::NB, &a.xcd_weave);
      a.xcd_rounds = a.xcd_chunk > 0 ? ((nblk) >> 3) / a.xcd_chunk : 0;
      a.xcd_inv = a.xcd_chunk > 1 ? (unsigned)(0x100000000ull / (unsigned)a.xcd_chunk) : 0xffffffffu;
      a.reverse = c->alt_order ? (int)(c->xe_counter & 1) : 0;
      // (the face speed table is made the first time a kernel that reads it is launched)
      {
         if (int rc = ensure_face_table<P4>(c)) { return rc; }
         a.fgeo = c->d_fgeo;
         a.face_rows = c->d_face_rows;
      }
      hipLaunchKernelGGL((ho_kernel2<P4, 3>), dim3(grid), dim3(C::NT), 0, c->stream, a);
   }
   else
   {
      using C = K2Cfg<P>;
      const int nblk = (e_end - e_begin + C::NB - 1) / C::NB;
      const int grid = nblk;
      a.xcd_chunk = xcd_chunk_for(c, nblk, C::NB, &a.xcd_weave);
      a.xcd_rounds = a.xcd_chunk > 0 ? ((nblk) >> 3) / a.xcd_chunk : 0;
      a.xcd_inv = a.xcd_chunk > 1 ? (unsigned)(0x100000000ull / (unsigned)a.xcd_chunk) : 0xffffffffu;
      a.reverse = c->alt_order ? (int)(c->xe_counter & 1) : 0;
      // (the face speed table is made the first time a kernel that reads it is launched)
      {
         if (int rc = ensure_face_table<P>(c)) { return rc; }
         a.fgeo = c->d_fgeo;
         a.face_rows = c->d_face_rows;
      }
      hipLaunchKernelGGL((ho_kernel2<P, 1>), dim3(grid), dim3(C::NT), 0, c->stream, a);
   }
   RMH_HIP(hipGetLastError());
   return 0;
}

#define RMH_DISPATCH(c, expr)                                                                  \
   switch ((c)->p)                                                                             \
   {                                                                                           \
      case 1: { constexpr int P = 1; expr; break; }                                            \
      case 2: { constexpr int P = 2; expr; break; }                                            \
      case 3: { constexpr int P = 3; expr; break; }                                            \
      case 4: { constexpr int P = 4; expr; break; }                                            \
      case 5: { constexpr int P = 5; expr; break; }                                            \
      case 6: { constexpr int P = 6; expr; break; }                                            \
      default: return fail(RMH_ERR_INVALID, "unsupported order");                              \
   }

// entry points that exist for hexahedra only
#define RMH_3D_ONLY(c, name)                                                                                              \
   if ((c)->dim != 3) { return fail(RMH_ERR_INVALID, name ": not available for dim = 2 (HO solver and granular limiter sequence only)"); }

// dim = 2: nodes [ne][2][9], face_nbr [ne][4]; the caller's 3 x 3 element stencil [ne][9] becomes the middle layer of the
// 27-entry table that the bounds kernels read (no neighbours along z)
template <int P>
int create_tables_2d(rmh_ctx *c)
{
   std::vector<double> tab = make_tables_q<P, P + 2>();
   return upload(&c->d_tab, tab.data(), tab.size());
}

int create_device_state_2d(rmh_ctx *c, const rmh_layout *L)
{
   const size_t ne = c->ne;
   int rc = 0;
   if ((rc = upload(&c->d_x0, L->x0, ne * 18))) { return rc; }
   if ((rc = upload(&c->d_vel, L->vel, ne * 18))) { return rc; }
   if ((rc = upload(&c->d_nbr, L->face_nbr, ne * 4))) { return rc; }
   std::vector<int> st(ne * 27, -1);
   for (size_t e = 0; e < ne; e++)
   {
      for (int k = 0; k < 9; k++) { st[e * 27 + 9 + k] = L->stencil27[e * 9 + k]; }
   }
   if ((rc = upload(&c->d_st27, st.data(), ne * 27))) { return rc; }
   if (L->subcell_vel && (rc = upload(&c->d_subvel, L->subcell_vel, ne * 2 * c->ndof))) { return rc; }
   RMH_DISPATCH(c, rc = create_tables_2d<P>(c));
   if (rc) { return rc; }
   RMH_HIP(hipMalloc((void **)&c->d_m, ne * c->ndof * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&c->d_xe_min, ne * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&c->d_xe_max, ne * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&c->d_cg, sizeof(int)));
   RMH_HIP(hipMemset(c->d_cg, 0, sizeof(int)));
   return 0;
}

template <int P, int MODE>
int launch_ho_2d(rmh_ctx *c, const double *u, double *du, double *m, double t)
{
   Ho2Args a;
   a.u = u;
   a.x0 = c->d_x0;
   a.vel = c->d_vel;
   a.face_nbr = c->d_nbr;
   a.tab = c->d_tab;
   a.du = du;
   a.m = m;
   a.xe_min = MODE != 4 ? c->d_xe_min : nullptr;
   a.xe_max = MODE != 4 ? c->d_xe_max : nullptr;
   a.cg_iters = c->d_cg;
   a.t = t;
   a.move = c->exec_mode == 1 ? 1 : 0;
   a.alpha = c->exec_mode == 1 ? 1.0 : -1.0; // remhos.cpp:648-657
   a.upw = c->exec_mode == 1 ? 1.0 : -1.0;
   a.rel2 = c->rel_tol * c->rel_tol;
   a.abs2 = c->abs_tol * c->abs_tol;
   a.max_iter = c->max_iter;
   a.jacobi_step = c->jacobi_step;
   a.mass_fix = c->mass_fix;
   a.subvel = c->d_subvel;
   a.rd_subcell = c->lo_type == 3 ? 0 : 1;
   hipLaunchKernelGGL((ho2d_kernel<P, MODE>), dim3(c->ne), dim3(64), 0, c->stream, a);
   RMH_HIP(hipGetLastError());
   return 0;
}

// device buffers, tables and lo 4 set-up data of a new context (rmh_create releases the context when this fails)
int create_device_state(rmh_ctx *c, const rmh_layout *L)
{
   const size_t ne = c->ne;
   int rc = 0;
   if ((rc = upload(&c->d_x0, L->x0, ne * 81))) { return rc; }
   if ((rc = upload(&c->d_vel, L->vel, ne * 81))) { return rc; }
   {
      // the same nodes in the hierarchical form of the quadratic Lagrange basis along the directions of the mask RMH_HIER, for
      // ho_kernel2: (n0, n1 - n0, n2 - n0).  The set-up kernels (face table, sub-mesh, lumped mass) keep the nodal arrays.
      std::vector<double> h(ne * 81);
      for (const double *src : {L->x0, L->vel})
      {
         std::copy(src, src + ne * 81, h.begin());
         for (size_t b = 0; b < ne * 3; b++) // (element, component) blocks of 27 nodes, index a + 3 b + 9 c
         {
            double *n = h.data() + b * 27;
            for (int dir = 0; dir < 3; dir++)
            {
               if (!((RMH_HIER >> dir) & 1)) { continue; }
               const int st = dir == 0 ? 1 : (dir == 1 ? 3 : 9);
               for (int k = 0; k < 27; k++)
               {
                  if ((k / st) % 3 == 0) { n[k + st] -= n[k]; n[k + 2 * st] -= n[k]; }
               }
            }
         }
         if ((rc = upload(src == L->x0 ? &c->d_x0h : &c->d_velh, h.data(), ne * 81))) { return rc; }
      }
   }
   if ((rc = upload(&c->d_nbr, L->face_nbr, ne * 6))) { return rc; }
   if ((rc = upload(&c->d_st27, L->stencil27, ne * 27))) { return rc; }
   {
      // Face speed table, one block per face (FaceGeo, rmh_ho2.hpp): the low face c of element e owns block 3 e + c; the
      // high face reads the block of the neighbour's low face with the sign flipped where that neighbour is an owned
      // element that names e back and the nine face nodes of x0 and v agree BIT BY BIT on the two sides (then the two rows
      // are the same numbers with opposite signs); otherwise -- boundary, ghost neighbour, a periodic seam of a remap run --
      // it keeps a block of its own behind the regular ones.
      std::vector<int> rows(ne * 6);
      long long slots = 3 * (long long)ne;
      const int st3[3] = {1, 3, 9};
      for (size_t e = 0; e < ne; e++)
      {
         for (int cdir = 0; cdir < 3; cdir++)
         {
            rows[e * 6 + 2 * cdir] = (int)(3 * e + cdir);
            const int nb = L->face_nbr[e * 6 + 2 * cdir + 1];
            bool shared = nb >= 0 && (size_t)nb < ne && L->face_nbr[(size_t)nb * 6 + 2 * cdir] == (int)e && (size_t)nb != e;
            if (shared)
            {
               const int n1 = st3[(cdir + 1) % 3], n2 = st3[(cdir + 2) % 3], hi = 2 * st3[cdir];
               for (const double *arr : {L->x0, L->vel})
               {
                  const double *me = arr + e * 81, *ot = arr + (size_t)nb * 81;
                  for (int comp = 0; comp < 3 && shared; comp++)
                  {
                     for (int a = 0; a < 9; a++)
                     {
                        const int off = comp * 27 + (a % 3) * n1 + (a / 3) * n2;
                        if (std::memcmp(&me[off + hi], &ot[off], sizeof(double)) != 0) { shared = false; break; }
                     }
                  }
               }
            }
            if (slots >= 0x7fffffffLL) { return fail(RMH_ERR_INVALID, "too many faces for the 31-bit block index of the face speed table"); }
            rows[e * 6 + 2 * cdir + 1] = shared ? (int)((3u * (unsigned)nb + (unsigned)cdir) | 0x80000000u) : (int)slots++;
         }
      }
      c->face_slots = slots;
      if ((rc = upload(&c->d_face_rows, rows.data(), ne * 6))) { return rc; }
   }
   if (L->subcell_vel)
   {
      if ((rc = upload(&c->d_subvel, L->subcell_vel, ne * 3 * c->ndof))) { return rc; }
   }
   RMH_DISPATCH(c, rc = create_tables<P>(c));
   if (rc) { return rc; }
   if (c->d_subvel)
   {
      const size_t ns = (size_t)c->p * c->p * c->p;
      RMH_HIP(hipMalloc((void **)&c->d_subx0, ne * 3 * c->ndof * sizeof(double)));
      RMH_HIP(hipMalloc((void **)&c->d_subvmid, ne * 3 * ns * sizeof(double)));
      RMH_DISPATCH(c, hipLaunchKernelGGL((subcell_setup_kernel<P>), dim3(c->ne), dim3(64), 0, nullptr, (const double *)c->d_x0,
                                         (const double *)c->d_subvel, (const double *)c->d_tab, c->d_subx0, c->d_subvmid));
      RMH_HIP(hipGetLastError());
      RMH_HIP(hipDeviceSynchronize());
   }
   RMH_HIP(hipMalloc((void **)&c->d_m, ne * c->ndof * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&c->d_xe_min, ne * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&c->d_xe_max, ne * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&c->d_xe_min2, ne * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&c->d_xe_max2, ne * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&c->d_cg, sizeof(int)));
   RMH_HIP(hipMemset(c->d_cg, 0, sizeof(int)));
   return 0;
}

} // namespace

#ifdef RMH_STAMPS
// sums the per-workgroup stamp rows into out[32]; reset != 0 clears them afterwards
extern "C" int rmh_debug_stamps(unsigned long long *out, int reset)
{
   const size_t n = (size_t)rmh::RMH_STAMP_MAXWG * 32;
   std::vector<unsigned long long> h(n);
   if (hipDeviceSynchronize() != hipSuccess) { return -1; }
   if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(rmh::g_stamps), n * sizeof(unsigned long long)) != hipSuccess) { return -1; }
   for (int k = 0; k < 32; k++) { out[k] = 0; }
   for (size_t i = 0; i < n; i++) { out[i % 32] += h[i]; }
   if (reset)
   {
      void *p = nullptr;
      if (hipGetSymbolAddress(&p, HIP_SYMBOL(rmh::g_stamps)) != hipSuccess) { return -1; }
      if (hipMemset(p, 0, n * sizeof(unsigned long long)) != hipSuccess) { return -1; }
   }
   return 0;
}
#endif

// Ghost data (neighbour traces, ghost element extrema) is valid between rmh_exchange_end and the next rmh_exchange_begin,
// and -- for the fused limiter and the one-kernel stage, which take the extrema of u from the ghost slots -- only if no
// rmh_exchange_minmax_* has put another field's extrema there since.  Only element ranges that reach a ghost reader are held to
// that: the interior range of a split stage is launched while the exchange is in flight.  The granular ghost readers
// (rmh_ho_apply, rmh_lo_rd*: neighbour traces; rmh_bounds: whatever extrema the caller exchanged last, incl. another field's)
// are refused while an exchange of u is in flight only.
static const char *ghosts_not_ready(const rmh_ctx *c, int e_begin); // (both defined behind rmh_comm.hpp, which has struct Exchange)
static const char *ghosts_in_flight(const rmh_ctx *c);

extern "C" {

const char *rmh_last_error(void) { return g_last_error.c_str(); }
const char *rmh_version(void) { return "remhos_amd 0.1 (gfx950)"; }

int rmh_create(const rmh_layout *L, rmh_ctx **out)
{
   if (!L || !out) { return fail(RMH_ERR_INVALID, "null argument"); }
   *out = nullptr;
   if (L->dim != 3 && L->dim != 2) { return fail(RMH_ERR_INVALID, "dim must be 3 (or 2: HO solver and granular limiter sequence)"); }
   const int nf = 2 * L->dim, nst = L->dim == 3 ? 27 : 9;
   if (L->dim == 2 && L->ne_ghost != 0) { return fail(RMH_ERR_INVALID, "dim = 2: single rank (ne_ghost = 0)"); }
   if (L->order < 1 || L->order > 6) { return fail(RMH_ERR_INVALID, "order must be in 1..6"); }
   if (L->mesh_order != 2) { return fail(RMH_ERR_INVALID, "mesh_order must be 2"); }
   if (L->ne_owned <= 0 || L->ne_ghost < 0) { return fail(RMH_ERR_INVALID, "bad element counts"); }
   if (!L->x0 || !L->vel || !L->face_nbr || !L->stencil27) { return fail(RMH_ERR_INVALID, "null mesh array"); }
   // the kernels index u, the ghost block and the element extrema with these tables: reject anything out of range
   {
      const long long ntot = (long long)L->ne_owned + L->ne_ghost;
      for (long long e = 0; e < L->ne_owned; e++)
      {
         for (int f = 0; f < nf; f++)
         {
            const int nb = L->face_nbr[e * nf + f];
            if (nb < -1 || nb >= ntot) { return fail(RMH_ERR_INVALID, "face_nbr entry out of range"); }
         }
         for (int k = 0; k < nst; k++)
         {
            const int nb = L->stencil27[e * nst + k];
            if (nb < -1 || nb >= ntot) { return fail(RMH_ERR_INVALID, "stencil27 entry out of range"); }
         }
         if (L->stencil27[e * nst + nst / 2] != e) { return fail(RMH_ERR_INVALID, "the centre entry of an element's stencil must be the element itself"); }
      }
   }
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
   {
      return fail(RMH_ERR_NO_DEVICE, "no HIP device: the remhos_amd hot path has no CPU fallback");
   }
   if (L->device < 0 || L->device >= ndev) { return fail(RMH_ERR_INVALID, "bad device ordinal"); }
   RMH_HIP(hipSetDevice(L->device)); // (stays current: every entry point makes its context's device current)
   rmh_ctx *c = new rmh_ctx;
   c->p = L->order;
   c->ne = L->ne_owned;
   c->ng = L->ne_ghost;
   c->exec_mode = L->exec_mode;
   c->device = L->device;
   c->dim = L->dim;
   c->ndof = (c->p + 1) * (c->p + 1) * (c->dim == 3 ? c->p + 1 : 1);
   c->gh_ustride = c->ndof;
   if (c->ng > 0)
   {
      const int nst = c->dim == 3 ? 27 : 9;
      for (int e = c->ne - 1; e >= 0 && c->ghost_readers_end == 0; e--)
      {
         for (int k = 0; k < nst; k++)
         {
            if (L->stencil27[(size_t)e * nst + k] >= c->ne) { c->ghost_readers_end = e + 1; break; }
         }
      }
   }
   if (c->dim == 3)
   {
      // the usual distance of an element to its +z neighbour (lattice numberings: the elements of one layer), for xcd_chunk_for
      std::unordered_map<int, int> hist;
      for (int e = 0; e < c->ne; e++)
      {
         const int nb = L->face_nbr[(size_t)e * 6 + 5];
         if (nb > e && nb < c->ne) { hist[nb - e]++; }
      }
      int best = 0, cnt = 0;
      for (const auto &kv : hist) { if (kv.second > cnt || (kv.second == cnt && kv.first < best)) { best = kv.first; cnt = kv.second; } }
      c->layer_stride = 2 * (long long)cnt > c->ne ? best : 0; // (only where most elements agree)
      if (const char *s = std::getenv("RMH_XCD_CHUNK")) { c->xcd_chunk_env = std::max(0, std::atoi(s)); }
      if (const char *s = std::getenv("RMH_ALT_ORDER")) { c->alt_order = std::atoi(s) != 0; }
      if (const char *s = std::getenv("RMH_XCD_WEAVE")) { c->xcd_weave = std::min(4, std::max(0, std::atoi(s))); }
   }
   // every failure past this point releases the context and whatever it already owns
   const int rc = c->dim == 3 ? create_device_state(c, L) : create_device_state_2d(c, L);
   if (rc) { rmh_destroy(c); return rc; }
   *out = c;
   return RMH_OK;
}

void rmh_destroy(rmh_ctx *c)
{
   if (!c) { return; }
   (void)hipSetDevice(c->device);
   exchange_free(c);
   void *bufs[] = {c->d_x0, c->d_vel, c->d_x0h, c->d_velh, c->d_tab, c->d_subvel, c->d_subx0, c->d_subvmid, c->d_fgeo, c->d_face_rows, c->d_m, c->d_scr_ho, c->d_scr_lo, c->d_xe_min, c->d_xe_max, c->d_xe_min2, c->d_xe_max2, c->d_nbr, c->d_st27, c->d_cg, c->d_dt_est, c->d_viol};
   for (void *b : bufs) { (void)hipFree(b); }
   for (int b = 0; b < 4; b++)
   {
      for (auto &ep : c->pending[b]) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
   }
   for (auto &ep : c->pool) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
   delete c;
}

int rmh_set_stream(rmh_ctx *c, void *s)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   c->stream = (hipStream_t)s;
   return RMH_OK;
}

int rmh_batch_order(rmh_ctx *c, int n_elements, int *layer_elements, int *batch_elements, int *chunk, int *weave)
{
   if (!c || !layer_elements || !batch_elements || !chunk || !weave || n_elements < 0) { return fail(RMH_ERR_INVALID, "rmh_batch_order: bad argument"); }
   RMH_3D_ONLY(c, "rmh_batch_order");
   int nb = 1;
   RMH_DISPATCH(c, nb = stage_batch_elements<P>(c));
   *layer_elements = c->layer_stride;
   *batch_elements = nb;
   *chunk = xcd_chunk_for(c, (n_elements + nb - 1) / nb, nb, weave);
   return RMH_OK;
}

int rmh_stream_create_reserving(int device, int reserve_cus, void **hip_stream)
{
   if (!hip_stream || reserve_cus < 0) { return fail(RMH_ERR_INVALID, "rmh_stream_create_reserving: bad argument"); }
   *hip_stream = nullptr;
   int ndev = 0;
   if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { return fail(RMH_ERR_INVALID, "bad device ordinal"); }
   RMH_HIP(hipSetDevice(device));
   hipStream_t s = nullptr;
   if (reserve_cus == 0)
   {
      RMH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      *hip_stream = s;
      return RMH_OK;
   }
   hipDeviceProp_t prop;
   RMH_HIP(hipGetDeviceProperties(&prop, device));
   const int ncu = prop.multiProcessorCount;
   if (reserve_cus >= ncu) { return fail(RMH_ERR_INVALID, "rmh_stream_create_reserving: cannot reserve every compute unit"); }
   std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
   for (int b = 0; b < ncu; b++) { mask[b / 32] |= 1u << (b % 32); }
   // bit i * (ncu / k) + i, i < k: one CU per XCD for k = 8 whether the mask enumerates the CUs XCD by XCD (32 consecutive
   // bits each) or round-robin over the XCDs
   const int stride = ncu / reserve_cus;
   for (int i = 0; i < reserve_cus; i++)
   {
      const int b = (i * stride + i % (stride > 0 ? stride : 1)) % ncu;
      mask[b / 32] &= ~(1u << (b % 32));
   }
   RMH_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
   *hip_stream = s;
   return RMH_OK;
}

int rmh_stream_destroy(void *hip_stream)
{
   if (hip_stream) { RMH_HIP(hipStreamDestroy((hipStream_t)hip_stream)); }
   return RMH_OK;
}

int rmh_setup(rmh_ctx *c, double t)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   c->t = t;
   c->ho_done = false;
   return RMH_OK;
}

int rmh_set_ghost_u(rmh_ctx *c, const double *ug)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   if (c->xch) { return fail(RMH_ERR_STATE, "the ghosts of this context belong to its exchange plan (rmh_exchange_setup)"); }
   c->u_ghost = ug;
   c->gh_ustride = c->ndof;
   return RMH_OK;
}

int rmh_set_ghost_minmax(rmh_ctx *c, const double *gmin, const double *gmax)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   if (c->xch) { return fail(RMH_ERR_STATE, "the ghosts of this context belong to its exchange plan (rmh_exchange_setup)"); }
   c->gh_min = gmin;
   c->gh_max = gmax;
   c->gh_mstride = 1;
   return RMH_OK;
}

int rmh_set_ghost_records(rmh_ctx *c, const double *rec)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   if (c->xch) { return fail(RMH_ERR_STATE, "the ghosts of this context belong to its exchange plan (rmh_exchange_setup)"); }
   c->u_ghost = rec;
   c->gh_min = rec ? rec + c->ndof : nullptr;
   c->gh_max = rec ? rec + c->ndof + 1 : nullptr;
   c->gh_ustride = c->gh_mstride = c->ndof + 2;
   return RMH_OK;
}

int rmh_ho_apply(rmh_ctx *c, const double *u, double *du)
{
   if (!c || !u || !du) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   if (c->ng > 0 && !c->u_ghost) { return fail(RMH_ERR_STATE, "ghost values of u not set"); }
   if (const char *why = ghosts_in_flight(c)) { return fail(RMH_ERR_STATE, why); }
   EventPair ep;
   int rc = timer_begin(c, 0, ep);
   if (rc) { return rc; }
   if (c->dim == 2) { RMH_DISPATCH(c, rc = (launch_ho_2d<P, 0>(c, u, du, c->d_m, c->t))); }
   else { RMH_DISPATCH(c, rc = (launch_ho<P, 0>(c, u, du, c->d_m, c->t))); }
   if (rc) { return rc; }
   rc = timer_end(c, 0, ep);
   c->ho_done = true;
   extrema_dropped(c);
   return rc;
}

const double *rmh_lumped_mass(rmh_ctx *c) { return c ? c->d_m : nullptr; }

int rmh_compute_lumped_mass(rmh_ctx *c, double t, double *m)
{
   if (!c || !m) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   int rc = 0;
   if (c->dim == 2)
   {
      RMH_DISPATCH(c, rc = (launch_ho_2d<P, 4>(c, nullptr, nullptr, m, t)));
      return rc;
   }
   RMH_DISPATCH(c, hipLaunchKernelGGL((lumped_mass_kernel<P>), dim3(c->ne), dim3(KCfg<P>::NT), 0, c->stream,
                                      (const double *)c->d_x0h, (const double *)c->d_velh, (const double *)c->d_tab, t,
                                      c->exec_mode == 1 ? 1 : 0, (int)RMH_HIER, m));
   RMH_HIP(hipGetLastError());
   return rc;
}

int rmh_lo_massavg(rmh_ctx *c, const double *u, const double *du_ho, double dt, double *du_lo)
{
   if (!c || !u || !du_ho || !du_lo) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   extrema_dropped(c);
   if (!c->ho_done) { return fail(RMH_ERR_STATE, "rmh_lo_massavg needs the lumped mass of rmh_ho_apply"); }
   EventPair ep;
   int rc = timer_begin(c, 2, ep);
   if (rc) { return rc; }
   if (c->dim == 2)
   {
      RMH_DISPATCH(c, hipLaunchKernelGGL((lo_massavg_kernel<P, 2>), dim3(c->ne), dim3(KCfg<P, 2>::NT), 0, c->stream, u,
                                         du_ho, (const double *)c->d_m, dt, du_lo));
   }
   else
   {
      RMH_DISPATCH(c, hipLaunchKernelGGL((lo_massavg_kernel<P>), dim3(c->ne), dim3(KCfg<P>::NT), 0, c->stream, u,
                                         du_ho, (const double *)c->d_m, dt, du_lo));
   }
   RMH_HIP(hipGetLastError());
   return timer_end(c, 2, ep);
}

namespace
{
int lo_rd(rmh_ctx *c, const double *u, double *du_lo, int lo_type)
{
   RMH_ENTER(c);
   extrema_dropped(c);
   EventPair ep;
   int rc = timer_begin(c, 2, ep);
   if (rc) { return rc; }
   const int keep = c->lo_type;
   c->lo_type = lo_type;
   if (c->dim == 2) { RMH_DISPATCH(c, rc = (launch_ho_2d<P, 2>(c, u, du_lo, c->d_m, c->t))); }
   else { RMH_DISPATCH(c, rc = (launch_ho<P, 2>(c, u, du_lo, c->d_m, c->t))); }
   c->lo_type = keep;
   if (rc) { return rc; }
   c->ho_done = true; // lumped mass and element extrema are current
   return timer_end(c, 2, ep);
}
} // namespace

int rmh_lo_rdsubcell(rmh_ctx *c, const double *u, double *du_lo)
{
   if (!c || !u || !du_lo) { return fail(RMH_ERR_INVALID, "null argument"); }
   if (c->p < 2) { return fail(RMH_ERR_INVALID, "Subcell schemes require FE order > 2."); } // remhos.cpp:612-616
   if (!c->d_subvel) { return fail(RMH_ERR_STATE, "rmh_lo_rdsubcell needs rmh_layout.subcell_vel"); }
   if (c->ng > 0 && !c->u_ghost) { return fail(RMH_ERR_STATE, "ghost values of u not set"); }
   return lo_rd(c, u, du_lo, 4);
}

int rmh_lo_rd(rmh_ctx *c, const double *u, double *du_lo)
{
   if (!c || !u || !du_lo) { return fail(RMH_ERR_INVALID, "null argument"); }
   if (c->p < 2) { return fail(RMH_ERR_INVALID, "rmh_lo_rd: the RD kernel is built for orders >= 2"); }
   if (c->ng > 0 && !c->u_ghost) { return fail(RMH_ERR_STATE, "ghost values of u not set"); }
   return lo_rd(c, u, du_lo, 3);
}

int rmh_elem_minmax(rmh_ctx *c, const double *u, double *xe_min, double *xe_max)
{
   if (!c || !u || !xe_min || !xe_max) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   if (c->dim == 2)
   {
      RMH_DISPATCH(c, hipLaunchKernelGGL((elem_minmax_kernel<P, 2>), dim3(SCfg<P, 2>::grid(c->ne, SCfg<P, 2>::U8)),
                                         dim3(SCfg<P, 2>::NT), 0, c->stream, u, xe_min, xe_max, c->ne));
   }
   else
   {
      RMH_DISPATCH(c, hipLaunchKernelGGL((elem_minmax_kernel<P>), dim3(SCfg<P>::grid(c->ne, SCfg<P>::U8)), dim3(SCfg<P>::NT), 0,
                                         c->stream, u, xe_min, xe_max, c->ne));
   }
   RMH_HIP(hipGetLastError());
   return RMH_OK;
}

int rmh_bounds(rmh_ctx *c, const double *xe_min, const double *xe_max, double *u_min, double *u_max)
{
   if (!c || !xe_min || !xe_max || !u_min || !u_max) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   extrema_dropped(c);
   if (c->ng > 0 && (!c->gh_min || !c->gh_max)) { return fail(RMH_ERR_STATE, "ghost extrema not set"); }
   if (const char *why = ghosts_in_flight(c)) { return fail(RMH_ERR_STATE, why); }
   const int wide = (((uintptr_t)u_min | (uintptr_t)u_max) & 15) == 0; // 16-byte stores
   if (c->dim == 2)
   {
      RMH_DISPATCH(c, hipLaunchKernelGGL((bounds_kernel<P, 2>), dim3(SCfg<P, 2>::grid(c->ne, SCfg<P, 2>::UB)), dim3(SCfg<P, 2>::NT), 0, c->stream,
                                         c->bounds_type, (const int *)c->d_st27, c->ne, xe_min, xe_max, c->gh_min, c->gh_max, c->gh_mstride, u_min,
                                         u_max, wide));
   }
   else
   {
      RMH_DISPATCH(c, hipLaunchKernelGGL((bounds_kernel<P>), dim3(SCfg<P>::grid(c->ne, SCfg<P>::UB)), dim3(SCfg<P>::NT), 0, c->stream,
                                         c->bounds_type, (const int *)c->d_st27, c->ne, xe_min, xe_max, c->gh_min, c->gh_max, c->gh_mstride, u_min,
                                         u_max, wide));
   }
   RMH_HIP(hipGetLastError());
   return RMH_OK;
}

int rmh_fct_clipscale(rmh_ctx *c, const double *u, const double *m, const double *du_ho, const double *du_lo,
                      const double *u_min, const double *u_max, double dt, double *du)
{
   if (!c || !u || !m || !du_ho || !du_lo || !u_min || !u_max || !du)
   {
      return fail(RMH_ERR_INVALID, "null argument");
   }
   RMH_ENTER(c);
   extrema_dropped(c);
   EventPair ep;
   int rc = timer_begin(c, 3, ep);
   if (rc) { return rc; }
   if (c->dim == 2)
   {
      RMH_DISPATCH(c, hipLaunchKernelGGL((fct_clipscale_kernel<P, 2>), dim3(c->ne), dim3(KCfg<P, 2>::NT), 0, c->stream, u, m,
                                         du_ho, du_lo, u_min, u_max, dt, du));
   }
   else
   {
      RMH_DISPATCH(c, hipLaunchKernelGGL((fct_clipscale_kernel<P>), dim3(c->ne), dim3(KCfg<P>::NT), 0, c->stream, u, m,
                                         du_ho, du_lo, u_min, u_max, dt, du));
   }
   RMH_HIP(hipGetLastError());
   return timer_end(c, 3, ep);
}

int rmh_product_ratio(rmh_ctx *c, const double *us, const double *u, double *s, unsigned char *active_el,
                      unsigned char *active_dofs)
{
   if (!c || !u || !active_el || !active_dofs || ((us != nullptr) != (s != nullptr))) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_3D_ONLY(c, "rmh_product_ratio");
   RMH_ENTER(c);
   extrema_dropped(c);
   RMH_DISPATCH(c, hipLaunchKernelGGL((product_ratio_kernel<P>), dim3(c->ne), dim3(KCfg<P>::NT), 0, c->stream, us, u, s,
                                      active_el, active_dofs));
   RMH_HIP(hipGetLastError());
   return RMH_OK;
}

int rmh_elem_minmax_masked(rmh_ctx *c, const double *u, const unsigned char *active_el, const unsigned char *active_dofs,
                           double *xe_min, double *xe_max)
{
   if (!c || !u || !active_el || !active_dofs || !xe_min || !xe_max) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_3D_ONLY(c, "rmh_elem_minmax_masked");
   RMH_ENTER(c);
   RMH_DISPATCH(c, hipLaunchKernelGGL((elem_minmax_masked_kernel<P>), dim3(c->ne), dim3(KCfg<P>::NT), 0, c->stream, u,
                                      active_el, active_dofs, xe_min, xe_max));
   RMH_HIP(hipGetLastError());
   return RMH_OK;
}

int rmh_fct_product(rmh_ctx *c, const double *us, const double *m, const double *d_us_ho, double *s_min, double *s_max,
                    const double *u_new, const unsigned char *active_el, const unsigned char *active_dofs, double dt,
                    double *d_us)
{
   if (!c || !us || !m || !d_us_ho || !s_min || !s_max || !u_new || !active_el || !active_dofs || !d_us)
   {
      return fail(RMH_ERR_INVALID, "null argument");
   }
   RMH_3D_ONLY(c, "rmh_fct_product");
   if (!(dt > 0.0)) { return fail(RMH_ERR_INVALID, "dt must be positive"); }
   RMH_ENTER(c);
   extrema_dropped(c);
   EventPair ep;
   int rc = timer_begin(c, 3, ep);
   if (rc) { return rc; }
   RMH_DISPATCH(c, hipLaunchKernelGGL((fct_product_kernel<P>), dim3(c->ne), dim3(KCfg<P>::NT), 0, c->stream, us, m, d_us_ho,
                                      s_min, s_max, u_new, active_el, active_dofs, dt, d_us));
   RMH_HIP(hipGetLastError());
   return timer_end(c, 3, ep);
}


static int limit_fused_impl(rmh_ctx *c, const double *u, const double *du_ho, const double *du_lo, double dt, double *du,
                            const double *x_base, double a, double b, double dt_rk, double *y_out)
{
   if (!c || !u || !du_ho || (!du && !y_out)) { return fail(RMH_ERR_INVALID, "null argument"); }
   if (!c->ho_done) { return fail(RMH_ERR_STATE, "rmh_limit_fused must follow rmh_ho_apply on the same u"); }
   RMH_ENTER(c);
   extrema_dropped(c);
   if (c->ng > 0 && (!c->gh_min || !c->gh_max)) { return fail(RMH_ERR_STATE, "ghost extrema not set"); }
   if (const char *why = ghosts_not_ready(c, 0)) { return fail(RMH_ERR_STATE, why); }
   LimitArgs la;
   la.u = u;
   la.du_ho = du_ho;
   la.m = c->d_m;
   la.du_lo = du_lo;
   la.stencil27 = c->d_st27;
   la.xe_min = c->d_xe_min;
   la.xe_max = c->d_xe_max;
   la.gh_min = c->gh_min;
   la.gh_max = c->gh_max;
   la.gh_mstride = c->gh_mstride;
   la.bounds_type = c->bounds_type;
   la.dt_est = c->dt_control ? c->d_dt_est : nullptr;
   la.ne_owned = c->ne;
   la.dt = dt;
   la.du = du;
   la.x_base = x_base;
   la.a = a;
   la.b = b;
   la.dt_rk = dt_rk;
   la.y_out = y_out;
   EventPair ep;
   int rc = timer_begin(c, 3, ep);
   if (rc) { return rc; }
   if (c->dim == 2) {   RMH_DISPATCH(c, hipLaunchKernelGGL((limit_fused_kernel<P, 2>), dim3(SCfg<P, 2>::grid(c->ne, SCfg<P, 2>::U4)), dim3(SCfg<P, 2>::NT), 0, c->stream, la)); }
   else {   RMH_DISPATCH(c, hipLaunchKernelGGL((limit_fused_kernel<P>), dim3(SCfg<P>::grid(c->ne, SCfg<P>::U4)), dim3(SCfg<P>::NT), 0, c->stream, la)); }
   RMH_HIP(hipGetLastError());
   return timer_end(c, 3, ep);
}

int rmh_limit_fused(rmh_ctx *c, const double *u, const double *du_ho, double dt, double *du, const double *x_base,
                    double a, double b, double dt_rk, double *y_out)
{
   return limit_fused_impl(c, u, du_ho, nullptr, dt, du, x_base, a, b, dt_rk, y_out);
}

int rmh_limit_fused_lo(rmh_ctx *c, const double *u, const double *du_ho, const double *du_lo, double dt, double *du,
                       const double *x_base, double a, double b, double dt_rk, double *y_out)
{
   if (!du_lo) { return fail(RMH_ERR_INVALID, "null du_lo"); }
   return limit_fused_impl(c, u, du_ho, du_lo, dt, du, x_base, a, b, dt_rk, y_out);
}

int rmh_stage_fused_chain(rmh_ctx *c, const double *u, double dt, const double *x_base, double a, double b, double dt_rk,
                          double *y_out, double *du, int e_begin, int e_end, int finish, unsigned long long in_token,
                          unsigned long long *out_token)
{
   if (out_token) { *out_token = 0; }
   if (!c || !u || !y_out) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   if (!(dt > 0.0)) { return fail(RMH_ERR_INVALID, "dt must be positive"); }
   if (y_out == u || du == u) { return fail(RMH_ERR_INVALID, "rmh_stage_fused: the output must not alias u"); }
   if (c->dim == 2)
   {
      // dim = 2: the whole stage as a sequence inside the library -- HO kernel (lumped mass, element extrema), the RD solver
      // for lo 3 / 4, then the fused limiter with the RK update (rmh_2d.hpp, rmh_stream.hpp); whole rank only, no tokens
      if (e_begin != 0 || e_end != c->ne || !finish) { return fail(RMH_ERR_INVALID, "rmh_stage_fused_range: dim = 2 runs the whole rank in one call"); }
      if ((c->lo_type == 3 || c->lo_type == 4) && c->p < 2) { return fail(RMH_ERR_STATE, "rmh_stage_fused with lo 3 / 4 needs order >= 2"); }
      if (c->lo_type == 4 && !c->d_subvel) { return fail(RMH_ERR_STATE, "rmh_stage_fused with lo 4 needs rmh_layout.subcell_vel"); }
      const size_t n = (size_t)c->ne * c->ndof;
      if (!c->d_scr_ho) { RMH_HIP(hipMalloc((void **)&c->d_scr_ho, n * sizeof(double))); }
      if (c->lo_type != 5 && !c->d_scr_lo) { RMH_HIP(hipMalloc((void **)&c->d_scr_lo, n * sizeof(double))); }
      int rc2 = rmh_ho_apply(c, u, c->d_scr_ho);
      if (rc2) { return rc2; }
      if (c->lo_type == 4) { rc2 = rmh_lo_rdsubcell(c, u, c->d_scr_lo); }
      else if (c->lo_type == 3) { rc2 = rmh_lo_rd(c, u, c->d_scr_lo); }
      if (rc2) { return rc2; }
      return c->lo_type == 5 ? rmh_limit_fused(c, u, c->d_scr_ho, dt, du, x_base, a, b, dt_rk, y_out)
                             : rmh_limit_fused_lo(c, u, c->d_scr_ho, c->d_scr_lo, dt, du, x_base, a, b, dt_rk, y_out);
   }
   if (e_begin < 0 || e_end > c->ne || e_begin > e_end) { return fail(RMH_ERR_INVALID, "rmh_stage_fused_range: bad element range"); }
   if (c->lo_type == 4 && (c->p < 2 || !c->d_subvel))
   {
      return fail(RMH_ERR_STATE, "rmh_stage_fused with lo 4 needs order >= 2 and rmh_layout.subcell_vel");
   }
   if (c->lo_type == 3 && c->p < 2) { return fail(RMH_ERR_STATE, "rmh_stage_fused with lo 3 needs order >= 2"); }
   if (c->ng > 0 && (!c->u_ghost || !c->gh_min || !c->gh_max)) { return fail(RMH_ERR_STATE, "ghost data not set"); }
   if (const char *why = ghosts_not_ready(c, e_begin)) { return fail(RMH_ERR_STATE, why); }
   int rc = 0;
   // Element extrema of the stage input.  They are at hand only if the caller PRESENTS the token the stage that wrote u
   // returned -- its statement that u is that stage's untouched output.  Anything else (no token, a stale one, a vector
   // that was modified since) costs one streaming pass.  The ranges of one stage share the extrema of its first call.
   if (c->stage_open && (u != c->stage_u || dt != c->stage_dt))
   {
      // the ranges of one stage must name the same input and step: the extrema at hand are those of the first call's u
      c->stage_open = false;
      return fail(RMH_ERR_STATE, "rmh_stage_fused_range: the ranges of one stage must pass the same u and dt (stage abandoned)");
   }
   if (!c->stage_open)
   {
      if (in_token == 0 || in_token != c->xe_token)
      {
         rc = rmh_elem_minmax(c, u, c->d_xe_min, c->d_xe_max);
         if (rc) { return rc; }
      }
      c->stage_open = true;
      c->stage_u = u;
      c->stage_dt = dt;
   }
   c->xe_token = 0; // (until this stage is finished, no token is valid)
   if (e_end > e_begin)
   {
      EventPair ep;
      rc = timer_begin(c, 0, ep);
      if (!rc) { RMH_DISPATCH(c, rc = launch_stage_fused<P>(c, u, dt, x_base, a, b, dt_rk, y_out, du, e_begin, e_end)); }
      if (!rc) { rc = timer_end(c, 0, ep); }
      if (rc)
      {
         c->stage_open = false; // a failed stage leaves no extrema behind: the next one recomputes them
         return rc;
      }
   }
   if (finish)
   {
      // the stage wrote the extrema of y_out beside it: they become the context's, named by a fresh token
      std::swap(c->d_xe_min, c->d_xe_min2);
      std::swap(c->d_xe_max, c->d_xe_max2);
      c->xe_token = ++c->xe_counter;
      c->stage_open = false;
      if (out_token) { *out_token = c->xe_token; }
   }
   c->ho_done = false; // the lumped mass vector is not refreshed by the fused stage
   return rc;
}

int rmh_stage_fused_range(rmh_ctx *c, const double *u, double dt, const double *x_base, double a, double b,
                          double dt_rk, double *y_out, double *du, int e_begin, int e_end, int finish)
{
   return rmh_stage_fused_chain(c, u, dt, x_base, a, b, dt_rk, y_out, du, e_begin, e_end, finish, 0, nullptr);
}

int rmh_stage_fused(rmh_ctx *c, const double *u, double dt, const double *x_base, double a, double b, double dt_rk,
                    double *y_out, double *du)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null argument"); }
   return rmh_stage_fused_chain(c, u, dt, x_base, a, b, dt_rk, y_out, du, 0, c->ne, 1, 0, nullptr);
}

int rmh_halo_pack(rmh_ctx *c, const double *u, const int *send_elems, int nsend, double *rows, double *out_min,
                  double *out_max)
{
   if (!c || !u || (nsend > 0 && (!send_elems || !rows || !out_min || !out_max))) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_3D_ONLY(c, "rmh_halo_pack");
   RMH_ENTER(c);
   if (nsend <= 0) { return RMH_OK; }
   RMH_DISPATCH(c, hipLaunchKernelGGL((halo_pack_kernel<P>), dim3(nsend), dim3(KCfg<P>::NT), 0, c->stream, u, send_elems,
                                      rows, out_min, out_max, c->ndof, 1));
   RMH_HIP(hipGetLastError());
   return RMH_OK;
}

int rmh_halo_pack_records(rmh_ctx *c, const double *u, const int *send_elems, int nsend, double *rec)
{
   if (!c || !u || (nsend > 0 && (!send_elems || !rec))) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_3D_ONLY(c, "rmh_halo_pack_records");
   RMH_ENTER(c);
   if (nsend <= 0) { return RMH_OK; }
   const int w = c->ndof + 2;
   RMH_DISPATCH(c, hipLaunchKernelGGL((halo_pack_kernel<P>), dim3(nsend), dim3(KCfg<P>::NT), 0, c->stream, u, send_elems,
                                      rec, rec + c->ndof, rec + c->ndof + 1, w, w));
   RMH_HIP(hipGetLastError());
   return RMH_OK;
}

int rmh_enable_timers(rmh_ctx *c, int on)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   c->timers_on = on != 0;
   return RMH_OK;
}

int rmh_reset_timers(rmh_ctx *c)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   RMH_ENTER(c);
   int rc = timers_resolve(c);
   for (int b = 0; b < 4; b++) { c->tacc[b] = 0.0; }
   return rc;
}

int rmh_timers(rmh_ctx *c, double t[4])
{
   if (!c || !t) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   int rc = timers_resolve(c);
   for (int b = 0; b < 4; b++) { t[b] = c->tacc[b]; }
   return rc;
}

int rmh_last_cg_iters(rmh_ctx *c, int *max_iters)
{
   if (!c || !max_iters) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   RMH_HIP(hipStreamSynchronize(c->stream));
   RMH_HIP(hipMemcpy(max_iters, c->d_cg, sizeof(int), hipMemcpyDeviceToHost));
   RMH_HIP(hipMemset(c->d_cg, 0, sizeof(int)));
   return RMH_OK;
}

int rmh_set_lo_type(rmh_ctx *c, int lo_type)
{
   if (!c || (lo_type != 3 && lo_type != 4 && lo_type != 5)) { return fail(RMH_ERR_INVALID, "lo_type must be 3, 4 or 5"); }
   if (lo_type == 3 && c->p < 2) { return fail(RMH_ERR_STATE, "lo 3: the batched RD kernel is built for orders >= 2"); }
   if (lo_type == 4 && (c->p < 2 || !c->d_subvel)) { return fail(RMH_ERR_STATE, "lo 4 needs order >= 2 and rmh_layout.subcell_vel"); }
   c->lo_type = lo_type;
   return RMH_OK;
}

int rmh_invalidate_extrema(rmh_ctx *c)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   extrema_dropped(c);
   return RMH_OK;
}

int rmh_set_bounds_type(rmh_ctx *c, int bounds_type)
{
   if (!c || (bounds_type != 0 && bounds_type != 1)) { return fail(RMH_ERR_INVALID, "Wrong option for bounds computation."); }
   c->bounds_type = bounds_type;
   return RMH_OK;
}

namespace
{
__global__ void set_scalar_kernel(double *p, double v) { *p = v; }
} // namespace

int rmh_dt_estimate_reset(rmh_ctx *c)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   RMH_ENTER(c);
   if (!c->d_dt_est) { RMH_HIP(hipMalloc((void **)&c->d_dt_est, sizeof(double))); }
   hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, c->stream, c->d_dt_est, (double)INFINITY);
   RMH_HIP(hipGetLastError());
   return RMH_OK;
}

int rmh_set_dt_control(rmh_ctx *c, int on)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null ctx"); }
   if (on && c->bounds_type != 1) { return fail(RMH_ERR_STATE, "Error: -dtc 1 requires -bt 1."); }
   c->dt_control = on != 0;
   return on ? rmh_dt_estimate_reset(c) : RMH_OK;
}

int rmh_dt_estimate_update(rmh_ctx *c, const double *x, const double *dx, const double *x_min, const double *x_max)
{
   if (!c || !x || !dx || !x_min || !x_max) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   if (!c->dt_control) { return RMH_OK; } // TimeStepControl::FixedTimeStep: nothing to do (remhos.cpp:1973)
   const size_t n = (size_t)c->ne * c->ndof;
   const int nblk = (int)std::min<size_t>((n + 255) / 256, 4096);
   hipLaunchKernelGGL((dt_estimate_kernel<0>), dim3(nblk), dim3(256), 0, c->stream, x, dx, x_min, x_max, n, c->d_dt_est);
   RMH_HIP(hipGetLastError());
   return RMH_OK;
}

int rmh_dt_estimate_get(rmh_ctx *c, double *dt)
{
   if (!c || !dt) { return fail(RMH_ERR_INVALID, "null argument"); }
   RMH_ENTER(c);
   if (!c->d_dt_est) { return fail(RMH_ERR_STATE, "dt control is off"); }
   RMH_HIP(hipStreamSynchronize(c->stream));
   RMH_HIP(hipMemcpy(dt, c->d_dt_est, sizeof(double), hipMemcpyDeviceToHost));
   return RMH_OK;
}

namespace
{
__global__ void violation_reset_kernel(unsigned long long *v)
{
   v[0] = 0;
   v[1] = ~0ull;
   v[2] = 0;
   v[3] = 0;
}
} // namespace

int rmh_check_violation(rmh_ctx *c, const double *u, double dt, const double *du, const double *u_min, const double *u_max,
                        const double *bound_scale, double tol, const unsigned char *active_dofs, rmh_violation *out)
{
   if (!c || !u || !u_min || !u_max || !out) { return fail(RMH_ERR_INVALID, "null argument"); }
   if (!(tol >= 0.0)) { return fail(RMH_ERR_INVALID, "rmh_check_violation: the tolerance must not be negative"); }
   RMH_ENTER(c);
   if (!c->d_viol) { RMH_HIP(hipMalloc((void **)&c->d_viol, 4 * sizeof(unsigned long long))); }
   const size_t n = (size_t)c->ne * c->ndof;
   const int nblk = (int)std::max<size_t>(1, std::min<size_t>((n + 255) / 256, 4096));
   hipLaunchKernelGGL(violation_reset_kernel, dim3(1), dim3(1), 0, c->stream, c->d_viol);
   hipLaunchKernelGGL(check_violation_kernel, dim3(nblk), dim3(256), 0, c->stream, u, dt, du, u_min, u_max, bound_scale, tol,
                      active_dofs, n, c->d_viol);
   RMH_HIP(hipGetLastError());
   RMH_HIP(hipStreamSynchronize(c->stream));
   unsigned long long v[4];
   RMH_HIP(hipMemcpy(v, c->d_viol, sizeof(v), hipMemcpyDeviceToHost));
   std::memset(out, 0, sizeof(*out));
   out->count = (long long)v[0];
   out->first = v[0] ? (long long)v[1] : -1;
   std::memcpy(&out->over, &v[2], 8);
   std::memcpy(&out->under, &v[3], 8);
   if (v[0])
   {
      // the three numbers the reference prints for that dof (remhos.cpp:1567-1569, 1587-1589)
      const size_t i = (size_t)v[1];
      double ui = 0.0, dui = 0.0, lo = 0.0, hi = 0.0, s = 1.0;
      RMH_HIP(hipMemcpy(&ui, u + i, 8, hipMemcpyDeviceToHost));
      if (du) { RMH_HIP(hipMemcpy(&dui, du + i, 8, hipMemcpyDeviceToHost)); }
      RMH_HIP(hipMemcpy(&lo, u_min + i, 8, hipMemcpyDeviceToHost));
      RMH_HIP(hipMemcpy(&hi, u_max + i, 8, hipMemcpyDeviceToHost));
      if (bound_scale) { RMH_HIP(hipMemcpy(&s, bound_scale + i, 8, hipMemcpyDeviceToHost)); }
      out->first_min = bound_scale ? lo * s : lo;
      out->first_max = bound_scale ? hi * s : hi;
      out->first_value = du ? ui + dt * dui : ui;
   }
   return RMH_OK;
}

int rmh_get_mass_tol(rmh_ctx *c, double *rel_tol, double *abs_tol, int *max_iter)
{
   if (!c || !rel_tol || !abs_tol || !max_iter) { return fail(RMH_ERR_INVALID, "null argument"); }
   *rel_tol = c->rel_tol;
   *abs_tol = c->abs_tol;
   *max_iter = c->max_iter;
   return RMH_OK;
}

int rmh_set_mass_tol(rmh_ctx *c, double rel_tol, double abs_tol, int max_iter)
{
   if (!c || rel_tol < 0 || abs_tol < 0 || max_iter < 1) { return fail(RMH_ERR_INVALID, "bad tolerance"); }
   c->rel_tol = rel_tol;
   c->abs_tol = abs_tol;
   c->max_iter = max_iter;
   return RMH_OK;
}

int rmh_set_mass_completion(rmh_ctx *c, int jacobi_step, int constant_mode)
{
   if (!c || jacobi_step < 0 || jacobi_step > 1 || constant_mode < 0 || constant_mode > 1)
   {
      return fail(RMH_ERR_INVALID, "mass completion: flags are 0 or 1");
   }
   c->jacobi_step = jacobi_step;
   c->mass_fix = constant_mode;
   return RMH_OK;
}

} // extern "C"

#include "rmh_comm.hpp"

static const char *ghosts_in_flight(const rmh_ctx *c)
{
   if (c->ng <= 0) { return nullptr; }
   if (c->xch && c->xch->gen_begin != c->xch->gen_end) { return "the exchange of u is still in flight (rmh_exchange_end first): this call reads ghost data"; }
   return nullptr;
}

static const char *ghosts_not_ready(const rmh_ctx *c, int e_begin)
{
   if (c->ng <= 0 || e_begin >= c->ghost_readers_end) { return nullptr; }
   if (c->xch && c->xch->gen_begin != c->xch->gen_end) { return "the exchange of u is still in flight (rmh_exchange_end first): this element range reads ghost data"; }
   if (c->gh_foreign) { return "the ghost extrema hold another field's values (rmh_exchange_minmax_*): exchange u first"; }
   return nullptr;
}
