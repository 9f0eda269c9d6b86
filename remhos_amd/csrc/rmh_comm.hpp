// Neighbour exchange of the box-partitioned stage (include/rmh.h, "Neighbour exchange inside the library"):
// plan, pack kernels, ghost records and three transports -- RCCL grouped send/recv (one process per GPU over xGMI),
// device copies between contexts of one process, or a caller-side transport.
//
// What it replaces in the reference, per RK stage: ParGridFunction::ExchangeFaceNbrData (remhos_ho.cpp:122 and
// remhos.cpp:1812-1813; full face-neighbour elements through MPI) and the GroupCommunicator min/max reduction of
// DofInfo::ComputeOverlapBounds (remhos_tools.cpp:449-466).  Here: one message per neighbour rank and stage.
// (part of the library's one translation unit: included at the end of rmh_api.hip, behind the kernels it launches)
#pragma once
#include "rmh_ctx.hpp"
#include "rmh_kernels.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#if !defined(HIPEMU)
#include <dlfcn.h>
#endif

namespace rmh
{

// ---- RCCL through dlopen: the library has no link-time dependency on it (contexts without neighbours, the local
// transport and the host-side tests never touch it); a process that already carries an RCCL (PyTorch) shares it.
struct Rccl
{
   typedef struct { char internal[128]; } UniqueId; // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
   typedef void *Comm;                              // ncclComm_t
   int (*GetUniqueId)(UniqueId *) = nullptr;
   int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
   int (*CommDestroy)(Comm) = nullptr;
   int (*CommCount)(Comm, int *) = nullptr;
   int (*GroupStart)() = nullptr;
   int (*GroupEnd)() = nullptr;
   int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
   int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
   int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
   const char *(*GetErrorString)(int) = nullptr;
   static constexpr int kDouble = 8;                   // ncclFloat64
   static constexpr int kSum = 0, kMax = 2, kMin = 3;  // ncclRedOp_t
   bool ok = false;
   std::string why;
};

static Rccl &rccl()
{
   static Rccl r;
   static bool tried = false;
   if (tried) { return r; }
   tried = true;
#if defined(HIPEMU)
   r.why = "host emulation build: no RCCL";
#else
   void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD); // the copy the process already loaded, if any
   if (!h) { h = dlopen("librccl.so.1", RTLD_NOW); }
   if (!h) { h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW); }
   if (!h) { r.why = std::string("librccl.so.1 not found: ") + dlerror(); return r; }
   auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) { r.why = std::string("RCCL symbol missing: ") + n; } return p; };
   r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
   r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
   r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
   r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
   r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
   r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
   r.Send = (decltype(r.Send))sym("ncclSend");
   r.Recv = (decltype(r.Recv))sym("ncclRecv");
   r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
   r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
   r.ok = r.why.empty();
#endif
   return r;
}

#define RMH_NCCL(call)                                                                                     \
   do {                                                                                                    \
      const int rc_ = (call);                                                                              \
      if (rc_ != 0) { return fail(RMH_ERR_HIP, std::string(#call) + ": " + rccl().GetErrorString(rc_)); } \
   } while (0)

struct PeerPlan
{
   int rank = -1;
   long long send_off = 0, send_n = 0; // segment of the send buffer (doubles)
   long long recv_off = 0, recv_n = 0; // segment of the ghost buffer (doubles)
   rmh_ctx *local = nullptr;           // same-process peer (rmh_comm_connect_local)
   int local_index = -1;               // this context's peer index on the other side
   int send_first_el = 0, send_count_el = 0; // its records in the send list / its ghost slots (rmh_exchange_minmax_*)
   int recv_first_el = 0, recv_count_el = 0;
};

struct Exchange
{
   bool compact = false;
   int nsend = 0;
   std::vector<PeerPlan> peers;
   int *d_send_elem = nullptr;      // [nsend]
   int *d_send_code = nullptr;      // [nsend] compact: face 0..5 of the sender whose layer goes out, -1: extrema only
   long long *d_send_off = nullptr; // [nsend] offset of the record in the send buffer (doubles)
   double *d_send = nullptr, *d_ghost = nullptr;
   long long send_doubles = 0, ghost_doubles = 0;
   // transport
   Rccl::Comm comm = nullptr;
   bool own_comm = false;
   int nranks = 1, rank = 0;
   hipStream_t xs = nullptr;               // exchange stream
   hipEvent_t ev_packed = nullptr;         // send buffer complete (context stream)
   hipEvent_t ev_done = nullptr;           // ghosts received (exchange stream)
   hipEvent_t ev_started = nullptr;        // the exchange stream has passed its wait for ev_packed: the RCCL kernel is next in its queue
   bool comm_first = false;                // the context's stream waits for ev_started behind rmh_exchange_begin (blocks of >= 16 M dofs; RMH_COMM_FIRST=0 / 1 overrides)
   unsigned long long gen_begin = 0, gen_end = 0;
   double *d_red = nullptr;                // scratch of rmh_allreduce
   // exchange of caller-given element extrema (rmh_exchange_minmax_*): [nsend][2] out, [ne_ghost][2] in, and where ghost
   // g's (min, max) pair sits in the ghost buffer (doubles)
   double *d_send2 = nullptr, *d_recv2 = nullptr;
   long long *d_ghost_pos = nullptr;
   int ng = 0;
   unsigned long long mm_begin = 0, mm_end = 0;
};

__global__ void minmax_pack_kernel(const double *xe_min, const double *xe_max, const int *send_elem, int n, double *out)
{
   for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
   {
      out[2 * k] = xe_min[send_elem[k]];
      out[2 * k + 1] = xe_max[send_elem[k]];
   }
}
__global__ void minmax_unpack_kernel(const double *in, const long long *ghost_pos, int ng, double *ghost)
{
   for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += gridDim.x * blockDim.x)
   {
      ghost[ghost_pos[g]] = in[2 * g];
      ghost[ghost_pos[g] + 1] = in[2 * g + 1];
   }
}

// one workgroup per send record
template <int P>
__global__ void __launch_bounds__(KCfg<P>::NT) halo_pack_compact_kernel(const double *u, const int *send_elem, const int *send_code,
                                                                        const long long *send_off, double *out)
{
   using C = KCfg<P>;
   constexpr int D = P + 1, D2 = D * D;
   __shared__ double s_red[4];
   const int k = blockIdx.x;
   const int e = send_elem[k], code = send_code[k];
   double *rec = out + send_off[k];
   const double *ue = u + (size_t)e * C::D3;
   double lmin = INFINITY, lmax = -INFINITY;
   for (int i = threadIdx.x; i < C::D3; i += C::NT)
   {
      const double v = ue[i];
      lmin = fmin(lmin, v);
      lmax = fmax(lmax, v);
   }
   lmin = block_min<C::NW>(lmin, s_red);
   lmax = block_max<C::NW>(lmax, s_red);
   if (threadIdx.x == 0) { rec[0] = lmin; rec[1] = lmax; }
   if (code >= 0)
   {
      // the layer on this element's face `code`, ordered (i1 along c+1, i2 along c+2): exactly what the receiver's
      // element on the other side reads as neighbour trace r = i1 + D*i2 (ho_kernel2 phase A)
      const int c = code >> 1, side = code & 1;
      const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
      const int strc = (c == 0) ? 1 : (c == 1 ? D : D2);
      const int str1 = (c1 == 0) ? 1 : (c1 == 1 ? D : D2);
      const int str2 = (c2 == 0) ? 1 : (c2 == 1 ? D : D2);
      for (int r = threadIdx.x; r < D2; r += C::NT)
      {
         rec[2 + r] = ue[(side ? P : 0) * strc + (r % D) * str1 + (r / D) * str2];
      }
   }
}

#define RMH_DISPATCH_P(p, expr)                                                                \
   switch (p)                                                                                  \
   {                                                                                           \
      case 1: { constexpr int P = 1; expr; break; }                                            \
      case 2: { constexpr int P = 2; expr; break; }                                            \
      case 3: { constexpr int P = 3; expr; break; }                                            \
      case 4: { constexpr int P = 4; expr; break; }                                            \
      case 5: { constexpr int P = 5; expr; break; }                                            \
      case 6: { constexpr int P = 6; expr; break; }                                            \
      default: return fail(RMH_ERR_INVALID, "unsupported order");                              \
   }

void exchange_free(rmh_ctx *c)
{
   Exchange *x = c->xch;
   if (!x) { return; }
   (void)hipFree(x->d_send_elem);
   (void)hipFree(x->d_send_code);
   (void)hipFree(x->d_send_off);
   (void)hipFree(x->d_send);
   (void)hipFree(x->d_ghost);
   (void)hipFree(x->d_red);
   (void)hipFree(x->d_send2);
   (void)hipFree(x->d_recv2);
   (void)hipFree(x->d_ghost_pos);
   if (x->ev_packed) { (void)hipEventDestroy(x->ev_packed); }
   if (x->ev_done) { (void)hipEventDestroy(x->ev_done); }
   if (x->ev_started) { (void)hipEventDestroy(x->ev_started); }
   if (x->xs) { (void)hipStreamDestroy(x->xs); }
   if (x->comm && x->own_comm && rccl().ok) { (void)rccl().CommDestroy(x->comm); }
   delete x;
   c->xch = nullptr;
}

} // namespace rmh

extern "C" {

} // extern "C"

namespace rmh
{
// (tables_rewritten: the device copies of face_nbr / stencil27 were re-indexed to ghost cells; saved_*: their originals)
static int exchange_setup_impl(rmh_ctx *c, const rmh_exchange_desc *d, int compact, bool &tables_rewritten, std::vector<int> &saved_nbr,
                               std::vector<int> &saved_st27);
}

extern "C" {

int rmh_exchange_setup(rmh_ctx *c, const rmh_exchange_desc *d, int compact)
{
   if (!c || !d || d->n_peers < 0) { return fail(RMH_ERR_INVALID, "null argument"); }
   if (c->dim != 3) { return fail(RMH_ERR_INVALID, "rmh_exchange_setup: not available for dim = 2 (single rank)"); }
   if (c->xch) { return fail(RMH_ERR_STATE, "rmh_exchange_setup: the exchange of this context is already set up"); }
   RMH_ENTER(c);
   bool rewritten = false;
   std::vector<int> saved_nbr, saved_st27;
   const int rc = exchange_setup_impl(c, d, compact, rewritten, saved_nbr, saved_st27);
   if (rc != RMH_OK)
   {
      // a failed set-up leaves the context as it was: no plan, the neighbour tables in ghost-slot indexing
      const std::string msg = rmh_last_error();
      exchange_free(c);
      if (rewritten)
      {
         (void)hipMemcpy(c->d_nbr, saved_nbr.data(), saved_nbr.size() * sizeof(int), hipMemcpyHostToDevice);
         (void)hipMemcpy(c->d_st27, saved_st27.data(), saved_st27.size() * sizeof(int), hipMemcpyHostToDevice);
      }
      return fail(rc, msg);
   }
   return RMH_OK;
}

} // extern "C"

namespace rmh
{
static int exchange_setup_impl(rmh_ctx *c, const rmh_exchange_desc *d, int compact, bool &tables_rewritten, std::vector<int> &saved_nbr,
                               std::vector<int> &saved_st27)
{
   const int ne = c->ne, ng = c->ng, D2 = (c->p + 1) * (c->p + 1), nd = c->ndof;
   // ---- validate the plan: ghost ranges of the neighbours tile [0, ne_ghost) -----------------------------------
   std::vector<int> ghost_peer(ng, -1);
   long long nsend = 0;
   for (int k = 0; k < d->n_peers; k++)
   {
      if (d->send_count[k] < 0 || d->recv_count[k] < 0 || d->recv_first[k] < 0 || d->recv_first[k] + d->recv_count[k] > ng)
      {
         return fail(RMH_ERR_INVALID, "rmh_exchange_setup: bad counts");
      }
      for (int j = 0; j < d->send_count[k]; j++)
      {
         const int e = d->send_elems[k][j];
         if (e < 0 || e >= ne) { return fail(RMH_ERR_INVALID, "rmh_exchange_setup: send element out of range"); }
      }
      for (int g = d->recv_first[k]; g < d->recv_first[k] + d->recv_count[k]; g++)
      {
         if (ghost_peer[g] != -1) { return fail(RMH_ERR_INVALID, "rmh_exchange_setup: ghost slot filled twice"); }
         ghost_peer[g] = k;
      }
      nsend += d->send_count[k];
   }
   for (int g = 0; g < ng; g++)
   {
      if (ghost_peer[g] < 0) { return fail(RMH_ERR_INVALID, "rmh_exchange_setup: a ghost slot is filled by no neighbour"); }
   }
   Exchange *x = new Exchange;
   c->xch = x;
   x->compact = compact != 0;
   x->nsend = (int)nsend;
   x->peers.resize(d->n_peers);
   std::vector<int> send_elem, send_code;
   std::vector<long long> send_off;
   send_elem.reserve(nsend);
   // host copies of the neighbour tables (only entries that point at ghosts are looked at / rewritten)
   std::vector<int> nbr((size_t)ne * 6), st27;
   RMH_HIP(hipMemcpy(nbr.data(), c->d_nbr, nbr.size() * sizeof(int), hipMemcpyDeviceToHost));
   const int full_rec = nd + 2;
   const int face_rec = 2 + D2 + (D2 & 1); // [min | max | D^2 layer], padded to whole 2-double cells
   std::vector<long long> ghost_cell(ng, 0);    // compact: first cell of ghost slot g
   if (x->compact)
   {
      // which ghosts share a face with this rank, and through which face each owned element sees which neighbour rank
      std::vector<char> ghost_face(ng, 0);
      std::map<std::pair<int, int>, int> elem_face_to_peer; // (element, peer) -> the element's face towards that peer
      for (int e = 0; e < ne; e++)
      {
         for (int f = 0; f < 6; f++)
         {
            const int nb = nbr[(size_t)e * 6 + f];
            if (nb < ne) { continue; }
            const int g = nb - ne;
            ghost_face[g]++;
            auto key = std::make_pair(e, ghost_peer[g]);
            if (elem_face_to_peer.count(key))
            {
               return fail(RMH_ERR_INVALID, "rmh_exchange_setup: an element is adjacent to one neighbour rank through two faces "
                                            "(block one element thin): use compact = 0 on all ranks");
            }
            elem_face_to_peer[key] = f;
         }
      }
      for (int g = 0; g < ng; g++)
      {
         if (ghost_face[g] > 1)
         {
            return fail(RMH_ERR_INVALID, "rmh_exchange_setup: a ghost element is seen through two faces: use compact = 0 on all ranks");
         }
      }
      // receive side: cells in ghost-slot order (the ranges of the neighbours are contiguous)
      long long cell = 0;
      std::vector<int> order(d->n_peers);
      for (int k = 0; k < d->n_peers; k++) { order[k] = k; }
      std::sort(order.begin(), order.end(), [&](int a, int b) { return d->recv_first[a] < d->recv_first[b]; });
      for (int k : order)
      {
         x->peers[k].recv_off = 2 * cell;
         for (int g = d->recv_first[k]; g < d->recv_first[k] + d->recv_count[k]; g++)
         {
            ghost_cell[g] = cell;
            cell += ghost_face[g] ? face_rec / 2 : 1;
         }
         x->peers[k].recv_n = 2 * cell - x->peers[k].recv_off;
      }
      x->ghost_doubles = 2 * cell;
      // send side: record type per send entry from this rank's own tables (the adjacency is symmetric: element e
      // is a face ghost of neighbour k exactly when e has a face neighbour owned by k)
      long long off = 0;
      for (int k = 0; k < d->n_peers; k++)
      {
         x->peers[k].send_off = off;
         for (int j = 0; j < d->send_count[k]; j++)
         {
            const int e = d->send_elems[k][j];
            auto it = elem_face_to_peer.find(std::make_pair(e, k));
            const int code = it == elem_face_to_peer.end() ? -1 : it->second;
            send_elem.push_back(e);
            send_code.push_back(code);
            send_off.push_back(off);
            off += code >= 0 ? face_rec : 2;
         }
         x->peers[k].send_n = off - x->peers[k].send_off;
      }
      x->send_doubles = off;
      // re-index the device tables: ghost slot g -> ne + first cell of its record
      st27.resize((size_t)ne * 27);
      RMH_HIP(hipMemcpy(st27.data(), c->d_st27, st27.size() * sizeof(int), hipMemcpyDeviceToHost));
      if (2 * cell / 2 + (long long)ne > 0x7fffffffLL) { return fail(RMH_ERR_INVALID, "rmh_exchange_setup: too many ghost cells"); }
      saved_nbr = nbr;
      saved_st27 = st27;
      for (int &v : nbr) { if (v >= ne) { v = ne + (int)ghost_cell[v - ne]; } }
      for (int &v : st27) { if (v >= ne) { v = ne + (int)ghost_cell[v - ne]; } }
      tables_rewritten = true;
      RMH_HIP(hipMemcpy(c->d_nbr, nbr.data(), nbr.size() * sizeof(int), hipMemcpyHostToDevice));
      RMH_HIP(hipMemcpy(c->d_st27, st27.data(), st27.size() * sizeof(int), hipMemcpyHostToDevice));
   }
   else
   {
      long long off = 0;
      for (int k = 0; k < d->n_peers; k++)
      {
         x->peers[k].send_off = off;
         x->peers[k].send_n = (long long)d->send_count[k] * full_rec;
         x->peers[k].recv_off = (long long)d->recv_first[k] * full_rec;
         x->peers[k].recv_n = (long long)d->recv_count[k] * full_rec;
         for (int j = 0; j < d->send_count[k]; j++) { send_elem.push_back(d->send_elems[k][j]); }
         off += x->peers[k].send_n;
      }
      x->send_doubles = off;
      x->ghost_doubles = (long long)ng * full_rec;
   }
   {
      int first = 0;
      for (int k = 0; k < d->n_peers; k++)
      {
         x->peers[k].rank = d->peer_rank[k];
         x->peers[k].send_first_el = first;
         x->peers[k].send_count_el = d->send_count[k];
         x->peers[k].recv_first_el = d->recv_first[k];
         x->peers[k].recv_count_el = d->recv_count[k];
         first += d->send_count[k];
      }
   }
   // where the (min, max) pair of ghost g lives: cells 2 c, 2 c + 1 of its compact record, or behind the values of its
   // whole-element record
   x->ng = ng;
   std::vector<long long> ghost_pos(ng);
   for (int g = 0; g < ng; g++) { ghost_pos[g] = x->compact ? 2 * ghost_cell[g] : (long long)g * full_rec + nd; }
   // ---- device side --------------------------------------------------------------------------------------------
   if (nsend > 0)
   {
      RMH_HIP(hipMalloc((void **)&x->d_send_elem, nsend * sizeof(int)));
      RMH_HIP(hipMemcpy(x->d_send_elem, send_elem.data(), nsend * sizeof(int), hipMemcpyHostToDevice));
      if (x->compact)
      {
         RMH_HIP(hipMalloc((void **)&x->d_send_code, nsend * sizeof(int)));
         RMH_HIP(hipMemcpy(x->d_send_code, send_code.data(), nsend * sizeof(int), hipMemcpyHostToDevice));
         RMH_HIP(hipMalloc((void **)&x->d_send_off, nsend * sizeof(long long)));
         RMH_HIP(hipMemcpy(x->d_send_off, send_off.data(), nsend * sizeof(long long), hipMemcpyHostToDevice));
      }
   }
   RMH_HIP(hipMalloc((void **)&x->d_send2, std::max<long long>(1, 2 * nsend) * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&x->d_recv2, std::max<long long>(1, 2 * (long long)ng) * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&x->d_ghost_pos, std::max<long long>(1, (long long)ng) * sizeof(long long)));
   if (ng > 0) { RMH_HIP(hipMemcpy(x->d_ghost_pos, ghost_pos.data(), ng * sizeof(long long), hipMemcpyHostToDevice)); }
   RMH_HIP(hipMalloc((void **)&x->d_send, std::max<long long>(1, x->send_doubles) * sizeof(double)));
   RMH_HIP(hipMalloc((void **)&x->d_ghost, std::max<long long>(1, x->ghost_doubles) * sizeof(double)));
   RMH_HIP(hipMemset(x->d_ghost, 0, std::max<long long>(1, x->ghost_doubles) * sizeof(double)));
   // (non-blocking: a blocking stream would synchronise implicitly with the legacy default stream, i.e. with the
   // interior launch of a caller that runs the context there, and the exchange would not overlap it; everything that
   // must be ordered is ordered by ev_packed / ev_done)
   // Highest priority, so that the RCCL send / recv kernel -- enqueued a few microseconds AFTER the interior launch has
   // filled every CU -- is placed ahead of the interior kernel's remaining workgroups.  (Kernel trace of the one-rank
   // self-loop, profiles/r03_rccl_selfloop_timeline.txt: the RCCL kernel lasts as long as the interior launch it runs
   // beside, with or without the priority; the halo shell is behind the interior launch on the context's stream anyway,
   // so the exchange costs nothing as long as it ends by then.)
   {
      int pr_least = 0, pr_greatest = 0;
      RMH_HIP(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
      RMH_HIP(hipStreamCreateWithPriority(&x->xs, hipStreamNonBlocking, pr_greatest));
   }
   RMH_HIP(hipEventCreateWithFlags(&x->ev_packed, hipEventDisableTiming));
   RMH_HIP(hipEventCreateWithFlags(&x->ev_done, hipEventDisableTiming));
   RMH_HIP(hipEventCreateWithFlags(&x->ev_started, hipEventDisableTiming));
   // (the hold costs two cross-stream wake-ups, ~30 us per stage: measured on the one-rank self-loop it is neutral for a 96^3 block
   // -- 2.7 ms interior launch, RCCL kernel 8 us instead of 2.7 ms in the queue, interior launch 2.4 % shorter without the
   // half-resident exchange kernel spinning beside it -- and -2.5 % for a 48^3 block; on by default for large blocks only)
   x->comm_first = (double)c->ne * c->ndof >= 16.0e6;
   if (const char *s = std::getenv("RMH_COMM_FIRST")) { x->comm_first = std::atoi(s) != 0; }
   // the kernels read the ghosts from the library's buffer from now on
   if (x->compact)
   {
      c->u_ghost = x->d_ghost + 2; // trace r of the record whose first cell is g: u_ghost[2 g + r]
      c->gh_ustride = 2;
      c->gh_min = x->d_ghost;      // [2 g], [2 g + 1]
      c->gh_max = x->d_ghost + 1;
      c->gh_mstride = 2;
      c->gh_compact = 1;
   }
   else
   {
      c->u_ghost = x->d_ghost;
      c->gh_min = x->d_ghost + nd;
      c->gh_max = x->d_ghost + nd + 1;
      c->gh_ustride = c->gh_mstride = full_rec;
      c->gh_compact = 0;
   }
   return RMH_OK;
}
} // namespace rmh

extern "C" {

int rmh_comm_unique_id(char id[128])
{
   if (!id) { return fail(RMH_ERR_INVALID, "null argument"); }
   if (!rccl().ok) { return fail(RMH_ERR_STATE, "RCCL unavailable: " + rccl().why); }
   Rccl::UniqueId u;
   RMH_NCCL(rccl().GetUniqueId(&u));
   std::memcpy(id, u.internal, 128);
   return RMH_OK;
}

int rmh_comm_init(rmh_ctx *c, const char id[128], int nranks, int rank)
{
   if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) { return fail(RMH_ERR_INVALID, "bad argument"); }
   if (!c->xch) { return fail(RMH_ERR_STATE, "rmh_comm_init: call rmh_exchange_setup first"); }
   if (!rccl().ok) { return fail(RMH_ERR_STATE, "RCCL unavailable: " + rccl().why); }
   RMH_ENTER(c);
   Rccl::UniqueId u;
   std::memcpy(u.internal, id, 128);
   Rccl::Comm comm = nullptr;
   RMH_NCCL(rccl().CommInitRank(&comm, nranks, u, rank));
   c->xch->comm = comm;
   c->xch->own_comm = true;
   c->xch->nranks = nranks;
   c->xch->rank = rank;
   return RMH_OK;
}

int rmh_comm_attach(rmh_ctx *c, void *nccl_comm)
{
   if (!c || !nccl_comm) { return fail(RMH_ERR_INVALID, "null argument"); }
   if (!c->xch) { return fail(RMH_ERR_STATE, "rmh_comm_attach: call rmh_exchange_setup first"); }
   if (!rccl().ok) { return fail(RMH_ERR_STATE, "RCCL unavailable: " + rccl().why); }
   c->xch->comm = (Rccl::Comm)nccl_comm;
   c->xch->own_comm = false;
   return RMH_OK;
}

int rmh_comm_connect_local(rmh_ctx *c, int k, rmh_ctx *other, int other_k)
{
   if (!c || !other || !c->xch || !other->xch) { return fail(RMH_ERR_INVALID, "rmh_comm_connect_local: both contexts need an exchange plan"); }
   if (k < 0 || k >= (int)c->xch->peers.size() || other_k < 0 || other_k >= (int)other->xch->peers.size())
   {
      return fail(RMH_ERR_INVALID, "rmh_comm_connect_local: bad peer index");
   }
   PeerPlan &me = c->xch->peers[k], &o = other->xch->peers[other_k];
   if (o.send_n != me.recv_n || o.recv_n != me.send_n)
   {
      return fail(RMH_ERR_INVALID, "rmh_comm_connect_local: the segment sizes of the two sides do not mirror each other");
   }
   me.local = other;
   me.local_index = other_k;
   o.local = c;
   o.local_index = k;
   return RMH_OK;
}

int rmh_exchange_begin(rmh_ctx *c, const double *u)
{
   if (!c || !u) { return fail(RMH_ERR_INVALID, "null argument"); }
   Exchange *x = c->xch;
   if (!x) { return fail(RMH_ERR_STATE, "rmh_exchange_begin: no exchange plan"); }
   if (x->gen_begin != x->gen_end) { return fail(RMH_ERR_STATE, "rmh_exchange_begin: the previous exchange was not ended"); }
   RMH_ENTER(c);
   x->gen_begin++;
   // (the exchange under way rewrites the ghost extrema with this vector's; they are valid -- and gh_foreign cleared -- at
   // rmh_exchange_end; the interior range of a split stage, launched in between, reads no ghost: ghosts_not_ready, rmh_api.hip)
   if (x->peers.empty()) { return RMH_OK; }
   // local peers read the send buffer on THEIR exchange streams: the previous copies out of it must be done before
   // it is overwritten (their ev_done is recorded behind those copies)
   for (PeerPlan &p : x->peers)
   {
      if (p.local && p.local->xch->gen_end > 0) { RMH_HIP(hipStreamWaitEvent(c->stream, p.local->xch->ev_done, 0)); }
   }
   if (x->nsend > 0)
   {
      if (x->compact)
      {
         RMH_DISPATCH_P(c->p, hipLaunchKernelGGL((halo_pack_compact_kernel<P>), dim3(x->nsend), dim3(KCfg<P>::NT), 0, c->stream, u,
                                                 (const int *)x->d_send_elem, (const int *)x->d_send_code,
                                                 (const long long *)x->d_send_off, x->d_send));
      }
      else
      {
         const int w = c->ndof + 2;
         RMH_DISPATCH_P(c->p, hipLaunchKernelGGL((halo_pack_kernel<P>), dim3(x->nsend), dim3(KCfg<P>::NT), 0, c->stream, u,
                                                 (const int *)x->d_send_elem, x->d_send, x->d_send + c->ndof, x->d_send + c->ndof + 1, w, w));
      }
      RMH_HIP(hipGetLastError());
   }
   RMH_HIP(hipEventRecord(x->ev_packed, c->stream));
   if (x->comm)
   {
      // the exchange stream starts behind the pack kernel -- and thereby behind every kernel of the previous stage
      // that still read the ghosts it is about to overwrite
      RMH_HIP(hipStreamWaitEvent(x->xs, x->ev_packed, 0));
      if (x->comm_first) { RMH_HIP(hipEventRecord(x->ev_started, x->xs)); }
      // (a failing call must not leave the group open or the generation counter advanced: the context stays usable)
      int rc = rccl().GroupStart();
      const char *what = "ncclGroupStart";
      if (rc == 0)
      {
         for (PeerPlan &p : x->peers)
         {
            if (p.local) { continue; }
            if (rc == 0 && p.send_n > 0) { rc = rccl().Send(x->d_send + p.send_off, (size_t)p.send_n, Rccl::kDouble, p.rank, x->comm, x->xs); what = "ncclSend"; }
            if (rc == 0 && p.recv_n > 0) { rc = rccl().Recv(x->d_ghost + p.recv_off, (size_t)p.recv_n, Rccl::kDouble, p.rank, x->comm, x->xs); what = "ncclRecv"; }
         }
         const int rc_end = rccl().GroupEnd();
         if (rc == 0 && rc_end != 0) { rc = rc_end; what = "ncclGroupEnd"; }
      }
      if (rc != 0)
      {
         x->gen_begin--;
         return fail(RMH_ERR_HIP, std::string(what) + ": " + rccl().GetErrorString(rc));
      }
      // The interior launch that follows on the context's stream is held until the exchange stream has passed its wait for the
      // pack kernel, i.e. until the grouped send / recv kernel is the next packet of ITS queue: in-stream the interior kernel
      // followed the pack kernel at once, the cross-stream wake-up of the exchange stream took ~7 us longer, and by then the
      // interior's workgroups had filled every CU -- the RCCL kernel (41 workgroups with a large LDS block) then found no CU
      // until that launch drained (measured: 2.7 ms in the queue at -rs 5, 10 us alone), so nothing of the exchange was hidden.
      // Now both become eligible together and the exchange kernel, first in its queue, is resident before the flood.
      if (x->comm_first) { RMH_HIP(hipStreamWaitEvent(c->stream, x->ev_started, 0)); }
   }
   return RMH_OK;
}

int rmh_exchange_end(rmh_ctx *c)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null argument"); }
   Exchange *x = c->xch;
   if (!x) { return fail(RMH_ERR_STATE, "rmh_exchange_end: no exchange plan"); }
   if (x->gen_begin != x->gen_end + 1) { return fail(RMH_ERR_STATE, "rmh_exchange_end without rmh_exchange_begin"); }
   RMH_ENTER(c);
   if (x->peers.empty()) { x->gen_end++; c->gh_foreign = false; return RMH_OK; }
   bool any_local = false;
   for (PeerPlan &p : x->peers)
   {
      if (!p.local) { continue; }
      Exchange *o = p.local->xch;
      if (o->gen_begin != x->gen_begin)
      {
         return fail(RMH_ERR_STATE, "rmh_exchange_end: a same-process neighbour has not begun this exchange (every context "
                                    "calls rmh_exchange_begin before any calls rmh_exchange_end)");
      }
      const PeerPlan &q = o->peers[p.local_index];
      if (q.send_n != p.recv_n) { return fail(RMH_ERR_STATE, "rmh_exchange_end: segment sizes of the two sides differ"); }
      if (!any_local) { RMH_HIP(hipStreamWaitEvent(x->xs, x->ev_packed, 0)); } // (orders the copies behind this context's earlier ghost readers)
      any_local = true;
      RMH_HIP(hipStreamWaitEvent(x->xs, o->ev_packed, 0));
      if (p.recv_n > 0)
      {
         RMH_HIP(hipMemcpyAsync(x->d_ghost + p.recv_off, o->d_send + q.send_off, (size_t)p.recv_n * sizeof(double),
                                hipMemcpyDeviceToDevice, x->xs));
      }
   }
   if (x->comm || any_local)
   {
      RMH_HIP(hipEventRecord(x->ev_done, x->xs));
      RMH_HIP(hipStreamWaitEvent(c->stream, x->ev_done, 0));
   }
   x->gen_end++;
   c->gh_foreign = false; // the ghost extrema are again those of the vector rmh_exchange_begin packed
   return RMH_OK;
}

int rmh_exchange_minmax_begin(rmh_ctx *c, const double *xe_min, const double *xe_max)
{
   if (!c || !xe_min || !xe_max) { return fail(RMH_ERR_INVALID, "null argument"); }
   Exchange *x = c->xch;
   if (!x) { return fail(RMH_ERR_STATE, "rmh_exchange_minmax_begin: no exchange plan"); }
   if (x->gen_begin != x->gen_end) { return fail(RMH_ERR_STATE, "rmh_exchange_minmax_begin: an exchange of u is still open"); }
   if (x->mm_begin != x->mm_end) { return fail(RMH_ERR_STATE, "rmh_exchange_minmax_begin: the previous one was not ended"); }
   RMH_ENTER(c);
   x->mm_begin++;
   if (x->peers.empty()) { return RMH_OK; }
   for (PeerPlan &p : x->peers)
   {
      if (p.local && p.local->xch->mm_end > 0) { RMH_HIP(hipStreamWaitEvent(c->stream, p.local->xch->ev_done, 0)); }
   }
   if (x->nsend > 0)
   {
      hipLaunchKernelGGL(minmax_pack_kernel, dim3(std::max(1, std::min(1024, (x->nsend + 255) / 256))), dim3(256), 0, c->stream, xe_min,
                         xe_max, (const int *)x->d_send_elem, x->nsend, x->d_send2);
      RMH_HIP(hipGetLastError());
   }
   RMH_HIP(hipEventRecord(x->ev_packed, c->stream));
   if (x->comm)
   {
      RMH_HIP(hipStreamWaitEvent(x->xs, x->ev_packed, 0));
      int rc = rccl().GroupStart();
      const char *what = "ncclGroupStart";
      if (rc == 0)
      {
         for (PeerPlan &p : x->peers)
         {
            if (p.local) { continue; }
            if (rc == 0 && p.send_count_el > 0) { rc = rccl().Send(x->d_send2 + 2 * (size_t)p.send_first_el, 2 * (size_t)p.send_count_el, Rccl::kDouble, p.rank, x->comm, x->xs); what = "ncclSend"; }
            if (rc == 0 && p.recv_count_el > 0) { rc = rccl().Recv(x->d_recv2 + 2 * (size_t)p.recv_first_el, 2 * (size_t)p.recv_count_el, Rccl::kDouble, p.rank, x->comm, x->xs); what = "ncclRecv"; }
         }
         const int rc_end = rccl().GroupEnd();
         if (rc == 0 && rc_end != 0) { rc = rc_end; what = "ncclGroupEnd"; }
      }
      if (rc != 0)
      {
         x->mm_begin--;
         return fail(RMH_ERR_HIP, std::string(what) + ": " + rccl().GetErrorString(rc));
      }
   }
   return RMH_OK;
}

int rmh_exchange_minmax_end(rmh_ctx *c)
{
   if (!c) { return fail(RMH_ERR_INVALID, "null argument"); }
   Exchange *x = c->xch;
   if (!x) { return fail(RMH_ERR_STATE, "rmh_exchange_minmax_end: no exchange plan"); }
   if (x->mm_begin != x->mm_end + 1) { return fail(RMH_ERR_STATE, "rmh_exchange_minmax_end without rmh_exchange_minmax_begin"); }
   RMH_ENTER(c);
   if (x->peers.empty()) { x->mm_end++; return RMH_OK; }
   bool any_local = false, any_remote = false;
   for (PeerPlan &p : x->peers)
   {
      if (!p.local) { any_remote = true; continue; }
      Exchange *o = p.local->xch;
      if (o->mm_begin != x->mm_begin)
      {
         return fail(RMH_ERR_STATE, "rmh_exchange_minmax_end: a same-process neighbour has not begun this exchange");
      }
      const PeerPlan &q = o->peers[p.local_index];
      if (q.send_count_el != p.recv_count_el) { return fail(RMH_ERR_STATE, "rmh_exchange_minmax_end: the two sides' lists differ"); }
      if (!any_local) { RMH_HIP(hipStreamWaitEvent(x->xs, x->ev_packed, 0)); }
      any_local = true;
      RMH_HIP(hipStreamWaitEvent(x->xs, o->ev_packed, 0));
      if (p.recv_count_el > 0)
      {
         RMH_HIP(hipMemcpyAsync(x->d_recv2 + 2 * (size_t)p.recv_first_el, o->d_send2 + 2 * (size_t)q.send_first_el,
                                2 * (size_t)p.recv_count_el * sizeof(double), hipMemcpyDeviceToDevice, x->xs));
      }
   }
   if (any_remote && !x->comm) { return fail(RMH_ERR_STATE, "rmh_exchange_minmax needs RCCL or same-process neighbours (no caller-side transport)"); }
   if (x->ng > 0)
   {
      hipLaunchKernelGGL(minmax_unpack_kernel, dim3(std::max(1, std::min(1024, (x->ng + 255) / 256))), dim3(256), 0, x->xs,
                         (const double *)x->d_recv2, (const long long *)x->d_ghost_pos, x->ng, x->d_ghost);
      RMH_HIP(hipGetLastError());
   }
   RMH_HIP(hipEventRecord(x->ev_done, x->xs));
   RMH_HIP(hipStreamWaitEvent(c->stream, x->ev_done, 0));
   x->mm_end++;
   // the ghost extrema now belong to the caller's field (the ratio s of product remap), not to the vector whose traces the
   // ghosts hold: the limiters of that vector refuse them until the next rmh_exchange_begin / _end refreshes them
   c->gh_foreign = true;
   return RMH_OK;
}

int rmh_exchange_buffers(rmh_ctx *c, double **send_buf, long long *send_doubles, double **ghost_buf, long long *ghost_doubles)
{
   if (!c || !c->xch) { return fail(RMH_ERR_STATE, "no exchange plan"); }
   if (send_buf) { *send_buf = c->xch->d_send; }
   if (send_doubles) { *send_doubles = c->xch->send_doubles; }
   if (ghost_buf) { *ghost_buf = c->xch->d_ghost; }
   if (ghost_doubles) { *ghost_doubles = c->xch->ghost_doubles; }
   return RMH_OK;
}

int rmh_exchange_peer(rmh_ctx *c, int k, int *rank, long long *send_offset, long long *send_doubles, long long *recv_offset,
                      long long *recv_doubles)
{
   if (!c || !c->xch || k < 0 || k >= (int)c->xch->peers.size()) { return fail(RMH_ERR_INVALID, "bad peer index"); }
   const PeerPlan &p = c->xch->peers[k];
   if (rank) { *rank = p.rank; }
   if (send_offset) { *send_offset = p.send_off; }
   if (send_doubles) { *send_doubles = p.send_n; }
   if (recv_offset) { *recv_offset = p.recv_off; }
   if (recv_doubles) { *recv_doubles = p.recv_n; }
   return RMH_OK;
}

int rmh_comm_count(rmh_ctx *c, int *nranks)
{
   if (!c || !nranks) { return fail(RMH_ERR_INVALID, "null argument"); }
   *nranks = 0;
   Exchange *x = c->xch;
   if (!x || !x->comm) { return RMH_OK; } // no communicator: 0 ranks
   RMH_NCCL(rccl().CommCount(x->comm, nranks));
   return RMH_OK;
}

int rmh_allreduce(rmh_ctx *c, double *vals, int n, int op)
{
   if (!c || !vals || n < 1 || n > 64 || op < 0 || op > 2) { return fail(RMH_ERR_INVALID, "bad argument"); }
   Exchange *x = c->xch;
   if (!x || !x->comm) { return fail(RMH_ERR_STATE, "rmh_allreduce needs an RCCL communicator (rmh_comm_init / rmh_comm_attach)"); }
   RMH_ENTER(c);
   if (!x->d_red) { RMH_HIP(hipMalloc((void **)&x->d_red, 64 * sizeof(double))); }
   RMH_HIP(hipMemcpyAsync(x->d_red, vals, n * sizeof(double), hipMemcpyHostToDevice, x->xs));
   const int rop = op == 0 ? Rccl::kSum : (op == 1 ? Rccl::kMin : Rccl::kMax);
   RMH_NCCL(rccl().AllReduce(x->d_red, x->d_red, (size_t)n, Rccl::kDouble, rop, x->comm, x->xs));
   RMH_HIP(hipMemcpyAsync(vals, x->d_red, n * sizeof(double), hipMemcpyDeviceToHost, x->xs));
   RMH_HIP(hipStreamSynchronize(x->xs));
   return RMH_OK;
}

} // extern "C"
