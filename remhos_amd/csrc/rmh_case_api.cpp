// C ABI of the host-side case builder (include/rmh_driver.h, rmhd_case_*).  No GPU code.
#include "../../include/rmh.h"
#include "../../include/rmh_driver.h"
#include "rmh_host.hpp"

#include <cstring>
#include <cstdio>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

struct rmhd_case
{
   remhos::CaseData d;
};

namespace remhos
{
thread_local std::string g_driver_error;

CaseConfig to_config(const rmhd_config &c)
{
   CaseConfig k;
   k.mesh = std::string(c.mesh, strnlen(c.mesh, sizeof(c.mesh)));
   k.rs = c.rs;
   k.order = c.order;
   k.problem = c.problem;
   k.dt = c.dt;
   k.t_final = c.t_final;
   k.max_steps = c.max_steps;
   k.lo_type = c.lo_type;
   k.px = c.px > 0 ? c.px : 1;
   k.py = c.py > 0 ? c.py : 1;
   k.pz = c.pz > 0 ? c.pz : 1;
   k.rank = c.rank;
   for (int d = 0; d < 3; d++) { k.rs_extra[d] = c.rs_extra[d]; }
   k.self_wrap = c.self_wrap;
   k.tile_rows = c.tile_rows > 0 ? c.tile_rows : 0;
   return k;
}
} // namespace remhos

extern "C" {

const char *rmhd_last_error(void) { return remhos::g_driver_error.c_str(); }

rmhd_case *rmhd_case_create(const rmhd_config *cfg)
{
   if (!cfg) { remhos::g_driver_error = "null config"; return nullptr; }
   rmhd_case *c = new rmhd_case;
   const std::string err = remhos::build_case(remhos::to_config(*cfg), c->d);
   if (!err.empty())
   {
      remhos::g_driver_error = err;
      delete c;
      return nullptr;
   }
   return c;
}

void rmhd_case_destroy(rmhd_case *c) { delete c; }

int rmhd_case_get_info(const rmhd_case *c, rmhd_case_info *info)
{
   if (!c || !info) { return -1; }
   const remhos::CaseData &d = c->d;
   info->order = d.order;
   info->exec_mode = d.exec_mode;
   info->ndof = d.ndof;
   info->ne_owned = d.ne_owned;
   info->ne_ghost = d.ne_ghost;
   info->ne_halo = d.ne_halo;
   info->dim = d.dim;
   info->n_peers = (int)d.peers.size();
   info->ne_global = d.ne_global;
   for (int k = 0; k < 3; k++)
   {
      info->n[k] = d.n[k];
      info->lo[k] = d.lo[k];
      info->nl[k] = d.nl[k];
      info->bb_min[k] = d.bb_min[k];
      info->bb_max[k] = d.bb_max[k];
   }
   info->dt = d.dt;
   return 0;
}

const double *rmhd_case_x0(const rmhd_case *c) { return c->d.x0.data(); }
const double *rmhd_case_vel(const rmhd_case *c) { return c->d.vel.data(); }
const double *rmhd_case_u0(const rmhd_case *c) { return c->d.u0.data(); }
const double *rmhd_case_s0(const rmhd_case *c) { return c->d.s0.data(); }
const double *rmhd_case_subcell_vel(const rmhd_case *c)
{
   return c->d.subcell_vel.empty() ? nullptr : c->d.subcell_vel.data();
}
const int *rmhd_case_face_nbr(const rmhd_case *c) { return c->d.face_nbr.data(); }
const int *rmhd_case_stencil27(const rmhd_case *c) { return c->d.stencil27.data(); }
const long long *rmhd_case_owned_gid(const rmhd_case *c) { return c->d.owned_gid.data(); }
const long long *rmhd_case_ghost_gid(const rmhd_case *c) { return c->d.ghost_gid.data(); }

int rmhd_case_peer(const rmhd_case *c, int k, int *rank, int *nsend, const int **send_elems, int *nrecv,
                   const int **recv_slots)
{
   if (!c || k < 0 || k >= (int)c->d.peers.size()) { return -1; }
   const remhos::Peer &p = c->d.peers[k];
   *rank = p.rank;
   *nsend = (int)p.send_elems.size();
   *send_elems = p.send_elems.data();
   *nrecv = (int)p.recv_slots.size();
   *recv_slots = p.recv_slots.data();
   return 0;
}

// -save (remhos.cpp:1015-1030, 1365-1380): the mesh at pseudo-time t and a DG field in MFEM's text formats
// ("MFEM mesh v1.0" with a nodal GridFunction, and a GridFunction file), so that a run can be opened in GLVis
// or diffed against meshHO_*.mesh / sltn_*.gf of a Remhos run elsewhere.  The mesh nodes are written as an L2
// (element-wise, Gauss-Lobatto = closed uniform at order 2) vector field for both meshes -- what the reference
// itself uses for the periodic ones (remhos.cpp:505-519); elements keep the lattice numbering (x fastest).
int rmhd_case_save(const rmhd_case *c, double t, const double *u, const char *mesh_path, const char *gf_path)
{
   if (!c) { return -1; }
   const std::string err = remhos::save_mfem(c->d, t, u, mesh_path, gf_path);
   if (!err.empty()) { remhos::g_driver_error = err; return -1; }
   return 0;
}


// include/rmh.h: neighbour tables from the vertex ids of the elements (any element numbering); DIM = 3: 8 corners, 6 faces,
// 27-entry stencil; DIM = 2: 4 corners, 4 faces, 9-entry stencil
extern "C++" {
template <int DIM>
static int build_tables_dim(int ne_owned, int ne_total, const int *ev, int *face_nbr, int *stencil)
{
   constexpr int NC = 1 << DIM, NF = 2 * DIM, NST = DIM == 3 ? 27 : 9, CENTRE = NST / 2, ALL = NC - 1;
   if (ne_owned <= 0 || ne_total < ne_owned || !ev || !face_nbr || !stencil) { return RMH_ERR_INVALID; }
   std::unordered_map<int, std::vector<int>> v2e; // vertex -> elements (owned and ghost)
   for (int e = 0; e < ne_total; e++) { for (int k = 0; k < NC; k++) { v2e[ev[(size_t)e * NC + k]].push_back(e); } }
   std::fill(stencil, stencil + (size_t)ne_owned * NST, -1);
   std::vector<int> cand;
   for (int e = 0; e < ne_owned; e++)
   {
      const int *ve = ev + (size_t)e * NC;
      cand.clear();
      for (int k = 0; k < NC; k++) { const auto &l = v2e[ve[k]]; cand.insert(cand.end(), l.begin(), l.end()); }
      std::sort(cand.begin(), cand.end());
      cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
      for (int o : cand)
      {
         if (o == e) { stencil[(size_t)e * NST + CENTRE] = e; continue; }
         const int *vo = ev + (size_t)o * NC;
         // corners of e shared with o, and the corners of o they coincide with
         int off[3] = {0, 0, 0}, n = 0, and_e = ALL, or_e = 0, and_o = ALL, or_o = 0;
         for (int k = 0; k < NC; k++)
         {
            for (int j = 0; j < NC; j++)
            {
               if (ve[k] == vo[j]) { n++; and_e &= k; or_e |= k; and_o &= j; or_o |= j; }
            }
         }
         bool ok = n == 1 || n == 2 || (DIM == 3 && n == 4);
         for (int d = 0; d < DIM && ok; d++)
         {
            const int be_and = (and_e >> d) & 1, be_or = (or_e >> d) & 1;
            const int bo_and = (and_o >> d) & 1, bo_or = (or_o >> d) & 1;
            if (be_and == 1) { off[d] = 1; ok = bo_or == 0; }         // all shared corners on e's high side: o's low side
            else if (be_or == 0) { off[d] = -1; ok = bo_and == 1; }   // all on e's low side: o's high side
            else { off[d] = 0; ok = bo_and == 0 && bo_or == 1; }      // spread over the direction on both elements
         }
         // not a face / edge / vertex contact with aligned axes: rotated neighbour, or a periodic direction with fewer
         // than 3 elements (the same element on both sides)
         if (!ok) { return RMH_ERR_INVALID; }
         stencil[(size_t)e * NST + (off[0] + 1) + 3 * (off[1] + 1) + (DIM == 3 ? 9 * (off[2] + 1) : 0)] = o;
      }
      const int fs3[6] = {12, 14, 10, 16, 4, 22}, fs2[4] = {3, 5, 1, 7};
      for (int f = 0; f < NF; f++) { face_nbr[(size_t)e * NF + f] = stencil[(size_t)e * NST + (DIM == 3 ? fs3[f] : fs2[f])]; }
   }
   return RMH_OK;
}
} // extern "C++"

int rmh_build_tables(int ne_owned, int ne_total, const int *ev, int *face_nbr, int *stencil27)
{
   return build_tables_dim<3>(ne_owned, ne_total, ev, face_nbr, stencil27);
}

int rmh_build_tables_2d(int ne_owned, int ne_total, const int *ev, int *face_nbr, int *stencil9)
{
   return build_tables_dim<2>(ne_owned, ne_total, ev, face_nbr, stencil9);
}

} // extern "C"
