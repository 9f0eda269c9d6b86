// Host-side case setup (see rmh_host.hpp).  Plain C++17, no GPU code.
#include "rmh_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <cstring>
#include <map>
#include <thread>

namespace remhos
{

namespace
{

struct MeshDef
{
   std::vector<double> coarse; // vertex coordinates per direction of the data/*.mesh lattice
   bool periodic;
};

// data/periodic-cube.mesh:114-328 (6-decimal coordinates are part of the
// reference's answers), data/cube01_hex.mesh:64-144
bool lookup_mesh(const std::string &name, MeshDef &m)
{
   if (name == "periodic-cube") { m = {{-1.0, -0.333333, 0.333333, 1.0}, true}; return true; }
   if (name == "cube01_hex") { m = {{0.0, 0.5, 1.0}, false}; return true; }
   return false;
}

std::vector<double> refine(std::vector<double> c, int levels)
{
   for (int l = 0; l < levels; l++)
   {
      std::vector<double> o(2 * c.size() - 1);
      for (size_t i = 0; i < c.size(); i++) { o[2 * i] = c[i]; }
      for (size_t i = 0; i + 1 < c.size(); i++) { o[2 * i + 1] = 0.5 * c[i] + 0.5 * c[i + 1]; }
      c.swap(o);
   }
   return c;
}

// Q2 Lagrange basis on {0, 1/2, 1}
inline void lag2(double t, double L[3])
{
   L[0] = 2.0 * (t - 0.5) * (t - 1.0);
   L[1] = -4.0 * t * (t - 1.0);
   L[2] = 2.0 * t * (t - 0.5);
}

void parallel_for(long long n, const std::function<void(long long, long long)> &body)
{
   unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
   if (n < 4096) { nt = 1; }
   std::vector<std::thread> pool;
   const long long chunk = (n + nt - 1) / nt;
   for (unsigned t = 0; t < nt; t++)
   {
      const long long a = t * chunk, b = std::min(n, a + chunk);
      if (a >= b) { break; }
      pool.emplace_back([=, &body]() { body(a, b); });
   }
   for (auto &th : pool) { th.join(); }
}

inline double box3D(double xmin, double xmax, double ymin, double ymax, double zmin, double zmax, double theta,
                    double ox, double oy, double x, double y, double z)
{
   const double s = std::sin(theta * M_PI / 180), c = std::cos(theta * M_PI / 180);
   const double xn = c * (x - ox) - s * (y - oy) + ox;
   const double yn = s * (x - ox) + c * (y - oy) + oy;
   return (xn > xmin && xn < xmax && yn > ymin && yn < ymax && z > zmin && z < zmax) ? 1.0 : 0.0;
}
inline double get_cross(double r1, double r2) { return r1 + r2 - r1 * r2; }
inline double ring(double rin, double rout, const double c[3], const double y[3])
{
   double r = 0.0;
   for (int i = 0; i < 3; i++) { r += (y[i] - c[i]) * (y[i] - c[i]); }
   r = std::sqrt(r);
   return (r > rin && r < rout) ? 1.0 : 0.0;
}

} // namespace

// remhos.cpp:2001-2120 (dim = 3)
void velocity_function(int problem, const double *bb_min, const double *bb_max, const double x[3], double v[3])
{
   double X[3];
   for (int i = 0; i < 3; i++)
   {
      const double center = (bb_min[i] + bb_max[i]) * 0.5;
      X[i] = 2 * (x[i] - center) / (bb_max[i] - bb_min[i]);
   }
   v[0] = v[1] = v[2] = 0.0;
   switch (problem % 20)
   {
      case 0:
         v[0] = std::sqrt(3. / 6.); v[1] = std::sqrt(2. / 6.); v[2] = std::sqrt(1. / 6.);
         break;
      case 1:
      case 2:
      case 4:
      {
         const double w = M_PI / 2;
         v[0] = -w * X[1]; v[1] = w * X[0]; v[2] = 0.0;
         break;
      }
      case 5:
         v[0] = 1.0; v[1] = 1.0; v[2] = 1.0;
         break;
      case 10:
      case 12:
      case 13:
      case 14:
      case 15:
      case 16:
      case 17:
      {
         // Taylor-Green deformation; map [-1,1] to [0,1]
         for (int d = 0; d < 3; d++) { X[d] = X[d] * 0.5 + 0.5; }
         v[0] = std::sin(M_PI * X[0]) * std::cos(M_PI * X[1]);
         v[1] = -std::cos(M_PI * X[0]) * std::sin(M_PI * X[1]);
         v[0] *= std::cos(M_PI * X[2]);
         v[1] *= std::cos(M_PI * X[2]);
         v[2] = 0.0;
         break;
      }
      default: break;
   }
}

// remhos.cpp:2201-2355 (dim = 3)
double u0_function(int problem, const double *bb_min, const double *bb_max, const double x[3])
{
   double X[3];
   for (int i = 0; i < 3; i++)
   {
      const double center = (bb_min[i] + bb_max[i]) * 0.5;
      X[i] = 2 * (x[i] - center) / (bb_max[i] - bb_min[i]);
   }
   switch (problem % 10)
   {
      case 0:
      case 1:
      {
         double rx = 0.45, ry = 0.25;
         const double cx = 0., cy = -0.2, w = 10.;
         const double s = (1. + 0.25 * std::cos(2 * M_PI * X[2]));
         rx *= s;
         ry *= s;
         return (std::erfc(w * (X[0] - cx - rx)) * std::erfc(-w * (X[0] - cx + rx)) *
                 std::erfc(w * (X[1] - cy - ry)) * std::erfc(-w * (X[1] - cy + ry))) / 16;
      }
      case 4:
      {
         const double scale = 0.0225;
         const double coef = (0.5 / std::sqrt(scale));
         const bool slit = (X[0] <= -0.05) || (X[0] >= 0.05) || (X[1] >= 0.7);
         const double cone = coef * std::sqrt(std::pow(X[0], 2.) + std::pow(X[1] + 0.5, 2.));
         const double hump = coef * std::sqrt(std::pow(X[0] + 0.5, 2.) + std::pow(X[1], 2.));
         return (slit && ((std::pow(X[0], 2.) + std::pow(X[1] - .5, 2.)) <= 4. * scale))
                   ? 1.
                   : 0. + (1. - cone) * (std::pow(X[0], 2.) + std::pow(X[1] + .5, 2.) <= 4. * scale) +
                        .25 * (1. + std::cos(M_PI * hump)) *
                           ((std::pow(X[0] + .5, 2.) + std::pow(X[1], 2.)) <= 4. * scale);
      }
      case 5:
      {
         double y[3];
         for (int i = 0; i < 3; i++) { y[i] = 50. * (x[i] + 1.); }
         double rect1 = box3D(7., 32., 10., 13., 10., 13., -45., 15.5, 11.5, y[0], y[1], y[2]);
         double rect2 = box3D(14., 17., 3., 26., 10., 13., -45., 15.5, 11.5, y[0], y[1], y[2]);
         double rect3 = box3D(14., 17., 10., 13., 3., 26., -45., 15.5, 11.5, y[0], y[1], y[2]);
         double cross = get_cross(get_cross(rect1, rect2), rect3);
         const double c1[3] = {40., 40., 40.}, c2[3] = {40., 20., 20.};
         const double dom2 = cross + ring(7., 10., c1, y) + ring(3., 7., c2, y);
         rect1 = box3D(2., 27., 30., 33., 30., 33., 0., 0., 0., y[0], y[1], y[2]);
         rect2 = box3D(9., 12., 23., 46., 30., 33., 0., 0., 0., y[0], y[1], y[2]);
         rect3 = box3D(9., 12., 30., 33., 23., 46., 0., 0., 0., y[0], y[1], y[2]);
         cross = get_cross(get_cross(rect1, rect2), rect3);
         const double dom3 = cross + ring(0., 7., c1, y) + ring(0., 3., c2, y) + ring(7., 10., c2, y);
         const double dom1 = 1. - get_cross(dom2, dom3);
         return dom1 + 2. * dom2 + 3. * dom3;
      }
      default: return 0.0;
   }
}

// Simple nonlinear function (remhos.cpp:2357-2361): physical coordinates, no bounding-box map
double s0_function(const double x[3]) { return 2.0 + std::sin(2 * M_PI * x[0]) * std::sin(2 * M_PI * x[1]); }

// ---- dim = 2 ------------------------------------------------------------------------------------------
// remhos.cpp:2001-2120 with dim = 2
void velocity_function_2d(int problem, const double *bb_min, const double *bb_max, const double x[2], double v[2])
{
   double X[2];
   for (int i = 0; i < 2; i++) { X[i] = 2 * (x[i] - (bb_min[i] + bb_max[i]) * 0.5) / (bb_max[i] - bb_min[i]); }
   v[0] = v[1] = 0.0;
   switch (problem % 20)
   {
      case 0: v[0] = std::sqrt(2. / 3.); v[1] = std::sqrt(1. / 3.); break;
      case 1:
      case 2:
      case 4: { const double w = M_PI / 2; v[0] = -w * X[1]; v[1] = w * X[0]; break; }
      case 5: v[0] = 1.0; v[1] = 1.0; break;
      case 10:
      case 12:
      case 13:
      case 14:
      case 15:
      case 16:
      case 17:
         // Taylor-Green deformation; map [-1,1] to [0,1]
         for (int d = 0; d < 2; d++) { X[d] = X[d] * 0.5 + 0.5; }
         v[0] = std::sin(M_PI * X[0]) * std::cos(M_PI * X[1]);
         v[1] = -std::cos(M_PI * X[0]) * std::sin(M_PI * X[1]);
         break;
      default: break;
   }
}

namespace
{
inline double box2D(double x0, double y0, double x1, double y1, double theta, double ox, double oy, double x, double y)
{
   const double s = std::sin(theta * M_PI / 180), c = std::cos(theta * M_PI / 180);
   const double xn = c * (x - ox) - s * (y - oy) + ox;
   const double yn = s * (x - ox) + c * (y - oy) + oy;
   return (xn > x0 && xn < x1 && yn > y0 && yn < y1) ? 1.0 : 0.0;
}
inline double ring2(double rin, double rout, double cx, double cy, const double y[2])
{
   const double r = std::sqrt((y[0] - cx) * (y[0] - cx) + (y[1] - cy) * (y[1] - cy));
   return (r > rin && r < rout) ? 1.0 : 0.0;
}
} // namespace

// remhos.cpp:2201-2355 with dim = 2
double u0_function_2d(int problem, const double *bb_min, const double *bb_max, const double x[2])
{
   double X[2];
   for (int i = 0; i < 2; i++) { X[i] = 2 * (x[i] - (bb_min[i] + bb_max[i]) * 0.5) / (bb_max[i] - bb_min[i]); }
   switch (problem % 10)
   {
      case 0:
      case 1:
      {
         const double rx = 0.45, ry = 0.25, cx = 0., cy = -0.2, w = 10.;
         return (std::erfc(w * (X[0] - cx - rx)) * std::erfc(-w * (X[0] - cx + rx)) * std::erfc(w * (X[1] - cy - ry)) *
                 std::erfc(-w * (X[1] - cy + ry))) / 16;
      }
      case 4:
      {
         // slotted disk, cone and hump ("pacman" remap runs use it as problem 14)
         const double scale = 0.0225, coef = 0.5 / std::sqrt(scale);
         const bool slit = (X[0] <= -0.05) || (X[0] >= 0.05) || (X[1] >= 0.7);
         const double r_disk = X[0] * X[0] + (X[1] - 0.5) * (X[1] - 0.5);
         const double r_cone = X[0] * X[0] + (X[1] + 0.5) * (X[1] + 0.5);
         const double r_hump = (X[0] + 0.5) * (X[0] + 0.5) + X[1] * X[1];
         if (slit && r_disk <= 4. * scale) { return 1.0; }
         return (1. - coef * std::sqrt(r_cone)) * (r_cone <= 4. * scale ? 1.0 : 0.0) +
                .25 * (1. + std::cos(M_PI * coef * std::sqrt(r_hump))) * (r_hump <= 4. * scale ? 1.0 : 0.0);
      }
      case 5:
      {
         // balls and jacks: a rotated cross and two rings, indicator values
         const double y[2] = {50. * (x[0] + 1.), 50. * (x[1] + 1.)};
         const double rect1 = box2D(14., 3., 17., 26., -45., 15.5, 11.5, y[0], y[1]);
         const double rect2 = box2D(7., 10., 32., 13., -45., 15.5, 11.5, y[0], y[1]);
         return (rect1 + rect2 - rect1 * rect2) + ring2(7., 10., 40., 40., y) + ring2(3., 7., 40., 20., y);
      }
      default: return 0.0;
   }
}

// The problem ids whose velocity field AND initial condition are implemented above (remhos.cpp:2001-2355 defines more:
// 2, 3, 6, 7 and 11 have no branch here).  An id outside the set used to run silently with v = 0 or u0 = 0 -- and with
// -dt -1 a CFL step of 0.25 h / 1e-7; now it is an error of the case builder.
static std::string check_problem(int problem)
{
   const int pv = problem % 20, pu = problem % 10;
   const bool vel_ok = pv == 0 || pv == 1 || pv == 2 || pv == 4 || pv == 5 || pv == 10 || (pv >= 12 && pv <= 17);
   const bool u0_ok = pu == 0 || pu == 1 || pu == 4 || pu == 5;
   if (problem < 0 || !vel_ok || !u0_ok)
   {
      return "problem " + std::to_string(problem) + " is not implemented (velocity: 0, 1, 2, 4, 5, 10, 12-17 mod 20; initial field: 0, 1, 4, 5 mod 10)";
   }
   return "";
}


// The reference's 2-D lattice meshes on ONE rank: data/inline-quad.mesh (4 x 4 quadrilaterals on [0,1]^2) and
// data/periodic-square.mesh (3 x 3 on [-1,1]^2, 9-decimal coordinates: part of the reference's answers).  Same steps as the
// 3-D builder below: CFL step, Q2 nodes and remap displacement (with the reference's "t advances first" loop), per-element
// node copies, nodal initial condition at the closed-uniform points, sub-mesh velocity for lo 4, neighbour tables.
static std::string build_case_2d(const CaseConfig &cfg, CaseData &out)
{
   std::vector<double> coarse;
   bool periodic = false;
   if (cfg.mesh == "inline-quad") { coarse = {0.0, 0.25, 0.5, 0.75, 1.0}; }
   else if (cfg.mesh == "periodic-square") { coarse = {-1.0, -0.333333333, 0.333333333, 1.0}; periodic = true; }
   else { return "unknown 2-D lattice mesh '" + cfg.mesh + "'"; }
   if (cfg.order < 1 || cfg.order > 6) { return "order must be in 1..6"; }
   if (cfg.px != 1 || cfg.py != 1 || cfg.pz != 1 || cfg.rank != 0 || cfg.self_wrap != 0) { return "dim = 2 runs on one rank"; }
   if (cfg.rs_extra[0] || cfg.rs_extra[1] || cfg.rs_extra[2]) { return "dim = 2: no rs_extra"; }
   if (cfg.lo_type != 3 && cfg.lo_type != 4 && cfg.lo_type != 5) { return "lo_type must be 3, 4 or 5"; }
   if (const std::string e = check_problem(cfg.problem); !e.empty()) { return e; }
   const std::vector<double> verts = refine(coarse, cfg.rs);
   const int N = (int)verts.size() - 1, p = cfg.order, D = p + 1, problem = cfg.problem;
   out = CaseData();
   out.dim = 2;
   out.order = p;
   out.exec_mode = problem < 10 ? 0 : 1;
   out.ndof = D * D;
   out.periodic = periodic;
   out.ne_global = (long long)N * N;
   out.ne_owned = N * N;
   for (int d = 0; d < 3; d++)
   {
      out.n[d] = d < 2 ? N : 1;
      out.lo[d] = 0;
      out.nl[d] = out.n[d];
      out.bb_min[d] = d < 2 ? verts.front() : 0.0;
      out.bb_max[d] = d < 2 ? verts.back() : 0.0;
   }
   const int ne = out.ne_owned, nn1 = 2 * N + 1;
   const double *bmin = out.bb_min, *bmax = out.bb_max;
   std::vector<double> n1(nn1);
   for (int i = 0; i <= N; i++) { n1[2 * i] = verts[i]; }
   for (int i = 0; i < N; i++) { n1[2 * i + 1] = 0.5 * verts[i] + 0.5 * verts[i + 1]; }
   // time step (remhos.cpp:538-553): 0.25 * |det J(center)|^(1/2) / |v(center)|
   double dt = cfg.dt;
   if (dt < 0.0)
   {
      dt = INFINITY;
      for (int ey = 0; ey < N; ey++)
      {
         for (int ex = 0; ex < N; ex++)
         {
            const double hx = n1[2 * ex + 2] - n1[2 * ex], hy = n1[2 * ey + 2] - n1[2 * ey];
            const double xc[2] = {n1[2 * ex + 1], n1[2 * ey + 1]};
            double v[2];
            velocity_function_2d(problem, bmin, bmax, xc, v);
            dt = std::fmin(dt, 0.25 * std::sqrt(std::fabs(hx * hy)) / std::sqrt(v[0] * v[0] + v[1] * v[1] + 1e-14));
         }
      }
   }
   out.dt = dt;
   // lattice nodes: positions and remap displacement (remhos.cpp:562-584)
   const bool remap = out.exec_mode == 1;
   std::vector<double> xn((size_t)2 * nn1 * nn1), vn((size_t)2 * nn1 * nn1);
   for (int iy = 0; iy < nn1; iy++)
   {
      for (int ix = 0; ix < nn1; ix++)
      {
         const double x0[2] = {n1[ix], n1[iy]};
         double x[2] = {x0[0], x0[1]}, v[2];
         velocity_function_2d(problem, bmin, bmax, x, v);
         if (remap)
         {
            double t = 0.0; // (the reference's order: t advances BEFORE min(dt, t_final - t) is taken)
            while (t < cfg.t_final)
            {
               t += dt;
               const double hh = std::min(dt, cfg.t_final - t);
               for (int c = 0; c < 2; c++) { x[c] = x[c] + hh * v[c]; }
               velocity_function_2d(problem, bmin, bmax, x, v);
            }
            for (int c = 0; c < 2; c++) { v[c] = x[c] - x0[c]; }
         }
         const size_t k = (size_t)ix + (size_t)nn1 * iy;
         for (int c = 0; c < 2; c++) { xn[2 * k + c] = x0[c]; vn[2 * k + c] = v[c]; }
      }
   }
   out.x0.resize((size_t)ne * 18);
   out.vel.resize((size_t)ne * 18);
   out.u0.resize((size_t)ne * out.ndof);
   out.s0.resize((size_t)ne * out.ndof);
   const bool lo4 = cfg.lo_type == 4;
   if (lo4) { out.subcell_vel.assign((size_t)ne * 2 * out.ndof, 0.0); }
   out.owned_gid.resize(ne);
   std::vector<double> Lcu(3 * D);
   for (int i = 0; i < D; i++) { lag2((double)i / p, &Lcu[3 * i]); }
   for (int e = 0; e < ne; e++)
   {
      const int ex = e % N, ey = e / N;
      out.owned_gid[e] = e;
      double *ex0 = &out.x0[(size_t)e * 18], *ev = &out.vel[(size_t)e * 18];
      for (int ay = 0; ay < 3; ay++)
      {
         for (int ax = 0; ax < 3; ax++)
         {
            const size_t k = (size_t)(2 * ex + ax) + (size_t)nn1 * (2 * ey + ay);
            for (int c = 0; c < 2; c++)
            {
               ex0[c * 9 + ax + 3 * ay] = xn[2 * k + c];
               ev[c * 9 + ax + 3 * ay] = vn[2 * k + c];
            }
         }
      }
      for (int iy = 0; iy < D; iy++)
      {
         for (int ix = 0; ix < D; ix++)
         {
            double x[3] = {0, 0, 0};
            for (int ay = 0; ay < 3; ay++)
            {
               for (int ax = 0; ax < 3; ax++)
               {
                  const double w = Lcu[3 * ix + ax] * Lcu[3 * iy + ay];
                  for (int c = 0; c < 2; c++) { x[c] += w * ex0[c * 9 + ax + 3 * ay]; }
               }
            }
            const int i = ix + D * iy;
            out.u0[(size_t)e * out.ndof + i] = u0_function_2d(problem, bmin, bmax, x); // remhos.cpp:878-884
            out.s0[(size_t)e * out.ndof + i] = s0_function(x);
            if (lo4)
            {
               // remap: v_sub_gf, the instantaneous velocity at the sub-mesh nodes, zero on the domain boundary of non-periodic
               // meshes (remhos.cpp:837-853); transport: the advection velocity at the sub-mesh nodes
               double v[2];
               velocity_function_2d(problem, bmin, bmax, x, v);
               const bool bdr = remap && !periodic && ((ex == 0 && ix == 0) || (ex == N - 1 && ix == p) || (ey == 0 && iy == 0) || (ey == N - 1 && iy == p));
               for (int c = 0; c < 2; c++) { out.subcell_vel[((size_t)e * 2 + c) * out.ndof + i] = bdr ? 0.0 : v[c]; }
            }
         }
      }
   }
   // neighbour tables: 3 x 3 stencil (entry (ox+1) + 3 (oy+1)), faces f = 2 c + side
   out.stencil27.assign((size_t)ne * 9, -1);
   out.face_nbr.assign((size_t)ne * 4, -1);
   if (periodic && N < 3) { return "a periodic direction needs at least 3 elements"; }
   for (int e = 0; e < ne; e++)
   {
      const int ex = e % N, ey = e / N;
      for (int oy = -1; oy <= 1; oy++)
      {
         for (int ox = -1; ox <= 1; ox++)
         {
            int gx = ex + ox, gy = ey + oy;
            if (periodic) { gx = (gx + N) % N; gy = (gy + N) % N; }
            if (gx < 0 || gx >= N || gy < 0 || gy >= N) { continue; }
            out.stencil27[(size_t)e * 9 + (ox + 1) + 3 * (oy + 1)] = gx + N * gy;
         }
      }
      const int fs[4] = {3, 5, 1, 7};
      for (int f = 0; f < 4; f++) { out.face_nbr[(size_t)e * 4 + f] = out.stencil27[(size_t)e * 9 + fs[f]]; }
   }
   return "";
}

std::string build_case(const CaseConfig &cfg, CaseData &out)
{
   if (cfg.mesh == "inline-quad" || cfg.mesh == "periodic-square") { return build_case_2d(cfg, out); }
   MeshDef md;
   if (!lookup_mesh(cfg.mesh, md)) { return "unknown lattice mesh '" + cfg.mesh + "' (periodic-cube, cube01_hex)"; }
   if (cfg.order < 1 || cfg.order > 6) { return "order must be in 1..6"; }
   if (cfg.px < 1 || cfg.py < 1 || cfg.pz < 1) { return "bad partition"; }
   const int nranks = cfg.px * cfg.py * cfg.pz;
   if (cfg.rank < 0 || cfg.rank >= nranks) { return "bad rank"; }
   if (cfg.lo_type != 3 && cfg.lo_type != 4 && cfg.lo_type != 5) { return "lo_type must be 3, 4 or 5"; }
   if (const std::string e = check_problem(cfg.problem); !e.empty()) { return e; }

   // per direction: -rs levels plus rs_extra[d] more (0 in the reference's meshes; the weak-scaling lattices of
   // bench.py refine the directions that carry two partition blocks once more)
   std::vector<double> vertsd[3];
   int Nd[3];
   const int P[3] = {cfg.px, cfg.py, cfg.pz};
   for (int d = 0; d < 3; d++)
   {
      if (cfg.rs_extra[d] < 0 || cfg.rs_extra[d] > 4) { return "bad rs_extra"; }
      vertsd[d] = refine(md.coarse, cfg.rs + cfg.rs_extra[d]);
      Nd[d] = (int)vertsd[d].size() - 1;
      if (P[d] > Nd[d]) { return "more partition blocks than elements in a direction"; }
   }
   const int sw = cfg.self_wrap - 1; // direction whose periodic wrap is exchanged with the own rank, or -1
   if (cfg.self_wrap < 0 || cfg.self_wrap > 3) { return "bad self_wrap"; }
   if (sw >= 0 && (!md.periodic || P[sw] != 1 || Nd[sw] < 3)) { return "self_wrap needs a periodic mesh, one block and >= 3 elements in that direction"; }
   const int Nx = Nd[0], Ny = Nd[1];
   out = CaseData();
   out.order = cfg.order;
   out.exec_mode = cfg.problem < 10 ? 0 : 1; // remhos.cpp:437-440
   const int D = cfg.order + 1, p = cfg.order;
   out.ndof = D * D * D;
   out.periodic = md.periodic;
   out.ne_global = (long long)Nd[0] * Nd[1] * Nd[2];
   for (int d = 0; d < 3; d++)
   {
      out.n[d] = Nd[d];
      out.bb_min[d] = vertsd[d].front();
      out.bb_max[d] = vertsd[d].back();
   }
   const int r3[3] = {cfg.rank % cfg.px, (cfg.rank / cfg.px) % cfg.py, cfg.rank / (cfg.px * cfg.py)};
   auto chunk_lo = [&](int d, int r) { return (int)(((long long)r * Nd[d]) / P[d]); };
   for (int d = 0; d < 3; d++)
   {
      out.lo[d] = chunk_lo(d, r3[d]);
      out.nl[d] = chunk_lo(d, r3[d] + 1) - out.lo[d];
   }
   auto owner_1d = [&](int d, int g)
   {
      // inverse of chunk_lo
      int r = (int)(((long long)(g + 1) * P[d] - 1) / Nd[d]);
      while (chunk_lo(d, r) > g) { r--; }
      while (chunk_lo(d, r + 1) <= g) { r++; }
      return r;
   };
   const int nlx = out.nl[0], nly = out.nl[1], nlz = out.nl[2];
   out.ne_owned = nlx * nly * nlz;
   const int ne = out.ne_owned;
   const double *bmin = out.bb_min, *bmax = out.bb_max;
   const int problem = cfg.problem;

   // mesh nodes per direction (Gauss-Lobatto order 2 = vertices and midpoints)
   std::vector<double> nodes1d[3];
   for (int d = 0; d < 3; d++)
   {
      nodes1d[d].resize(2 * Nd[d] + 1);
      for (int i = 0; i <= Nd[d]; i++) { nodes1d[d][2 * i] = vertsd[d][i]; }
      for (int i = 0; i < Nd[d]; i++) { nodes1d[d][2 * i + 1] = 0.5 * vertsd[d][i] + 0.5 * vertsd[d][i + 1]; }
   }

   // ---- time step (remhos.cpp:538-553): 0.25 * h_e / |v(center)|, min over ALL elements ------------
   double dt = cfg.dt;
   if (dt < 0.0)
   {
      dt = INFINITY;
      for (int ez = 0; ez < Nd[2]; ez++)
      {
         for (int ey = 0; ey < Nd[1]; ey++)
         {
            for (int ex = 0; ex < Nd[0]; ex++)
            {
               // J(center) of the Q2 lattice element: differences of the mid-face nodes
               const double hx = nodes1d[0][2 * ex + 2] - nodes1d[0][2 * ex];
               const double hy = nodes1d[1][2 * ey + 2] - nodes1d[1][2 * ey];
               const double hz = nodes1d[2][2 * ez + 2] - nodes1d[2][2 * ez];
               const double length = std::pow(std::fabs(hx * hy * hz), 1. / 3.);
               const double xc[3] = {nodes1d[0][2 * ex + 1], nodes1d[1][2 * ey + 1], nodes1d[2][2 * ez + 1]};
               double v[3];
               velocity_function(problem, bmin, bmax, xc, v);
               const double speed = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + 1e-14);
               dt = std::fmin(dt, 0.25 * length / speed);
            }
         }
      }
   }
   out.dt = dt;

   // ---- local lattice nodes: positions and remap displacement (remhos.cpp:562-584) ----------------
   const int nnx = 2 * nlx + 1, nny = 2 * nly + 1, nnz = 2 * nlz + 1;
   const long long nn = (long long)nnx * nny * nnz;
   std::vector<double> xn(3 * nn), vn(3 * nn);
   const bool remap = out.exec_mode == 1;
   const double t_final = cfg.t_final;
   parallel_for(nn, [&](long long a, long long b)
   {
      for (long long k = a; k < b; k++)
      {
         const int ix = (int)(k % nnx), iy = (int)((k / nnx) % nny), iz = (int)(k / ((long long)nnx * nny));
         const double x0[3] = {nodes1d[0][2 * out.lo[0] + ix], nodes1d[1][2 * out.lo[1] + iy], nodes1d[2][2 * out.lo[2] + iz]};
         double x[3] = {x0[0], x0[1], x0[2]}, v[3];
         velocity_function(problem, bmin, bmax, x, v);
         if (remap)
         {
            // note the reference's order: t advances BEFORE min(dt, t_final - t) is taken
            double t = 0.0;
            while (t < t_final)
            {
               t += dt;
               const double h = std::min(dt, t_final - t);
               for (int c = 0; c < 3; c++) { x[c] = x[c] + h * v[c]; }
               velocity_function(problem, bmin, bmax, x, v);
            }
            for (int c = 0; c < 3; c++) { v[c] = x[c] - x0[c]; }
         }
         for (int c = 0; c < 3; c++)
         {
            xn[3 * k + c] = x0[c];
            vn[3 * k + c] = v[c];
         }
      }
   });

   // ---- per-element copies (L2 nodal E-vector), initial condition, subcell velocity --------------
   out.x0.resize((size_t)ne * 81);
   out.vel.resize((size_t)ne * 81);
   out.u0.resize((size_t)ne * out.ndof);
   out.s0.resize((size_t)ne * out.ndof);
   const bool lo4 = cfg.lo_type == 4;
   if (lo4) { out.subcell_vel.assign((size_t)ne * 3 * out.ndof, 0.0); }
   out.owned_gid.resize(ne);
   std::vector<double> Lcu(3 * D); // Q2 Lagrange at the closed-uniform points i/p
   for (int i = 0; i < D; i++) { lag2((double)i / p, &Lcu[3 * i]); }
   parallel_for(ne, [&](long long a, long long b)
   {
      for (long long e = a; e < b; e++)
      {
         const int lx = (int)(e % nlx), ly = (int)((e / nlx) % nly), lz = (int)(e / ((long long)nlx * nly));
         out.owned_gid[e] = (long long)(out.lo[0] + lx) + (long long)Nx * ((out.lo[1] + ly) + (long long)Ny * (out.lo[2] + lz));
         double *ex0 = &out.x0[(size_t)e * 81], *ev = &out.vel[(size_t)e * 81];
         for (int az = 0; az < 3; az++)
         {
            for (int ay = 0; ay < 3; ay++)
            {
               for (int ax = 0; ax < 3; ax++)
               {
                  const long long k = (2 * lx + ax) + (long long)nnx * ((2 * ly + ay) + (long long)nny * (2 * lz + az));
                  const int a3 = ax + 3 * (ay + 3 * az);
                  for (int c = 0; c < 3; c++)
                  {
                     ex0[c * 27 + a3] = xn[3 * k + c];
                     ev[c * 27 + a3] = vn[3 * k + c];
                  }
               }
            }
         }
         // u_i = u0(x(xi_i)) at the closed-uniform points (remhos.cpp:878-884)
         for (int iz = 0; iz < D; iz++)
         {
            for (int iy = 0; iy < D; iy++)
            {
               for (int ix = 0; ix < D; ix++)
               {
                  double x[3] = {0, 0, 0};
                  for (int az = 0; az < 3; az++)
                  {
                     for (int ay = 0; ay < 3; ay++)
                     {
                        for (int ax = 0; ax < 3; ax++)
                        {
                           const double w = Lcu[3 * ix + ax] * Lcu[3 * iy + ay] * Lcu[3 * iz + az];
                           const int a3 = ax + 3 * (ay + 3 * az);
                           for (int c = 0; c < 3; c++) { x[c] += w * ex0[c * 27 + a3]; }
                        }
                     }
                  }
                  const int i = ix + D * (iy + D * iz);
                  out.u0[(size_t)e * out.ndof + i] = u0_function(problem, bmin, bmax, x);
                  out.s0[(size_t)e * out.ndof + i] = s0_function(x);
                  if (lo4)
                  {
                     // remap: v_sub_gf, the instantaneous velocity at the sub-mesh nodes, zero on the
                     // domain boundary of non-periodic meshes (remhos.cpp:837-853);
                     // transport: the advection velocity at the sub-mesh nodes (the subcell midpoint
                     // value lom.subcellCoeff->Eval of remhos_lo.cpp:1051-1082 is their mean)
                     double v[3];
                     velocity_function(problem, bmin, bmax, x, v);
                     bool bdr = false;
                     if (remap && !md.periodic)
                     {
                        const int g[3] = {out.lo[0] + lx, out.lo[1] + ly, out.lo[2] + lz};
                        const int id[3] = {ix, iy, iz};
                        for (int c = 0; c < 3; c++)
                        {
                           bdr = bdr || (g[c] == 0 && id[c] == 0) || (g[c] == Nd[c] - 1 && id[c] == p);
                        }
                     }
                     for (int c = 0; c < 3; c++)
                     {
                        out.subcell_vel[((size_t)e * 3 + c) * out.ndof + i] = bdr ? 0.0 : v[c];
                     }
                  }
               }
            }
         }
      }
   });

   // ---- topology: 27-stencil with ghosts, face neighbours, halo lists ------------------------------
   out.stencil27.assign((size_t)ne * 27, -1);
   out.face_nbr.assign((size_t)ne * 6, -1);
   std::map<std::pair<int, long long>, int> ghost_slot; // (owner, gid) -> slot, ordered
   std::map<int, std::vector<long long>> send_sets;       // peer -> gids of owned elements it needs
   bool seam = false; // did the last stencil entry cross the seam of the self-wrapped direction?
   auto wrap = [&](int d, int g, bool &ok)
   {
      if (g >= 0 && g < Nd[d]) { return g; }
      if (!md.periodic) { ok = false; return 0; }
      if (d == sw) { seam = true; }
      return (g % Nd[d] + Nd[d]) % Nd[d];
   };
   // pass 1: collect ghosts and send sets (only elements near the box surface have remote neighbours)
   for (int lz = 0; lz < nlz; lz++)
   {
      for (int ly = 0; ly < nly; ly++)
      {
         for (int lx = 0; lx < nlx; lx++)
         {
            const bool surf = lx == 0 || ly == 0 || lz == 0 || lx == nlx - 1 || ly == nly - 1 || lz == nlz - 1;
            if (!surf) { continue; }
            const long long mygid = out.owned_gid[lx + (long long)nlx * (ly + (long long)nly * lz)];
            for (int s = 0; s < 27; s++)
            {
               const int o[3] = {s % 3 - 1, (s / 3) % 3 - 1, s / 9 - 1};
               bool ok = true;
               seam = false;
               const int g[3] = {wrap(0, out.lo[0] + lx + o[0], ok), wrap(1, out.lo[1] + ly + o[1], ok),
                                 wrap(2, out.lo[2] + lz + o[2], ok)};
               if (!ok) { continue; }
               const int owner = owner_1d(0, g[0]) + cfg.px * (owner_1d(1, g[1]) + cfg.py * owner_1d(2, g[2]));
               // (self-loop: an element reached across the seam is a ghost copy owned by this very rank; the layers on
               // the two sides of the seam are distinct elements, so the global id still names the copy)
               if (owner == cfg.rank && !seam) { continue; }
               const long long gid = g[0] + (long long)Nx * (g[1] + (long long)Ny * g[2]);
               ghost_slot[{owner, gid}] = 0;
               send_sets[owner].push_back(mygid);
            }
         }
      }
   }
   int slot = 0;
   out.ghost_gid.clear();
   for (auto &kv : ghost_slot)
   {
      kv.second = slot++;
      out.ghost_gid.push_back(kv.first.second);
   }
   out.ne_ghost = slot;
   auto local_of_gid = [&](long long gid)
   {
      const int gx = (int)(gid % Nx), gy = (int)((gid / Nx) % Ny), gz = (int)(gid / ((long long)Nx * Ny));
      return (gx - out.lo[0]) + nlx * ((gy - out.lo[1]) + nly * (gz - out.lo[2]));
   };
   std::map<int, Peer> peers;
   for (auto &kv : ghost_slot)
   {
      Peer &pr = peers[kv.first.first];
      pr.rank = kv.first.first;
      pr.recv_slots.push_back(kv.second);
   }
   for (auto &kv : send_sets)
   {
      std::vector<long long> &g = kv.second;
      std::sort(g.begin(), g.end());
      g.erase(std::unique(g.begin(), g.end()), g.end());
      Peer &pr = peers[kv.first];
      pr.rank = kv.first;
      for (long long gid : g) { pr.send_elems.push_back(local_of_gid(gid)); }
   }
   for (auto &kv : peers) { out.peers.push_back(kv.second); }
   // pass 2: fill the tables
   for (int lz = 0; lz < nlz; lz++)
   {
      for (int ly = 0; ly < nly; ly++)
      {
         for (int lx = 0; lx < nlx; lx++)
         {
            const int e = lx + nlx * (ly + nly * lz);
            for (int s = 0; s < 27; s++)
            {
               const int o[3] = {s % 3 - 1, (s / 3) % 3 - 1, s / 9 - 1};
               bool ok = true;
               seam = false;
               const int g[3] = {wrap(0, out.lo[0] + lx + o[0], ok), wrap(1, out.lo[1] + ly + o[1], ok),
                                 wrap(2, out.lo[2] + lz + o[2], ok)};
               if (!ok) { continue; }
               int idx;
               const int l[3] = {g[0] - out.lo[0], g[1] - out.lo[1], g[2] - out.lo[2]};
               if (!seam && l[0] >= 0 && l[0] < nlx && l[1] >= 0 && l[1] < nly && l[2] >= 0 && l[2] < nlz)
               {
                  idx = l[0] + nlx * (l[1] + nly * l[2]);
               }
               else
               {
                  const int owner = owner_1d(0, g[0]) + cfg.px * (owner_1d(1, g[1]) + cfg.py * owner_1d(2, g[2]));
                  const long long gid = g[0] + (long long)Nx * (g[1] + (long long)Ny * g[2]);
                  idx = ne + ghost_slot.at({owner, gid});
               }
               out.stencil27[(size_t)e * 27 + s] = idx;
            }
            // face neighbours are the six axis entries of the stencil
            const int fs[6] = {12, 14, 10, 16, 4, 22};
            for (int f = 0; f < 6; f++) { out.face_nbr[(size_t)e * 6 + f] = out.stencil27[(size_t)e * 27 + fs[f]]; }
         }
      }
   }
   // ---- element order ----------------------------------------------------------------------------
   // Halo first: owned elements whose 27-stencil reaches a ghost come first (they form the shell of the box), so
   // that one stage can run as two element ranges: [ne_halo, ne) while the neighbour exchange is in
   // flight, [0, ne_halo) after it.  Behind that, the lattice order (x, y, z) -- or, with cfg.tile_rows = T > 0, strips of
   // T lattice rows in y with z running before y inside a strip: a stage kernel works on a few hundred consecutive elements
   // per XCD at a time, and in that order they contain each other's y- AND z-neighbours (in lattice order the z-neighbour
   // is a whole layer away, so its traces, extrema and the shared face-table block come from HBM a second time).
   out.ne_halo = 0;
   if (out.ne_ghost > 0 || cfg.tile_rows > 0)
   {
      std::vector<int> new_of_old(ne), old_of_new(ne);
      int nh = 0;
      std::vector<char> halo(ne, 0);
      if (out.ne_ghost > 0)
      {
         for (int e = 0; e < ne; e++)
         {
            for (int s = 0; s < 27; s++) { halo[e] = halo[e] || out.stencil27[(size_t)e * 27 + s] >= ne; }
            nh += halo[e];
         }
      }
      const int T = cfg.tile_rows > 0 ? cfg.tile_rows : 0;
      auto key = [&](int e)
      {
         // (the order so far is the local lattice order: e = lx + nlx (ly + nly lz))
         const long long lx = e % nlx, ly = (e / nlx) % nly, lz = e / (nlx * nly);
         const long long pos = T ? lx + nlx * ((ly % T) + (long long)T * (lz + (long long)nlz * (ly / T))) : (long long)e;
         return std::make_pair(halo[e] ? 0 : 1, pos);
      };
      for (int e = 0; e < ne; e++) { old_of_new[e] = e; }
      std::sort(old_of_new.begin(), old_of_new.end(), [&](int a, int b) { return key(a) < key(b); });
      for (int k = 0; k < ne; k++) { new_of_old[old_of_new[k]] = k; }
      out.ne_halo = nh;
      auto permute_rows = [&](auto &v, size_t w)
      {
         if (v.empty()) { return; }
         auto tmp = v;
         for (int k = 0; k < ne; k++)
         {
            std::copy(tmp.begin() + (size_t)old_of_new[k] * w, tmp.begin() + ((size_t)old_of_new[k] + 1) * w,
                      v.begin() + (size_t)k * w);
         }
      };
      permute_rows(out.x0, 81);
      permute_rows(out.vel, 81);
      permute_rows(out.u0, (size_t)out.ndof);
      permute_rows(out.s0, (size_t)out.ndof);
      permute_rows(out.subcell_vel, (size_t)3 * out.ndof);
      permute_rows(out.owned_gid, 1);
      permute_rows(out.stencil27, 27);
      permute_rows(out.face_nbr, 6);
      for (int &idx : out.stencil27) { if (idx >= 0 && idx < ne) { idx = new_of_old[idx]; } }
      for (int &idx : out.face_nbr) { if (idx >= 0 && idx < ne) { idx = new_of_old[idx]; } }
      for (Peer &pr : out.peers) { for (int &el : pr.send_elems) { el = new_of_old[el]; } }
   }
   return "";
}

// ---- error norms (remhos.cpp:1438-1470) -------------------------------------------------------------------------------
namespace
{
// n-point Gauss-Legendre rule on [0, 1] (Newton iteration on P_n)
void gauss_legendre_01(int n, std::vector<double> &x, std::vector<double> &w)
{
   x.resize(n);
   w.resize(n);
   for (int i = 0; i < n; i++)
   {
      double z = std::cos(M_PI * (i + 0.75) / (n + 0.5)), dp = 1.0;
      for (int it = 0; it < 100; it++)
      {
         double p0 = 1.0, p1 = z;
         for (int k = 2; k <= n; k++) { const double p2 = ((2 * k - 1) * z * p1 - (k - 1) * p0) / k; p0 = p1; p1 = p2; }
         dp = n * (z * p1 - p0) / (z * z - 1.0);
         const double dz = p1 / dp;
         z -= dz;
         if (std::fabs(dz) < 1e-16) { break; }
      }
      x[n - 1 - i] = 0.5 * (z + 1.0);
      w[n - 1 - i] = 1.0 / ((1.0 - z * z) * dp * dp);
   }
}
} // namespace

std::string lp_error_sums(const CaseData &d, int problem, double t_exact, const double *u, double err[3])
{
   err[0] = err[1] = err[2] = 0.0;
   const int prob = problem % 10;
   if (prob != 0 && prob != 4) { return "no exact solution is defined for this problem (4: rotation, 0: translation)"; }
   if (prob == 0 && !d.periodic) { return "the translated exact field is defined on the periodic meshes"; }
   if (d.exec_mode != 0) { return "error norms are for the transport problems"; }
   const int p = d.order, D = p + 1, nq = p + 2; // order 2 p + 3 -> p + 2 points per direction
   std::vector<double> xq, wq;
   gauss_legendre_01(nq, xq, wq);
   // 1-D tables: Bernstein values of u_h, Q2 Lagrange values and derivatives of the mesh map
   std::vector<double> B((size_t)nq * D), L((size_t)nq * 3), dL((size_t)nq * 3);
   for (int q = 0; q < nq; q++)
   {
      const double t = xq[q];
      double binom = 1.0;
      for (int k = 0; k <= p; k++)
      {
         B[(size_t)q * D + k] = binom * std::pow(t, k) * std::pow(1.0 - t, p - k);
         binom = binom * (p - k) / (k + 1);
      }
      lag2(t, &L[3 * q]);
      dL[3 * q + 0] = 4.0 * t - 3.0;
      dL[3 * q + 1] = 4.0 - 8.0 * t;
      dL[3 * q + 2] = 4.0 * t - 1.0;
   }
   double vel[3] = {0, 0, 0};
   if (prob == 0)
   {
      const double x0[3] = {0, 0, 0};
      if (d.dim == 2) { velocity_function_2d(problem, d.bb_min, d.bb_max, x0, vel); }
      else { velocity_function(problem, d.bb_min, d.bb_max, x0, vel); } // (a constant)
   }
   if (d.dim == 2)
   {
      // the same sums on quadrilaterals: nodes [ne][2][9], (p + 1)^2 Bernstein coefficients per element
      long double a1 = 0.0L, a2 = 0.0L;
      for (int e = 0; e < d.ne_owned; e++)
      {
         const double *X = &d.x0[(size_t)e * 18];
         const double *ue = u + (size_t)e * d.ndof;
         for (int qy = 0; qy < nq; qy++)
         {
            for (int qx = 0; qx < nq; qx++)
            {
               double x[2] = {0, 0}, J[2][2] = {{0, 0}, {0, 0}};
               for (int ay = 0; ay < 3; ay++)
               {
                  for (int ax = 0; ax < 3; ax++)
                  {
                     const double l = L[3 * qx + ax] * L[3 * qy + ay];
                     const double gx = dL[3 * qx + ax] * L[3 * qy + ay], gy = L[3 * qx + ax] * dL[3 * qy + ay];
                     for (int c = 0; c < 2; c++)
                     {
                        const double xc = X[c * 9 + ax + 3 * ay];
                        x[c] += l * xc;
                        J[c][0] += gx * xc;
                        J[c][1] += gy * xc;
                     }
                  }
               }
               const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
               double uh = 0.0;
               for (int iy = 0; iy < D; iy++)
               {
                  double row = 0.0;
                  for (int ix = 0; ix < D; ix++) { row += B[(size_t)qx * D + ix] * ue[ix + D * iy]; }
                  uh += row * B[(size_t)qy * D + iy];
               }
               double xe[2] = {x[0], x[1]};
               if (prob == 0)
               {
                  for (int c = 0; c < 2; c++)
                  {
                     const double len = d.bb_max[c] - d.bb_min[c];
                     double r = std::fmod(x[c] - vel[c] * t_exact - d.bb_min[c], len);
                     if (r < 0.0) { r += len; }
                     xe[c] = d.bb_min[c] + r;
                  }
               }
               const double ex = std::fabs(uh - u0_function_2d(problem, d.bb_min, d.bb_max, xe));
               const double w = wq[qx] * wq[qy] * std::fabs(det);
               a1 += w * ex;
               a2 += w * ex * ex;
               err[2] = std::fmax(err[2], ex);
            }
         }
      }
      err[0] = (double)a1;
      err[1] = (double)a2;
      return "";
   }
   std::vector<double> partial((size_t)d.ne_owned * 3, 0.0);
   parallel_for(d.ne_owned, [&](long long e_begin, long long e_end)
   {
   for (long long e = e_begin; e < e_end; e++)
   {
      const double *X = &d.x0[(size_t)e * 81];
      const double *ue = u + (size_t)e * d.ndof;
      double s1 = 0.0, s2 = 0.0, sm = 0.0;
      for (int qz = 0; qz < nq; qz++)
      {
         for (int qy = 0; qy < nq; qy++)
         {
            for (int qx = 0; qx < nq; qx++)
            {
               double x[3] = {0, 0, 0}, J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
               for (int az = 0; az < 3; az++)
               {
                  for (int ay = 0; ay < 3; ay++)
                  {
                     for (int ax = 0; ax < 3; ax++)
                     {
                        const int a3 = ax + 3 * (ay + 3 * az);
                        const double l = L[3 * qx + ax] * L[3 * qy + ay] * L[3 * qz + az];
                        const double gx = dL[3 * qx + ax] * L[3 * qy + ay] * L[3 * qz + az];
                        const double gy = L[3 * qx + ax] * dL[3 * qy + ay] * L[3 * qz + az];
                        const double gz = L[3 * qx + ax] * L[3 * qy + ay] * dL[3 * qz + az];
                        for (int c = 0; c < 3; c++)
                        {
                           const double xc = X[c * 27 + a3];
                           x[c] += l * xc;
                           J[c][0] += gx * xc; J[c][1] += gy * xc; J[c][2] += gz * xc;
                        }
                     }
                  }
               }
               const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                                  J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
               double uh = 0.0;
               for (int iz = 0; iz < D; iz++)
               {
                  for (int iy = 0; iy < D; iy++)
                  {
                     double row = 0.0;
                     for (int ix = 0; ix < D; ix++) { row += B[(size_t)qx * D + ix] * ue[ix + D * (iy + D * iz)]; }
                     uh += row * B[(size_t)qy * D + iy] * B[(size_t)qz * D + iz];
                  }
               }
               double xe[3] = {x[0], x[1], x[2]};
               if (prob == 0)
               {
                  // u0(x - v t), wrapped periodically into the bounding box
                  for (int c = 0; c < 3; c++)
                  {
                     const double len = d.bb_max[c] - d.bb_min[c];
                     double r = std::fmod(x[c] - vel[c] * t_exact - d.bb_min[c], len);
                     if (r < 0.0) { r += len; }
                     xe[c] = d.bb_min[c] + r;
                  }
               }
               const double ex = std::fabs(uh - u0_function(problem, d.bb_min, d.bb_max, xe));
               const double w = wq[qx] * wq[qy] * wq[qz] * std::fabs(det);
               s1 += w * ex;
               s2 += w * ex * ex;
               sm = std::fmax(sm, ex);
            }
         }
      }
      partial[(size_t)e * 3 + 0] = s1; partial[(size_t)e * 3 + 1] = s2; partial[(size_t)e * 3 + 2] = sm;
   }
   });
   long double a1 = 0.0L, a2 = 0.0L;
   for (int e = 0; e < d.ne_owned; e++)
   {
      a1 += partial[(size_t)e * 3];
      a2 += partial[(size_t)e * 3 + 1];
      err[2] = std::fmax(err[2], partial[(size_t)e * 3 + 2]);
   }
   err[0] = (double)a1;
   err[1] = (double)a2;
   return "";
}

// -save (remhos.cpp:1015-1030, 1365-1380): see include/rmh_driver.h (rmhd_case_save)
std::string save_mfem(const CaseData &d, double t, const double *u, const char *mesh_path, const char *gf_path)
{
   if (!mesh_path) { return "null mesh path"; }
   if (d.dim != 3) { return "-save writes the 3-D meshes"; }
   if ((long long)d.ne_owned != d.ne_global)
   {
      return "rmhd_case_save writes the mesh of a single-rank case (PrintAsOne / SaveAsOne)";
   }
   const int N[3] = {d.n[0], d.n[1], d.n[2]};
   const bool per = d.periodic;
   const int nv1[3] = {per ? N[0] : N[0] + 1, per ? N[1] : N[1] + 1, per ? N[2] : N[2] + 1};
   auto vid = [&](int ix, int iy, int iz)
   {
      if (per) { ix %= N[0]; iy %= N[1]; iz %= N[2]; }
      return ix + nv1[0] * (iy + (long long)nv1[1] * iz);
   };
   FILE *f = std::fopen(mesh_path, "w");
   if (!f) { return std::string("cannot open ") + mesh_path; }
   std::fprintf(f, "MFEM mesh v1.0\n\ndimension\n3\n\nelements\n%d\n", d.ne_owned);
   for (int e = 0; e < d.ne_owned; e++)
   {
      const long long g = d.owned_gid[e];
      const int ex = (int)(g % N[0]), ey = (int)((g / N[0]) % N[1]), ez = (int)(g / ((long long)N[0] * N[1]));
      // geometry 5 = CUBE; vertex order of mfem::Hexahedron
      std::fprintf(f, "1 5 %lld %lld %lld %lld %lld %lld %lld %lld\n", vid(ex, ey, ez), vid(ex + 1, ey, ez),
                   vid(ex + 1, ey + 1, ez), vid(ex, ey + 1, ez), vid(ex, ey, ez + 1), vid(ex + 1, ey, ez + 1),
                   vid(ex + 1, ey + 1, ez + 1), vid(ex, ey + 1, ez + 1));
   }
   if (per) { std::fprintf(f, "\nboundary\n0\n"); }
   else
   {
      std::fprintf(f, "\nboundary\n%d\n", 2 * (N[0] * N[1] + N[1] * N[2] + N[0] * N[2]));
      // geometry 3 = SQUARE, attributes 1..6 by side (-x, +x, -y, +y, -z, +z), outward orientation
      for (int j = 0; j < N[2]; j++) { for (int i = 0; i < N[1]; i++) {
         std::fprintf(f, "1 3 %lld %lld %lld %lld\n", vid(0, i, j), vid(0, i, j + 1), vid(0, i + 1, j + 1), vid(0, i + 1, j));
         std::fprintf(f, "2 3 %lld %lld %lld %lld\n", vid(N[0], i, j), vid(N[0], i + 1, j), vid(N[0], i + 1, j + 1), vid(N[0], i, j + 1)); } }
      for (int j = 0; j < N[2]; j++) { for (int i = 0; i < N[0]; i++) {
         std::fprintf(f, "3 3 %lld %lld %lld %lld\n", vid(i, 0, j), vid(i + 1, 0, j), vid(i + 1, 0, j + 1), vid(i, 0, j + 1));
         std::fprintf(f, "4 3 %lld %lld %lld %lld\n", vid(i, N[1], j), vid(i, N[1], j + 1), vid(i + 1, N[1], j + 1), vid(i + 1, N[1], j)); } }
      for (int j = 0; j < N[1]; j++) { for (int i = 0; i < N[0]; i++) {
         std::fprintf(f, "5 3 %lld %lld %lld %lld\n", vid(i, j, 0), vid(i, j + 1, 0), vid(i + 1, j + 1, 0), vid(i + 1, j, 0));
         std::fprintf(f, "6 3 %lld %lld %lld %lld\n", vid(i, j, N[2]), vid(i + 1, j, N[2]), vid(i + 1, j + 1, N[2]), vid(i, j + 1, N[2])); } }
   }
   std::fprintf(f, "\nvertices\n%lld\n\nnodes\nFiniteElementSpace\nFiniteElementCollection: L2_T1_3D_P2\nVDim: 3\nOrdering: 0\n\n",
                (long long)nv1[0] * nv1[1] * nv1[2]);
   // Ordering 0 = byNODES: all x components (element by element, 27 nodes, x fastest), then y, then z.
   // Elements are written in lattice order: position k of the file is the element with global id k.
   std::vector<int> local_of_gid(d.ne_owned);
   for (int e = 0; e < d.ne_owned; e++) { local_of_gid[d.owned_gid[e]] = e; }
   for (int comp = 0; comp < 3; comp++)
   {
      for (int g = 0; g < d.ne_owned; g++)
      {
         const int e = local_of_gid[g];
         for (int a = 0; a < 27; a++)
         {
            const size_t k = (size_t)e * 81 + comp * 27 + a;
            std::fprintf(f, "%.14g\n", d.x0[k] + (d.exec_mode == 1 ? t * d.vel[k] : 0.0));
         }
      }
   }
   std::fclose(f);
   if (u && gf_path)
   {
      f = std::fopen(gf_path, "w");
      if (!f) { return std::string("cannot open ") + gf_path; }
      // DG_FECollection(order, dim, BasisType::Positive) (remhos.cpp:588-590): basis type 2
      std::fprintf(f, "FiniteElementSpace\nFiniteElementCollection: L2_T2_3D_P%d\nVDim: 1\nOrdering: 0\n\n", d.order);
      for (int g = 0; g < d.ne_owned; g++)
      {
         const int e = local_of_gid[g];
         for (int i = 0; i < d.ndof; i++) { std::fprintf(f, "%.14g\n", u[(size_t)e * d.ndof + i]); }
      }
      std::fclose(f);
   }
   return "";
}


} // namespace remhos
