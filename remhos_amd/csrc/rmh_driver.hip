// Host classes of include/remhos_amd/solvers.hpp and the single-GPU time loop rmhd_run().
#include "../../include/remhos_amd/solvers.hpp"
#include "../../include/rmh_driver.h"
#include "rmh_host.hpp"

#include <hip/hip_runtime.h>

#include <chrono>
#include <ctime>
#include <sys/stat.h>
#include <unistd.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace remhos
{

extern thread_local std::string g_driver_error;
CaseConfig to_config(const rmhd_config &c);

void rmh_abort(const char *msg, const char *file, int line)
{
   std::fprintf(stderr, "\nremhos_amd abort: %s\n ... in %s:%d\n ... last rmh error: %s\n", msg, file, line,
                rmh_last_error());
   std::abort();
}

#define RMH_CALL(expr) RMH_VERIFY((expr) == 0, #expr)
#define HIP_CALL(expr) RMH_VERIFY((expr) == hipSuccess, #expr)

// ---- Vector ------------------------------------------------------------------------------
Vector::Vector(int n) { SetSize(n); }
Vector::Vector(const Vector &o)
{
   SetSize(o.size);
   if (size) { HIP_CALL(hipMemcpyAsync(data, o.data, sizeof(double) * size, hipMemcpyDeviceToDevice, nullptr)); }
}
Vector &Vector::operator=(const Vector &o)
{
   if (this == &o) { return *this; }
   if (size != o.size) { SetSize(o.size); }
   if (size) { HIP_CALL(hipMemcpyAsync(data, o.data, sizeof(double) * size, hipMemcpyDeviceToDevice, nullptr)); }
   return *this;
}
Vector::~Vector()
{
   if (own && data) { (void)hipFree(data); }
}
void Vector::SetSize(int n)
{
   if (own && data) { (void)hipFree(data); }
   data = nullptr;
   size = n;
   own = true;
   if (n > 0) { HIP_CALL(hipMalloc((void **)&data, sizeof(double) * n)); }
}
void Vector::CopyFromHost(const double *h)
{
   HIP_CALL(hipMemcpy(data, h, sizeof(double) * size, hipMemcpyHostToDevice));
}
void Vector::CopyToHost(double *h) const
{
   HIP_CALL(hipMemcpy(h, data, sizeof(double) * size, hipMemcpyDeviceToHost));
}

__global__ void fill_kernel(double *z, double v, int n)
{
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { z[i] = v; }
}
__global__ void axpby_kernel(double a, const double *x, double b, const double *y, double *z, int n)
{
   for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
   {
      z[i] = a * x[i] + b * y[i];
   }
}
static int grid_for(int n) { return std::max(1, std::min(2048, (n + 255) / 256)); }

Vector &Vector::operator=(double value)
{
   if (size) { hipLaunchKernelGGL(fill_kernel, dim3(grid_for(size)), dim3(256), 0, nullptr, data, value, size); }
   return *this;
}
void add(const Vector &x, double a, const Vector &y, Vector &z)
{
   const int n = x.Size();
   hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, nullptr, 1.0, x.Read(), a, y.Read(), z.Write(),
                      n);
}
void add(double a, const Vector &x, double b, const Vector &y, Vector &z)
{
   const int n = x.Size();
   hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, nullptr, a, x.Read(), b, y.Read(), z.Write(), n);
}

void TimingData::Update(rmh_ctx *ctx)
{
   double t[4];
   RMH_CALL(rmh_timers(ctx, t));
   sw_rhs = t[0];
   sw_L2inv = t[1];
   sw_LO = t[2];
   sw_FCT = t[3];
}

// ---- solvers: forward to the C ABI ------------------------------------------------------------
void SpaceLayout::ExchangeFaceNbrData(const double *u) const
{
   if (!exchange) { return; }
   RMH_CALL(rmh_exchange_begin(ctx, u));
   RMH_CALL(rmh_exchange_end(ctx));
}
void SpaceLayout::ExchangeElementExtrema(const double *el_min, const double *el_max) const
{
   if (!exchange) { return; }
   RMH_CALL(rmh_exchange_minmax_begin(ctx, el_min, el_max));
   RMH_CALL(rmh_exchange_minmax_end(ctx));
}

LocalInverseHOSolver::LocalInverseHOSolver(ParFiniteElementSpace &space, bool partial_assembly) : HOSolver(space)
{
   // remhos_ho.cpp:77-81 (M_inv->SetAbsTol(1e-8), M_inv->SetRelTol(0.0)) / :104-115 (exact element inverse)
   if (partial_assembly)
   {
      RMH_CALL(rmh_set_mass_tol(pfes.Ctx(), 0.0, 1e-8, 100));
      RMH_CALL(rmh_set_mass_completion(pfes.Ctx(), 1, 1));
   }
   else
   {
      RMH_CALL(rmh_set_mass_tol(pfes.Ctx(), 1e-14, 0.0, 100));
      RMH_CALL(rmh_set_mass_completion(pfes.Ctx(), 0, 0));
   }
}

void LocalInverseHOSolver::CalcHOSolution(const Vector &u, Vector &du) const
{
   RMH_VERIFY(timer, "Timer not set."); // remhos_ho.cpp:86
   pfes.ExchangeFaceNbrData(u.Read());  // remhos_ho.cpp:122 (inside K.Mult for the PA branch)
   RMH_CALL(rmh_ho_apply(pfes.Ctx(), u.Read(), du.Write()));
}

void CGHOSolver::CalcHOSolution(const Vector &u, Vector &du) const
{
   RMH_VERIFY(timer, "Timer not set.");
   double rel, abs;
   int maxit;
   RMH_CALL(rmh_get_mass_tol(pfes.Ctx(), &rel, &abs, &maxit));
   RMH_CALL(rmh_set_mass_tol(pfes.Ctx(), 1e-12, 0.0, 500)); // remhos_ho.cpp:60-63
   pfes.ExchangeFaceNbrData(u.Read());
   RMH_CALL(rmh_ho_apply(pfes.Ctx(), u.Read(), du.Write()));
   RMH_CALL(rmh_set_mass_tol(pfes.Ctx(), rel, abs, maxit));
}

void MassBasedAvg::CalcLOSolution(const Vector &u, Vector &du) const
{
   // remhos_lo.cpp:247-324
   if (du_HO)
   {
      RMH_CALL(rmh_lo_massavg(pfes.Ctx(), u.Read(), du_HO->Read(), dt, du.Write()));
      du_HO = nullptr;
   }
   else
   {
      Vector du_HO_tmp(u.Size());
      ho_solver.CalcHOSolution(u, du_HO_tmp);
      RMH_CALL(rmh_lo_massavg(pfes.Ctx(), u.Read(), du_HO_tmp.Read(), dt, du.Write()));
   }
}

void PAResidualDistribution::CalcLOSolution(const Vector &u, Vector &du) const
{
   RMH_CALL(rmh_lo_rd(pfes.Ctx(), u.Read(), du.Write()));
}

void PAResidualDistributionSubcell::CalcLOSolution(const Vector &u, Vector &du) const
{
   RMH_CALL(rmh_lo_rdsubcell(pfes.Ctx(), u.Read(), du.Write()));
}

void ClipScaleSolver::CalcFCTSolution(const ParGridFunction &u, const Vector &m, const Vector &du_ho,
                                      const Vector &du_lo, const Vector &u_min, const Vector &u_max,
                                      Vector &du) const
{
   RMH_CALL(rmh_fct_clipscale(pfes.Ctx(), u.Read(), m.Read(), du_ho.Read(), du_lo.Read(), u_min.Read(), u_max.Read(),
                              dt, du.Write()));
}

void ClipScaleSolver::CalcFCTProduct(const ParGridFunction &us, const Vector &m, const Vector &d_us_HO, const Vector &d_us_LO,
                                     Vector &s_min, Vector &s_max, const Vector &u_new, const Array<bool> &active_el,
                                     const Array<bool> &active_dofs, Vector &d_us)
{
   (void)d_us_LO; // (only solvers with NeedsLOProductInput() read it, remhos.cpp:1865-1869)
   static_assert(sizeof(bool) == 1, "Array<bool> is handed to the kernels as a byte array");
   RMH_CALL(rmh_fct_product(pfes.Ctx(), us.Read(), m.Read(), d_us_HO.Read(), s_min.ReadWrite(), s_max.ReadWrite(), u_new.Read(),
                            (const unsigned char *)active_el.Read(), (const unsigned char *)active_dofs.Read(), dt, d_us.Write()));
   if (verify_bounds)
   {
      // Check the bounds of the final solution (remhos_fct.cpp:568-610): us + dt d_us against the scaled bounds
      // (s_min u_new, s_max u_new) of ScaleProductBounds on the active dofs, eps = 1e-12.  (The kernel keeps the scaled
      // bounds in registers; s_min / s_max come back updated like the reference's, so the product below IS us_min / us_max.)
      rmh_violation v;
      RMH_CALL(rmh_check_violation(pfes.Ctx(), us.Read(), dt, d_us.Read(), s_min.Read(), s_max.Read(), u_new.Read(), 1e-12,
                                   (const unsigned char *)active_dofs.Read(), &v));
      if (v.count > 0)
      {
         const int nd = pfes.GetNDofs();
         std::printf("Final us %lld %lld %.17g %.17g %.17g\n---\n", v.first % nd, v.first / nd, v.first_min, v.first_value, v.first_max);
         RMH_VERIFY(false, "Bounds violation FCT us.");
      }
   }
}

// remhos.cpp:1557-1594
static void report_violation(const rmh_violation &v, const std::string &info)
{
   if (v.count == 0) { return; }
   std::printf("%s bounds violation: %lld %.17g %.17g %.17g\n%.17g %.17g\n(%lld dofs out of bounds; largest overshoot %.3e, undershoot %.3e)\n",
               info.c_str(), v.first, v.first_min, v.first_value, v.first_max, v.first_max - v.first_value,
               v.first_value - v.first_min, v.count, v.over, v.under);
   std::fflush(stdout);
   RMH_VERIFY(false, "Aborted due to bounds violation.");
}
void check_violation(ParFiniteElementSpace &pfes, const Vector &u_new, const Vector &u_min, const Vector &u_max, std::string info,
                     double tol, const Array<bool> *active_dofs)
{
   rmh_violation v;
   RMH_CALL(rmh_check_violation(pfes.Ctx(), u_new.Read(), 0.0, nullptr, u_min.Read(), u_max.Read(), nullptr, tol,
                                active_dofs ? (const unsigned char *)active_dofs->Read() : nullptr, &v));
   report_violation(v, info);
}
void check_violation(ParFiniteElementSpace &pfes, const Vector &u, double dt, const Vector &du_new, const Vector &u_min,
                     const Vector &u_max, std::string info, double tol, const Array<bool> *active_dofs)
{
   rmh_violation v;
   RMH_CALL(rmh_check_violation(pfes.Ctx(), u.Read(), dt, du_new.Read(), u_min.Read(), u_max.Read(), nullptr, tol,
                                active_dofs ? (const unsigned char *)active_dofs->Read() : nullptr, &v));
   report_violation(v, info);
}

// remhos_sync.cpp:23-96
void ComputeBoolIndicators(ParFiniteElementSpace &pfes, const Vector &u, Array<bool> &ind_elem, Array<bool> &ind_dofs)
{
   RMH_CALL(rmh_product_ratio(pfes.Ctx(), nullptr, u.Read(), nullptr, (unsigned char *)ind_elem.Write(),
                              (unsigned char *)ind_dofs.Write()));
}
void ComputeRatio(ParFiniteElementSpace &pfes, const Vector &us, const Vector &u, Vector &s, Array<bool> &bool_el,
                  Array<bool> &bool_dof)
{
   RMH_CALL(rmh_product_ratio(pfes.Ctx(), us.Read(), u.Read(), s.Write(), (unsigned char *)bool_el.Write(),
                              (unsigned char *)bool_dof.Write()));
}

DofInfo::DofInfo(ParFiniteElementSpace &space)
   : pfes(space), xe_min(space.GetNE()), xe_max(space.GetNE()), xi_min(space.GetVSize()), xi_max(space.GetVSize())
{
}
void DofInfo::ComputeElementsMinMax(const Vector &u, Vector &u_min, Vector &u_max, Array<bool> *active_el,
                                    Array<bool> *active_dof) const
{
   if (active_el && active_dof)
   {
      RMH_CALL(rmh_elem_minmax_masked(pfes.Ctx(), u.Read(), (const unsigned char *)active_el->Read(),
                                      (const unsigned char *)active_dof->Read(), u_min.Write(), u_max.Write()));
      return;
   }
   RMH_CALL(rmh_elem_minmax(pfes.Ctx(), u.Read(), u_min.Write(), u_max.Write()));
}
void DofInfo::ComputeBounds(const Vector &el_min, const Vector &el_max, Vector &dof_min, Vector &dof_max) const
{
   pfes.ExchangeElementExtrema(el_min.Read(), el_max.Read()); // remhos_tools.cpp:461-466
   RMH_CALL(rmh_bounds(pfes.Ctx(), el_min.Read(), el_max.Read(), dof_min.Write(), dof_max.Write()));
}

// ---- AdvectionOperator (remhos.cpp:1596-1739, 1798-1916) -----------------------------------------
AdvectionOperator::AdvectionOperator(ParFiniteElementSpace &space, DofInfo &dofs_, HOSolver *hos, LOSolver *los,
                                     FCTSolver *fct, bool fused_limiter, bool product_sync)
   : LimitedTimeDependentOperator((product_sync ? 2 : 1) * space.GetVSize()), pfes(space), dofs(dofs_), ho_solver(hos),
     lo_solver(los), fct_solver(fct),
     lumpedM(rmh_lumped_mass(space.Ctx()) ? const_cast<double *>(rmh_lumped_mass(space.Ctx())) : nullptr, space.GetVSize()),
     du_HO(space.GetVSize()), du_LO(fused_limiter ? 0 : space.GetVSize()), d_us_HO(product_sync ? space.GetVSize() : 0),
     s_ratio(product_sync ? space.GetVSize() : 0), u_new(product_sync ? space.GetVSize() : 0), fused(fused_limiter),
     product(product_sync)
{
   if (ho_solver) { ho_solver->timer = &timer; }
   if (lo_solver) { lo_solver->timer = &timer; }
   if (fct_solver) { fct_solver->timer = &timer; }
   if (product)
   {
      // flag arrays of product remap (mfem::Array<bool> on the device)
      static_assert(sizeof(bool) == 1, "Array<bool> is handed to the kernels as a byte array");
      const int ne = space.GetNE(), n = space.GetVSize();
      bool *p = nullptr;
      HIP_CALL(hipMalloc((void **)&p, (size_t)2 * (ne + n)));
      s_bool_el.MakeRef(p, ne);
      s_bool_dofs.MakeRef(p + ne, n);
      s_bool_el_new.MakeRef(p + ne + n, ne);
      s_bool_dofs_new.MakeRef(p + 2 * ne + n, n);
   }
}

AdvectionOperator::~AdvectionOperator()
{
   if (product) { (void)hipFree(s_bool_el.Write()); }
}

void AdvectionOperator::MultUnlimited(const Vector &X, Vector &Y) const
{
   // remap: move the mesh to the stage time and re-set-up M_HO, K_HO, lumpedM
   // (remhos.cpp:1598-1637) -- matrix-free here: the kernels evaluate x0 + t*v themselves
   RMH_CALL(rmh_setup(pfes.Ctx(), GetTime()));
   RMH_VERIFY(ho_solver && lo_solver && fct_solver, "FCT requires HO and LO solvers."); // remhos.cpp:1690
   const int n = pfes.GetVSize();
   if (product)
   {
      // Remap the product field (remhos.cpp:1709-1738).  Its HO rate is formed FIRST: the context keeps the element
      // extrema of the last vector rmh_ho_apply saw, and the fused limiter of u relies on them.
      const Vector us(const_cast<double *>(X.Read()) + n, n);
      Vector d_us(Y.Write() + n, n);
      ho_solver->CalcHOSolution(us, d_us);
   }
   const Vector u(const_cast<double *>(X.Read()), n);
   Vector d_u(Y.Write(), n);
   ho_solver->CalcHOSolution(u, d_u);
   // Limiting is deferred to LimitMult()
}

void AdvectionOperator::LimitMult(const Vector &X, Vector &Y) const
{
   const int n = pfes.GetVSize();
   const Vector u(const_cast<double *>(X.Read()), n);
   Vector d_u(Y.Write(), n);
   if (fused)
   {
      if (verify_bounds) { du_HO = d_u; } // (the fused kernel overwrites the HO rate; the check re-forms the LO rate from it)
      // d_u holds du_HO on entry; the fused kernel reads du_HO and writes d_u element by element
      RMH_CALL(rmh_limit_fused(pfes.Ctx(), u.Read(), d_u.Read(), dt, d_u.Write(), nullptr, 0.0, 1.0, 0.0, nullptr));
      if (verify_bounds)
      {
         // remhos.cpp:1824-1837 for the fused limiter: it keeps du_LO and the dof bounds in registers, so the granular
         // kernels form them once more for the check
         if (du_LO.Size() != n) { du_LO.SetSize(n); }
         auto mba = dynamic_cast<MassBasedAvg *>(lo_solver);
         if (mba) { mba->SetHOSolution(du_HO); }
         lo_solver->CalcLOSolution(u, du_LO);
         dofs.ComputeElementsMinMax(u, dofs.xe_min, dofs.xe_max);
         dofs.ComputeBounds(dofs.xe_min, dofs.xe_max, dofs.xi_min, dofs.xi_max);
         check_violation(pfes, u, dt, du_LO, dofs.xi_min, dofs.xi_max, "LimitMult LO u", 1e-12, nullptr);
         check_violation(pfes, u, dt, d_u, dofs.xi_min, dofs.xi_max, "LimitMult FCT solution u", 1e-12, nullptr);
      }
   }
   else
   {
      // the reference's sequence, remhos.cpp:1812-1831 (the x_gf face-neighbour exchange at
      // :1812-1813 is dead weight for ClipScale and is skipped)
      du_HO = d_u; // Vector du_HO(d_u)
      auto mba = dynamic_cast<MassBasedAvg *>(lo_solver);
      if (mba) { mba->SetHOSolution(du_HO); }
      lo_solver->CalcLOSolution(u, du_LO);
      dofs.ComputeElementsMinMax(u, dofs.xe_min, dofs.xe_max);
      dofs.ComputeBounds(dofs.xe_min, dofs.xe_max, dofs.xi_min, dofs.xi_max);
      if (verify_bounds) { check_violation(pfes, u, dt, du_LO, dofs.xi_min, dofs.xi_max, "LimitMult LO u", 1e-12, nullptr); } // :1824-1828
      fct_solver->CalcFCTSolution(u, lumpedM, du_HO, du_LO, dofs.xi_min, dofs.xi_max, d_u);
      if (verify_bounds) { check_violation(pfes, u, dt, d_u, dofs.xi_min, dofs.xi_max, "LimitMult FCT solution u", 1e-12, nullptr); } // :1833-1837
      UpdateTimeStepEstimate(u, du_LO, dofs.xi_min, dofs.xi_max); // remhos.cpp:1839-1842 (no-op with a fixed dt)
   }
   if (!product) { return; }
   // Remap the product field (remhos.cpp:1848-1915)
   const Vector us(const_cast<double *>(X.Read()) + n, n);
   Vector d_us(Y.Write() + n, n);
   d_us_HO = d_us; // Vector d_us_HO(d_us)
   Vector d_us_LO;
   RMH_VERIFY(!fct_solver->NeedsLOProductInput(), "no FCT solver on this path needs the LO product rate"); // :1866-1870
   // Compute the ratio s = us_old / u_old, and old active dofs.
   ComputeRatio(pfes, us, u, s_ratio, s_bool_el, s_bool_dofs);
   // Bounds for s, based on the old values (and old active dofs).
   dofs.ComputeElementsMinMax(s_ratio, dofs.xe_min, dofs.xe_max, &s_bool_el, &s_bool_dofs);
   dofs.ComputeBounds(dofs.xe_min, dofs.xe_max, dofs.xi_min, dofs.xi_max); // (+inf, -inf) of inactive elements = "&s_bool_el"
   // Evolve u and get the new active dofs.
   add(u, dt, d_u, u_new);
   ComputeBoolIndicators(pfes, u_new, s_bool_el_new, s_bool_dofs_new);
   fct_solver->CalcFCTProduct(us, lumpedM, d_us_HO, d_us_LO, dofs.xi_min, dofs.xi_max, u_new, s_bool_el_new, s_bool_dofs_new,
                              d_us);
}

void AdvectionOperator::UpdateTimeStepEstimate(const Vector &x, const Vector &dx, const Vector &x_min,
                                               const Vector &x_max) const
{
   RMH_CALL(rmh_dt_estimate_update(pfes.Ctx(), x.Read(), dx.Read(), x_min.Read(), x_max.Read()));
}
void AdvectionOperator::ResetTimeStepRatio() const { RMH_CALL(rmh_dt_estimate_reset(pfes.Ctx())); }
real_t AdvectionOperator::GetTimeStepRatio() const
{
   double est = 0.0;
   RMH_CALL(rmh_dt_estimate_get(pfes.Ctx(), &est));
   return (dt != 0.) ? est / dt : 0.; // remhos.cpp:1997
}

// ---- IDP solvers (remhos_solvers.cpp) ---------------------------------------------------------------------------------
void ForwardEulerIDPSolver::Init(LimitedTimeDependentOperator &f_)
{
   IDPODESolver::Init(f_);
   dx.SetSize(f_.Height());
}
void ForwardEulerIDPSolver::Step(Vector &x, real_t &t, real_t &dt)
{
   // one limited leg over the whole step (the reference's -s 11, remhos_solvers.cpp:30-40)
   f->SetTime(t);
   f->SetDt(dt);
   f->MultUnlimited(x, dx);
   f->LimitMult(x, dx);
   add(x, dt, dx, x);
   t += dt;
}

// An explicit Runge-Kutta method  X_i = x_n + dt sum_j A_ij k_j  (last row: the weights b, abscissa 1) run as a chain of
// LIMITED forward Euler legs (the reference's RKIDPSolver, remhos_solvers.hpp:94-126; masks absent, remhos.cpp:502-507).
// The chain stands at a state that is itself a row of the tableau, "here"; leg i has to carry it to row i over the
// pseudo-time span = abscissa_i - at.  As an Euler rate that is
//      U_i = sum_j (A_ij - here_j) / span * k_j ,
// but the earlier k_j are gone: what the chain kept of stage j is its limited rate L_j = limit(U_j).  Reading U_j = L_j
// backwards gives every earlier k_j as a combination of L_0 .. L_j ("worth"), and U_i becomes
//      U_i = w_i k_i + sum_{k<i} w_k L_k .
// A leg whose successor's abscissa does not lie ahead of its own is limited and remembered, but the chain stays put.
std::vector<RKIDPSolver::EulerLeg> RKIDPSolver::PlanLegs(int stages, const real_t *a_packed, const real_t *b_row, const real_t *c_abs)
{
   auto row = [&](int i) { return (i < stages - 1) ? a_packed + i * (i + 1) / 2 : b_row; };
   auto abscissa = [&](int i) { return (i < stages - 1) ? c_abs[i] : real_t(1); };
   std::vector<std::vector<real_t>> worth(stages, std::vector<real_t>(stages, 0.)); // k_j = sum_k worth[j][k] L_k
   std::vector<real_t> here(stages, 0.);
   real_t at = 0.;
   std::vector<EulerLeg> legs(stages);
   for (int i = 0; i < stages; i++)
   {
      EulerLeg &leg = legs[i];
      leg.from = at;
      leg.span = abscissa(i) - at;
      RMH_VERIFY(leg.span > 0. && row(i)[i] != 0., "tableau cannot be run as forward Euler legs");
      leg.w.assign(i + 1, 0.);
      for (int j = 0; j < i; j++)
      {
         const real_t g = (row(i)[j] - here[j]) / leg.span;
         if (g == 0.) { continue; }
         for (int k = 0; k <= j; k++) { leg.w[k] += g * worth[j][k]; }
      }
      leg.w[i] = row(i)[i] / leg.span;
      for (int k = 0; k < i; k++) { worth[i][k] = -leg.w[k] / leg.w[i]; }
      worth[i][i] = 1. / leg.w[i];
      leg.lands = (i == stages - 1) || abscissa(i + 1) > abscissa(i);
      if (leg.lands)
      {
         at = abscissa(i);
         for (int j = 0; j < stages; j++) { here[j] = (j <= i) ? row(i)[j] : 0.; }
      }
   }
   return legs;
}

RKIDPSolver::RKIDPSolver(int stages, const real_t a_packed[], const real_t b_row[], const real_t c_abs[])
   : legs(PlanLegs(stages, a_packed, b_row, c_abs)), limited(stages)
{
}
void RKIDPSolver::Init(LimitedTimeDependentOperator &f_)
{
   IDPODESolver::Init(f_);
   for (Vector &v : limited) { v.SetSize(f->Height()); }
}
void RKIDPSolver::Step(Vector &x, real_t &t, real_t &dt)
{
   f->SetTime(t);
   for (size_t i = 0; i < legs.size(); i++)
   {
      const EulerLeg &leg = legs[i];
      Vector &rate = limited[i];
      f->SetDt(leg.span * dt);
      f->MultUnlimited(x, rate); // k_i at the state (and mesh position) the chain stands at
      if (i > 0 || leg.w[0] != 1.)
      {
         add(leg.w[i], rate, (i > 0) ? leg.w[0] : 0., limited[0], rate);
         for (size_t k = 1; k < i; k++) { add(rate, leg.w[k], limited[k], rate); }
      }
      f->LimitMult(x, rate); // rate: U_i in, L_i out
      if (leg.lands)
      {
         add(x, leg.span * dt, rate, x);
         f->SetTime(t + (leg.from + leg.span) * dt);
      }
   }
   t += dt;
}

// midpoint rule and the three-stage third-order method that the reference runs as -s 12 / -s 13 (remhos_solvers.cpp:252-260):
// lower-triangular rows packed one after the other, weights, abscissae
const real_t RK2IDPSolver::a[] = {1. / 2.};
const real_t RK2IDPSolver::b[] = {0., 1.};
const real_t RK2IDPSolver::c[] = {1. / 2.};
const real_t RK3IDPSolver::a[] = {1. / 3., /**/ 0., 2. / 3.};
const real_t RK3IDPSolver::b[] = {1. / 4., 0., 3. / 4.};
const real_t RK3IDPSolver::c[] = {1. / 3., 2. / 3.};

// ---- RK3 SSP [MFEM RK3SSPSolver::Step] -------------------------------------------------------------
void RK3SSPSolver::Init(LimitedTimeDependentOperator &op)
{
   f = &op;
   y.SetSize(op.Height());
   k.SetSize(op.Height());
}
void RK3SSPSolver::Step(Vector &x, real_t &t, real_t &dt)
{
   // x0 = x, t0 = t, k0 = dt*f(t0, x0)
   f->SetTime(t);
   f->Mult(x, k);
   // x1 = x + k0, t1 = t + dt, k1 = dt*f(t1, x1)
   add(x, dt, k, y);
   f->SetTime(t + dt);
   f->Mult(y, k);
   // x2 = 3/4*x + 1/4*(x1 + k1), t2 = t + 1/2*dt, k2 = dt*f(t2, x2)
   add(y, dt, k, y);
   add(3. / 4, x, 1. / 4, y, y);
   f->SetTime(t + dt / 2);
   f->Mult(y, k);
   // x3 = 1/3*x + 2/3*(x2 + k2), t3 = t + dt
   add(y, dt, k, y);
   add(1. / 3, x, 2. / 3, y, x);
   t += dt;
}

} // namespace remhos

using namespace remhos;

namespace
{
bool read_or_write_id(const char *path, bool writer, char id[128]); // (below, with rmhd_run_partitioned)
}

extern "C" int rmhd_run(const rmhd_config *cfg, rmhd_result *res) { return rmhd_run_rank(cfg, nullptr, 0, res, nullptr, nullptr); }

extern "C" int rmhd_run_state(const rmhd_config *cfg, rmhd_result *res, double *u_final, double *us_final)
{
   return rmhd_run_rank(cfg, nullptr, 0, res, u_final, us_final);
}

extern "C" int rmhd_run_rank(const rmhd_config *cfg, const char *comm_id_file, int device, rmhd_result *res, double *u_final,
                             double *us_final)
{
   if (!cfg || !res) { g_driver_error = "null argument"; return -1; }
   std::memset(res, 0, sizeof(*res));
   CaseConfig cc = to_config(*cfg);
   const int nranks = cc.px * cc.py * cc.pz;
   const bool rccl_ranks = comm_id_file && comm_id_file[0];
   if (nranks != 1 && !rccl_ranks)
   {
      g_driver_error = "rmhd_run drives one block; box-partitioned runs go through rmhd_run_partitioned, or through "
                       "rmhd_run_rank with one process per block";
      return -1;
   }
   CaseData cd;
   const std::string err = build_case(cc, cd);
   if (!err.empty()) { g_driver_error = err; return -1; }

   rmh_layout L;
   L.dim = cd.dim; // (2: the reference's quadrilateral lattices, one rank -- build_case_2d)
   L.order = cd.order;
   L.mesh_order = 2;
   L.exec_mode = cd.exec_mode;
   L.ne_owned = cd.ne_owned;
   L.ne_ghost = cd.ne_ghost; // (> 0 only for a self-wrapped block: rmhd_config.self_wrap)
   L.x0 = cd.x0.data();
   L.vel = cd.vel.data();
   L.face_nbr = cd.face_nbr.data();
   L.stencil27 = cd.stencil27.data();
   L.subcell_vel = cd.subcell_vel.empty() ? nullptr : cd.subcell_vel.data();
   L.device = device;
   rmh_ctx *ctx = nullptr;
   if (rmh_create(&L, &ctx) != 0) { g_driver_error = rmh_last_error(); return -1; }
   rmh_enable_timers(ctx, 1);
   bool reduce_over_ranks = false;
   if (cd.ne_ghost > 0)
   {
      // The block has neighbours: the other blocks of the partition, one process each, over RCCL (comm_id_file) -- or,
      // for a self-wrapped block, this rank itself.  The solver classes then run the exchanges the reference's would
      // (SpaceLayout::ExchangeFaceNbrData / ExchangeElementExtrema) through the library's plan; a self-wrapped block
      // without an id file uses a one-rank RCCL communicator, or device copies (RMH_EXCHANGE=local; always under the
      // host emulation).
      const int np = (int)cd.peers.size();
      std::vector<int> prank(np), scount(np), rfirst(np), rcount(np);
      std::vector<const int *> selems(np);
      for (int j = 0; j < np; j++)
      {
         const Peer &pr = cd.peers[j];
         prank[j] = pr.rank;
         scount[j] = (int)pr.send_elems.size();
         selems[j] = pr.send_elems.data();
         rfirst[j] = pr.recv_slots.empty() ? 0 : pr.recv_slots.front();
         rcount[j] = (int)pr.recv_slots.size();
      }
      rmh_exchange_desc d = {np, prank.data(), scount.data(), selems.data(), rfirst.data(), rcount.data()};
      bool thin = false;
      const int P3[3] = {cc.px, cc.py, cc.pz};
      for (int dd = 0; dd < 3; dd++) { thin = thin || (P3[dd] > 1 && cd.n[dd] / P3[dd] < 2); }
      const char *tr = std::getenv("RMH_EXCHANGE");
      char id[128];
      bool ok = rmh_exchange_setup(ctx, &d, thin ? 0 : 1) == 0;
      if (ok && rccl_ranks)
      {
         if (cc.rank == 0) { std::remove(comm_id_file); ok = rmh_comm_unique_id(id) == 0; }
         std::string why;
         if (ok && !read_or_write_id(comm_id_file, cc.rank == 0, id)) { ok = false; why = "cannot exchange the RCCL unique id through the file"; }
         if (!why.empty()) { g_driver_error = "neighbour exchange: " + why; rmh_destroy(ctx); return -1; }
         ok = ok && rmh_comm_init(ctx, id, nranks, cc.rank) == 0;
         if (ok && cc.rank == 0) { std::remove(comm_id_file); }
         reduce_over_ranks = ok;
      }
      else if (ok && np == 1 && prank[0] == cc.rank && !(tr && std::string(tr) == "local") && rmh_comm_unique_id(id) == 0)
      {
         ok = rmh_comm_init(ctx, id, 1, 0) == 0;
      }
      else if (ok && np == 1 && prank[0] == cc.rank) { ok = rmh_comm_connect_local(ctx, 0, ctx, 0) == 0; }
      else if (ok)
      {
         g_driver_error = "neighbour exchange: a block with neighbour ranks needs the RCCL id file";
         rmh_destroy(ctx);
         return -1;
      }
      if (!ok)
      {
         g_driver_error = std::string("neighbour exchange: ") + rmh_last_error();
         rmh_destroy(ctx);
         return -1;
      }
      if (cfg->fused && !cfg->ps && (cfg->ode_solver == 0 || cfg->ode_solver == 3))
      {
         g_driver_error = "a block with neighbours and the one-kernel stage runs through rmhd_run_partitioned";
         rmh_destroy(ctx);
         return -1;
      }
   }
   if (rmh_set_bounds_type(ctx, cfg->bounds_type) != 0 || (cfg->dt_control && rmh_set_dt_control(ctx, 1) != 0))
   {
      g_driver_error = rmh_last_error();
      rmh_destroy(ctx);
      return -1;
   }
   const bool dtc = cfg->dt_control != 0;

   const int vsize = cd.ne_owned * cd.ndof;
   int rc = 0;
   {
      ParFiniteElementSpace pfes(ctx, cd.ne_owned, cd.ndof, (long long)cd.ne_global * cd.ndof, cd.ne_ghost > 0);
      DofInfo dofs(pfes);
      // solver factory of remhos.cpp:912-925, 927-995 for the options on the path
      HOSolver *ho_solver = nullptr;
      if (cfg->ho_type == 2) { ho_solver = new CGHOSolver(pfes); }
      else { ho_solver = new LocalInverseHOSolver(pfes, cfg->pa != 0); }
      LOSolver *lo_solver = nullptr;
      if (cc.lo_type == 5) { lo_solver = new MassBasedAvg(pfes, *ho_solver, nullptr); }
      else if (cc.lo_type == 3) { lo_solver = new PAResidualDistribution(pfes); }
      else { lo_solver = new PAResidualDistributionSubcell(pfes); }
      double dt = cd.dt;
      FCTSolver *fct_solver = new ClipScaleSolver(pfes, nullptr, dt);
      // -ps / -s 11|12|13 (remhos.cpp:484-507, 875-904): block vector [u | us], IDP solvers.  They limit a COMBINATION of
      // the stage's HO rate and the earlier limited updates, so the stage cannot be the one-kernel rmh_stage_fused:
      // HO kernel + fused limiter kernel (fused = 1) or the reference's call sequence (fused = 0).
      const bool ps = cfg->ps != 0;
      const int ode_type = cfg->ode_solver ? cfg->ode_solver : 3;
      const bool idp = ode_type > 10;
      RMH_VERIFY(ode_type == 3 || ode_type == 11 || ode_type == 12 || ode_type == 13, "-s must be 3, 11, 12 or 13");
      RMH_VERIFY(!ps || cd.exec_mode == 1, "Products are processed only in remap mode.");                         // remhos.cpp:1713
      RMH_VERIFY(!ps || !cfg->dt_control, "Automatic time step is not implemented for product remap.");           // remhos.cpp:1714
      RMH_VERIFY(!ps || cc.lo_type == 5, "product remap is built for -lo 5 (what the fused limiter kernel takes)");
      // fused = 1: one kernel per RK stage (rmh_stage_fused with the LO solver and the mass tolerance of the options)
      const bool fused = cfg->fused != 0 && !ps && !idp;
      if (fused)
      {
         RMH_CALL(rmh_set_lo_type(ctx, cc.lo_type));
         if (cfg->ho_type == 2) { RMH_CALL(rmh_set_mass_tol(ctx, 1e-12, 0.0, 500)); }
      }
      AdvectionOperator adv(pfes, dofs, ho_solver, lo_solver, fct_solver, (ps || idp) && cfg->fused != 0 && cc.lo_type == 5, ps);
      const bool vb = cfg->verify_bounds != 0;
      adv.verify_bounds = vb;         // remhos.cpp:1115-1116
      fct_solver->verify_bounds = vb;

      // Primary scalar field is u; for product remap we also evolve us (remhos.cpp:875-904): S = [u | us]
      Vector S((ps ? 2 : 1) * vsize);
      Vector u(S.ReadWrite(), vsize);
      u.CopyFromHost(cd.u0.data());
      std::vector<double> h_u(vsize), h_m(vsize), h_us(ps ? vsize : 0);
      if (ps)
      {
         // s = s0 where the element is active (BoolFunctionCoefficient on ComputeBoolIndicators(u), remhos.cpp:890-894),
         // us = u * s node by node ("we don't target conservation at initialization", :899-900)
         for (int e = 0; e < cd.ne_owned; e++)
         {
            bool active = false;
            for (int i = 0; i < cd.ndof; i++) { active = active || cd.u0[(size_t)e * cd.ndof + i] > 1e-12; } // EMPTY_ZONE_TOL
            for (int i = 0; i < cd.ndof; i++)
            {
               const size_t k = (size_t)e * cd.ndof + i;
               h_us[k] = cd.u0[k] * (active ? cd.s0[k] : 0.0);
            }
         }
         Vector us(S.ReadWrite() + vsize, vsize);
         us.CopyFromHost(h_us.data());
      }
      // initial mass (remhos.cpp:1073-1076)
      Vector masses(vsize);
      RMH_CALL(rmh_compute_lumped_mass(ctx, 0.0, masses.Write()));
      masses.CopyToHost(h_m.data());
      // (sums of ~1e7 terms: accumulated in extended precision so that the printed mass carries no summation error of its
      // own -- the conservation claims are at the 1e-12 level)
      long double mass0_acc = 0.0L;
      for (int i = 0; i < vsize; i++) { mass0_acc += (long double)h_m[i] * cd.u0[i]; }
      // MPI_Allreduce of the report (remhos.cpp:1074-1081, 1412-1421) over the ranks of a partition: rmh_allreduce
      auto reduce = [&](double v, int op) -> double
      {
         if (reduce_over_ranks) { RMH_CALL(rmh_allreduce(ctx, &v, 1, op)); }
         return v;
      };
      const double mass0 = reduce((double)mass0_acc, 0);
      long double mass0_us_acc = 0.0L;
      for (int i = 0; i < (ps ? vsize : 0); i++) { mass0_us_acc += (long double)h_m[i] * h_us[i]; } // remhos.cpp:1077-1081

      // Print the starting mesh and initial condition (remhos.cpp:1015-1030)
      if (cfg->save)
      {
         const std::string e = save_mfem(cd, 0.0, cd.u0.data(), "meshHO_init.mesh", "sltn_init.gf");
         RMH_VERIFY(e.empty(), e.c_str());
      }
      ODESolver *ode_solver_p = nullptr; // remhos.cpp:486-500
      switch (ode_type)
      {
         case 11: ode_solver_p = new ForwardEulerIDPSolver(); break;
         case 12: ode_solver_p = new RK2IDPSolver(); break;
         case 13: ode_solver_p = new RK3IDPSolver(); break;
         default: ode_solver_p = new RK3SSPSolver(); break;
      }
      ODESolver &ode_solver = *ode_solver_p;
      const int stages_per_step = ode_type == 11 ? 1 : (ode_type == 12 ? 2 : 3);
      double t = 0.0;
      adv.SetTime(t);
      ode_solver.Init(adv);
      // For remap, the pseudo-time always evolves from 0 to 1 (remhos.cpp:1128-1134)
      const double t_final = cd.exec_mode == 1 ? 1.0 : cc.t_final;
      bool done = false;
      int ti = 0, ti_total = 0; // accepted steps / all steps incl. repeated ones (remhos.cpp:1142)
      HIP_CALL(hipDeviceSynchronize());
      const auto w0 = std::chrono::steady_clock::now();
      Vector y1(fused ? vsize : 0), y2(fused ? vsize : 0);
      unsigned long long tok_u = 0;
      Vector Sold(dtc ? vsize : 0);
      int repeats = 0;
      // -vb: global extrema of u (GetMinMax, remhos.cpp:1124) and of s = us / u on its active dofs (ComputeMinMaxS,
      // remhos_sync.cpp:116-140) -- element extrema on the device, the ne values reduced on the host (a debug mode)
      Vector vb_du(vb && fused ? vsize : 0), vb_s(vb && ps ? vsize : 0);
      bool *vb_flags = nullptr;
      if (vb && ps) { HIP_CALL(hipMalloc((void **)&vb_flags, (size_t)cd.ne_owned + vsize)); }
      std::vector<double> h_emin(vb ? cd.ne_owned : 0), h_emax(vb ? cd.ne_owned : 0);
      auto global_minmax = [&](const Vector &v, Array<bool> *el, Array<bool> *df, double &lo, double &hi)
      {
         dofs.ComputeElementsMinMax(v, dofs.xe_min, dofs.xe_max, el, df);
         dofs.xe_min.CopyToHost(h_emin.data());
         dofs.xe_max.CopyToHost(h_emax.data());
         lo = INFINITY;
         hi = -INFINITY;
         for (int e = 0; e < cd.ne_owned; e++) { lo = std::fmin(lo, h_emin[e]); hi = std::fmax(hi, h_emax[e]); }
         lo = reduce(lo, 1);
         hi = reduce(hi, 2);
      };
      auto s_minmax = [&](double &lo, double &hi)
      {
         Vector us(S.ReadWrite() + vsize, vsize);
         Array<bool> el(vb_flags, cd.ne_owned), df(vb_flags + cd.ne_owned, vsize);
         ComputeRatio(pfes, us, u, vb_s, el, df);
         global_minmax(vb_s, &el, &df, lo, hi);
      };
      double u_min_glob = 0., u_max_glob = 0., s_min_glob = 0., s_max_glob = 0.;
      if (vb)
      {
         global_minmax(u, nullptr, nullptr, u_min_glob, u_max_glob);
         if (ps) { s_minmax(s_min_glob, s_max_glob); }
      }
      while (!done)
      {
         // Monotonicity check for debug purposes mainly (remhos.cpp:1218-1262; forced_bounds holds for every LO solver here)
         if (vb)
         {
            const double eps = 1e-10;
            double u_min_new, u_max_new, s_min_new = s_min_glob, s_max_new = s_max_glob;
            global_minmax(u, nullptr, nullptr, u_min_new, u_max_new);
            if (ps) { s_minmax(s_min_new, s_max_new); }
            const bool unit_range = cc.problem % 10 == 6 || cc.problem % 10 == 7;
            const double ulo = unit_range ? 0.0 : u_min_glob, uhi = unit_range ? 1.0 : u_max_glob;
            const double slo = unit_range ? 0.0 : s_min_glob, shi = unit_range ? 1.0 : s_max_glob;
            char msg[96];
            std::snprintf(msg, sizeof(msg), "Undershoot of %.6e", ulo - u_min_new);
            RMH_VERIFY(u_min_new > ulo - eps, msg);
            std::snprintf(msg, sizeof(msg), "Overshoot of %.6e", u_max_new - uhi);
            RMH_VERIFY(u_max_new < uhi + eps, msg);
            std::snprintf(msg, sizeof(msg), "Undershoot in s of %.6e", slo - s_min_new);
            RMH_VERIFY(s_min_new > slo - eps, msg);
            std::snprintf(msg, sizeof(msg), "Overshoot in s of %.6e", s_max_new - shi);
            RMH_VERIFY(s_max_new < shi + eps, msg);
            if (!unit_range) { u_min_glob = u_min_new; u_max_glob = u_max_new; } // (s_min / s_max stay the initial ones, like the reference's)
         }
         double dt_real = std::min(dt, t_final - t);
         // This also resets the time step estimate when automatic dt is on (remhos.cpp:1150-1152)
         adv.SetDt(dt_real);
         if (dtc)
         {
            adv.ResetTimeStepRatio();
            Sold = u;
         }
         if (fused && vb)
         {
            // the one-kernel stage under -vb: the kernel also hands out the limited rate, and the granular kernels form the
            // dof bounds of the stage input for the check of remhos.cpp:1833-1837 (the LO rate never leaves the kernel:
            // its check, :1824-1828, is the granular sequence's -- rmhd_config.fused = 0)
            const int ne = cd.ne_owned;
            auto stage = [&](const Vector &in, double ts, const double *xb, double a, double b, Vector &out)
            {
               RMH_CALL(rmh_setup(ctx, ts));
               RMH_CALL(rmh_stage_fused_chain(ctx, in.Read(), dt_real, xb, a, b, dt_real, out.Write(), vb_du.Write(), 0, ne, 1, 0, nullptr));
               dofs.ComputeElementsMinMax(in, dofs.xe_min, dofs.xe_max);
               dofs.ComputeBounds(dofs.xe_min, dofs.xe_max, dofs.xi_min, dofs.xi_max);
               check_violation(pfes, in, dt_real, vb_du, dofs.xi_min, dofs.xi_max, "LimitMult FCT solution u", 1e-12, nullptr);
            };
            stage(u, t, nullptr, 0.0, 1.0, y1);
            stage(y1, t + dt_real, u.Read(), 3. / 4, 1. / 4, y2);
            stage(y2, t + dt_real / 2, u.Read(), 1. / 3, 2. / 3, u); // (y_out may alias x_base)
            tok_u = 0;
            t += dt_real;
         }
         else if (fused)
         {
            // one kernel per RK stage (rmh_stage_fused): same stage times and combinations as
            // RK3SSPSolver::Step; input and output vectors of a stage differ
            // (tok_u names the element extrema of u left by the stage that wrote it: rmh_stage_fused_chain)
            unsigned long long tok = 0;
            const int ne = cd.ne_owned;
            RMH_CALL(rmh_setup(ctx, t));
            RMH_CALL(rmh_stage_fused_chain(ctx, u.Read(), dt_real, nullptr, 0.0, 1.0, dt_real, y1.Write(), nullptr, 0, ne, 1, tok_u, &tok));
            RMH_CALL(rmh_setup(ctx, t + dt_real));
            RMH_CALL(rmh_stage_fused_chain(ctx, y1.Read(), dt_real, u.Read(), 3. / 4, 1. / 4, dt_real, y2.Write(), nullptr, 0, ne, 1, tok, &tok));
            RMH_CALL(rmh_setup(ctx, t + dt_real / 2));
            RMH_CALL(rmh_stage_fused_chain(ctx, y2.Read(), dt_real, u.Read(), 1. / 3, 2. / 3, dt_real, u.Write(), nullptr, 0, ne, 1, tok, &tok_u));
            t += dt_real;
         }
         else { ode_solver.Step(S, t, dt_real); }
         ti++;
         ti_total++;
         if (dtc)
         {
            // remhos.cpp:1178-1197; the estimate is the minimum over the ranks of a partition (MPI_Allreduce(MIN),
            // remhos.cpp:1993): every rank takes the same accept / repeat decision, hence issues the same exchanges
            const double dt_ratio = reduce(adv.GetTimeStepRatio(), 1);
            if (dt_ratio < 1.)
            {
               // Repeat with the proper time step.
               ti--;
               t -= dt_real;
               u = Sold;
               tok_u = 0; // (u is no longer the output of the last stage)
               dt = 0.85 * dt;
               repeats++;
               RMH_VERIFY(dt >= 1e-12, "The time step crashed!");
               continue;
            }
            else if (dt_ratio > 1.25) { dt *= 1.02; }
         }
         done = (t >= t_final - 1.e-8 * dt);
         if (ti_total == cc.max_steps) { done = true; } // -ms counts repeated steps too (remhos.cpp:1296)
      }
      HIP_CALL(hipDeviceSynchronize());
      const auto w1 = std::chrono::steady_clock::now();
      if (vb_flags) { (void)hipFree(vb_flags); }

      // final mass: remap uses the lumped mass at the final position (remhos.cpp:1382-1413)
      if (cd.exec_mode == 1)
      {
         RMH_CALL(rmh_compute_lumped_mass(ctx, t, masses.Write()));
         masses.CopyToHost(h_m.data());
      }
      u.CopyToHost(h_u.data());
      // Print the final mesh and solution (remhos.cpp:1365-1380)
      if (cfg->save)
      {
         const std::string e = save_mfem(cd, t, h_u.data(), "meshHO_final.mesh", "sltn_final.gf");
         RMH_VERIFY(e.empty(), e.c_str());
      }
      long double mass_acc = 0.0L;
      double umax = -INFINITY;
      for (int i = 0; i < vsize; i++)
      {
         mass_acc += (long double)h_m[i] * h_u[i];
         umax = std::fmax(umax, h_u[i]);
      }
      const double mass = reduce((double)mass_acc, 0);
      umax = reduce(umax, 2);
      if (u_final) { std::copy(h_u.begin(), h_u.end(), u_final); }
      {
         // Compute errors, if the exact solution is known (remhos.cpp:1438-1470)
         double es[3];
         if (lp_error_sums(cd, cc.problem, t, h_u.data(), es).empty())
         {
            res->has_errors = 1;
            res->err_l1 = reduce(es[0], 0);
            res->err_l2 = std::sqrt(reduce(es[1], 0));
            res->err_linf = reduce(es[2], 2);
         }
      }
      if (ps)
      {
         // remhos.cpp:1404, 1416-1434: mass of us with the same lumped masses; max of the ratio s = us / u
         Vector us(S.ReadWrite() + vsize, vsize), s_fin(vsize);
         us.CopyToHost(h_us.data());
         if (us_final) { std::copy(h_us.begin(), h_us.end(), us_final); }
         long double acc = 0.0L;
         for (int i = 0; i < vsize; i++) { acc += (long double)h_m[i] * h_us[i]; }
         bool *flags = nullptr;
         HIP_CALL(hipMalloc((void **)&flags, (size_t)cd.ne_owned + vsize));
         Array<bool> el(flags, cd.ne_owned), dfl(flags + cd.ne_owned, vsize);
         ComputeRatio(pfes, us, u, s_fin, el, dfl);
         std::vector<double> h_s(vsize);
         s_fin.CopyToHost(h_s.data());
         (void)hipFree(flags);
         double smax = -INFINITY;
         for (int i = 0; i < vsize; i++) { smax = std::fmax(smax, h_s[i]); }
         res->final_mass_us = reduce((double)acc, 0);
         res->mass0_us = reduce((double)mass0_us_acc, 0);
         res->mass_loss_us = std::fabs(res->mass0_us - res->final_mass_us);
         res->s_max = reduce(smax, 2);
      }
      adv.Timer().Update(ctx);
      const TimingData &T = adv.Timer();
      res->final_mass = mass;
      res->max_value = umax;
      res->mass0 = mass0;
      res->mass_loss = std::fabs(mass0 - mass);
      res->dt = dt;
      res->t_end = t;
      res->steps = ti;
      res->stages = stages_per_step * ti_total; // the FOMs count repeated steps too (remhos.cpp:1340-1348)
      res->global_dofs = pfes.GlobalVSize();
      res->t_rhs = T.sw_rhs;
      res->t_inv = T.sw_L2inv;
      res->t_lo = T.sw_LO;
      res->t_fct = T.sw_FCT;
      res->t_total = T.sw_rhs + T.sw_LO + T.sw_FCT; // remhos.cpp:1933 (omits INV)
      const double dofs_steps = 1e-6 * (double)res->global_dofs * res->stages;
      res->fom_rhs = T.sw_rhs > 0 ? dofs_steps / T.sw_rhs : 0;
      res->fom_inv = T.sw_L2inv > 0 ? dofs_steps / T.sw_L2inv : 0;
      res->fom_lo = T.sw_LO > 0 ? dofs_steps / T.sw_LO : 0;
      res->fom_fct = T.sw_FCT > 0 ? dofs_steps / T.sw_FCT : 0;
      res->fom = res->t_total > 0 ? dofs_steps / res->t_total : 0;
      res->wall = std::chrono::duration<double>(w1 - w0).count();
      res->fom_wall = dofs_steps / res->wall;
      int it = 0;
      rmh_last_cg_iters(ctx, &it);
      res->cg_iters_max = it;
      res->repeats = repeats;
      delete ode_solver_p;
      delete fct_solver;
      delete lo_solver;
      delete ho_solver;
   }
   rmh_destroy(ctx);
   return rc;
}


// ---- box-partitioned runs: the time loop of remhos() (remhos.cpp:1146-1330) over the blocks of a ParMesh-like
// partition, one fused kernel per RK stage and block, one neighbour exchange per stage (rmh_exchange_begin / _end) ----
namespace
{
struct Block
{
   CaseData cd;
   rmh_ctx *ctx = nullptr;
   double *x = nullptr, *y1 = nullptr, *y2 = nullptr, *xold = nullptr, *m = nullptr;
   int vsize = 0;
   unsigned long long tok = 0; // token of the extrema of the vector the next stage reads (rmh_stage_fused_chain)
   hipStream_t stream = nullptr; // the context's stream: a non-blocking one, so that the exchange stream overlaps it
};

// ncclUniqueId rendezvous through a file.  Record = magic, launch tag, id.  The launch tag is what tells this launch's
// file from a stale one (a path reused by a later run, or left by a crashed one): RMH_COMM_NONCE if the launcher exports
// it (bench.py does), else the parent process id (ranks started by one shell loop / one torchrun agent share it; ranks started
// through per-rank wrappers have different parents and NEED the nonce).  Rank 0 removes a stale file first, writes <path>.tmp
// and renames it; the others poll until they read a record with their tag that is fresh: not older than five minutes with a
// nonce, not older than the reader's own start (less 30 s) without.  Rank 0 deletes the file once ncclCommInitRank -- a
// collective -- has returned.
struct IdRecord
{
   char magic[8];
   long long tag;
   char id[128];
};

long long launch_tag()
{
   if (const char *v = std::getenv("RMH_COMM_NONCE")) { return std::atoll(v); }
   return (long long)getppid();
}
// (without a nonce of the launcher the parent pid can repeat across launches of one shell: a reader then also refuses a
// record that is older than its own start by more than half a minute -- a file left by an earlier, crashed launch)
const std::time_t g_library_loaded = std::time(nullptr);

bool read_or_write_id(const char *path, bool writer, char id[128])
{
   IdRecord rec;
   if (writer)
   {
      std::memcpy(rec.magic, "RMHNCCL1", 8);
      rec.tag = launch_tag();
      std::memcpy(rec.id, id, 128);
      const std::string tmp = std::string(path) + ".tmp";
      FILE *f = std::fopen(tmp.c_str(), "wb");
      if (!f) { return false; }
      const bool ok = std::fwrite(&rec, sizeof(rec), 1, f) == 1;
      std::fclose(f);
      return ok && std::rename(tmp.c_str(), path) == 0;
   }
   for (int tries = 0; tries < 12000; tries++)
   {
      struct stat sb;
      FILE *f = std::fopen(path, "rb");
      if (f)
      {
         const bool ok = std::fread(&rec, sizeof(rec), 1, f) == 1 && fstat(fileno(f), &sb) == 0;
         std::fclose(f);
         const bool fresh = std::getenv("RMH_COMM_NONCE") ? std::time(nullptr) - sb.st_mtime < 300 : sb.st_mtime + 30 >= g_library_loaded;
         if (ok && std::memcmp(rec.magic, "RMHNCCL1", 8) == 0 && rec.tag == launch_tag() && fresh)
         {
            std::memcpy(id, rec.id, 128);
            return true;
         }
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
   }
   return false;
}
} // namespace

extern "C" int rmhd_axpby(double a, const double *x, double b, const double *y, double *z, long long n, void *stream)
{
   if (n < 0 || n > 0x7fffffffLL || (n > 0 && (!x || !y || !z))) { g_driver_error = "rmhd_axpby: bad argument"; return -1; }
   if (n == 0) { return 0; }
   hipLaunchKernelGGL(remhos::axpby_kernel, dim3(remhos::grid_for((int)n)), dim3(256), 0, (hipStream_t)stream, a, x, b, y, z, (int)n);
   if (hipGetLastError() != hipSuccess) { g_driver_error = "rmhd_axpby: launch failed"; return -1; }
   return 0;
}

extern "C" int rmhd_id_file_exchange(const char *path, int writer, char id[128])
{
   if (!path || !path[0] || !id) { g_driver_error = "rmhd_id_file_exchange: null argument"; return -1; }
   if (!read_or_write_id(path, writer != 0, id)) { g_driver_error = std::string("rmhd_id_file_exchange: ") + (writer ? "cannot write " : "no fresh record of this launch at ") + path; return -1; }
   return 0;
}

extern "C" int rmhd_run_partitioned(const rmhd_config *cfg, const char *comm_id_file, int device, rmhd_result *res)
{
   if (!cfg || !res) { g_driver_error = "null argument"; return -1; }
   CaseConfig cc0 = to_config(*cfg);
   const int nranks = cc0.px * cc0.py * cc0.pz;
   const bool rccl = comm_id_file && comm_id_file[0];
   if (!cfg->fused) { g_driver_error = "rmhd_run_partitioned runs the one-kernel stage (fused = 1)"; return -1; }
   std::vector<Block> blocks(rccl ? 1 : nranks);
   auto cleanup = [&]()
   {
      for (Block &b : blocks)
      {
         (void)hipFree(b.x); (void)hipFree(b.y1); (void)hipFree(b.y2); (void)hipFree(b.xold); (void)hipFree(b.m);
         if (b.ctx) { rmh_destroy(b.ctx); }
         if (b.stream) { (void)hipStreamDestroy(b.stream); }
      }
   };
#define RMHD_TRY(expr)                                                          \
   do {                                                                         \
      if ((expr) != 0) { g_driver_error = std::string(#expr) + ": " + rmh_last_error(); cleanup(); return -1; } \
   } while (0)
#define RMHD_HIP(expr)                                                          \
   do {                                                                         \
      if ((expr) != hipSuccess) { g_driver_error = #expr; cleanup(); return -1; } \
   } while (0)
   // ---- set-up of every block this process owns ------------------------------------------------------------------
   for (size_t k = 0; k < blocks.size(); k++)
   {
      Block &b = blocks[k];
      CaseConfig cc = cc0;
      cc.rank = rccl ? cc0.rank : (int)k;
      const std::string err = build_case(cc, b.cd);
      if (!err.empty()) { g_driver_error = err; cleanup(); return -1; }
      rmh_layout L;
      if (b.cd.dim != 3) { g_driver_error = "the partitioned stage loop is for the 3-D cases"; cleanup(); return -1; }
      L.dim = 3; L.order = b.cd.order; L.mesh_order = 2; L.exec_mode = b.cd.exec_mode;
      L.ne_owned = b.cd.ne_owned; L.ne_ghost = b.cd.ne_ghost;
      L.x0 = b.cd.x0.data(); L.vel = b.cd.vel.data(); L.face_nbr = b.cd.face_nbr.data(); L.stencil27 = b.cd.stencil27.data();
      L.subcell_vel = b.cd.subcell_vel.empty() ? nullptr : b.cd.subcell_vel.data();
      L.device = device;
      RMHD_TRY(rmh_create(&L, &b.ctx));
      {
         // RMH_COMM_CUS = k: the stage kernels leave k compute units to the exchange stream (rmh_stream_create_reserving; only
         // where RCCL kernels run beside them -- same-process blocks exchange by device copies)
         const char *ecu = std::getenv("RMH_COMM_CUS");
         const int kcu = (rccl && ecu) ? std::atoi(ecu) : 0;
         void *sv = nullptr;
         RMHD_TRY(rmh_stream_create_reserving(device, kcu > 0 ? kcu : 0, &sv));
         b.stream = (hipStream_t)sv;
      }
      RMHD_TRY(rmh_set_stream(b.ctx, b.stream));
      RMHD_TRY(rmh_set_lo_type(b.ctx, cc.lo_type));
      RMHD_TRY(rmh_set_bounds_type(b.ctx, cfg->bounds_type));
      if (cfg->dt_control) { RMHD_TRY(rmh_set_dt_control(b.ctx, 1)); }
      if (cfg->ho_type == 2) { RMHD_TRY(rmh_set_mass_tol(b.ctx, 1e-12, 0.0, 500)); }
      else if (cfg->pa)
      {
         // LocalInverseHOSolver on a partially assembled M (remhos_ho.cpp:77-81): DGMassInverse's rule, completed
         RMHD_TRY(rmh_set_mass_tol(b.ctx, 0.0, 1e-8, 100));
         RMHD_TRY(rmh_set_mass_completion(b.ctx, 1, 1));
      }
      b.vsize = b.cd.ne_owned * b.cd.ndof;
      const size_t bytes = sizeof(double) * (size_t)b.vsize;
      RMHD_HIP(hipMalloc((void **)&b.x, bytes));
      RMHD_HIP(hipMalloc((void **)&b.y1, bytes));
      RMHD_HIP(hipMalloc((void **)&b.y2, bytes));
      RMHD_HIP(hipMalloc((void **)&b.m, bytes));
      if (cfg->dt_control) { RMHD_HIP(hipMalloc((void **)&b.xold, bytes)); }
      RMHD_HIP(hipMemcpy(b.x, b.cd.u0.data(), bytes, hipMemcpyHostToDevice));
      // exchange plan from the case builder's halo lists; compact records unless a block is one element thin
      const int np = (int)b.cd.peers.size();
      std::vector<int> prank(np), scount(np), rfirst(np), rcount(np);
      std::vector<const int *> selems(np);
      for (int j = 0; j < np; j++)
      {
         const Peer &pr = b.cd.peers[j];
         prank[j] = pr.rank;
         scount[j] = (int)pr.send_elems.size();
         selems[j] = pr.send_elems.data();
         rfirst[j] = pr.recv_slots.empty() ? 0 : pr.recv_slots.front();
         rcount[j] = (int)pr.recv_slots.size();
      }
      rmh_exchange_desc d = {np, prank.data(), scount.data(), selems.data(), rfirst.data(), rcount.data()};
      bool thin = false;
      const int P3[3] = {cc.px, cc.py, cc.pz};
      for (int dd = 0; dd < 3; dd++) { thin = thin || (P3[dd] > 1 && b.cd.n[dd] / P3[dd] < 2); }
      RMHD_TRY(rmh_exchange_setup(b.ctx, &d, thin ? 0 : 1));
   }
   if (rccl)
   {
      char id[128];
      if (cc0.rank == 0)
      {
         std::remove(comm_id_file); // (a stale record: see read_or_write_id)
         RMHD_TRY(rmh_comm_unique_id(id));
      }
      if (!read_or_write_id(comm_id_file, cc0.rank == 0, id)) { g_driver_error = "cannot exchange the RCCL unique id through the file"; cleanup(); return -1; }
      RMHD_TRY(rmh_comm_init(blocks[0].ctx, id, nranks, cc0.rank));
      if (cc0.rank == 0) { std::remove(comm_id_file); } // (every rank has read it: ncclCommInitRank is collective)
   }
   else
   {
      for (int r = 0; r < nranks; r++)
      {
         for (size_t j = 0; j < blocks[r].cd.peers.size(); j++)
         {
            const int q = blocks[r].cd.peers[j].rank;
            if (q < r) { continue; } // (each pair once; a block may be its own neighbour's neighbour only through q != r)
            int jq = -1;
            for (size_t i = 0; i < blocks[q].cd.peers.size(); i++) { if (blocks[q].cd.peers[i].rank == r) { jq = (int)i; } }
            if (jq < 0) { g_driver_error = "halo lists of two blocks do not mirror each other"; cleanup(); return -1; }
            RMHD_TRY(rmh_comm_connect_local(blocks[r].ctx, (int)j, blocks[q].ctx, jq));
         }
      }
   }
   bool reduce_failed = false;
   auto reduce = [&](double v, int op) -> double
   {
      // over the blocks of this process, then over the ranks of the communicator
      if (rccl) { if (rmh_allreduce(blocks[0].ctx, &v, 1, op) != 0) { reduce_failed = true; } }
      return v;
   };
   bool want_errors = false, have_errors = false;
   double err_sums[3] = {0, 0, 0};
   auto mass_and_max = [&](double t, double &mass, double &umax) -> bool
   {
      mass = 0.0; umax = -INFINITY;
      for (Block &b : blocks)
      {
         if (rmh_compute_lumped_mass(b.ctx, b.cd.exec_mode == 1 ? t : 0.0, b.m) != 0) { return false; }
         if (hipStreamSynchronize(b.stream) != hipSuccess) { return false; } // (a non-blocking stream: the copies below do not wait for it)
         std::vector<double> hm(b.vsize), hu(b.vsize);
         if (hipMemcpy(hm.data(), b.m, sizeof(double) * b.vsize, hipMemcpyDeviceToHost) != hipSuccess) { return false; }
         if (hipMemcpy(hu.data(), b.x, sizeof(double) * b.vsize, hipMemcpyDeviceToHost) != hipSuccess) { return false; }
         long double acc = 0.0L; // (extended precision: see rmhd_run)
         for (int i = 0; i < b.vsize; i++) { acc += (long double)hm[i] * hu[i]; umax = std::fmax(umax, hu[i]); }
         mass += (double)acc;
         double es[3];
         if (want_errors && lp_error_sums(b.cd, cc0.problem, t, hu.data(), es).empty())
         {
            have_errors = true;
            err_sums[0] += es[0]; err_sums[1] += es[1]; err_sums[2] = std::fmax(err_sums[2], es[2]);
         }
      }
      mass = reduce(mass, 0);  // MPI_Allreduce SUM, remhos.cpp:1412
      umax = reduce(umax, 2);  // MPI_Allreduce MAX, remhos.cpp:1415
      return true;
   };
   double mass0 = 0, umax0 = 0;
   if (!mass_and_max(0.0, mass0, umax0)) { g_driver_error = "initial mass"; cleanup(); return -1; }
   // ---- time loop ----------------------------------------------------------------------------------------------
   const double t_final = blocks[0].cd.exec_mode == 1 ? 1.0 : cc0.t_final;
   double dt = blocks[0].cd.dt, t = 0.0;
   int ti = 0, ti_total = 0, repeats = 0;
   bool done = false;
   // TimingData buckets: HIP events around the launches of every `timer_every`-th step (default 4; RMH_DRIVER_TIMERS = k, 0 =
   // never), scaled to all steps in the report.  Events around EVERY launch cost the 48^3-element blocks of an 8-rank strong run
   // 8 % (measured with all eight blocks on one GPU: 14.8 k -> 16.1 k MDOFs*stage/s without them); the wall clock is exact.
   const char *tenv = std::getenv("RMH_DRIVER_TIMERS");
   const int timer_every = tenv ? std::max(0, std::atoi(tenv)) : 4;
   int timed_steps = 0, counted_steps = 0;
   for (Block &b : blocks) { RMHD_TRY(rmh_enable_timers(b.ctx, 0)); }
   RMHD_HIP(hipDeviceSynchronize());
   auto w0 = std::chrono::steady_clock::now();
   int timed_from = 0; // steps taken before the stopwatches (re)started (rmhd_config.warmup_steps)
   // one RK stage of all blocks: post every exchange, run the elements that reach no ghost while the messages are
   // in flight, complete the exchanges, run the halo-dependent shells
   auto stage = [&](int which, double ts, double dt_real, double ra, double rb) -> bool
   {
      auto in = [&](Block &b) { return which == 0 ? b.x : (which == 1 ? b.y1 : b.y2); };
      auto out = [&](Block &b) { return which == 0 ? b.y1 : (which == 1 ? b.y2 : b.x); };
      for (Block &b : blocks) { if (rmh_exchange_begin(b.ctx, in(b)) != 0) { return false; } }
      for (Block &b : blocks)
      {
         if (rmh_setup(b.ctx, ts) != 0) { return false; }
         if (rmh_stage_fused_chain(b.ctx, in(b), dt_real, which == 0 ? nullptr : b.x, ra, rb, dt_real, out(b), nullptr,
                                   b.cd.ne_halo, b.cd.ne_owned, 0, b.tok, nullptr) != 0) { return false; }
      }
      for (Block &b : blocks) { if (rmh_exchange_end(b.ctx) != 0) { return false; } }
      for (Block &b : blocks)
      {
         if (rmh_stage_fused_chain(b.ctx, in(b), dt_real, which == 0 ? nullptr : b.x, ra, rb, dt_real, out(b), nullptr, 0,
                                   b.cd.ne_halo, 1, b.tok, &b.tok) != 0) { return false; }
      }
      return true;
   };
   while (!done)
   {
      if (cfg->warmup_steps > 0 && ti_total == cfg->warmup_steps)
      {
         // the timed region starts here: everything before is finished on every rank (the reduction is the barrier)
         RMHD_HIP(hipDeviceSynchronize());
         (void)reduce(0.0, 0);
         for (Block &b : blocks) { RMHD_TRY(rmh_reset_timers(b.ctx)); }
         timed_from = ti_total;
         timed_steps = counted_steps = 0;
         w0 = std::chrono::steady_clock::now();
      }
      {
         const bool sample = timer_every > 0 && counted_steps % timer_every == 0;
         for (Block &b : blocks) { RMHD_TRY(rmh_enable_timers(b.ctx, sample ? 1 : 0)); }
         timed_steps += sample ? 1 : 0;
         counted_steps++;
      }
      const double dt_real = std::min(dt, t_final - t);
      if (cfg->dt_control)
      {
         for (Block &b : blocks)
         {
            RMHD_TRY(rmh_dt_estimate_reset(b.ctx));
            RMHD_HIP(hipMemcpyAsync(b.xold, b.x, sizeof(double) * b.vsize, hipMemcpyDeviceToDevice, b.stream));
         }
         RMHD_HIP(hipDeviceSynchronize());
      }
      // RK3SSPSolver::Step: stage times t, t + dt, t + dt/2
      if (!stage(0, t, dt_real, 0.0, 1.0) || !stage(1, t + dt_real, dt_real, 3. / 4, 1. / 4) ||
          !stage(2, t + dt_real / 2, dt_real, 1. / 3, 2. / 3))
      {
         g_driver_error = std::string("stage: ") + rmh_last_error(); cleanup(); return -1;
      }
      t += dt_real;
      ti++;
      ti_total++;
      if (cfg->dt_control)
      {
         double est = INFINITY;
         for (Block &b : blocks) { double e = 0; RMHD_TRY(rmh_dt_estimate_get(b.ctx, &e)); est = std::fmin(est, e); }
         est = reduce(est, 1); // MPI_Allreduce MIN, remhos.cpp:1993
         const double ratio = dt_real != 0. ? est / dt_real : 0.;
         if (ratio < 1.)
         {
            ti--;
            t -= dt_real;
            for (Block &b : blocks)
            {
               RMHD_HIP(hipMemcpyAsync(b.x, b.xold, sizeof(double) * b.vsize, hipMemcpyDeviceToDevice, b.stream));
               b.tok = 0; // (x is no longer the output of the last stage)
            }
            RMHD_HIP(hipDeviceSynchronize());
            dt = 0.85 * dt;
            repeats++;
            if (dt < 1e-12) { g_driver_error = "The time step crashed!"; cleanup(); return -1; }
            continue;
         }
         else if (ratio > 1.25) { dt *= 1.02; }
      }
      done = (t >= t_final - 1.e-8 * dt);
      if (ti_total == cc0.max_steps) { done = true; }
   }
   RMHD_HIP(hipDeviceSynchronize());
   (void)reduce(0.0, 0); // (every rank has finished its last stage)
   const auto w1 = std::chrono::steady_clock::now();
   double mass = 0, umax = 0;
   want_errors = true; // (remhos.cpp:1438-1470; MPI-reduced like ComputeLpError)
   if (!mass_and_max(t, mass, umax)) { g_driver_error = "final mass"; cleanup(); return -1; }
   if (have_errors)
   {
      err_sums[0] = reduce(err_sums[0], 0);
      err_sums[1] = reduce(err_sums[1], 0);
      err_sums[2] = reduce(err_sums[2], 2);
   }
   double tk = 0.0;
   int itmax = 0;
   for (Block &b : blocks)
   {
      double tm[4];
      RMHD_TRY(rmh_timers(b.ctx, tm));
      tk = std::fmax(tk, tm[0]);
      int it = 0;
      rmh_last_cg_iters(b.ctx, &it);
      itmax = std::max(itmax, it);
   }
   if (timed_steps > 0) { tk *= (double)counted_steps / timed_steps; } // (sampled steps -> all steps of the timed region)
   tk = reduce(tk, 2); // MPI_Reduce MAX of the stopwatches, remhos.cpp:1934
   if (reduce_failed) { g_driver_error = std::string("rmh_allreduce: ") + rmh_last_error(); cleanup(); return -1; }
   std::memset(res, 0, sizeof(*res));
   res->timer_every = timer_every;
   res->timer_steps = timed_steps;
   res->final_mass = mass;
   res->max_value = umax;
   res->mass0 = mass0;
   res->mass_loss = std::fabs(mass0 - mass);
   if (have_errors)
   {
      res->has_errors = 1;
      res->err_l1 = err_sums[0];
      res->err_l2 = std::sqrt(err_sums[1]);
      res->err_linf = err_sums[2];
   }
   res->dt = dt;
   res->t_end = t;
   res->steps = ti;
   res->stages = 3 * ti_total;
   res->global_dofs = blocks[0].cd.ne_global * blocks[0].cd.ndof;
   res->t_rhs = tk; // the whole stage is one kernel: everything is in the RHS bucket
   res->t_total = tk;
   res->timed_stages = 3 * (ti_total - timed_from);
   res->n_peers = (int)blocks[0].cd.peers.size();
   res->transport = rccl ? 1 : (nranks > 1 ? 2 : 0);
   if (rccl) { (void)rmh_comm_count(blocks[0].ctx, &res->comm_ranks); }
   {
      long long sd = 0, gd = 0;
      if (res->n_peers > 0) { (void)rmh_exchange_buffers(blocks[0].ctx, nullptr, &sd, nullptr, &gd); }
      res->send_bytes_per_stage = 8 * sd;
      res->recv_bytes_per_stage = 8 * gd;
   }
   const double dofs_steps = 1e-6 * (double)res->global_dofs * res->timed_stages;
   res->fom_rhs = tk > 0 ? dofs_steps / tk : 0;
   res->fom = res->fom_rhs;
   res->wall = reduce(std::chrono::duration<double>(w1 - w0).count(), 2);
   res->fom_wall = dofs_steps / res->wall;
   res->cg_iters_max = itmax;
   res->repeats = repeats;
   cleanup();
   return 0;
#undef RMHD_TRY
#undef RMHD_HIP
}
