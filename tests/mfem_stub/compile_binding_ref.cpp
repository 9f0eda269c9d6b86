// compile-only: the binding header behind the REFERENCE'S OWN interface headers (found through -I <reference checkout>; only
// mfem.hpp and general/forall.hpp are the stubs of this directory) -- tests/test_binding_compiles.py, skipped where the reference
// checkout is absent.  What the binding derives from and overrides is then what Remhos declares, not a restatement of it.
#include "remhos_ho.hpp"
#include "remhos_lo.hpp"
#include "remhos_fct.hpp"
#include "remhos_tools.hpp"

#include "remhos_amd/mfem_binding.hpp"

// what the solver factory of remhos.cpp:912-995, 1083-1108 would do with the plugins (the reference's own TimingData, DofInfo
// and SmoothnessIndicator types in the signatures)
void factory(mfem::ParFiniteElementSpace &pfes, const mfem::GridFunction &x0, const mfem::GridFunction &v, mfem::real_t dt,
             mfem::HOSolver *&ho, mfem::LOSolver *&lo, mfem::FCTSolver *&fct, int ho_type, int lo_type, bool pa,
             mfem::SmoothnessIndicator *si, mfem::TimingData &timer, bool verify_bounds)
{
   static mfem::RMHContext rmh(pfes, x0, v, 1);
   ho = ho_type == 2 ? (mfem::HOSolver *)new mfem::RMHCGHOSolver(pfes, rmh) : new mfem::RMHLocalInverseHOSolver(pfes, rmh, pa);
   if (lo_type == 5) { lo = new mfem::RMHMassBasedAvg(pfes, *ho, &v, rmh); }
   else { lo = new mfem::RMHResidualDistribution(pfes, rmh, lo_type == 4); }
   fct = new mfem::RMHClipScaleSolver(pfes, si, dt, rmh);
   // remhos.cpp:1115-1116, 1550-1552
   ho->timer = &timer;
   lo->timer = &timer;
   fct->timer = &timer;
   fct->verify_bounds = verify_bounds;
   // remhos.cpp:1818-1831: LimitMult's calls on the abstract interfaces
   mfem::Vector u, du_HO, du_LO, m, lo_b, hi_b, d_u;
   mfem::ParGridFunction x_gf;
   auto mba = dynamic_cast<mfem::MassBasedAvg *>(lo);
   if (mba) { mba->SetHOSolution(du_HO); }
   lo->UpdateTimeStep(dt);
   fct->UpdateTimeStep(dt);
   ho->CalcHOSolution(u, du_HO);
   lo->CalcLOSolution(u, du_LO);
   fct->CalcFCTSolution(x_gf, m, du_HO, du_LO, lo_b, hi_b, d_u);
   mfem::Array<bool> el, dofs;
   if (!fct->NeedsLOProductInput()) { fct->CalcFCTProduct(x_gf, m, du_HO, du_LO, lo_b, hi_b, u, el, dofs, d_u); }
}
