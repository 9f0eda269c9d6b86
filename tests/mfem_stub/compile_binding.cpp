// compile-only: the binding header against the stub declarations (tests/test_binding_compiles.py)
#include "remhos_interfaces.hpp"

#include "remhos_amd/mfem_binding.hpp"

// what the solver factory of remhos.cpp:912-995, 1083-1108 would do with the plugins
void factory(mfem::ParFiniteElementSpace &pfes, const mfem::GridFunction &x0, const mfem::GridFunction &v, mfem::real_t dt,
             mfem::HOSolver *&ho, mfem::LOSolver *&lo, mfem::FCTSolver *&fct, int ho_type, int lo_type, bool pa)
{
   static mfem::RMHContext rmh(pfes, x0, v, 1);
   ho = ho_type == 2 ? (mfem::HOSolver *)new mfem::RMHCGHOSolver(pfes, rmh) : new mfem::RMHLocalInverseHOSolver(pfes, rmh, pa); // pa: remhos.cpp's -pa option (remhos_ho.cpp:72-82)
   if (lo_type == 5) { lo = new mfem::RMHMassBasedAvg(pfes, *ho, &v, rmh); }
   else { lo = new mfem::RMHResidualDistribution(pfes, rmh, lo_type == 4); }
   fct = new mfem::RMHClipScaleSolver(pfes, nullptr, dt, rmh);
}
