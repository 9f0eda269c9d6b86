// TEST INFRASTRUCTURE: the abstract interfaces the binding derives from, restated as DECLARATIONS from the reference's
// headers (remhos_ho.hpp:29-42, remhos_lo.hpp:28-44, 87-109, remhos_fct.hpp:31-90, remhos_tools.hpp:52-64) so that
// include/remhos_amd/mfem_binding.hpp can be type-checked without the Remhos tree.  In a Remhos build the real
// remhos_ho.hpp / remhos_lo.hpp / remhos_fct.hpp are included instead.
#pragma once
#include "mfem.hpp"

namespace mfem
{
struct TimingData;
class SmoothnessIndicator;

class HOSolver
{
protected:
   ParFiniteElementSpace &pfes;

public:
   HOSolver(ParFiniteElementSpace &space) : pfes(space) {}
   virtual ~HOSolver() {}
   virtual void CalcHOSolution(const Vector &u, Vector &du) const = 0;
   TimingData *timer = nullptr;
};

class LOSolver
{
protected:
   ParFiniteElementSpace &pfes;
   real_t dt = -1.0;

public:
   LOSolver(ParFiniteElementSpace &space) : pfes(space) {}
   virtual ~LOSolver() {}
   virtual void UpdateTimeStep(real_t dt_new) { dt = dt_new; }
   virtual void CalcLOSolution(const Vector &u, Vector &du) const = 0;
   TimingData *timer = nullptr;
};

class MassBasedAvg : public LOSolver
{
protected:
   HOSolver &ho_solver;
   const GridFunction *mesh_v;
   mutable const Vector *du_HO = nullptr;

public:
   MassBasedAvg(ParFiniteElementSpace &space, HOSolver &hos, const GridFunction *mesh_vel)
      : LOSolver(space), ho_solver(hos), mesh_v(mesh_vel) {}
   void SetHOSolution(Vector &du) { du_HO = &du; }
   virtual void CalcLOSolution(const Vector &u, Vector &du) const;
};

class FCTSolver
{
protected:
   ParFiniteElementSpace &pfes;
   SmoothnessIndicator *smth_indicator;
   real_t dt;
   const bool needs_LO_input_for_products;

public:
   FCTSolver(ParFiniteElementSpace &space, SmoothnessIndicator *si, real_t dt_, bool needs_LO_prod)
      : pfes(space), smth_indicator(si), dt(dt_), needs_LO_input_for_products(needs_LO_prod) {}
   virtual ~FCTSolver() {}
   virtual void UpdateTimeStep(real_t dt_new) { dt = dt_new; }
   bool NeedsLOProductInput() const { return needs_LO_input_for_products; }
   virtual void CalcFCTSolution(const ParGridFunction &u, const Vector &m, const Vector &du_ho, const Vector &du_lo,
                                const Vector &u_min, const Vector &u_max, Vector &du) const = 0;
   virtual void CalcFCTProduct(const ParGridFunction &us, const Vector &m, const Vector &d_us_HO, const Vector &d_us_LO,
                               Vector &s_min, Vector &s_max, const Vector &u_new, const Array<bool> &active_el,
                               const Array<bool> &active_dofs, Vector &d_us);
   TimingData *timer = nullptr;
   bool verify_bounds = false;
};
} // namespace mfem
