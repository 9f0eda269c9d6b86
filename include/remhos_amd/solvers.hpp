// C++ host side of the MI355X-native Remhos hot path.
//
// The abstract classes HOSolver / LOSolver / FCTSolver keep the reference's virtual signatures
// (remhos_ho.hpp:29-42, remhos_lo.hpp:28-44, remhos_fct.hpp:31-90) so that the concrete
// classes below can be handed to an AdvectionOperator exactly like the reference's
// LocalInverseHOSolver / MassBasedAvg / PAResidualDistributionSubcell / ClipScaleSolver
// (remhos.cpp:912-925, 927-995, 1083-1113).  They own nothing numeric: every Calc* call
// forwards to the C ABI in include/rmh.h, i.e. to a HIP kernel.
//
// Stand-ins for the MFEM types that appear in those signatures:
//   Vector              device-resident vector (mfem::Vector with a valid device pointer)
//   ParGridFunction     = Vector (DG L-vector == E-vector, remhos_lo.cpp:274)
//   SpaceLayout         what the solvers need from ParFiniteElementSpace: sizes + the rmh_ctx
// Errors abort like MFEM_VERIFY / MFEM_ABORT do (remhos_ho.cpp:86, remhos_fct.cpp:465).
#pragma once
#include "../rmh.h"

#include <cstddef>
#include <string>
#include <vector>

namespace remhos
{

typedef double real_t;

[[noreturn]] void rmh_abort(const char *msg, const char *file, int line);
#define RMH_VERIFY(cond, msg)                                   \
   do {                                                         \
      if (!(cond)) { ::remhos::rmh_abort(msg, __FILE__, __LINE__); } \
   } while (0)

// Device vector.  Read()/Write()/ReadWrite() return DEVICE pointers (mfem::Vector::Read() with a
// device configured, remhos.cpp:1679-1680); there is no implicit host mirror.
class Vector
{
   double *data = nullptr;
   int size = 0;
   bool own = false;

public:
   Vector() {}
   explicit Vector(int n);
   Vector(double *device_ptr, int n) : data(device_ptr), size(n), own(false) {}
   Vector(const Vector &o);
   Vector &operator=(const Vector &o);
   Vector &operator=(double value);
   ~Vector();
   void SetSize(int n);
   int Size() const { return size; }
   const double *Read() const { return data; }
   double *Write() { return data; }
   double *ReadWrite() { return data; }
   void CopyFromHost(const double *h);
   void CopyToHost(double *h) const;
};
typedef Vector ParGridFunction;
typedef Vector GridFunction;   // (only as the unused mesh-velocity argument of MassBasedAvg, remhos_lo.hpp:91)
class SmoothnessIndicator;     // remhos_tools.hpp:66-112: not on the hot path, only named by the FCTSolver constructors

// mfem::Array<T> as far as the FCTSolver interface needs it (the active-element / active-dof flags of product remap,
// remhos_fct.hpp:72-86): a device-resident flag array.
template <class T>
class Array
{
   T *data = nullptr;
   int size = 0;

public:
   Array() {}
   Array(T *device_ptr, int n) : data(device_ptr), size(n) {}
   int Size() const { return size; }
   const T *Read() const { return data; }
   T *Write() { return data; }
   void MakeRef(T *device_ptr, int n) { data = device_ptr; size = n; }
};

// y = a*x + b*y' helpers used by the RK solver (device axpys)
void add(const Vector &x, double a, const Vector &y, Vector &z);           // z = x + a y
void add(double a, const Vector &x, double b, const Vector &y, Vector &z); // z = a x + b y

class SpaceLayout
{
   rmh_ctx *ctx;
   int ne, ndof;
   long long global_vsize;
   bool exchange; // the context has neighbour ranks (rmh_exchange_setup): the solvers exchange what MFEM's would

public:
   SpaceLayout(rmh_ctx *c, int ne_, int ndof_, long long gvs, bool has_neighbours = false)
      : ctx(c), ne(ne_), ndof(ndof_), global_vsize(gvs), exchange(has_neighbours)
   {
   }
   rmh_ctx *Ctx() const { return ctx; }
   // ParGridFunction::ExchangeFaceNbrData (remhos_ho.cpp:122): the neighbours' face layers and element extrema of u
   void ExchangeFaceNbrData(const double *u) const;
   // the GroupCommunicator min / max of DofInfo::ComputeOverlapBounds (remhos_tools.cpp:461-466) for given extrema
   void ExchangeElementExtrema(const double *el_min, const double *el_max) const;
   int GetNE() const { return ne; }
   int GetNDofs() const { return ndof; }
   int GetVSize() const { return ne * ndof; }
   long long GlobalVSize() const { return global_vsize; }
};
typedef SpaceLayout ParFiniteElementSpace;

// remhos_tools.hpp:52-64.  The stopwatches are HIP-event buckets inside the rmh_ctx; Update()
// pulls the accumulated seconds.
struct TimingData
{
   double sw_rhs = 0, sw_L2inv = 0, sw_LO = 0, sw_FCT = 0;
   void Update(rmh_ctx *ctx);
};

// High-Order Solver (remhos_ho.hpp:29-42)
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

// remhos_ho.hpp:56-68, remhos_ho.cpp:72-128.  The reference's constructor looks at the assembly level of M: partial
// assembly -> DGMassInverse with SetAbsTol(1e-8), SetRelTol(0) (remhos_ho.cpp:79-80), otherwise the exact dense inverse of
// every element (:104-115).  Here the level is a constructor argument and selects the rule of the same element-local PCG:
//   partial_assembly = true : stop at (D^-1 r, r) <= (1e-8)^2 like DGMassInverse, then the two completion steps of
//                             rmh_set_mass_completion (one Jacobi step on the left-over residual, constant mode) -- no
//                             further mass apply, mass conserved to round-off per stage;
//   partial_assembly = false: converged to rel. 1e-14, the stand-in for the exact inverse.
class LocalInverseHOSolver : public HOSolver
{
public:
   LocalInverseHOSolver(ParFiniteElementSpace &space, bool partial_assembly); // (no default: the mass rule is the caller's statement, and it replaces the context's tolerance)
   void CalcHOSolution(const Vector &u, Vector &du) const override;
};

// remhos_ho.hpp:44-54, remhos_ho.cpp:30-70: du = M^-1 K u by Jacobi-preconditioned CG to a relative tolerance of
// 1e-12 (abs 0, at most 500 iterations).  M is block diagonal, so the element-local PCG of the HO kernel run to
// that relative tolerance PER ELEMENT satisfies the reference's global stopping test a fortiori.
class CGHOSolver : public HOSolver
{
public:
   CGHOSolver(ParFiniteElementSpace &space) : HOSolver(space) {}
   void CalcHOSolution(const Vector &u, Vector &du) const override;
};

// Low-Order Solver (remhos_lo.hpp:28-44)
class LOSolver
{
protected:
   ParFiniteElementSpace &pfes;
   real_t dt = -1.0; // usually not known at creation, updated later.

public:
   LOSolver(ParFiniteElementSpace &space) : pfes(space) {}
   virtual ~LOSolver() {}
   virtual void UpdateTimeStep(real_t dt_new) { dt = dt_new; }
   virtual void CalcLOSolution(const Vector &u, Vector &du) const = 0;
   TimingData *timer = nullptr;
};

// remhos_lo.hpp:87-109
class MassBasedAvg : public LOSolver
{
protected:
   HOSolver &ho_solver;
   const GridFunction *mesh_v;
   // Temporary HO solution, used only in the next call to CalcLOSolution().
   mutable const Vector *du_HO = nullptr;

public:
   // (mesh_vel is unused by the reference as well: SURVEY.md Appendix B.11)
   MassBasedAvg(ParFiniteElementSpace &space, HOSolver &hos, const GridFunction *mesh_vel)
      : LOSolver(space), ho_solver(hos), mesh_v(mesh_vel)
   {
   }
   void SetHOSolution(Vector &du) { du_HO = &du; }
   void CalcLOSolution(const Vector &u, Vector &du) const override;
};

// remhos_lo.hpp:111-140 (-lo 3)
class PAResidualDistribution : public LOSolver
{
public:
   PAResidualDistribution(ParFiniteElementSpace &space) : LOSolver(space) {}
   void CalcLOSolution(const Vector &u, Vector &du) const override;
};

// remhos_lo.hpp:142-171
class PAResidualDistributionSubcell : public LOSolver
{
public:
   PAResidualDistributionSubcell(ParFiniteElementSpace &space) : LOSolver(space) {}
   void CalcLOSolution(const Vector &u, Vector &du) const override;
};

// Monotone, High-order, Conservative Solver (remhos_fct.hpp:31-90)
class FCTSolver
{
protected:
   ParFiniteElementSpace &pfes;
   SmoothnessIndicator *smth_indicator;
   real_t dt;
   const bool needs_LO_input_for_products;

public:
   FCTSolver(ParFiniteElementSpace &space, SmoothnessIndicator *si, real_t dt_, bool needs_LO_prod)
      : pfes(space), smth_indicator(si), dt(dt_), needs_LO_input_for_products(needs_LO_prod)
   {
   }
   virtual ~FCTSolver() {}
   virtual void UpdateTimeStep(real_t dt_new) { dt = dt_new; }
   bool NeedsLOProductInput() const { return needs_LO_input_for_products; }
   // Calculate du that satisfies the following:
   // bounds preservation: u_min_i <= u_i + dt du_i <= u_max_i,
   // conservation:        sum m_i (u_i + dt du_ho_i) = sum m_i (u_i + dt du_i).
   virtual void CalcFCTSolution(const ParGridFunction &u, const Vector &m, const Vector &du_ho,
                                const Vector &du_lo, const Vector &u_min, const Vector &u_max,
                                Vector &du) const = 0;
   // Used in the case of product remap (remhos_fct.hpp:72-86): given the input, calculates d_us, so that
   // bounds preservation: s_min_i <= (us_i + dt d_us_i) / u_new_i <= s_max_i,
   // conservation: sum m_i (us_i + dt d_us_HO_i) = sum m_i (us_i + dt d_us_i).
   virtual void CalcFCTProduct(const ParGridFunction &us, const Vector &m, const Vector &d_us_HO, const Vector &d_us_LO,
                               Vector &s_min, Vector &s_max, const Vector &u_new, const Array<bool> &active_el,
                               const Array<bool> &active_dofs, Vector &d_us)
   {
      RMH_VERIFY(false, "Product remap is not implemented for the chosen solver");
   }
   TimingData *timer = nullptr;
   bool verify_bounds = false;
};

// remhos_fct.hpp:137-155
class ClipScaleSolver : public FCTSolver
{
public:
   ClipScaleSolver(ParFiniteElementSpace &space, SmoothnessIndicator *si, real_t dt_) : FCTSolver(space, si, dt_, false) {}
   void CalcFCTSolution(const ParGridFunction &u, const Vector &m, const Vector &du_ho, const Vector &du_lo,
                        const Vector &u_min, const Vector &u_max, Vector &du) const override;
   // remhos_fct.cpp:543-611: compatible low-order product, scaled bounds, clip + scale, empty dofs zeroed -- one kernel
   // (rmh_fct_product)
   void CalcFCTProduct(const ParGridFunction &us, const Vector &m, const Vector &d_us_HO, const Vector &d_us_LO,
                       Vector &s_min, Vector &s_max, const Vector &u_new, const Array<bool> &active_el,
                       const Array<bool> &active_dofs, Vector &d_us) override;
};

// Local bounds (remhos_tools.hpp:114-189): element extrema and overlap bounds
class DofInfo
{
   ParFiniteElementSpace &pfes;

public:
   Vector xe_min, xe_max, xi_min, xi_max;
   DofInfo(ParFiniteElementSpace &space);
   // (with the masks of product remap, remhos_tools.cpp:497-523: inactive elements get (+inf, -inf), which makes
   // ComputeBounds on the result the reference's ComputeBounds(..., &active_el))
   void ComputeElementsMinMax(const Vector &u, Vector &u_min, Vector &u_max, Array<bool> *active_el = nullptr,
                              Array<bool> *active_dof = nullptr) const;
   void ComputeBounds(const Vector &el_min, const Vector &el_max, Vector &dof_min, Vector &dof_max) const;
};

// remhos_sync.hpp: active element / dof flags and the ratio s = us / u of product remap (device kernels)
void ComputeBoolIndicators(ParFiniteElementSpace &pfes, const Vector &u, Array<bool> &ind_elem, Array<bool> &ind_dofs);
void ComputeRatio(ParFiniteElementSpace &pfes, const Vector &us, const Vector &u, Vector &s, Array<bool> &bool_el,
                  Array<bool> &bool_dof);

// -vb (remhos.cpp:1557-1594): abort with the reference's message when a dof of u_new (first overload) or of u + dt * du_new
// (second) lies outside [u_min - tol, u_max + tol]; active_dofs restricts the check like the reference's.  One streaming
// kernel (rmh_check_violation) instead of a host loop; the space is the first argument because a device vector does not know
// its context.
void check_violation(ParFiniteElementSpace &pfes, const Vector &u_new, const Vector &u_min, const Vector &u_max, std::string info,
                     double tol, const Array<bool> *active_dofs);
void check_violation(ParFiniteElementSpace &pfes, const Vector &u, double dt, const Vector &du_new, const Vector &u_min,
                     const Vector &u_max, std::string info, double tol, const Array<bool> *active_dofs);

// remhos_solvers.hpp:25-63
class LimitedTimeDependentOperator
{
protected:
   real_t dt = 0.0, t = 0.0;
   int height;

public:
   LimitedTimeDependentOperator(int n) : height(n) {}
   virtual ~LimitedTimeDependentOperator() {}
   int Height() const { return height; }
   virtual void SetDt(real_t dt_) { dt = dt_; }
   real_t GetDt() const { return dt; }
   void SetTime(real_t t_) { t = t_; }
   real_t GetTime() const { return t; }
   virtual void MultUnlimited(const Vector &x, Vector &y) const = 0;
   virtual void LimitMult(const Vector &x, Vector &y) const = 0;
   virtual void Mult(const Vector &x, Vector &y) const
   {
      MultUnlimited(x, y);
      LimitMult(x, y);
   }
};

// The stage operator (remhos.cpp:115-198, 1596-1739, 1798-1916).  One field u, or -- product remap, -ps -- the block
// vector [u | us] (remhos.cpp:875-878: block_offsets of size 3): MultUnlimited then also forms the HO rate of us
// (:1709-1738) and LimitMult limits it so that s = us / u stays in its local bounds (:1848-1915).
class AdvectionOperator : public LimitedTimeDependentOperator
{
   ParFiniteElementSpace &pfes;
   DofInfo &dofs;
   HOSolver *ho_solver;
   LOSolver *lo_solver;
   FCTSolver *fct_solver;
   mutable Vector lumpedM, du_HO, du_LO;
   mutable Vector d_us_HO, s_ratio, u_new;                                     // product remap
   mutable Array<bool> s_bool_el, s_bool_dofs, s_bool_el_new, s_bool_dofs_new; // product remap
   mutable TimingData timer;
   const bool fused;   // LimitMult through rmh_limit_fused (no du_LO / bounds vectors)
   const bool product; // block vector [u | us]

public:
   // -vb (remhos.cpp:195, 1115): LimitMult checks the LO and the limited update against the dof bounds (:1824-1837).  With
   // the fused limiter the bounds and the LO rate are formed a second time by the granular kernels for the check.
   bool verify_bounds = false;

   AdvectionOperator(ParFiniteElementSpace &space, DofInfo &dofs_, HOSolver *hos, LOSolver *los, FCTSolver *fct,
                     bool fused_limiter, bool product_sync = false);
   ~AdvectionOperator();
   void SetDt(real_t dt_) override
   {
      LimitedTimeDependentOperator::SetDt(dt_);
      if (lo_solver) { lo_solver->UpdateTimeStep(dt_); }
      if (fct_solver) { fct_solver->UpdateTimeStep(dt_); }
   }
   void MultUnlimited(const Vector &x, Vector &y) const override;
   void LimitMult(const Vector &x, Vector &y) const override;
   // -dtc 1 (remhos.cpp:166-186, 1968-1998): the running minimum lives on the device (rmh_set_dt_control);
   // the fused limiter folds it in by itself, the granular sequence calls UpdateTimeStepEstimate
   void UpdateTimeStepEstimate(const Vector &x, const Vector &dx, const Vector &x_min, const Vector &x_max) const;
   void ResetTimeStepRatio() const;
   real_t GetTimeStepRatio() const;
   TimingData &Timer() const { return timer; }
};

// mfem::ODESolver as far as the driver needs it
class ODESolver
{
public:
   virtual ~ODESolver() {}
   virtual void Init(LimitedTimeDependentOperator &op) = 0;
   virtual void Step(Vector &x, real_t &t, real_t &dt) = 0;
};

// mfem::RK3SSPSolver (remhos.cpp:490; stage times t, t+dt, t+dt/2: SURVEY A.6)
class RK3SSPSolver : public ODESolver
{
   LimitedTimeDependentOperator *f = nullptr;
   Vector y, k;

public:
   void Init(LimitedTimeDependentOperator &op) override;
   void Step(Vector &x, real_t &t, real_t &dt) override;
};

// remhos_solvers.hpp:65-84: the IDP ("invariant domain preserving") solvers -- every stage is a LIMITED forward Euler
// update; what the reference runs product remap with (-s 11 / 12 / 13).
class IDPODESolver : public ODESolver
{
protected:
   LimitedTimeDependentOperator *f = nullptr;

public:
   void Init(LimitedTimeDependentOperator &f_) override { f = &f_; }
};

// remhos_solvers.hpp:86-92, remhos_solvers.cpp:24-40
class ForwardEulerIDPSolver : public IDPODESolver
{
   Vector dx;

public:
   void Init(LimitedTimeDependentOperator &f_) override;
   void Step(Vector &x, real_t &t, real_t &dt) override;
};

// remhos_solvers.hpp:94-126: an explicit Runge-Kutta tableau run as a chain of limited forward Euler legs (PlanLegs, in
// rmh_driver.hip, derives the legs from the tableau).  The masks (UseMask / AddMasked / UpdateMask) are not built: the
// reference's driver switches them off for every run (remhos.cpp:502-507).
class RKIDPSolver : public IDPODESolver
{
public:
   struct EulerLeg
   {
      real_t from = 0., span = 0.; // pseudo-time fractions of the step: the leg starts at from and is span long
      std::vector<real_t> w;       // unlimited rate of leg i = w[i] * (HO rate at the chain's state) + sum_{k<i} w[k] * limited rate k
      bool lands = true;           // the state advances by this leg
   };
   static std::vector<EulerLeg> PlanLegs(int stages, const real_t *a_packed, const real_t *b_row, const real_t *c_abs);

private:
   const std::vector<EulerLeg> legs;
   std::vector<Vector> limited; // limited rates of the legs of the current step

public:
   RKIDPSolver(int stages, const real_t a_packed[], const real_t b_row[], const real_t c_abs[]);
   void Init(LimitedTimeDependentOperator &f_) override;
   void Step(Vector &x, real_t &t, real_t &dt) override;
};

class RK2IDPSolver : public RKIDPSolver
{
   static const real_t a[], b[], c[];

public:
   RK2IDPSolver() : RKIDPSolver(2, a, b, c) {}
};

class RK3IDPSolver : public RKIDPSolver
{
   static const real_t a[], b[], c[];

public:
   RK3IDPSolver() : RKIDPSolver(3, a, b, c) {}
};

} // namespace remhos
