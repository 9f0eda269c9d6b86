// MI355X plugins for Remhos: the ONE file a Remhos maintainer adds.  Subclasses of Remhos's own
// abstract HOSolver / LOSolver / FCTSolver (remhos_ho.hpp:29-42, remhos_lo.hpp:28-44, remhos_fct.hpp:31-90) that
// forward to the C ABI of include/rmh.h, plus the context that builds an rmh_ctx from the ParFiniteElementSpace /
// ParMesh Remhos already has.
//
// In a Remhos build:   #include "mfem.hpp", "remhos_ho.hpp", "remhos_lo.hpp", "remhos_fct.hpp" BEFORE this header and
// link -lrmh.  In this repository the header is type-checked against stub declarations of exactly the MFEM / Remhos
// names it uses (tests/mfem_stub/, tests/test_binding_compiles.py) -- MFEM itself is not in the image.
#pragma once
#include "../rmh.h"

#include <vector>

namespace mfem
{

// One rmh_ctx per rank (per GPU), built once after the mesh, the DG space and the mesh-velocity field exist
// (remhos.cpp:442-584): replaces the construction of the PA bilinear forms M_HO / K_HO and of DofInfo's H1 bounds
// space for the hot path (remhos.cpp:638-730).
struct RMHContext
{
   rmh_ctx *ctx = nullptr;

   // x0: start positions of the mesh nodes (remhos.cpp:530-531); v_gf: remap displacement (remhos.cpp:562-584) or the
   // advection velocity sampled at the nodes (transport); both in the order-2 nodal space of the mesh.
   // ghost_vertices: the corner vertex ids of the face-, edge- and vertex-neighbour elements owned by other ranks,
   // [n_ghost][8] in the element's lexicographic corner order with GLOBAL vertex ids (empty on one rank, and for dim = 2); the owned
   // elements' ids are taken from the mesh.  For periodic meshes pass the identified (periodic) vertex ids through
   // owned_vertices instead of letting the constructor read them.
   RMHContext(ParFiniteElementSpace &pfes, const GridFunction &x0, const GridFunction &v_gf, int exec_mode,
              const std::vector<int> &ghost_vertices = std::vector<int>(),
              const std::vector<int> *owned_vertices = nullptr)
   {
      Mesh &mesh = *pfes.GetMesh();
      const int dim = mesh.Dimension();
      // 3: hexahedra (the whole API); 2: quadrilaterals (HO / LO solvers and the granular limiter sequence, rmh.h "dim = 2")
      MFEM_VERIFY(dim == 3 || dim == 2, "the MI355X hot path is built for hexahedral and quadrilateral tensor-lattice meshes");
      const int ne = pfes.GetNE();
      const int nc = 1 << dim, nf = 2 * dim, nst = dim == 3 ? 27 : 9; // corners, faces, stencil entries per element
      const int n_ghost = (int)ghost_vertices.size() / nc;
      MFEM_VERIFY(dim == 3 || n_ghost == 0, "dim = 2 runs on one rank");

      // E-vectors of the nodes: [ne][dim][3^dim], node a = ax + 3 (ay + 3 az)
      const FiniteElementSpace &nfes = *x0.FESpace();
      MFEM_VERIFY(nfes.GetOrder(0) == 2 && nfes.GetVDim() == dim, "mesh nodes must be order 2 (-mo 2, remhos.cpp:222)");
      const Operator *R = nfes.GetElementRestriction(ElementDofOrdering::LEXICOGRAPHIC);
      Vector x0_e(R->Height()), v_e(R->Height());
      R->Mult(x0, x0_e);
      R->Mult(v_gf, v_e);
      // MFEM's E-vector layout is (3^dim nodes, vdim, ne) with the node index fastest: exactly [ne][dim][3^dim]
      // corner vertex ids in lexicographic order; MFEM numbers the corners of a quadrilateral, and of the bottom and the top
      // face of a hexahedron, counter-clockwise
      static const int lex_of_mfem[8] = {0, 1, 3, 2, 4, 5, 7, 6};
      std::vector<int> ev((size_t)(ne + n_ghost) * nc);
      if (owned_vertices)
      {
         MFEM_VERIFY((int)owned_vertices->size() == nc * ne, "owned_vertices must hold 2^dim ids per element");
         for (int i = 0; i < nc * ne; i++) { ev[i] = (*owned_vertices)[i]; }
      }
      else
      {
         Array<int> v;
         for (int e = 0; e < ne; e++)
         {
            mesh.GetElementVertices(e, v);
            for (int k = 0; k < nc; k++) { ev[(size_t)e * nc + lex_of_mfem[k]] = v[k]; }
         }
      }
      for (size_t i = 0; i < ghost_vertices.size(); i++) { ev[(size_t)ne * nc + i] = ghost_vertices[i]; }
      // neighbour tables for any element numbering (replaces the H1 bounds space of DofInfo, remhos_tools.cpp:355-379)
      std::vector<int> face_nbr((size_t)ne * nf), stencil((size_t)ne * nst);
      MFEM_VERIFY((dim == 3 ? rmh_build_tables(ne, ne + n_ghost, ev.data(), face_nbr.data(), stencil.data())
                            : rmh_build_tables_2d(ne, ne, ev.data(), face_nbr.data(), stencil.data())) == 0,
                  "rmh_build_tables: the mesh is not an aligned tensor lattice");

      rmh_layout L = {};
      L.dim = dim;
      L.order = pfes.GetOrder(0);
      L.mesh_order = 2;
      L.exec_mode = exec_mode;
      L.ne_owned = ne;
      L.ne_ghost = n_ghost;
      L.x0 = x0_e.HostRead();
      L.vel = v_e.HostRead();
      L.face_nbr = face_nbr.data();
      L.stencil27 = stencil.data();
      L.subcell_vel = nullptr;
      L.device = Device::GetId();
      MFEM_VERIFY(rmh_create(&L, &ctx) == 0, rmh_last_error());
   }
   ~RMHContext() { rmh_destroy(ctx); }
   RMHContext(const RMHContext &) = delete;
   RMHContext &operator=(const RMHContext &) = delete;
};

// replaces LocalInverseHOSolver (remhos_ho.hpp:56-68, remhos_ho.cpp:72-128).  Like the reference's constructor it looks at
// the assembly level: partial assembly -> DGMassInverse's stopping rule (abs 1e-8, rel 0: remhos_ho.cpp:79-80), completed by
// the two steps of rmh_set_mass_completion (Jacobi step on the left-over residual, constant mode: every stage conserves the
// mass to round-off); otherwise the element-local solve converged to rel. 1e-14, the stand-in for the exact inverse (:104-115).
class RMHLocalInverseHOSolver : public HOSolver
{
   RMHContext &rmh;

public:
   RMHLocalInverseHOSolver(ParFiniteElementSpace &space, RMHContext &c, bool partial_assembly) : HOSolver(space), rmh(c)
   {
      const int rc = partial_assembly ? (rmh_set_mass_tol(rmh.ctx, 0.0, 1e-8, 100) | rmh_set_mass_completion(rmh.ctx, 1, 1))
                                      : (rmh_set_mass_tol(rmh.ctx, 1e-14, 0.0, 100) | rmh_set_mass_completion(rmh.ctx, 0, 0));
      MFEM_VERIFY(rc == 0, rmh_last_error());
   }
   void CalcHOSolution(const Vector &u, Vector &du) const override
   {
      MFEM_VERIFY(timer, "Timer not set."); // remhos_ho.cpp:86
      MFEM_VERIFY(rmh_ho_apply(rmh.ctx, u.Read(), du.Write()) == 0, rmh_last_error());
   }
};

// replaces CGHOSolver (-ho 2, remhos_ho.hpp:44-54, remhos_ho.cpp:30-70): the same kernel at rel. tolerance 1e-12
class RMHCGHOSolver : public HOSolver
{
   RMHContext &rmh;

public:
   RMHCGHOSolver(ParFiniteElementSpace &space, RMHContext &c) : HOSolver(space), rmh(c) {}
   void CalcHOSolution(const Vector &u, Vector &du) const override
   {
      MFEM_VERIFY(timer, "Timer not set.");
      double rel, abs;
      int maxit;
      MFEM_VERIFY(rmh_get_mass_tol(rmh.ctx, &rel, &abs, &maxit) == 0, rmh_last_error());
      MFEM_VERIFY(rmh_set_mass_tol(rmh.ctx, 1e-12, 0.0, 500) == 0, rmh_last_error()); // remhos_ho.cpp:60-63
      MFEM_VERIFY(rmh_ho_apply(rmh.ctx, u.Read(), du.Write()) == 0, rmh_last_error());
      MFEM_VERIFY(rmh_set_mass_tol(rmh.ctx, rel, abs, maxit) == 0, rmh_last_error());
   }
};

// MassBasedAvg on the device (remhos_lo.hpp:87-109, remhos_lo.cpp:247-324)
class RMHMassBasedAvg : public MassBasedAvg
{
   RMHContext &rmh;

public:
   RMHMassBasedAvg(ParFiniteElementSpace &space, HOSolver &hos, const GridFunction *mesh_vel, RMHContext &c)
      : MassBasedAvg(space, hos, mesh_vel), rmh(c)
   {
   }
   void CalcLOSolution(const Vector &u, Vector &du) const override
   {
      if (du_HO)
      {
         MFEM_VERIFY(rmh_lo_massavg(rmh.ctx, u.Read(), du_HO->Read(), dt, du.Write()) == 0, rmh_last_error());
         du_HO = nullptr; // valid only until the next CalcLOSolution (remhos_lo.hpp:93-94, remhos_lo.cpp:256)
         return;
      }
      Vector du_HO_tmp(u.Size()); // remhos_lo.cpp:258-262
      ho_solver.CalcHOSolution(u, du_HO_tmp);
      MFEM_VERIFY(rmh_lo_massavg(rmh.ctx, u.Read(), du_HO_tmp.Read(), dt, du.Write()) == 0, rmh_last_error());
   }
};

// replaces PAResidualDistributionSubcell (-lo 4, remhos_lo.hpp:142-171) and PAResidualDistribution (-lo 3, :111-140)
class RMHResidualDistribution : public LOSolver
{
   RMHContext &rmh;
   const bool subcell;

public:
   RMHResidualDistribution(ParFiniteElementSpace &space, RMHContext &c, bool subcell_scheme)
      : LOSolver(space), rmh(c), subcell(subcell_scheme)
   {
   }
   void CalcLOSolution(const Vector &u, Vector &du) const override
   {
      const int rc = subcell ? rmh_lo_rdsubcell(rmh.ctx, u.Read(), du.Write()) : rmh_lo_rd(rmh.ctx, u.Read(), du.Write());
      MFEM_VERIFY(rc == 0, rmh_last_error());
   }
};

// replaces ClipScaleSolver (remhos_fct.hpp:137-155)
class RMHClipScaleSolver : public FCTSolver
{
   RMHContext &rmh;

public:
   RMHClipScaleSolver(ParFiniteElementSpace &space, SmoothnessIndicator *si, real_t dt_, RMHContext &c)
      : FCTSolver(space, si, dt_, false), rmh(c)
   {
   }
   void CalcFCTSolution(const ParGridFunction &u, const Vector &m, const Vector &du_ho, const Vector &du_lo,
                        const Vector &u_min, const Vector &u_max, Vector &du) const override
   {
      MFEM_VERIFY(rmh_fct_clipscale(rmh.ctx, u.Read(), m.Read(), du_ho.Read(), du_lo.Read(), u_min.Read(), u_max.Read(), dt,
                                    du.Write()) == 0,
                  rmh_last_error());
   }
   void CalcFCTProduct(const ParGridFunction &us, const Vector &m, const Vector &d_us_HO, const Vector &d_us_LO,
                       Vector &s_min, Vector &s_max, const Vector &u_new, const Array<bool> &active_el,
                       const Array<bool> &active_dofs, Vector &d_us) override
   {
      (void)d_us_LO;
      static_assert(sizeof(bool) == 1, "Array<bool> is handed to the kernels as a byte array");
      MFEM_VERIFY(rmh_fct_product(rmh.ctx, us.Read(), m.Read(), d_us_HO.Read(), s_min.ReadWrite(), s_max.ReadWrite(),
                                  u_new.Read(), (const unsigned char *)active_el.Read(),
                                  (const unsigned char *)active_dofs.Read(), dt, d_us.Write()) == 0,
                  rmh_last_error());
   }
};

} // namespace mfem
