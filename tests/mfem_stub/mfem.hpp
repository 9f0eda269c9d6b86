// TEST INFRASTRUCTURE: the few MFEM declarations include/remhos_amd/mfem_binding.hpp uses, so that the binding can be
// compiled (syntax + types, -fsyntax-only) in an image without MFEM (tests/test_binding_compiles.py).  Declarations
// only, written from MFEM's public interface; nothing here is linked or run.
#pragma once
#include <cstdio>
#include <cstdlib>

namespace mfem
{
typedef double real_t;
#define MFEM_VERIFY(cond, msg)                                     \
   do {                                                            \
      if (!(cond)) { std::fprintf(stderr, "%s\n", (const char *)(msg)); std::abort(); } \
   } while (0)
#define MFEM_ABORT(msg) MFEM_VERIFY(false, msg)

template <class T>
class Array
{
public:
   Array();
   Array(const Array &);
   explicit Array(int n);
   void SetSize(int n);
   int Size() const;
   T *GetData();
   const T *GetData() const;
   T &operator[](int i);
   const T &operator[](int i) const;
   const T *Read() const;
   T *Write();
};

class Vector
{
public:
   Vector();
   explicit Vector(int n);
   int Size() const;
   const real_t *Read() const;
   real_t *Write();
   real_t *ReadWrite();
   const real_t *HostRead() const;
};

class Operator
{
public:
   int Height() const;
   virtual void Mult(const Vector &x, Vector &y) const = 0;
   virtual ~Operator();
};

enum class ElementDofOrdering { NATIVE, LEXICOGRAPHIC };
enum class Ordering_Type { byNODES, byVDIM };

class Mesh
{
public:
   int GetNE() const;
   int Dimension() const;
   void GetElementVertices(int i, Array<int> &v) const;
};
class ParMesh : public Mesh
{
public:
   int GetNFaceNeighborElements() const;
};

class FiniteElementSpace
{
public:
   int GetNE() const;
   int GetVSize() const;
   int GetOrder(int i) const;
   int GetVDim() const;
   Ordering_Type GetOrdering() const;
   Mesh *GetMesh() const;
   const Operator *GetElementRestriction(ElementDofOrdering o) const;
};
class ParFiniteElementSpace : public FiniteElementSpace
{
public:
   ParMesh *GetParMesh() const;
};

class GridFunction : public Vector
{
public:
   GridFunction();
   FiniteElementSpace *FESpace() const;
};
class ParGridFunction : public GridFunction
{
public:
   ParGridFunction();
   explicit ParGridFunction(ParFiniteElementSpace *pf);
   ParFiniteElementSpace *ParFESpace() const;
   void ExchangeFaceNbrData();
   Vector &FaceNbrData();
};

class Device
{
public:
   static int GetId();
};

// ---- names that only the reference's OWN headers use (remhos_ho.hpp, remhos_lo.hpp, remhos_fct.hpp, remhos_tools.hpp), so that
// tests/test_binding_compiles.py can put those headers themselves -- read from /root/reference where it exists -- in front of
// the binding.  Complete types where the headers hold them by value or derive from them, forward declarations otherwise.
typedef int HYPRE_Int;
enum class FaceType : bool { Interior, Boundary };
struct Geometry
{
   enum Type { INVALID = -1, POINT = 0, SEGMENT, TRIANGLE, SQUARE, TETRAHEDRON, CUBE, PRISM, PYRAMID };
};
class DenseMatrix
{
public:
   DenseMatrix();
};
class DenseTensor
{
public:
   DenseTensor();
};
class SparseMatrix
{
public:
   SparseMatrix();
   SparseMatrix(const SparseMatrix &);
};
class StopWatch
{
public:
   StopWatch();
};
class socketstream;
class IntegrationRule;
class VectorCoefficient;
class FiniteElement;
class ElementTransformation;
class FaceElementTransformations;
class H1_FECollection
{
public:
   H1_FECollection(int p, int dim);
};
class ParBilinearForm
{
public:
   explicit ParBilinearForm(ParFiniteElementSpace *pf);
};
class DGMassInverse
{
public:
   ~DGMassInverse();
};
class BilinearFormIntegrator
{
public:
   virtual void AssembleElementMatrix(const FiniteElement &, ElementTransformation &, DenseMatrix &);
   virtual void AssembleElementMatrix2(const FiniteElement &, const FiniteElement &, ElementTransformation &, DenseMatrix &);
   virtual ~BilinearFormIntegrator();
};
} // namespace mfem
