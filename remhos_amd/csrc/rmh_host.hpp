// Host-side setup of a Remhos case on a tensor-lattice mesh: what remhos() does between
// option parsing and the time loop (remhos.cpp:442-584, 878-884), re-stated without MFEM for
// the reference's benchmark meshes, plus the box partition that stands
// in for ParMesh(comm, mesh, partitioning) (remhos.cpp:459-463).
#pragma once
#include <string>
#include <vector>

namespace remhos
{

struct CaseConfig
{
   std::string mesh = "periodic-cube"; // -m   (name of a data/*.mesh lattice)
   int rs = 1;                         // -rs
   int order = 3;                      // -o
   int problem = 10;                   // -p
   double dt = -1.0;                   // -dt  (< 0: CFL rule, remhos.cpp:538-553)
   double t_final = 0.5;               // -tf
   int max_steps = -1;                 // -ms
   int lo_type = 5;                    // -lo  (3, 4 or 5)
   int px = 1, py = 1, pz = 1;         // box partition of the element lattice
   int rank = 0;
   int rs_extra[3] = {0, 0, 0};        // additional refinement levels per direction (weak-scaling lattices)
   // Self-loop (validation of the neighbour exchange on ONE rank): 1 + d makes the periodic wrap of direction d a halo
   // -- the elements beyond the seam become ghosts whose owner is this rank itself, so that the exchange (pack kernels,
   // ghost records, RCCL send / recv to the own rank) runs for real and must reproduce the plain periodic run bit for
   // bit.  Needs a periodic mesh, one block in that direction and at least three elements across it.  0: off.
   int self_wrap = 0;
   int tile_rows = 0; // element numbering: 0 lattice order; T > 0: y-strips of T rows, z before y inside a strip (rmh_driver.h)
};

// neighbour rank in the halo exchange: which owned elements it needs, which ghost slots it fills
struct Peer
{
   int rank;
   std::vector<int> send_elems; // local indices of owned elements, ascending global id
   std::vector<int> recv_slots; // ghost slot indices (0-based within the ghost block), ascending global id
};

struct CaseData
{
   int dim = 3; // 3: hexahedra; 2: quadrilaterals (single rank: x0 / vel [ne][2][9], face_nbr [ne][4], stencil27 = 3 x 3 [ne][9])
   int order = 0, exec_mode = 0, ndof = 0;
   int ne_owned = 0, ne_ghost = 0;
   int ne_halo = 0;        // owned elements [0, ne_halo) have a ghost in their 27-stencil (ordered first)
   long long ne_global = 0;
   int n[3] = {0, 0, 0};   // global elements per direction
   int lo[3] = {0, 0, 0};  // first owned element per direction
   int nl[3] = {0, 0, 0};  // owned elements per direction
   bool periodic = false;
   double bb_min[3], bb_max[3];
   double dt = 0.0;
   std::vector<double> x0;          // [ne_owned][3][27]
   std::vector<double> vel;         // [ne_owned][3][27]
   std::vector<double> u0;          // [ne_owned][ndof]
   std::vector<double> s0;          // [ne_owned][ndof] s0_function at the same nodes (product remap, remhos.cpp:892-894)
   std::vector<double> subcell_vel; // [ne_owned][3][ndof] (lo 4 only)
   std::vector<int> face_nbr;       // [ne_owned][6]
   std::vector<int> stencil27;      // [ne_owned][27]
   std::vector<long long> owned_gid, ghost_gid;
   std::vector<Peer> peers;
};

// problem definitions (remhos.cpp:2001-2120, 2201-2355)
void velocity_function(int problem, const double *bb_min, const double *bb_max, const double x[3], double v[3]);
double u0_function(int problem, const double *bb_min, const double *bb_max, const double x[3]);
double s0_function(const double x[3]); // remhos.cpp:2357-2361

// returns an empty string on success, an error message otherwise
std::string build_case(const CaseConfig &cfg, CaseData &out);
// problem definitions for dim = 2 (the same functions of remhos.cpp restricted to two coordinates)
void velocity_function_2d(int problem, const double *bb_min, const double *bb_max, const double x[2], double v[2]);
double u0_function_2d(int problem, const double *bb_min, const double *bb_max, const double x[2]);
// Error norms the way remhos() reports them (remhos.cpp:1438-1470: ParGridFunction::ComputeLpError against the initial
// condition for the solid-body rotation, problem 4): L1, L2, L-infinity of |u_h - u_ex| by Gauss-Legendre quadrature
// of order 2 p + 3 on every element [MFEM: GridFunction::ComputeLpError], the maximum over the quadrature points.  Also
// defined here for the translation (problem 0) on the periodic meshes, against u0(x - v t) wrapped into the box
// (SURVEY.md 8(d)).  u: HOST pointer, [ne_owned][ndof]; the mesh at pseudo-time t_mesh (remap) or the static mesh;
// t_exact: the time of the exact field; err[3] = L1, L2, Linf of THIS rank's elements before the reduction -- (sum of
// |e| w, sum of e^2 w, max |e|).  Returns "" or why the problem has no exact solution here.
std::string lp_error_sums(const CaseData &d, int problem, double t_exact, const double *u, double err[3]);

// MFEM text formats of the mesh at pseudo-time t and of a DG field (host pointer); "" on success
std::string save_mfem(const CaseData &d, double t, const double *u, const char *mesh_path, const char *gf_path);

} // namespace remhos
