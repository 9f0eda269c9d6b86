/*
 * rmh.h -- C ABI of the MI355X-native Remhos hot path (HO -> LO -> bounds -> FCT RK stage).
 *
 * This is the drop-in boundary.  Every entry point replaces one call the reference's
 * AdvectionOperator makes into its HOSolver / LOSolver / FCTSolver / DofInfo objects
 * (reference file:line cited per function; paths are relative to the CEED/Remhos checkout).
 * The C++ classes in include/remhos_amd/solvers.hpp keep the reference's virtual
 * signatures and forward to these functions; INTEGRATION.md shows the binding a Remhos
 * maintainer would add.
 *
 * Conventions
 *   - one rmh_ctx per GPU / per MPI rank; calls on one ctx are serialised on one HIP stream;
 *   - every vector argument of a hot-path call is a DEVICE pointer, non-owning, never aliased
 *     with an output; layout = MFEM L2 E-vector: `ndof = (p+1)^3` contiguous doubles per
 *     element, lexicographic with x fastest (remhos.cpp:588-590, remhos_lo.cpp:274);
 *   - arrays handed to rmh_create() are HOST pointers and are copied;
 *   - functions return 0 on success and a negative rmh_status otherwise (the reference aborts
 *     through MFEM_VERIFY/MFEM_ABORT instead, e.g. remhos_ho.cpp:86, remhos_fct.cpp:465);
 *   - there is no CPU fallback: without a HIP device rmh_create() fails with RMH_ERR_NO_DEVICE.
 */
#ifndef RMH_H
#define RMH_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rmh_ctx rmh_ctx;

typedef enum {
   RMH_OK = 0,
   RMH_ERR_INVALID = -1,     /* bad argument / unsupported order or dimension      */
   RMH_ERR_NO_DEVICE = -2,   /* no HIP device: the product path refuses to run     */
   RMH_ERR_HIP = -3,         /* a HIP runtime call failed (see rmh_last_error)     */
   RMH_ERR_NOT_CONVERGED = -4, /* local mass solve hit the iteration cap           */
   RMH_ERR_STATE = -5        /* call order violated (e.g. FCT before HO)           */
} rmh_status;

/* Mesh + space description of one rank ("SpaceLayout", stands in for the
 * ParFiniteElementSpace / ParMesh the reference's solvers receive: remhos_ho.hpp:32,
 * remhos_lo.hpp:31, remhos_fct.hpp:34).  Elements [0, ne_owned) are owned, elements
 * [ne_owned, ne_owned + ne_ghost) are ghosts owned by neighbour ranks (MFEM's face-neighbour
 * elements, plus edge/vertex neighbours for the bounds stencil). */
typedef struct {
   int dim;          /* 3: hexahedra (the whole API).  2: quadrilaterals -- HO solver + granular limiter sequence *
                      *    (see "dim = 2" below)                                                */
   int order;        /* polynomial order p of the Bernstein DG space (-o, remhos.cpp:261)   */
   int mesh_order;   /* order of the nodal mesh (-mo, remhos.cpp:222); 2                    */
   int exec_mode;    /* 0 transport, 1 remap (remhos.cpp:437-440)                           */
   int ne_owned;
   int ne_ghost;
   /* Mesh nodes as per-element copies (MFEM L2 nodal E-vector; for H1 meshes the caller
    * expands shared nodes), [ne_owned][3 components][27 nodes], node a = ax + 3*(ay + 3*az):
    *   x0  : start positions                                   (remhos.cpp:530-531)
    *   vel : remap: pseudo-time mesh displacement v_gf          (remhos.cpp:562-584)
    *         transport: advection velocity sampled at the nodes (remhos.cpp:534)          */
   const double *x0;
   const double *vel;
   /* face neighbours [ne_owned][6], face f = 2*c + side (c = direction, side 0: xi_c = 0,
    * side 1: xi_c = 1); value = element index in [0, ne_owned + ne_ghost) or -1 on the
    * domain boundary.  The neighbour sees the face as (c, 1 - side) with identical tangential
    * orientation (tensor-lattice meshes: SURVEY.md section 7 "Hard parts").                 */
   const int *face_nbr;
   /* 27-point element stencil for the overlap bounds [ne_owned][27], entry
    * (ox+1) + 3*(oy+1) + 9*(oz+1): element sharing the corresponding vertex/edge/face,
    * -1 if none (ComputeOverlapBounds, remhos_tools.cpp:432-495).                           */
   const int *stencil27;
   /* lo 4 only (may be NULL): sub-mesh node velocity at the closed-uniform points
    * [ne_owned][3][(p+1)^3] (v_sub_gf, remhos.cpp:837-853).                                 */
   const double *subcell_vel;
   int device;       /* HIP device ordinal */
} rmh_layout;

/* dim = 2 (quadrilateral tensor lattices; remhos_amd/csrc/rmh_2d.hpp): the reference's 2-D runs -- its ctest table
 * (remhos_tests.cpp:38-107: inline-quad, -ho 3 -lo 5 -fct 2, incl. the -pa entries and its CUDA-device entry #10) and
 * BASELINE configs[0]'s mesh family.  Layout: x0 / vel [ne][2][9] (node a = ax + 3*ay), face_nbr [ne][4] (f = 2*c + side),
 * `stencil27` = the 3 x 3 element stencil [ne][9] (entry (ox+1) + 3*(oy+1)), ne_ghost = 0, subcell_vel [ne][2][(p+1)^2] or NULL; E-vectors
 * carry (p+1)^2 doubles per element; Q = p + 2 quadrature points per direction.  Entry points: rmh_setup,
 * rmh_ho_apply, rmh_lumped_mass, rmh_compute_lumped_mass, rmh_lo_massavg, rmh_lo_rd, rmh_lo_rdsubcell, rmh_elem_minmax, rmh_bounds, rmh_fct_clipscale,
 * rmh_limit_fused, rmh_limit_fused_lo, rmh_stage_fused (the whole rank: HO kernel, RD solver for lo 3 / 4 and the fused limiter run
 * as a sequence inside the library; no tokens), the mass-rule / bounds-type / dt-control setters and getters, timers.  Everything
 * else (element ranges of a stage, product fields, exchange) returns RMH_ERR_INVALID for a 2-D context. */

/* Neighbour tables from mesh topology (host, no GPU): face_nbr and stencil27 of rmh_layout for ANY element numbering,
 * from the vertex ids of the elements -- what a binding has at hand (Mesh::GetElementVertices; periodic meshes: the
 * identified vertex ids).  elem_vertices[ne_total][8]: the 8 corner vertex ids of every element in LEXICOGRAPHIC local
 * order (corner k = kx + 2 ky + 4 kz along the element's reference axes; MFEM's hexahedron order 0..7 maps to
 * {0, 1, 3, 2, 4, 5, 7, 6}).  Elements [0, ne_owned) get table rows; elements [ne_owned, ne_total) (ghosts: the
 * face-, edge- and vertex-neighbour elements of other ranks) only appear as entries.  Two elements are neighbours at
 * stencil offset (ox, oy, oz) when the corners they share are exactly the corners of the first element's face / edge /
 * vertex in that direction -- the CG-node sharing that DofInfo::ComputeOverlapBounds reduces over
 * (remhos_tools.cpp:449-494), without the H1 bounds space.  Requires the neighbours' reference axes to be aligned
 * with the element's and at least 3 elements per periodic direction (tensor-lattice meshes in any numbering;
 * checked: RMH_ERR_INVALID otherwise).
 * face_nbr[ne_owned][6], stencil27[ne_owned][27]: outputs (caller-allocated). */
int rmh_build_tables(int ne_owned, int ne_total, const int *elem_vertices, int *face_nbr, int *stencil27);
/* The same for quadrilaterals (dim = 2): elem_vertices[ne_total][4], corner k = kx + 2 ky (MFEM's quadrilateral order 0..3 maps
 * to {0, 1, 3, 2}); outputs face_nbr[ne_owned][4] (f = 2*c + side) and the 3 x 3 stencil [ne_owned][9]. */
int rmh_build_tables_2d(int ne_owned, int ne_total, const int *elem_vertices, int *face_nbr, int *stencil9);

/* Creation / destruction.  Replaces the construction of LocalInverseHOSolver, MassBasedAvg /
 * PAResidualDistributionSubcell, ClipScaleSolver and DofInfo (remhos.cpp:730, 912-995,
 * 1083-1108). */
int  rmh_create(const rmh_layout *layout, rmh_ctx **out);
void rmh_destroy(rmh_ctx *ctx);
const char *rmh_last_error(void);
const char *rmh_version(void);

/* All work of ctx is enqueued on `hip_stream` (a hipStream_t; NULL = default stream). */
int rmh_set_stream(rmh_ctx *ctx, void *hip_stream);

/* A stream for a context whose kernels leave `reserve_cus` compute units of `device` alone.  reserve_cus = 0: an ordinary
 * NON-BLOCKING stream (hipStreamNonBlocking).  reserve_cus > 0: hipExtStreamCreateWithCUMask, which has no flags argument and
 * makes a default-flag, i.e. BLOCKING stream: work on it synchronises implicitly with the NULL stream (synchronous hipMemcpy,
 * kernels a caller launches on stream 0) -- an ordering and performance property only, results are the same; the library itself
 * launches nothing on the NULL stream after rmh_create (tests/test_gpu_stream.py asserts both flag values).  Why: the halo exchange of a partitioned stage
 * (rmh_exchange_begin, RCCL send / recv kernels on the library's exchange stream -- the reference's
 * ParGridFunction::ExchangeFaceNbrData inside K.Mult, remhos_ho.cpp:122) is launched microseconds AFTER the interior
 * stage kernel has taken every workgroup slot of the chip, and then only finds a CU as that kernel drains
 * (profiles/r04_rccl_selfloop_timeline.txt: 10 us alone, 406 us behind a full-chip launch).  With a few CUs kept free of
 * the context stream the exchange kernels start at once.  The cleared mask bits are spread so that every XCD gives up
 * the same number of CUs.  The C++ stage loops (rmhd_run_partitioned, rmhd_run_rank) create their context streams
 * through this call with reserve_cus = $RMH_COMM_CUS (default 0).  Destroy with rmh_stream_destroy. */
int rmh_stream_create_reserving(int device, int reserve_cus, void **hip_stream);
int rmh_stream_destroy(void *hip_stream);

/* Diagnostic: the batch order the one-kernel stage (rmh_stage_fused*) uses for a launch over `n_elements` elements of this
 * context.  The kernel maps its workgroups to element batches XCD-aware (each of the MI355X's eight XCDs has its own L2): where
 * the element numbering shows lattice layers -- most owned elements find their +z face neighbour the same distance e' - e away --
 * the launch is cut into chunks of 2^(*weave) layers, woven batch by batch, and the chunks are dealt round-robin to the XCDs, so
 * that z-neighbours run at the same time in one L2 or meet in the Infinity Cache; otherwise (*chunk = 0) every XCD takes one
 * contiguous eighth.  Outputs: *layer_elements (0: no layers found), *batch_elements (elements per workgroup at this order and
 * LO solver), *chunk (batches per chunk), *weave.  Results never depend on the order.  No reference
 * counterpart: MFEM's forall leaves the mapping to the back-end. */
int rmh_batch_order(rmh_ctx *ctx, int n_elements, int *layer_elements, int *batch_elements, int *chunk, int *weave);

/* Remap re-setup: AdvectionOperator::MultUnlimited moves the mesh to pseudo-time t and
 * re-assembles M_HO, K_HO and the lumped mass (remhos.cpp:1598-1637).  Here the geometry is
 * recomputed inside the kernels from x0 + t*vel (matrix-free), so this call only records t. */
int rmh_setup(rmh_ctx *ctx, double t);

/* Ghost data for multi-rank runs: device arrays holding the ghost elements' values of u
 * ([ne_ghost][ndof]; ParGridFunction::ExchangeFaceNbrData, remhos_ho.cpp:122 via
 * ParL2FaceRestriction) and ghost element extrema ([ne_ghost] each; the GroupCommunicator
 * min/max reduction of remhos_tools.cpp:461-466).  Pointers are remembered, not copied. */
int rmh_set_ghost_u(rmh_ctx *ctx, const double *u_ghost);
int rmh_set_ghost_minmax(rmh_ctx *ctx, const double *xe_min_ghost, const double *xe_max_ghost);

/* Halo pack (device): for the nsend owned elements listed in send_elems (device int array; the send
 * lists of all neighbour ranks concatenated) write their ndof values of u to rows[nsend][ndof] and their
 * min / max to out_min/out_max[nsend].  The receiver stores the rows in its ghost block and the extrema in
 * its ghost min/max arrays (one contiguous slice per neighbour rank), i.e. what
 * ParGridFunction::ExchangeFaceNbrData (remhos_ho.cpp:122) and the GroupCommunicator min/max reduction
 * (remhos_tools.cpp:461-466) deliver in the reference. */
int rmh_halo_pack(rmh_ctx *ctx, const double *u, const int *send_elems, int nsend, double *rows,
                  double *out_min, double *out_max);

/* The same exchange with ONE message per neighbour rank: a ghost record is [ndof values of u | min | max]
 * (ndof + 2 doubles).  rmh_halo_pack_records writes the records of the send list to rec[nsend][ndof + 2];
 * the receiver's ghost block is rec[ne_ghost][ndof + 2], announced with rmh_set_ghost_records (replaces
 * rmh_set_ghost_u + rmh_set_ghost_minmax). */
int rmh_halo_pack_records(rmh_ctx *ctx, const double *u, const int *send_elems, int nsend, double *rec);
int rmh_set_ghost_records(rmh_ctx *ctx, const double *rec);

/* ---- Neighbour exchange inside the library: RCCL send/recv over xGMI -------------------------------------------
 * Replaces, per RK stage, ParGridFunction::ExchangeFaceNbrData (remhos_ho.cpp:122; again in LimitMult,
 * remhos.cpp:1812-1813) and the GroupCommunicator min/max reduction of DofInfo::ComputeOverlapBounds
 * (remhos_tools.cpp:449-466) by ONE grouped ncclSend/ncclRecv exchange with the neighbour ranks of the box
 * partition (what ParMesh(comm, mesh, partitioning) sets up in the reference, remhos.cpp:459-463).
 *
 * rmh_exchange_setup takes the plan (HOST arrays, copied): for neighbour k, the owned elements it needs
 * (send_elems[k], in the order of ITS ghost slots) and the contiguous range of this rank's ghost slots it fills.
 * The library owns the send and ghost buffers from then on (rmh_set_ghost_* must not be used).
 *   compact = 0: a ghost record is [ndof values | min | max] (every neighbour element in full);
 *   compact = 1: a ghost record is [min | max | D^2 face layer facing this rank] for elements that share a face
 *                with this rank and [min | max] for edge / vertex neighbours (only the bounds stencil reads those):
 *                the payload SURVEY.md 8(e) asks for.  The face_nbr / stencil27 tables on the device are re-indexed
 *                to the variable-size records.  Fails with RMH_ERR_INVALID when an element is adjacent to the same
 *                neighbour rank through two faces (blocks one element thin): use compact = 0 on all ranks then.
 * Transports:
 *   rmh_comm_init / rmh_comm_attach -- RCCL: a communicator created from a broadcast ncclUniqueId
 *       (rmh_comm_unique_id on one rank), or the application's own ncclComm_t; the exchange runs on a stream of
 *       its own, ordered against the context's stream by events (no host synchronisation);
 *   rmh_comm_connect_local          -- neighbour k is another context of THIS process (several blocks on one GPU,
 *       or one process driving several GPUs; peer_ctx_index = this context's index among ITS neighbours):
 *       device-to-device copies instead of RCCL, same plan and buffers;
 *   neither                         -- the caller moves the bytes itself between rmh_exchange_begin and
 *       rmh_exchange_end (rmh_exchange_buffers / rmh_exchange_peer describe the segments; used with
 *       torch.distributed/gloo in the CPU tests).
 * rmh_exchange_begin(u): packs the send records of u on the context's stream and posts the exchange;
 * rmh_exchange_end: makes the context's stream wait for the received ghosts.  Kernels launched in between must
 * not read ghosts (rmh_stage_fused_range over the elements that reach none).  With local peers every context of
 * the process calls begin before any calls end. */
typedef struct {
   int n_peers;
   const int *peer_rank;         /* [n_peers] rank of neighbour k in the communicator                           */
   const int *send_count;        /* [n_peers]                                                                   */
   const int *const *send_elems; /* [n_peers][send_count[k]] owned element indices, in the receiver's ghost order */
   const int *recv_first;        /* [n_peers] first ghost slot (0-based within the ghost block) neighbour k fills */
   const int *recv_count;        /* [n_peers] number of consecutive ghost slots it fills                         */
} rmh_exchange_desc;
int rmh_exchange_setup(rmh_ctx *ctx, const rmh_exchange_desc *desc, int compact);
int rmh_comm_unique_id(char id[128]);
int rmh_comm_init(rmh_ctx *ctx, const char id[128], int nranks, int rank);
int rmh_comm_attach(rmh_ctx *ctx, void *nccl_comm);
int rmh_comm_connect_local(rmh_ctx *ctx, int peer_index, rmh_ctx *peer_ctx, int peer_ctx_index);
int rmh_exchange_begin(rmh_ctx *ctx, const double *u);
int rmh_exchange_end(rmh_ctx *ctx);
/* The same neighbours exchange CALLER-GIVEN element extrema (device arrays [ne_owned]): the ghost extrema -- what rmh_bounds
 * reads beside its arguments -- become the neighbours' values of xe_min / xe_max for the elements of the send lists.  This is
 * the GroupCommunicator min / max reduction of DofInfo::ComputeOverlapBounds (remhos_tools.cpp:461-466) for a field other
 * than the u whose extrema travel with rmh_exchange_begin: the masked extrema of s = us / u in product remap
 * (remhos.cpp:1883-1886; (+inf, -inf) of inactive elements are carried as they are).  RCCL or same-process neighbours; no
 * exchange of u may be open; with same-process neighbours every context calls begin before any calls end. */
int rmh_exchange_minmax_begin(rmh_ctx *ctx, const double *xe_min, const double *xe_max);
int rmh_exchange_minmax_end(rmh_ctx *ctx);
/* segments of the library-owned buffers (device pointers; counts in doubles) for a caller-side transport */
int rmh_exchange_buffers(rmh_ctx *ctx, double **send_buf, long long *send_doubles, double **ghost_buf, long long *ghost_doubles);
int rmh_exchange_peer(rmh_ctx *ctx, int peer_index, int *rank, long long *send_offset, long long *send_doubles,
                      long long *recv_offset, long long *recv_doubles);
/* Reductions over the ranks of the RCCL communicator for the driver's report (remhos.cpp:1412-1421 mass / max,
 * :1934 stopwatch maxima, :1993 dt estimate): vals[n] HOST doubles, in place; op 0 = sum, 1 = min, 2 = max.
 * Synchronises.  Not on the hot path. */
int rmh_allreduce(rmh_ctx *ctx, double *vals, int n, int op);
/* Number of ranks of the context's RCCL communicator as the communicator itself reports it (ncclCommCount; MPI_Comm_size
 * of the reference's pmesh.GetComm(), remhos.cpp:459-463); 0 without one. */
int rmh_comm_count(rmh_ctx *ctx, int *nranks);

/* HOSolver::CalcHOSolution (remhos_ho.hpp:38, LocalInverseHOSolver remhos_ho.cpp:84-129):
 * du = M^-1 (K_vol + K_face) u with an element-local, tightly converged mass solve.
 * Also refreshes the lumped mass vector (remhos.cpp:1632) and the element extrema of u. */
int rmh_ho_apply(rmh_ctx *ctx, const double *u, double *du);

/* Lumped mass M_HO * 1 at the pseudo-time of the last rmh_setup (remhos.cpp:719-727, 1625-1632).
 * rmh_lumped_mass returns the ctx-owned device vector written by rmh_ho_apply;
 * rmh_compute_lumped_mass evaluates it on its own (initial / final mass, remhos.cpp:1394-1403). */
const double *rmh_lumped_mass(rmh_ctx *ctx);
int rmh_compute_lumped_mass(rmh_ctx *ctx, double t, double *m);

/* LOSolver::CalcLOSolution:
 *   MassBasedAvg (lo 5, remhos_lo.cpp:247-324; du_ho is what SetHOSolution hands over,
 *   remhos_lo.hpp:102);  PAResidualDistributionSubcell (lo 4, remhos_lo.cpp:1620-1802). */
int rmh_lo_massavg(rmh_ctx *ctx, const double *u, const double *du_ho, double dt, double *du_lo);
int rmh_lo_rdsubcell(rmh_ctx *ctx, const double *u, double *du_lo);

/* LOSolver::CalcLOSolution of PAResidualDistribution (-lo 3, remhos_lo.hpp:111-140, remhos_lo.cpp:965-1034):
 * the same element residual distribution and lumped upwind face fluxes without the subcell fluctuations
 * (no rmh_layout.subcell_vel needed).  Same kernel as rmh_lo_rdsubcell; orders >= 2. */
int rmh_lo_rd(rmh_ctx *ctx, const double *u, double *du_lo);

/* DofInfo::ComputeElementsMinMax (remhos_tools.cpp:497-523) and DofInfo::ComputeBounds ->
 * ComputeOverlapBounds (remhos_tools.cpp:432-495).  rmh_bounds uses the element extrema of
 * the last rmh_elem_minmax / rmh_ho_apply call plus the ghost extrema. */
int rmh_elem_minmax(rmh_ctx *ctx, const double *u, double *xe_min, double *xe_max);
int rmh_bounds(rmh_ctx *ctx, const double *xe_min, const double *xe_max,
               double *u_min, double *u_max);

/* FCTSolver::CalcFCTSolution, ClipScaleSolver (remhos_fct.hpp:83-86, remhos_fct.cpp:449-541). */
int rmh_fct_clipscale(rmh_ctx *ctx, const double *u, const double *m,
                      const double *du_ho, const double *du_lo,
                      const double *u_min, const double *u_max, double dt, double *du);

/* ---- Product-field remap (-ps; second block of AdvectionOperator::LimitMult, remhos.cpp:1848-1915) ----------------
 * Flags are device byte arrays (mfem::Array<bool>): active_el[ne], active_dofs[ne * ndof].
 * rmh_product_ratio: ComputeBoolIndicators (remhos_sync.cpp:23-47: u > EMPTY_ZONE_TOL = 1e-12) and, when us and s are
 *   given, ComputeRatio (remhos_sync.cpp:50-96): s = us / u on active dofs, the element's mean active ratio on its
 *   other dofs, 0 in empty elements.
 * rmh_elem_minmax_masked: DofInfo::ComputeElementsMinMax(s, xe_min, xe_max, &active_el, &active_dofs)
 *   (remhos_tools.cpp:497-523); empty elements get (+inf, -inf), so that rmh_bounds on the result IS
 *   DofInfo::ComputeBounds(..., &active_el) (remhos_tools.cpp:432-495: inactive elements do not affect the bounds).
 * rmh_fct_product: ClipScaleSolver::CalcFCTProduct (remhos_fct.cpp:543-566) = CalcCompatibleLOProduct (:26-115; s_min /
 *   s_max are updated in place like the reference's) + ScaleProductBounds (:117-153) + ClipScale (:449-541) +
 *   ZeroOutEmptyDofs (remhos_sync.cpp:98-116) in one kernel.  d_us_HO comes from rmh_ho_apply on us.
 * All three work element by element; across ranks the bounds of s need the neighbours' masked extrema:
 * rmh_exchange_minmax_begin / _end on the output of rmh_elem_minmax_masked, then rmh_bounds (what DofInfo::ComputeBounds of
 * include/remhos_amd/solvers.hpp does). */
int rmh_product_ratio(rmh_ctx *ctx, const double *us, const double *u, double *s, unsigned char *active_el,
                      unsigned char *active_dofs);
int rmh_elem_minmax_masked(rmh_ctx *ctx, const double *u, const unsigned char *active_el, const unsigned char *active_dofs,
                           double *xe_min, double *xe_max);
int rmh_fct_product(rmh_ctx *ctx, const double *us, const double *m, const double *d_us_ho, double *s_min, double *s_max,
                    const double *u_new, const unsigned char *active_el, const unsigned char *active_dofs, double dt,
                    double *d_us);

/* Fused LimitMult for -lo 5 -fct 2 (remhos.cpp:1798-1845): mass-based average, overlap bounds
 * and clip+scale in one pass over the element, nothing but du written.  Uses the lumped mass
 * and element extrema left by rmh_ho_apply on the same u.  If y_out != NULL the RK update
 *   y_out = a * x_base + b * (u + dt_rk * du)      (RK3SSPSolver::Step [MFEM])
 * is applied as well and du may be NULL. */
int rmh_limit_fused(rmh_ctx *ctx, const double *u, const double *du_ho, double dt, double *du,
                    const double *x_base, double a, double b, double dt_rk, double *y_out);

/* Same limiter pass for an LO rate that was computed by another LOSolver (lo 4: rmh_lo_rdsubcell):
 * overlap bounds + ClipScale (+ RK update) without materialising u_min / u_max. */
int rmh_limit_fused_lo(rmh_ctx *ctx, const double *u, const double *du_ho, const double *du_lo, double dt,
                       double *du, const double *x_base, double a, double b, double dt_rk, double *y_out);

/* The whole RK stage in ONE kernel for -ho 3 -lo 5|4 -fct 2 (LO solver: rmh_set_lo_type): AdvectionOperator::Mult = MultUnlimited +
 * LimitMult (remhos_solvers.hpp:46-50, remhos.cpp:1596-1916) and the RK vector update
 *   y_out = a * x_base + b * (u + dt_rk * du)          (x_base may be NULL: a is ignored)
 * du_HO, du_LO, the lumped mass and the per-dof bounds never leave the compute unit; the element extrema
 * of y_out are kept for the next stage (rmh_stage_fused_chain).  du (may be NULL) receives the limited rate.
 * y_out must not alias u (other workgroups still read neighbour traces of u); it may alias x_base.
 * Ghost values of u and ghost extrema must be set for multi-rank runs. */
int rmh_stage_fused(rmh_ctx *ctx, const double *u, double dt, const double *x_base, double a, double b,
                    double dt_rk, double *y_out, double *du);

/* The same stage for the owned elements [e_begin, e_end) only, so that a multi-rank caller can run
 * the elements that touch no ghost while the halo exchange (ParGridFunction::ExchangeFaceNbrData,
 * remhos_ho.cpp:122, and the min/max GroupCommunicator of remhos_tools.cpp:461-466) is in flight, and
 * the halo-dependent ones after it.  All ranges of one stage take the same arguments; finish != 0 on
 * the last one (the element extrema of y_out then become the context's, see rmh_stage_fused_chain). */
int rmh_stage_fused_range(rmh_ctx *ctx, const double *u, double dt, const double *x_base, double a, double b,
                          double dt_rk, double *y_out, double *du, int e_begin, int e_end, int finish);

/* Chained stages.  A finished stage leaves the element extrema of y_out in the context and names them with a token
 * (never 0).  The next stage needs the extrema of ITS input for the bounds: presenting that token (in_token) is the
 * caller's statement "u is the untouched y_out of the stage that returned this token" and saves the streaming pass that
 * recomputes them; in_token = 0, a stale token, or any other entry point called in between costs that pass and nothing
 * else -- there is no way to get bounds from extrema that do not belong to u.  All range calls of one stage pass the same
 * in_token; *out_token (may be NULL) is set by the finishing call (0 before).  rmh_stage_fused / rmh_stage_fused_range
 * are this call with in_token = 0.  rmh_invalidate_extrema drops the context's token (kept for callers of the earlier,
 * pointer-keyed interface; not needed with tokens: a caller that modified the vector simply does not present one). */
int rmh_stage_fused_chain(rmh_ctx *ctx, const double *u, double dt, const double *x_base, double a, double b, double dt_rk,
                          double *y_out, double *du, int e_begin, int e_end, int finish, unsigned long long in_token,
                          unsigned long long *out_token);
int rmh_invalidate_extrema(rmh_ctx *ctx);

/* Which LOSolver rmh_stage_fused runs inside the stage kernel: 5 = MassBasedAvg (default), 4 =
 * PAResidualDistributionSubcell, 3 = PAResidualDistribution (-lo, remhos.cpp:268-276); lo 4 needs
 * rmh_layout.subcell_vel. */
int rmh_set_lo_type(rmh_ctx *ctx, int lo_type);

/* DofInfo bounds type (-bt, remhos.cpp:289; DofInfo::ComputeBounds, remhos_tools.hpp:168-182) used by rmh_bounds,
 * rmh_limit_fused, rmh_limit_fused_lo and rmh_stage_fused: 0 = overlap bounds (default), 1 = the element and its
 * face neighbours, one interval per element (ComputeMatrixSparsityBounds, remhos_tools.cpp:381-430). */
int rmh_set_bounds_type(rmh_ctx *ctx, int bounds_type);

/* Time step control -dtc 1 (TimeStepControl::LOBoundsError; requires bounds type 1, remhos.cpp:617-620).
 * While on, every limiter pass (rmh_limit_fused, rmh_limit_fused_lo, rmh_stage_fused) also folds
 * AdvectionOperator::UpdateTimeStepEstimate(u, du_LO, u_min, u_max) (remhos.cpp:1968-1998, called at :1839-1842)
 * into a device scalar: the largest dt with u_min <= u + dt * du_LO <= u_max over all dofs seen since
 * rmh_dt_estimate_reset (AdvectionOperator::SetDt resets it before each step, remhos.cpp:176-182).
 * rmh_dt_estimate_update does the same for the granular call sequence; rmh_dt_estimate_get synchronises and
 * returns the rank-local minimum (the caller min-reduces over ranks, as remhos.cpp:1993 does). */
int rmh_set_dt_control(rmh_ctx *ctx, int on);
int rmh_dt_estimate_reset(rmh_ctx *ctx);
int rmh_dt_estimate_update(rmh_ctx *ctx, const double *x, const double *dx, const double *x_min, const double *x_max);
int rmh_dt_estimate_get(rmh_ctx *ctx, double *dt);

/* check_violation (-vb, remhos.cpp:1557-1594; called around the limiter at remhos.cpp:1824-1837 and at the end of
 * FCTSolver::CalcFCTProduct, remhos_fct.cpp:568-610): the reference's only in-loop correctness guard on this path.  One
 * streaming pass over the rank's dofs tests
 *      u_new + tol < lo   ||   u_new > hi + tol,      u_new = u + dt * du   (du == NULL: u_new = u, the first overload)
 * with (lo, hi) = (u_min, u_max), or (u_min * bound_scale, u_max * bound_scale) when bound_scale != NULL -- the bounds
 * ScaleProductBounds forms from (s_min, s_max) and u_new for the product field (remhos_fct.cpp:117-153); active_dofs (device
 * bytes, may be NULL) selects the dofs like the reference's Array<bool>.  The reference aborts at the first violating dof and
 * prints its index and three values; here the verdict comes back (synchronises the stream) and the caller decides --
 * remhos::check_violation of include/remhos_amd/solvers.hpp aborts with the reference's message.
 *   count: violating dofs; first: the smallest violating index (-1: none) and u_min / u_new / u_max there;
 *   over / under: the largest u_new - hi and lo - u_new among the violating dofs (0 if none). */
typedef struct {
   long long count, first;
   double over, under;
   double first_min, first_value, first_max;
} rmh_violation;
int rmh_check_violation(rmh_ctx *ctx, const double *u, double dt, const double *du, const double *u_min, const double *u_max,
                        const double *bound_scale, double tol, const unsigned char *active_dofs, rmh_violation *out);

/* Stopwatch buckets of TimingData (remhos_tools.hpp:52-64; printed by
 * AdvectionOperator::PrintTimingData, remhos.cpp:1918-1966): seconds in
 * t[0]=RHS (K u), t[1]=L2inv (mass solve), t[2]=LO, t[3]=FCT since the last reset, measured
 * with HIP events on the ctx stream.  RHS and L2inv run in one kernel here, so t[0] holds the
 * fused HO kernel and t[1] is 0. Synchronises the stream. */
int rmh_timers(rmh_ctx *ctx, double t[4]);
int rmh_reset_timers(rmh_ctx *ctx);
/* enable/disable event timing (off by default: events between kernels cost launch gaps) */
int rmh_enable_timers(rmh_ctx *ctx, int on);

/* Diagnostics: max PCG iterations of the last rmh_ho_apply (synchronises). */
int rmh_last_cg_iters(rmh_ctx *ctx, int *max_iters);
/* Local mass solve controls (DGMassInverse::SetRelTol/SetAbsTol/SetMaxIter, remhos_ho.cpp:79-80).
 * Default: rel_tol 1e-14, abs_tol 0, max_iter 100 -- see DESIGN.md for why this is tighter than
 * the reference's abs 1e-8. */
int rmh_set_mass_tol(rmh_ctx *ctx, double rel_tol, double abs_tol, int max_iter);
int rmh_get_mass_tol(rmh_ctx *ctx, double *rel_tol, double *abs_tol, int *max_iter);
/* Completion of the local mass solve (no counterpart in the reference: DGMassInverse::Mult, remhos_ho.cpp:126, returns
 * whatever its stopping rule leaves).  Two steps behind the PCG loop, neither needs another mass apply:
 *   jacobi_step   = 1: x += D^-1 r with the residual r the PCG recurrence leaves (D: the Jacobi diagonal it preconditions
 *                      with).  On the nearly affine elements of a refined mesh D^-1 M = I + O(h): about one more order
 *                      of accuracy for free; a converged solve is not changed.
 *   constant_mode = 1: du_HO += (1^T b - sum_i m_i du_HO,i) / |element|.  Both bases sum to one, so 1^T b is the exact
 *                      integral of M^-1 b over the element and sum m du_HO that of the computed one: every stage then
 *                      conserves the mass to round-off for ANY stopping rule (rmh_set_mass_tol).
 * Default: both off (the solve is converged to rel_tol 1e-14 instead). */
int rmh_set_mass_completion(rmh_ctx *ctx, int jacobi_step, int constant_mode);

#ifdef __cplusplus
}
#endif
#endif /* RMH_H */
