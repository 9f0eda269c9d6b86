/*
 * rmh_driver.h -- C ABI of the host-side harness around the hot path: case setup on the
 * reference's lattice meshes (what remhos() does before its time loop, remhos.cpp:442-584,
 * 878-884) with a box partition standing in for ParMesh (remhos.cpp:459-463), and a
 * single-GPU restatement of the time loop (remhos.cpp:1146-1330) + final report
 * (remhos.cpp:1340-1436) driving the HOSolver / LOSolver / FCTSolver classes of
 * include/remhos_amd/solvers.hpp.
 *
 * The rmhd_case_* functions are pure host code (also built into librmh_host.so for the CPU
 * tests); rmhd_run needs the GPU.
 */
#ifndef RMH_DRIVER_H
#define RMH_DRIVER_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
   char mesh[32];     /* -m : "periodic-cube" | "cube01_hex" (data/ lattice meshes)             */
   int rs;            /* -rs                                                                  */
   int order;         /* -o                                                                   */
   int problem;       /* -p  (0 translation transport, 10 Taylor-Green remap, ...)            */
   double dt;         /* -dt (< 0: CFL rule remhos.cpp:538-553)                               */
   double t_final;    /* -tf                                                                  */
   int max_steps;     /* -ms (< 0: none)                                                      */
   int lo_type;       /* -lo : 3 RD, 4 subcell RD, 5 mass-based average                       */
   int fused;         /* 1: LimitMult through rmh_limit_fused, 0: the reference's call sequence */
   int px, py, pz;    /* box partition of the element lattice                                 */
   int rank;          /* which block this process owns                                        */
   int bounds_type;   /* -bt : 0 overlap bounds, 1 face-neighbour bounds                        */
   int dt_control;    /* -dtc: 0 fixed dt, 1 LOBoundsError (needs -bt 1; remhos.cpp:1178-1197)  */
   int ho_type;       /* -ho : 3 local inverse (0 means 3), 2 CG to rel. tolerance 1e-12        */
   int save;          /* -save: write meshHO_init/final.mesh and sltn_init/final.gf (cwd)        */
   int rs_extra[3];   /* additional uniform refinements per direction (0: the reference's meshes;
                         bench.py's weak-scaling lattices refine the partitioned directions once more) */
   int pa;            /* -pa : which local mass solve -ho 3 stands for (remhos_ho.cpp:72-82, 90-128):
                         0 full assembly -- the exact element inverse (here: PCG converged to rel. 1e-14);
                         1 partial assembly -- DGMassInverse with abs. tolerance 1e-8, rel. 0 (here: the same stopping
                           rule, completed by rmh_set_mass_completion(1, 1))                                        */
   int self_wrap;     /* validation of the neighbour exchange on ONE rank: 1 + d turns the periodic wrap of direction d
                         into a halo whose ghosts are owned by this rank itself (RCCL send / recv to the own rank); the
                         run must reproduce the plain periodic one bit for bit.  0: off                               */
   int warmup_steps;  /* rmhd_run_partitioned: steps taken before the stopwatches and the wall clock start (bench.py)  */
   int ps;            /* -ps : product-field remap, (u, us) evolved together with s = us / u kept in its local bounds
                         (remhos.cpp:888-904, 1709-1738, 1848-1915); remap mode, -fct 2, fixed dt, an IDP solver        */
   int ode_solver;    /* -s  : 3 (or 0) RK3 SSP; 11 / 12 / 13 forward Euler / RK2 / RK3 IDP solvers
                         (remhos_solvers.cpp; what -ps runs with in the reference's tests)                              */
   int tile_rows;     /* element NUMBERING of the case builder (dim = 3): 0 = the lattice order x, y, z; T > 0 = strips of T
                         lattice rows in y, and inside a strip z before y -- the elements a stage kernel has in flight at a
                         time then hold each other's face neighbours in all three directions (L2 hits instead of HBM reads of
                         the neighbour traces, extrema and shared face-table blocks).  The same mesh and the same results,
                         element for element (owned_gid maps the numbering); halo elements still come first.               */
   int verify_bounds; /* -vb : the reference's debug guards (remhos.cpp:324): check_violation on the LO and the limited update of
                         every stage (remhos.cpp:1824-1837), on the limited product field (remhos_fct.cpp:568-610), and the
                         monotonicity check of the global extrema at the top of every step (remhos.cpp:1218-1262).  A violation
                         prints the reference's message and aborts.  rmhd_run / rmhd_run_rank; costs the granular bounds + LO
                         kernels beside a fused stage.                                                                       */
} rmhd_config;

typedef struct {
   int order, exec_mode, ndof, ne_owned, ne_ghost, n_peers;
   int ne_halo;       /* owned elements [0, ne_halo) reach a ghost through their 27-stencil */
   int dim;           /* 3: hexahedra; 2: quadrilaterals (inline-quad, periodic-square: one rank; arrays [ne][2][9], 4 faces, 3 x 3 stencil) */
   long long ne_global;
   int n[3], lo[3], nl[3];
   double dt;
   double bb_min[3], bb_max[3];
} rmhd_case_info;

typedef struct rmhd_case rmhd_case;

/* returns NULL on error (message in rmhd_last_error) */
rmhd_case *rmhd_case_create(const rmhd_config *cfg);
void rmhd_case_destroy(rmhd_case *c);
const char *rmhd_last_error(void);
int rmhd_case_get_info(const rmhd_case *c, rmhd_case_info *info);
const double *rmhd_case_x0(const rmhd_case *c);          /* [ne_owned][3][27] */
const double *rmhd_case_vel(const rmhd_case *c);         /* [ne_owned][3][27] */
const double *rmhd_case_u0(const rmhd_case *c);          /* [ne_owned][ndof]  */
const double *rmhd_case_s0(const rmhd_case *c);          /* [ne_owned][ndof] s0_function at the nodes (-ps, remhos.cpp:892-894) */
const double *rmhd_case_subcell_vel(const rmhd_case *c); /* [ne_owned][3][ndof] or NULL */
const int *rmhd_case_face_nbr(const rmhd_case *c);       /* [ne_owned][6]  */
const int *rmhd_case_stencil27(const rmhd_case *c);      /* [ne_owned][27] */
const long long *rmhd_case_owned_gid(const rmhd_case *c);
const long long *rmhd_case_ghost_gid(const rmhd_case *c);
/* k-th neighbour rank of the halo exchange: owned elements it needs / ghost slots it fills */
int rmhd_case_peer(const rmhd_case *c, int k, int *rank, int *nsend, const int **send_elems,
                   int *nrecv, const int **recv_slots);

/* -save (remhos.cpp:1015-1030, 1365-1380): write the mesh at pseudo-time t (remap: x0 + t * v) as an "MFEM mesh
 * v1.0" file with L2 Gauss-Lobatto order-2 nodes and, when u (HOST pointer, [ne][ndof]) and gf_path are given, the
 * field as an MFEM GridFunction file (L2_T2_3D_P<order>, the positive basis of remhos.cpp:588-590).  Elements are
 * written in lattice order (global id).  Single-rank cases only (PrintAsOne / SaveAsOne).  0 on success. */
int rmhd_case_save(const rmhd_case *c, double t, const double *u, const char *mesh_path, const char *gf_path);

typedef struct {
   double final_mass, max_value, mass0, mass_loss; /* remhos.cpp:1423-1428 */
   double dt, t_end;
   int steps, stages;
   long long global_dofs;
   double t_rhs, t_inv, t_lo, t_fct, t_total; /* TimingData buckets, remhos.cpp:1928-1933 */
   double fom_rhs, fom_inv, fom_lo, fom_fct, fom; /* remhos.cpp:1947-1951 (fom omits INV) */
   double wall, fom_wall;                     /* whole stage loop, everything included    */
   int cg_iters_max;
   int repeats;                               /* steps repeated by the dt control         */
   /* rmhd_run_partitioned: */
   int timed_stages;                          /* stages inside the stopwatches / the wall clock (all but the warm-up) */
   int n_peers;                               /* neighbour ranks of this rank's block                                  */
   int transport;                             /* 0 none, 1 RCCL send/recv, 2 same-process device copies               */
   int comm_ranks;                            /* ranks of the RCCL communicator as it reports them (ncclCommCount); 0: none */
   long long send_bytes_per_stage, recv_bytes_per_stage; /* this rank's halo records per RK stage                     */
   /* -ps (remhos.cpp:1404, 1416-1434): */
   double final_mass_us, mass0_us, mass_loss_us, s_max;
   /* error norms against the exact field where one exists (remhos.cpp:1438-1470, ComputeLpError: quadrature of order
    * 2 p + 3, L-infinity over the quadrature points): problem 4 (rotation) against the initial condition like the
    * reference, problem 0 on the periodic meshes against u0(x - v t) wrapped into the box; has_errors = 0 otherwise */
   int has_errors, pad2_;
   double err_l1, err_l2, err_linf;
   /* rmhd_run_partitioned: the TimingData buckets above (t_rhs ... fom) are SAMPLED there -- HIP events around the launches of
    * every timer_every-th step only (RMH_DRIVER_TIMERS, default 4; events around every launch cost small blocks 8 %), scaled by
    * steps / timer_steps; wall and fom_wall are exact.  0 / 0: every step is timed (rmhd_run, rmhd_run_rank). */
   int timer_every, timer_steps;
} rmhd_result;

/* remhos() on one GPU (px = py = pz = 1): setup, time loop (RK3 SSP, or the IDP solvers -s 11 / 12 / 13; with -ps
 * the product field us beside u), report.  0 on success.
 * rmhd_run_state also hands back the final fields (HOST arrays of ne * ndof doubles in the case's element order, may be
 * NULL; us_final only with -ps) -- what -save writes, for parity checks of whole runs. */
int rmhd_run(const rmhd_config *cfg, rmhd_result *res);
int rmhd_run_state(const rmhd_config *cfg, rmhd_result *res, double *u_final, double *us_final);
/* The same loop for ONE BLOCK of a px x py x pz partition, one process per block (mpirun -np N ./remhos in the reference): the
 * solver classes exchange through the library's plan over RCCL (unique id through comm_id_file like rmhd_run_partitioned),
 * the report is reduced over the ranks (rmh_allreduce).  Everything rmhd_run takes, incl. -ps with the IDP solvers and the
 * granular solver sequence; the one-kernel stage of a partitioned run is rmhd_run_partitioned's.  comm_id_file = NULL:
 * rmhd_run_state.  The fields handed back are this rank's elements. */
int rmhd_run_rank(const rmhd_config *cfg, const char *comm_id_file, int device, rmhd_result *res, double *u_final, double *us_final);

/* The same run on a px x py x pz box partition (ParMesh(comm, mesh, partitioning), remhos.cpp:459-463), one fused
 * kernel per RK stage and block, one neighbour exchange per stage inside the library (rmh_exchange_begin / _end):
 *   comm_id_file == NULL : ALL blocks live in this process on `device`, exchanged by device copies (validation,
 *                          or one process driving the blocks of one GPU);
 *   comm_id_file != NULL : this process owns block cfg->rank on `device`; the blocks talk through RCCL
 *                          (ncclSend / ncclRecv over xGMI).  Rank 0 writes the ncclUniqueId to that file, the others
 *                          read it -- any launcher that starts px*py*pz copies with distinct -rank works.
 * Reductions of the report (mass, max, dt estimate, stopwatches: remhos.cpp:1412-1421, 1934, 1993) go through
 * rmh_allreduce.  Fixed dt or -dtc 1; -lo 3|4|5.  0 on success. */
int rmhd_run_partitioned(const rmhd_config *cfg, const char *comm_id_file, int device, rmhd_result *res);

/* The rendezvous the two calls above use for the 128-byte ncclUniqueId (the reference gets its communicator from MPI_Init,
 * remhos.cpp:217-220; here any launcher works): writer != 0 publishes `id` at `path` (written to a temporary and renamed, stamped
 * with a magic and the launch tag -- $RMH_COMM_NONCE, else the parent pid); writer == 0 polls for up to two minutes for a record
 * with the SAME tag that is fresh (a file left by an earlier launch on the same path is never taken) and copies it to `id`.
 * 0 on success.  Exported so that the protocol can be exercised without a GPU (tests/test_id_file.py). */
int rmhd_id_file_exchange(const char *path, int writer, char id[128]);

/* z = a x + b y on n doubles of device memory, one pass, on `stream` (a hipStream_t; NULL = the default stream): the vector
 * combination the reference's time integrators make between two Mult calls -- MFEM's add(a, x, b, y, z) in RK3SSPSolver::Step (the
 * solver remhos.cpp:490 selects for -s 3) and in remhos_solvers.cpp:134, 194 (add(x, c dt, dx, x_new)) -- i.e. remhos::add of
 * include/remhos_amd/solvers.hpp for callers that hold raw pointers (the Python stepper's granular call sequence; z may alias x or y).
 * 0 on success. */
int rmhd_axpby(double a, const double *x, double b, const double *y, double *z, long long n, void *stream);

#ifdef __cplusplus
}
#endif
#endif
