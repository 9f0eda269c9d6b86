// Private to the library: the context behind the opaque rmh_ctx of include/rmh.h, shared by the entry points
// (rmh_api.hip) and the neighbour exchange (rmh_comm.hip).
#pragma once
#include "../../include/rmh.h"
#include <hip/hip_runtime.h>

#include <string>
#include <vector>


namespace rmh
{
int fail(int code, const std::string &msg); // records the message of rmh_last_error and returns code

#define RMH_HIP(call)                                                                          \
   do {                                                                                        \
      hipError_t err_ = (call);                                                                \
      if (err_ != hipSuccess)                                                                  \
      {                                                                                        \
         return rmh::fail(RMH_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err_));   \
      }                                                                                        \
   } while (0)

// every entry point that touches the device makes its context's device current first (two contexts on different
// devices in one process would otherwise launch on whatever device the caller left current)
#define RMH_ENTER(c) RMH_HIP(hipSetDevice((c)->device))

struct EventPair
{
   hipEvent_t a, b;
};

struct Exchange; // rmh_comm.hpp
} // namespace rmh

struct rmh_ctx
{
   int dim = 3; // 3: hexahedra (the whole API); 2: quadrilaterals (HO solver + the granular limiter sequence, rmh_2d.hpp)
   int p = 0, ne = 0, ng = 0, exec_mode = 0, device = 0;
   int ndof = 0;
   hipStream_t stream = nullptr;
   double t = 0.0;
   double *d_x0 = nullptr, *d_vel = nullptr, *d_tab = nullptr, *d_subvel = nullptr;
   double *d_x0h = nullptr, *d_velh = nullptr; // x0, vel in the hierarchical node basis (what ho_kernel2 reads)
   double *d_subx0 = nullptr, *d_subvmid = nullptr; // lo 4 set-up data (subcell_setup_kernel)
   double *d_fgeo = nullptr;                        // face speed coefficients (face_geom_kernel)
   int *d_face_rows = nullptr;                      // [ne][6] table block of every element face (FaceGeo, rmh_ho2.hpp)
   long long face_slots = 0;                        // blocks in d_fgeo: 3 ne + the high faces that keep their own
   double *d_m = nullptr, *d_xe_min = nullptr, *d_xe_max = nullptr;
   double *d_scr_ho = nullptr, *d_scr_lo = nullptr; // dim = 2: du_HO / du_LO between the kernels of rmh_stage_fused (made on first use)
   double *d_xe_min2 = nullptr, *d_xe_max2 = nullptr; // extrema of the fused stage's output (swapped in)
   // d_xe_min / d_xe_max hold the element extrema of the output of the last FINISHED fused stage iff xe_token != 0; the
   // stage returned that token, and only a caller that presents it gets them reused (rmh_stage_fused_chain)
   unsigned long long xe_token = 0, xe_counter = 0;
   bool stage_open = false; // between the first range call of a stage and its finishing call
   const double *stage_u = nullptr; // input vector and step of the open stage (its ranges must agree)
   double stage_dt = 0.0;
   int *d_nbr = nullptr, *d_st27 = nullptr, *d_cg = nullptr;
   const double *u_ghost = nullptr, *gh_min = nullptr, *gh_max = nullptr;
   int layer_stride = 0;      // dim = 3: the usual distance e' - e of an element to its +z face neighbour (the elements of one lattice layer);
                              // 0: no such structure -- the stage kernel then keeps contiguous eighths per XCD (xcd_chunk_for, rmh_api.hip)
   int xcd_chunk_env = -1;    // RMH_XCD_CHUNK: -1 unset; 0: contiguous eighths; n: chunks of n batches (experiments)
   int xcd_weave = 1;         // log2 of the layers woven into one chunk (HoArgs::xcd_weave); RMH_XCD_WEAVE overrides (experiments)
   int alt_order = 1;         // every other stage launch walks its batches backwards: the end of the last stage's output -- this stage's input --
                              // is what the L2s and the Infinity Cache still hold (p = 4, 5 +0.5 %, 48^3 meshes +2 %; RMH_ALT_ORDER=0 switches it off)
   int ghost_readers_end = 0; // 1 + the last owned element whose stencil reaches a ghost: ranges from here on read no ghost data
   bool gh_foreign = false; // ghost extrema overwritten by rmh_exchange_minmax_* (another field's) since the last exchange of u
   int gh_ustride = 0, gh_mstride = 1; // element strides of the ghost arrays (0: ndof)
   int gh_compact = 0; // 1: ghost records are [min | max | D^2 face trace] cells (rmh_exchange_setup, compact)
   double rel_tol = 1e-14, abs_tol = 0.0;
   int max_iter = 100;
   int jacobi_step = 0, mass_fix = 0; // completion of the local mass solve (rmh_set_mass_completion)
   bool ho_done = false;
   int bounds_type = 0; // DofInfo bounds type (-bt): 0 overlap, 1 face neighbours
   double *d_dt_est = nullptr; // running minimum of UpdateTimeStepEstimate; null while dt control is off
   bool dt_control = false;
   unsigned long long *d_viol = nullptr; // verdict words of rmh_check_violation (made on first use)
   int lo_type = 5;    // LO solver inside rmh_stage_fused: 5 mass-based average, 4 subcell residual distribution
   // stopwatches (TimingData, remhos_tools.hpp:52-64)
   bool timers_on = false;
   double tacc[4] = {0, 0, 0, 0};
   std::vector<rmh::EventPair> pending[4];
   std::vector<rmh::EventPair> pool;
   rmh::Exchange *xch = nullptr; // neighbour exchange (rmh_exchange_setup), or null
};

namespace rmh
{
void exchange_free(rmh_ctx *c); // rmh_comm.hpp
}
