// Diagnostic instrumentation of ho_kernel2 (rmh_ho2.hpp), compiled in only by the diagnostic build
// `make stamps` (-DRMH_STAMPS -> remhos_amd/librmh_stamps.so, read by tools/stamps.py); in the product build both
// macros are empty.
// (included inside namespace rmh by rmh_ho2.hpp)
#pragma once

#ifdef RMH_STAMPS
// diagnostic build only: per-phase cycle shares of workgroup-thread 0.  The deltas are accumulated in LDS and
// written once at the end of the kernel (a global atomic per stamp would be waited for by the next
// s_waitcnt vmcnt(0) of the workgroup and show up as a phantom wait); one row of 32 counters per workgroup,
// summed by the host.
#ifndef RMH_STAMP_TID
#define RMH_STAMP_TID 0
#endif
constexpr int RMH_STAMP_MAXWG = 1 << 18;
__device__ unsigned long long g_stamps[RMH_STAMP_MAXWG][32];
#define RMH_STAMP(k)                                                                   \
   do {                                                                                \
      if (threadIdx.x == RMH_STAMP_TID)                                                \
      {                                                                                \
         const unsigned long long now_ = clock64();                                    \
         s_stamp[k] += now_ - stamp_prev_;                                             \
         stamp_prev_ = now_;                                                           \
      }                                                                                \
   } while (0)
#define RMH_STAMP_FLUSH()                                                              \
   do {                                                                                \
      __syncthreads();                                                                 \
      if (threadIdx.x < 32 && blockIdx.x < RMH_STAMP_MAXWG) { g_stamps[blockIdx.x][threadIdx.x] += s_stamp[threadIdx.x]; } \
   } while (0)
#elif defined(RMH_STOP_AT)
// diagnostic build only (tools/pmc_variants.sh): the kernel ends at phase mark RMH_STOP_AT, so that counters can be
// attributed to phases by differences between builds.  The test on a kernel argument (always true) keeps the compiler
// from discarding the work in front of the mark; the outputs are garbage.
#define RMH_STAMP(k)                                                  \
   do {                                                               \
      if ((k) == RMH_STOP_AT && a.max_iter != -12345) { return; }     \
   } while (0)
#define RMH_STAMP_FLUSH()
#elif defined(RMH_PHASE_MARKS)
// diagnostic compile to assembly only (tools/isa_phases.py): a comment per phase boundary in the .s file
#define RMH_STAMP(k) asm volatile("; RMH_PHASE " #k)
#define RMH_STAMP_FLUSH()
#else
#define RMH_STAMP(k)
#define RMH_STAMP_FLUSH()
#endif

