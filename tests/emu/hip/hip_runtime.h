// TEST INFRASTRUCTURE ONLY -- a minimal host emulation of the HIP constructs used by
// remhos_amd/csrc so that the kernel sources can be compiled with g++ and run under
// sanitizers / against the oracle on a machine without a GPU (tests/test_emu_cpu.py).
// The product never includes this file: hipcc resolves <hip/hip_runtime.h> to ROCm's header.
// A workgroup is emulated by blockDim.x OS threads that meet at a std::barrier for
// __syncthreads(); workgroups run one after another.
#pragma once
#define HIPEMU 1
#include <algorithm>
#include <barrier>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __constant__ static
#define HIP_SYMBOL(x) (x)
#define __launch_bounds__(...)

struct dim3
{
   unsigned x, y, z;
   dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
typedef void *hipStream_t;
typedef struct { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

namespace hipemu
{
inline thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
inline std::barrier<> *g_barrier = nullptr;
inline std::barrier<> *g_wave_barrier[16] = {nullptr}; // one per wavefront: cross-lane operations only meet their own 64 lanes
// cross-lane exchange slots, double-buffered: a lane's call k writes buffer k & 1, meets its wavefront ONCE and reads.  The
// next write to the same buffer is call k + 2, behind the barrier of call k + 1, which no lane passes before every lane
// has finished the reads of call k -- so one barrier per call is enough (all lanes of a wavefront make the same calls).
inline unsigned long long g_xchg_buf[2][1024], g_xchg2_buf[2][1024];
inline thread_local unsigned t_xchg_call = 0;
} // namespace hipemu

#define threadIdx (hipemu::t_threadIdx)
#define blockIdx (hipemu::t_blockIdx)
#define blockDim (hipemu::t_blockDim)
#define gridDim (hipemu::t_gridDim)

inline void __syncthreads() { hipemu::g_barrier->arrive_and_wait(); }
inline void hipemu_wave_sync() { hipemu::g_wave_barrier[hipemu::t_threadIdx.x >> 6]->arrive_and_wait(); }

#define HIPEMU_XCHG_SLOTS                                              \
   const unsigned xpar_ = hipemu::t_xchg_call++ & 1u;                  \
   unsigned long long *const xa = hipemu::g_xchg_buf[xpar_];           \
   unsigned long long *const xb = hipemu::g_xchg2_buf[xpar_];          \
   (void)xb

inline double __shfl_xor(double v, int off)
{
   const unsigned t = threadIdx.x;
   HIPEMU_XCHG_SLOTS;
   std::memcpy(&xa[t], &v, 8);
   hipemu_wave_sync();
   double r;
   std::memcpy(&r, &xa[t ^ (unsigned)off], 8);
   return r;
}

// value of lane `src` of the caller's wavefront (ds_bpermute)
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline double __shfl(double v, int src)
{
   const unsigned t = threadIdx.x, base = t - (t & 63u);
   HIPEMU_XCHG_SLOTS;
   std::memcpy(&xa[t], &v, 8);
   hipemu_wave_sync();
   double r;
   std::memcpy(&r, &xa[base + ((unsigned)src & 63u)], 8);
   return r;
}

inline int __double2loint(double v) { long long b; std::memcpy(&b, &v, 8); return (int)(b & 0xffffffffll); }
inline int __double2hiint(double v) { long long b; std::memcpy(&b, &v, 8); return (int)(b >> 32); }
inline double __hiloint2double(int hi, int lo)
{
   const unsigned long long b = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
   double v;
   std::memcpy(&v, &b, 8);
   return v;
}

// data-parallel primitive (DPP) emulation for the controls the kernels use
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int /*bank_mask*/, bool /*bc*/)
{
   const unsigned t = threadIdx.x, lane = t & 63u, base = t - lane;
   HIPEMU_XCHG_SLOTS;
   xa[t] = (unsigned)src;
   hipemu_wave_sync();
   const unsigned row = lane >> 4, inrow = lane & 15u;
   int srclane = -1;
   if (ctrl >= 0 && ctrl <= 0xFF) { srclane = (int)((lane & ~3u) + ((ctrl >> (2 * (lane & 3u))) & 3)); } // quad_perm
   else if (ctrl > 0x100 && ctrl <= 0x10F) { srclane = (inrow + (ctrl & 15) < 16) ? (int)(lane + (ctrl & 15)) : -1; }     // row_shl:n
   else if (ctrl == 0x140) { srclane = (int)(row * 16 + (15 - inrow)); }                                  // row_mirror
   else if (ctrl == 0x141) { srclane = (int)(row * 16 + (inrow < 8 ? 7 - inrow : 23 - inrow)); }          // row_half_mirror
   else if (ctrl == 0x142) { srclane = (row >= 1) ? (int)(row * 16 - 1) : -1; }                           // row_bcast:15
   else if (ctrl == 0x143) { srclane = (row >= 2) ? 31 : -1; }                                            // row_bcast:31
   int r = old;
   if (((row_mask >> row) & 1) && srclane >= 0) { r = (int)xa[base + srclane]; }
   return r;
}

// v_permlane32_swap: a = {a[0..31], b[0..31]}, b = {a[32..63], b[32..63]}
inline void hipemu_permlane32_swap(double &a, double &b)
{
   const unsigned t = threadIdx.x, lane = t & 63u, base = t - lane;
   HIPEMU_XCHG_SLOTS;
   std::memcpy(&xa[t], &a, 8);
   std::memcpy(&xb[t], &b, 8);
   hipemu_wave_sync();
   double na = a, nb = b;
   if (lane >= 32) { std::memcpy(&na, &xb[base + lane - 32], 8); }
   else { std::memcpy(&nb, &xa[base + lane + 32], 8); }
   a = na;
   b = nb;
}

// v_permlane16_swap: a = {a.row0, b.row0, a.row2, b.row2}, b = {a.row1, b.row1, a.row3, b.row3}
inline void hipemu_permlane16_swap(double &a, double &b)
{
   const unsigned t = threadIdx.x, lane = t & 63u, base = t - lane, row = lane >> 4, in = lane & 15u;
   HIPEMU_XCHG_SLOTS;
   std::memcpy(&xa[t], &a, 8);
   std::memcpy(&xb[t], &b, 8);
   hipemu_wave_sync();
   double na = a, nb = b;
   if (row & 1u) { std::memcpy(&na, &xb[base + (row - 1) * 16 + in], 8); } // odd rows of a <- even rows of b
   else { std::memcpy(&nb, &xa[base + (row + 1) * 16 + in], 8); }          // even rows of b <- odd rows of a
   a = na;
   b = nb;
}

template <typename T>
inline T __builtin_nontemporal_load(const T *p) { return *p; }

using std::max;
using std::min;

inline double atomicAdd(double *p, double v)
{
   unsigned long long *ip = reinterpret_cast<unsigned long long *>(p);
   unsigned long long old = __atomic_load_n(ip, __ATOMIC_RELAXED), neu;
   double o;
   do {
      std::memcpy(&o, &old, 8);
      const double n = o + v;
      std::memcpy(&neu, &n, 8);
   } while (!__atomic_compare_exchange_n(ip, &old, neu, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
   return o;
}

inline long long __double_as_longlong(double v) { long long b; std::memcpy(&b, &v, 8); return b; }

inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v)
{
   unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
   while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
   return old;
}

inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v)
{
   unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
   while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
   return old;
}

inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }

inline int atomicMax(int *p, int v)
{
   int old = __atomic_load_n(p, __ATOMIC_RELAXED);
   while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
   return old;
}

inline const char *hipGetErrorString(hipError_t) { return "hip emulation error"; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? 0 : 1; }
inline hipError_t hipFree(void *p) { std::free(p); return 0; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
template <typename S>
inline hipError_t hipMemcpyToSymbol(S &sym, const void *src, size_t n, size_t off, hipMemcpyKind)
{
   std::memcpy((char *)&sym + off, src, n);
   return 0;
}
inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return 0; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new std::remove_pointer_t<hipEvent_t>; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return 0; }
enum { hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { return hipStreamCreate(s); }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return 0; }
struct hipDeviceProp_t { int multiProcessorCount = 256; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { *p = hipDeviceProp_t(); return 0; }
// (the emulation has no compute units to mask: the stream is an ordinary one; the mask itself is checked by the caller's test)
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, unsigned n, const unsigned *mask) { return (n && mask) ? hipStreamCreate(s) : 1; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }

template <typename K, typename... Args>
inline void hipemu_launch(K kernel, dim3 grid, dim3 block, Args... args)
{
   std::barrier<> bar(block.x);
   hipemu::g_barrier = &bar;
   std::vector<std::unique_ptr<std::barrier<>>> wbar;
   for (unsigned w = 0; w * 64 < block.x; w++)
   {
      wbar.emplace_back(new std::barrier<>(std::min(64u, block.x - w * 64)));
      hipemu::g_wave_barrier[w] = wbar.back().get();
   }
   std::vector<std::thread> pool;
   for (unsigned t = 0; t < block.x; t++)
   {
      pool.emplace_back([=, &bar]()
      {
         hipemu::t_blockDim = block;
         hipemu::t_gridDim = grid;
         hipemu::t_threadIdx = dim3(t, 0, 0);
         for (unsigned b = 0; b < grid.x; b++)
         {
            hipemu::t_blockIdx = dim3(b, 0, 0);
            kernel(args...);
            bar.arrive_and_wait();
         }
      });
   }
   for (auto &th : pool) { th.join(); }
   hipemu::g_barrier = nullptr;
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
   hipemu_launch(kernel, grid, block, __VA_ARGS__)
