// Microbenchmark (diagnostic): do FP64 MFMA and FP64 VALU FMA execute concurrently on gfx950?
//   mode 0: VALU only   mode 1: MFMA only   mode 2: both interleaved in every wave   mode 3: waves split by parity
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(double *out, int iters, double a, double b)
{
   double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
   double4_t c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
   const double av = a + threadIdx.x, bv = b;
   const bool do_valu = MODE == 0 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 0);
   const bool do_mfma = MODE == 1 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 1);
   for (int i = 0; i < iters; i++)
   {
      if (do_mfma)
      {
         c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
         c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
      }
      if (do_valu)
      {
         x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
         x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
         x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
         x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
      }
      if (do_mfma)
      {
         c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c2, 0, 0, 0);
         c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c3, 0, 0, 0);
      }
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + c0[0] + c1[1] + c2[2] + c3[3];
}

template <int MODE>
void run(double *d, const char *name)
{
   const int blocks = 256 * 8, threads = 256, iters = 2048;
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   for (int rep = 0; rep < 2; rep++)
   {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0000001, 1e-9);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double waves = (double)blocks * threads / 64;
      const double fv = (MODE == 0 || MODE == 2) ? waves : (MODE == 3 ? waves / 2 : 0);
      const double fm = (MODE == 1 || MODE == 2) ? waves : (MODE == 3 ? waves / 2 : 0);
      const double valu = 2.0 * 16 * 64 * iters * fv, mfma = 2.0 * 1024 * 4 * iters * fm;
      if (rep) printf("%-28s %.3f ms  VALU %.1f TF + MFMA %.1f TF = %.1f TF\n", name, ms, valu / ms * 1e-9, mfma / ms * 1e-9, (valu + mfma) / ms * 1e-9);
   }
}

int main()
{
   double *d;
   hipMalloc(&d, sizeof(double) * 256 * 8 * 256);
   run<0>(d, "VALU only");
   run<1>(d, "MFMA only");
   run<2>(d, "both in every wave");
   run<3>(d, "waves split MFMA / VALU");
   return 0;
}
