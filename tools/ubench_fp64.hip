// Microbenchmark (diagnostic, not part of the product): FP64 VALU FMA and FP64 MFMA issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void fma_kernel(double *out, int iters, double a, double b)
{
   double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
   for (int i = 0; i < iters; i++)
   {
      x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
      x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void mfma_kernel(double *out, int iters, double a, double b)
{
   double4_t c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
   const double av = a + threadIdx.x, bv = b;
   for (int i = 0; i < iters; i++)
   {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c3, 0, 0, 0);
   }
   out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main()
{
   double *d;
   const int blocks = 256 * 8, threads = 256, iters = 4096;
   hipMalloc(&d, sizeof(double) * blocks * threads);
   hipEvent_t e0, e1;
   hipEventCreate(&e0); hipEventCreate(&e1);
   for (int rep = 0; rep < 2; rep++)
   {
      hipEventRecord(e0);
      hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0000001, 1e-9);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = 2.0 * 8 * iters * (double)blocks * threads;
      printf("VALU v_fma_f64: %.3f ms, %.2f TFLOP/s\n", ms, flops / ms * 1e-9);
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0000001, 1e-9);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      const double mflops = 2.0 * 16 * 16 * 4 * 4 * iters * (double)blocks * (threads / 64);
      printf("MFMA v_mfma_f64_16x16x4: %.3f ms, %.2f TFLOP/s\n", ms, mflops / ms * 1e-9);
   }
   return 0;
}
