// Sustained rate of v_mfma_f32_32x32x16_bf16 on this GPU: every wave issues independent MFMAs from registers only
// (no LDS, no memory), 2 waves per SIMD.  Prints TFLOP/s for a few run lengths — the ceiling the GEMM kernels can
// be measured against (nominal dense bf16 peak: 2500 TFLOP/s at the boost clock).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (__bf16)(float)(threadIdx.x + e);
    b[e] = (__bf16)(float)(threadIdx.x * 3 + e);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float* out;
  const int blocks = 256 * 2;   // 2 workgroups of 4 waves per CU = 2 waves per SIMD
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t s, e;
  hipEventCreate(&s);
  hipEventCreate(&e);
  for (int iters : {2000, 20000, 200000}) {
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms = 0.f;
    hipEventElapsedTime(&ms, s, e);
    const double flops = (double)blocks * 4 /*waves*/ * (double)iters * 24.0 * (2.0 * 32 * 32 * 16);
    printf("iters %7d  %8.3f ms  %8.1f TFLOP/s  (%.1f%% of 2500)\n", iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
  }
  return 0;
}
