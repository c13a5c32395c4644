// VALU issue-rate probe for gfx950: cycles per wave64 instruction for the ops the operand split is made of.
// Each kernel runs a long dependent-free stream of one opcode (8 independent chains) on ONE wave per SIMD and reports
// s_memtime cycles / instruction.   hipcc --offload-arch=gfx950 -O3 tools/native/valu_rate.hip -o tools/native/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 256
#define CHAINS 8

template <int OP>
__global__ void probe(unsigned* out, long long* cycles, unsigned seed) {
  unsigned a[CHAINS], b[CHAINS];
  for (int i = 0; i < CHAINS; ++i) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i + 0x3f800000u; }
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP / CHAINS; ++r)
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (OP == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        if (OP == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        if (OP == 2) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
        if (OP == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        if (OP == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        if (OP == 5) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(a[i]));
        if (OP == 6) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  unsigned s = 0;
  for (int i = 0; i < CHAINS; ++i) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_block) {
  unsigned* out; long long* cyc; long long h = 0;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  probe<OP><<<256, 64 * waves_per_block>>>(out, cyc, 1u);
  probe<OP><<<256, 64 * waves_per_block>>>(out, cyc, 2u);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-22s waves/block %d: %.2f cycles per instruction per wave\n", name, waves_per_block, (double)h / (64.0 * REP));
  hipFree(out); hipFree(cyc);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// one MFMA 32x32x16 bf16 followed by NV independent VALU ops (v_and / v_cvt_pk / v_sub mix), 4 accumulators in turn
template <int NV, int KIND>
__global__ void mix(float* out, long long* cycles, unsigned seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 fa, fb;
  for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)(float)(threadIdx.x + e); fb[e] = (__bf16)(float)(seed + e); }
  unsigned a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i + 0x3f800000u; }
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int i = (m * NV + v) & 7;
        if (KIND == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
        if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        if (KIND == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  unsigned x = 0;
  for (int i = 0; i < 8; ++i) x ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)x;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int NV, int KIND>
void run_mix(const char* kind, int threads) {
  float* out; long long* cyc; long long h = 0;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
  mix<NV, KIND><<<256, threads>>>(out, cyc, 1u);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) mix<NV, KIND><<<256 * 4, threads>>>(out, cyc, 2u);   // 4 blocks per CU in turn
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double flop = 20.0 * 256 * 4 * (threads / 64) * 64.0 * 16 * 32768.0;
  printf("MFMA + %2d x %-18s %d waves/SIMD: %.1f counter ticks per MFMA (wave 0), %.0f TFLOP/s by events\n", NV, kind,
         threads / 256, (double)h / (64.0 * 16), flop / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

template <int KIND>
void sweep(const char* kind) {
  for (int threads : {256, 512}) {
    run_mix<0, KIND>(kind, threads); run_mix<2, KIND>(kind, threads); run_mix<4, KIND>(kind, threads);
    run_mix<5, KIND>(kind, threads); run_mix<6, KIND>(kind, threads); run_mix<7, KIND>(kind, threads);
    run_mix<8, KIND>(kind, threads); run_mix<10, KIND>(kind, threads); run_mix<12, KIND>(kind, threads);
  }
}

int main() {
  sweep<0>("v_and_b32");
  sweep<1>("v_cvt_pk_bf16_f32");
  sweep<2>("v_sub_f32");
  for (int w : {1, 4, 8}) {
    run<0>("v_and_b32", w);
    run<1>("v_cvt_pk_bf16_f32", w);
    run<2>("v_perm_b32", w);
    run<3>("v_sub_f32", w);
    run<4>("v_mul_lo_u32", w);
    run<5>("v_lshlrev_b32", w);
    run<6>("v_and_or_b32", w);
  }
  return 0;
}
