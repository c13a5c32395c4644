// Bottom-up skeleton of the split-bf16 GEMM K loop on gfx950: which ingredient costs the matrix pipe its time?
// One workgroup = 4 waves (256 threads), 61 KB of LDS so that two workgroups share a CU, as the production kernel.
// Per k16 step and wave: [R] 12 ds_read_b128 fragment reads, 24 MFMAs 32x32x16 bf16 on 4 accumulators, [V] n VALU per MFMA,
// [W] 6 ds_write_b64, [B] barriers (2 per 2 steps, like the single-buffer kernel), [L] 4 buffer loads per step.
//   hipcc --offload-arch=gfx950 -O3 -w tools/native/gemm_skeleton.hip -o tools/native/gemm_skeleton
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int R, int V, int W, int B, int L>
__global__ __launch_bounds__(256, 2) void skel(const float4* __restrict__ g, float* out, int steps, unsigned seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* lds = reinterpret_cast<__bf16*>(smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // seed bit 31 set: random operand bits (bf16 values with random sign / mantissa and exponents around 1.0) — the matrix
  // pipe's power draw, hence its clock, depends on operand bit activity
  for (int i = t; i < 61440 / 4; i += 256) {
    unsigned v = seed + i;
    if (seed & 0x80000000u) {
      unsigned h = (unsigned)i * 2654435761u ^ (seed * 40503u);
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      v = (h & 0x807F807Fu) | 0x3F003F00u | ((h >> 3) & 0x00800080u);
    }
    reinterpret_cast<unsigned*>(smem)[i] = v;
  }
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 fa[6], fb[6];
  for (int i = 0; i < 6; ++i)
    for (int e = 0; e < 8; ++e) { fa[i][e] = (__bf16)(float)(lane + e + i); fb[i][e] = (__bf16)(float)(seed + e + i); }
  unsigned va[8], vb[8];
  for (int i = 0; i < 8; ++i) { va[i] = seed + t * 7 + i; vb[i] = seed * 3 + i + 0x3f800000u; }
  float4 ld[4] = {};
  const int frow = (wave >> 1) * 64 + (lane & 31), fk = (lane >> 5) * 8;
  const __bf16* ab = lds + frow * 40 + fk;
  for (int s = 0; s < steps; ++s) {
    if (R) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[2 * p] = *reinterpret_cast<const bf16x8*>(ab + p * 5120 + (s & 1) * 16);
        fa[2 * p + 1] = *reinterpret_cast<const bf16x8*>(ab + p * 5120 + 32 * 40 + (s & 1) * 16);
        fb[2 * p] = *reinterpret_cast<const bf16x8*>(ab + 15360 + p * 5120 + (s & 1) * 16);
        fb[2 * p + 1] = *reinterpret_cast<const bf16x8*>(ab + 15360 + p * 5120 + 32 * 40 + (s & 1) * 16);
      }
    }
    if (L) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ld[i] = g[(size_t)((s * 4 + i) * 256 + t) & 0xFFFFF];
    }
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(m >> 2) % 6], fb[(m >> 1) % 6], acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int i = (m * V + v) & 7;
        if (v & 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(va[i]) : "v"(vb[i]));
        else asm volatile("v_and_b32 %0, %1, %0" : "+v"(va[i]) : "v"(vb[i]));
      }
    }
    if (B && (s & 1)) __syncthreads();
    if (W && (s & 1)) {
#pragma unroll
      for (int i = 0; i < 12; ++i)
        *reinterpret_cast<uint2*>(lds + (i * 32 + (t >> 3)) * 40 + (t & 7) * 4) =
            make_uint2(va[i & 7] + __builtin_bit_cast(unsigned, ld[i & 3].x), vb[i & 7]);
    }
    if (B && (s & 1)) __syncthreads();
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][9];
  unsigned x = 0;
  for (int i = 0; i < 8; ++i) x ^= va[i];
  out[blockIdx.x * 256 + t] = r + (float)x + ld[0].x + ld[3].w;
}

static unsigned g_seed_flag = 0;

template <int R, int V, int W, int B, int L>
void run(const char* what, const float4* g, float* out) {
  const int steps = 576, blocks = 512 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(skel<R, V, W, B, L>), hipFuncAttributeMaxDynamicSharedMemorySize, 61440);
  skel<R, V, W, B, L><<<blocks, 256, 61440>>>(g, out, steps, 1u | g_seed_flag);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) skel<R, V, W, B, L><<<blocks, 256, 61440>>>(g, out, steps, (2u + r) | g_seed_flag);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 5.0 * blocks * 4 * steps * 24.0 * 32768.0;
  printf("%-58s %7.0f TFLOP/s executed\n", what, flop / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'r') { g_seed_flag = 0x80000000u; printf("random operand bits\n"); }
  float4* g; float* out;
  hipMalloc(&g, 16u << 20); hipMalloc(&out, 4u << 20);
  hipMemset(g, 0, 16u << 20);
  run<0, 0, 0, 0, 0>("MFMA only", g, out);
  run<1, 0, 0, 0, 0>("+ 12 ds_read_b128 per 24 MFMA", g, out);
  run<0, 5, 0, 0, 0>("MFMA + 5 VALU per MFMA", g, out);
  run<1, 5, 0, 0, 0>("+ reads + 5 VALU", g, out);
  run<1, 5, 1, 0, 0>("+ reads + VALU + 12 ds_write_b64 per 2 steps", g, out);
  run<1, 5, 1, 1, 0>("+ reads + VALU + writes + 2 barriers per 2 steps", g, out);
  run<1, 5, 1, 1, 1>("+ reads + VALU + writes + barriers + 4 loads per step", g, out);
  run<1, 5, 0, 1, 0>("reads + VALU + barriers (no writes)", g, out);
  run<1, 0, 1, 1, 1>("reads + writes + barriers + loads (no VALU)", g, out);
  run<1, 8, 1, 1, 1>("everything with 8 VALU per MFMA", g, out);
  return 0;
}
