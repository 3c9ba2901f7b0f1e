// Probe (round 4, for round 5): can ONE wave hide its VALU work behind its own bf16 MFMAs?
// The split-product row GEMM (csrc/mlp.hip unit 4) alternates, per 32-deep chunk and wave, a block of ~110 VALU instructions (operand
// prologue + three-part split + LDS writes of the NEXT values) with a block of 12 v_mfma_f32_32x32x16_bf16 issued back to back; the
// what-if builds (profiles/r04/gemm_split3_whatif.txt) show the MFMA block fully exposed.  A v_mfma_f32_32x32x16_bf16 occupies the
// SIMD's matrix pipe for 32 cycles = 8 issue slots in which the same wave may issue independent VALU / LDS instructions.
// Variants of one loop trip (12 MFMAs on one accumulator + the split of 12 value pairs + 6 ds_write_b64):
//   0  VALU block, then MFMA block (program order of the product kernel)
//   1  the same instructions, the scheduler told to emit 1 MFMA : 9 VALU : (every second group) 1 DS write  (__builtin_amdgcn_sched_group_barrier)
//   2  MFMAs only          3  VALU + LDS writes only
// Prints shader cycles per trip for 1, 2 and 3 waves per SIMD.       hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
// Static check (no GPU needed): llvm-objdump -d of the code object -- the number of VALU instructions between consecutive MFMAs of variant 1.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float pack_bf16(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(float, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ void split_bf16(float x0, float x1, float (&d)[3]) {
  d[0] = pack_bf16(x0, x1);
  unsigned u = __float_as_uint(d[0]);
  float r0 = x0 - __uint_as_float(u << 16), r1 = x1 - __uint_as_float(u & 0xffff0000u);
  d[1] = pack_bf16(r0, r1);
  u = __float_as_uint(d[1]);
  r0 -= __uint_as_float(u << 16);
  r1 -= __uint_as_float(u & 0xffff0000u);
  d[2] = pack_bf16(r0, r1);
}

constexpr int PAIRS = 12;        // value pairs split per trip (the 64 x 64 tile: 8 operand + 8 weight values per thread = 8 pairs; 12: the 64 x 128 tile)

template <int VAR>
__global__ void __launch_bounds__(256) probe(const float *in, float *out, long long *cyc, int trips) {
  __shared__ float2 lds[2][6][256];
  const int tid = threadIdx.x;
  float x[2 * PAIRS];
#pragma unroll
  for (int i = 0; i < 2 * PAIRS; ++i) x[i] = in[(i * 256 + tid) & 4095];
  float4 af = *reinterpret_cast<const float4 *>(in + 4 * tid), bf = *reinterpret_cast<const float4 *>(in + 1024 + 4 * tid);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float keep = 0.f;
  const long long t0 = clock64();
  for (int it = 0; it < trips; ++it) {
    float d[PAIRS][3];
    if (VAR != 2) {
#pragma unroll
      for (int p = 0; p < PAIRS; ++p) {
        const float s = 1.0f + 0.001f * (float)it;                      // (the prologue's scale / shift / ReLU)
        split_bf16(fmaxf(fmaf(x[2 * p], s, 0.5f), 0.f), fmaxf(fmaf(x[2 * p + 1], s, 0.25f), 0.f), d[p]);
      }
#pragma unroll
      for (int w = 0; w < 6; ++w) lds[it & 1][w][tid] = make_float2(d[2 * w][w % 3], d[2 * w + 1][w % 3]);
#pragma unroll
      for (int p = 0; p < PAIRS; ++p) keep += d[p][0] + d[p][1] + d[p][2];
    }
    if (VAR != 3) {
#pragma unroll
      for (int m = 0; m < 12; ++m)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc, 0, 0, 0);
    }
    if (VAR == 1) {
      // 12 groups: one MFMA, nine VALU, and a DS write behind every second one
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);              // 9 VALU
        if (g & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // 1 DS write
      }
    }
  }
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  float r = keep;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += acc[i];
  if (r == 12345.678f) out[blockIdx.x * 256 + tid] = r + lds[0][0][tid].x;
}

template <int VAR>
static void run(const char *name, const float *in, float *out, long long *cyc, int blocks_per_cu) {
  const int trips = 2000, nb = 256 * blocks_per_cu;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<VAR>, dim3(nb), dim3(256), 0, 0, in, out, cyc, trips);
  CK(hipDeviceSynchronize());
  long long *h = (long long *)malloc(nb * sizeof(long long));
  CK(hipMemcpy(h, cyc, nb * sizeof(long long), hipMemcpyDeviceToHost));
  double s = 0;
  for (int i = 0; i < nb; ++i) s += (double)h[i];
  printf("%-44s %d wave(s) per SIMD: %8.1f cycles per trip\n", name, blocks_per_cu, s / nb / trips);
  free(h);
}

int main() {
  float *in, *out; long long *cyc;
  CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 256 * 3 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 3 * 8));
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (float)(i % 977) - 0.4f;
  CK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice));
  for (int w = 1; w <= 3; ++w) {
    run<2>("12 MFMAs", in, out, cyc, w);
    run<3>("split of 12 pairs + 6 ds_write_b64", in, out, cyc, w);
    run<0>("VALU block, then MFMA block", in, out, cyc, w);
    run<1>("interleaved by sched_group_barrier", in, out, cyc, w);
  }
  return 0;
}
