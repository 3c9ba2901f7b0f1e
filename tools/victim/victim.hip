// Synthetic victims for the two-stream hazard (diagnosis; tools/victim_probe.py): long dependent VALU chains, one thread per value, no LDS,
// no scratch, coalesced store.  KIND 0: fused multiply-adds only.  1: v_rcp_f32 + v_sqrt_f32 in the chain.  2: IEEE division and sqrt
// (v_div_scale / v_div_fmas / v_div_fixup).  3: atan2f / acosf.  4: gather loads through an index table, no arithmetic to speak of.
// 5 / 6: a 9-key sorting network with lane-mask selects / with VGPR-mask selects.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(256) victim_kernel(float *__restrict__ out, const int *__restrict__ table, const float *__restrict__ src,
                                                     int n, int iters, float scale) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float x = 0.25f + (float)(t % 977) * (1.f / 1024.f), y = 0.5f + (float)(t % 313) * (1.f / 512.f), acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      x = __builtin_fmaf(x, 0.99f, 0.013f); y = __builtin_fmaf(y, 0.98f, x * 0.01f); acc = __builtin_fmaf(x, y, acc * 0.5f);
    } else if (KIND == 1) {
      x = __builtin_amdgcn_rcpf(x + 1.5f) + 0.3f; y = __builtin_amdgcn_sqrtf(y + x); acc = __builtin_fmaf(x, y, acc * 0.5f);
    } else if (KIND == 2) {
      x = 1.0f / (x + 1.5f) + 0.3f; y = sqrtf(y + x) / (1.0f + x); acc = __builtin_fmaf(x, y, acc * 0.5f);
    } else if (KIND == 3) {
      x = atan2f(y + 0.1f, x + 0.2f) * 0.3f + 0.3f; y = acosf(x * 0.5f) * 0.4f + 0.1f; acc = __builtin_fmaf(x, y, acc * 0.5f);
    } else if (KIND == 4) {
      const int p = table[(t * 9 + i) % n];
      acc += src[p * 3] - src[(p * 3 + 1) % n];
    } else if (KIND == 5 || KIND == 6) {
      // odd-even transposition sort of 9 keys with a payload, as the fan kernel's: 5 = compare -> lane mask (SGPR pair) -> v_cndmask selects;
      // 6 = the same decisions as integer sign bits in VGPRs and v_bfi selects (no lane mask anywhere)
      float k[9], v[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        k[j] = __builtin_fmaf((float)((t * 31 + j * 17 + i * 7) % 1009), 1.f / 1009.f, x * 0.001f);
        v[j] = (float)j + y;
      }
#pragma unroll
      for (int round = 0; round < 9; ++round) {
#pragma unroll
        for (int j = (round & 1); j + 1 < 9; j += 2) {
          if (KIND == 5) {
            const bool sw = k[j + 1] < k[j];
            const float tk = k[j], tv = v[j];
            k[j] = sw ? k[j + 1] : tk; v[j] = sw ? v[j + 1] : tv;
            k[j + 1] = sw ? tk : k[j + 1]; v[j + 1] = sw ? tv : v[j + 1];
          } else {
            const unsigned m = (unsigned)(__float_as_int(k[j + 1] - k[j]) >> 31);      // all ones where k[j+1] < k[j] (keys are finite, distinct or equal)
            const unsigned a = __float_as_uint(k[j]), b = __float_as_uint(k[j + 1]), c = __float_as_uint(v[j]), d = __float_as_uint(v[j + 1]);
            k[j] = __uint_as_float((b & m) | (a & ~m)); k[j + 1] = __uint_as_float((a & m) | (b & ~m));
            v[j] = __uint_as_float((d & m) | (c & ~m)); v[j + 1] = __uint_as_float((c & m) | (d & ~m));
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 9; ++j) acc = __builtin_fmaf(acc, 0.5f, v[j] * (float)(j + 1));
      x = __builtin_fmaf(x, 0.99f, 0.013f); y = __builtin_fmaf(y, 0.98f, 0.007f);
    } else if (KIND == 7 || KIND == 10 || KIND == 11) {
      // packed fp32 arithmetic written as 2-vectors: 7 = v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on VGPR operands only; 10 = one operand a
      // wave-uniform value (an SGPR pair / op_sel broadcast); 11 = inline constants and negated operands (neg_lo / neg_hi)
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f a = {x, y}, b = {y, x + 0.25f}, c = {acc, 0.5f};
#pragma unroll 8
      for (int u = 0; u < 8; ++u) {
        if (KIND == 7) {
          const v2f w = {b.y, a.x};
          c = __builtin_elementwise_fma(a, w, c); a = a * b + w; b = (b + a) * w;
        } else if (KIND == 10) {
          const v2f s2 = {scale, scale};
          c = __builtin_elementwise_fma(a, s2, c); a = a * s2 + b; b = (b + a) * s2;
        } else {
          c = __builtin_elementwise_fma(a, (v2f){0.5f, 0.5f}, -c); a = a * (v2f){-0.5f, -0.5f} - b; b = (b - a) * (v2f){0.5f, 0.5f};
        }
        a = {a.x - (float)(int)a.x, a.y - (float)(int)a.y}; b = {b.x - (float)(int)b.x, b.y - (float)(int)b.y};
      }
      x = a.x * 0.5f + 0.25f; y = b.y * 0.5f + 0.5f; acc = c.x * 0.5f + c.y;
    } else if (KIND >= 12 && KIND <= 21) {
      // the operand forms the SLP vectorizer produced in the fan kernel, as assembly statements (values stay in [0, 2): a contraction):
      // 12 v_pk_fma_f32 with an SGPR-pair operand broadcast by op_sel_hi; 13 v_pk_mul_f32 by an SGPR pair; 14 v_pk_add_f32 with op_sel:[0,1]
      // (the high half of a source feeding the low lane element) and negation; 15 v_pk_mul_f32 by an inline constant; 16 all four in turn
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f a = {x, y}, b = {y * 0.5f, x * 0.5f};
      const v2f s2 = {scale, scale * 0.5f};
#pragma unroll 4
      for (int u = 0; u < 16; ++u) {
        if (KIND == 12 || KIND == 16) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(a) : "v"(a), "s"(s2), "v"(b));
        if (KIND == 13 || KIND == 16) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(b) : "v"(a), "s"(s2));
        if (KIND == 14 || KIND == 16) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a) : "v"(a), "v"(b));
        if (KIND == 15 || KIND == 16) asm volatile("v_pk_mul_f32 %0, %1, 0.5 op_sel_hi:[1,0]" : "=v"(b) : "v"(a));
        // 17 op_sel:[0,1] alone; 18 the negation alone; 19 v_pk_mul_f32 with op_sel:[0,1]; 20 v_pk_fma_f32 with op_sel:[0,1,0]; 21 op_sel:[1,0]
        if (KIND == 17) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(a) : "v"(a), "v"(b));
        if (KIND == 18) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a) : "v"(a), "v"(b));
        if (KIND == 19) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(a) : "v"(a), "v"(b));
        if (KIND == 20) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(a) : "v"(a), "v"(b), "v"(b));
        if (KIND == 21) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(a) : "v"(a), "v"(b));
        if (KIND == 14 || KIND >= 17) { b = b * 0.75f; a = a * 0.5f; }
      }
      x = a.x - (float)(int)a.x + 0.25f; y = b.y - (float)(int)b.y + 0.5f; acc = __builtin_fmaf(a.y, b.x, acc * 0.5f);
    }
  }
  out[t] = acc + x + y;
}

extern "C" int victim_launch(int kind, float *out, const int *table, const float *src, int n, int iters, void *stream) {
  const dim3 grid((n + 255) / 256), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case 0: hipLaunchKernelGGL(victim_kernel<0>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 1: hipLaunchKernelGGL(victim_kernel<1>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 2: hipLaunchKernelGGL(victim_kernel<2>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 3: hipLaunchKernelGGL(victim_kernel<3>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 4: hipLaunchKernelGGL(victim_kernel<4>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 5: hipLaunchKernelGGL(victim_kernel<5>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 6: hipLaunchKernelGGL(victim_kernel<6>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 7: hipLaunchKernelGGL(victim_kernel<7>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 10: hipLaunchKernelGGL(victim_kernel<10>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 11: hipLaunchKernelGGL(victim_kernel<11>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 12: hipLaunchKernelGGL(victim_kernel<12>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 13: hipLaunchKernelGGL(victim_kernel<13>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 14: hipLaunchKernelGGL(victim_kernel<14>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 15: hipLaunchKernelGGL(victim_kernel<15>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 16: hipLaunchKernelGGL(victim_kernel<16>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 17: hipLaunchKernelGGL(victim_kernel<17>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 18: hipLaunchKernelGGL(victim_kernel<18>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 19: hipLaunchKernelGGL(victim_kernel<19>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    case 20: hipLaunchKernelGGL(victim_kernel<20>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
    default: hipLaunchKernelGGL(victim_kernel<21>, grid, block, 0, st, out, table, src, n, iters, 0.7071f); break;
  }
  return (int)hipGetLastError();
}


// ---- aggressors (tools/victim_probe.py AGG=custom:<kind>): candidate properties of the GEMM kernels, one at a time, launched on the main stream
__device__ __forceinline__ void glds16(const void *base, unsigned voff, const float *lds) {
  const unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)lds;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
}

template <int KIND>
__global__ void __launch_bounds__(256) aggressor(float *sink, int iters, const float *src) {
  v16f c = {};
  __shared__ __attribute__((aligned(16))) float stage[2][4096];           // 2 x 16 KB
  if (KIND == 4 || KIND == 5) {
    // LDS-DMA (global -> LDS without registers) of 16 KB per step, read back through ds_read; 5: with bf16 MFMAs on the fragments
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v8bf a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
    for (int i = 0; i < iters; ++i) {
      float *st = stage[i & 1];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        glds16(src + (size_t)((blockIdx.x * 7 + i) % 64) * 4096, (unsigned)((wave * 4 + q) * 1024 + lane * 16), st + (wave * 4 + q) * 256);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const float4 f = reinterpret_cast<const float4 *>(st)[threadIdx.x];
      if (KIND == 5) {
        a[0] = (__bf16)f.x; b[0] = (__bf16)f.y;
#pragma unroll
        for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
      } else {
        c[0] += f.x; c[1] += f.y; c[2] += f.z; c[3] += f.w;
      }
    }
  } else if (KIND == 9) {
    // everything at once: LDS-DMA staging, fragments read from LDS, v_cvt_pk_bf16_f32 of loaded values written back to LDS, four independent
    // bf16 MFMA chains accumulating in architectural VGPRs, barriers
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v16f c1 = {}, c2 = {}, c3 = {};
    for (int j = threadIdx.x; j < 2 * 4096; j += 256) (&stage[0][0])[j] = 0.001f * (j % 251);
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
      float *st = stage[i & 1];
      glds16(src + (size_t)((blockIdx.x * 7 + i) % 64) * 4096, (unsigned)(wave * 1024 + lane * 16), st + wave * 256);
      const float g0 = src[(blockIdx.x * 64 + i + lane) & 65535], g1 = src[(blockIdx.x * 64 + i + lane + 64) & 65535];
      unsigned pk;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(g0), "v"(g1));
      reinterpret_cast<unsigned *>(st)[2048 + threadIdx.x] = pk;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const v4f fa = *reinterpret_cast<const v4f *>(&st[((threadIdx.x + u * 64) & 1023) * 4]);
        const v4f fb = *reinterpret_cast<const v4f *>(&stage[(i + 1) & 1][((threadIdx.x + u * 32) & 1023) * 4]);
        const v8bf a = __builtin_bit_cast(v8bf, fa), b = __builtin_bit_cast(v8bf, fb);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(b), "v"(a));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(a));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(b), "v"(b));
      }
      __syncthreads();
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    for (int u = 0; u < 16; ++u) c[u] += c1[u] + c2[u] + c3[u];
  } else if (KIND == 10) {
    // as 8 (bf16 MFMAs accumulating in VGPRs), with operands whose bits change every step (xorshift: every operand bit toggles at random, as
    // real activations / weights do) instead of constant tiny values
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u ra = {threadIdx.x * 2654435761u + 1u, threadIdx.x * 40503u + 7u, blockIdx.x * 2246822519u + 3u, 0x9E3779B9u ^ threadIdx.x};
    v4u rb = {threadIdx.x * 3266489917u + 5u, threadIdx.x * 668265263u + 11u, blockIdx.x * 374761393u + 13u, 0x85EBCA6Bu ^ threadIdx.x};
    v16f c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ra ^= ra << 13; ra ^= ra >> 17; ra ^= ra << 5; rb ^= rb << 13; rb ^= rb >> 17; rb ^= rb << 5;
        const v4u ma = (ra & 0x807F807Fu) | 0x3F003F00u, mb = (rb & 0x807F807Fu) | 0x3F003F00u;      // bf16 values of magnitude 0.5 .. 1, random sign / mantissa
        const v8bf a = __builtin_bit_cast(v8bf, ma), b = __builtin_bit_cast(v8bf, mb);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(b), "v"(a));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(a));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(b), "v"(b));
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    for (int u = 0; u < 16; ++u) c[u] += c1[u] + c2[u] + c3[u];
  } else if (KIND == 8) {
    // bf16 MFMAs whose accumulator lives in ARCHITECTURAL VGPRs (v_mfma ... v[..], v[..], v[..], v[..]: what this package's GEMM kernels
    // issue) instead of AGPRs (what the compiler picks for the loops above, and what library GEMMs use)
    v8bf a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
    v16f c1 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(b), "v"(a));
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    for (int u = 0; u < 16; ++u) c[u] += c1[u];
  } else if (KIND == 7) {
    // no MFMA at all: v_cvt_pk_bf16_f32 (two fp32 -> packed bf16, new on gfx950) in a VALU loop
    float a = 0.001f * threadIdx.x, b = 0.5f + 0.002f * threadIdx.x;
    unsigned r = 0;
    for (int i = 0; i < iters * 16; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        unsigned p;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));
        r ^= p; a = __builtin_fmaf(a, 0.999f, 0.001f); b = __builtin_fmaf(b, 0.998f, 0.002f);
      }
    }
    c[0] = (float)r;
  } else if (KIND == 6) {
    // closer to a GEMM main loop: four independent accumulators, back-to-back bf16 MFMAs, the fragments re-read from LDS every step
    v16f c1 = {}, c2 = {}, c3 = {};
    typedef float v4f __attribute__((ext_vector_type(4)));
    for (int j = threadIdx.x; j < 2 * 4096; j += 256) (&stage[0][0])[j] = 0.001f * (j % 251);
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const v4f fa = *reinterpret_cast<const v4f *>(&stage[(i + u) & 1][((threadIdx.x + u * 64) & 1023) * 4]);
        const v4f fb = *reinterpret_cast<const v4f *>(&stage[(i + u + 1) & 1][((threadIdx.x + u * 32) & 1023) * 4]);
        const v8bf a = __builtin_bit_cast(v8bf, fa), b = __builtin_bit_cast(v8bf, fb);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, c3, 0, 0, 0);
      }
    }
    for (int u = 0; u < 16; ++u) c[u] += c1[u] + c2[u] + c3[u];
  } else if (KIND == 1) {
    v8bf a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
  } else if (KIND == 2) {
    const float a = 0.001f * threadIdx.x, b = 0.002f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
  } else {
    float a = 0.001f * threadIdx.x;
    for (int i = 0; i < iters * 64; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) c[u] = __builtin_fmaf(c[u], 0.999f, a);
    }
  }
  float s = 0.f;
  for (int u = 0; u < 16; ++u) s += c[u];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}


extern "C" int aggressor_launch(int kind, float *sink, const float *src, int grid, int iters, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case 1: hipLaunchKernelGGL(aggressor<1>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
    case 2: hipLaunchKernelGGL(aggressor<2>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
    case 3: hipLaunchKernelGGL(aggressor<3>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
    case 4: hipLaunchKernelGGL(aggressor<4>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
    case 5: hipLaunchKernelGGL(aggressor<5>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
    case 6: hipLaunchKernelGGL(aggressor<6>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
    case 7: hipLaunchKernelGGL(aggressor<7>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
    case 8: hipLaunchKernelGGL(aggressor<8>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
    case 10: hipLaunchKernelGGL(aggressor<10>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
    default: hipLaunchKernelGGL(aggressor<9>, dim3(grid), dim3(256), 0, st, sink, iters, src); break;
  }
  return (int)hipGetLastError();
}
