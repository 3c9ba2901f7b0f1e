// ATTEMPT at a standalone reproducer (no torch, no library of this repository) of the round-6 two-stream hazard on MI355X / gfx950 -- it does
// NOT reproduce, and cannot: on the round's boxes its two streams never overlapped (wall time = aggressor + victims, whatever the queue
// count, grid size or launching thread), so its zeros say nothing.  Kept for whoever finds out why.  The working setup is
// tools/victim_probe.py (torch streams).  What it was meant to show:
//   v_pk_add_f32 with op_sel on its SECOND source, in a kernel on one stream, beside a kernel that issues bf16 (or fp32) MFMAs on another.
//   hipcc --offload-arch=gfx950 -O3 -o tools/victim/standalone tools/victim/standalone.hip && tools/victim/standalone [launches] [aggressor]
//   aggressor: 0 none, 1 v_mfma_f32_32x32x16_bf16 loop, 2 v_mfma_f32_32x32x2_f32 loop, 3 fp32 FMA loop (no MFMA), 4 LDS-DMA loop,
//   5 LDS-DMA + bf16 MFMA loop, 6 four independent bf16 MFMA chains fed from LDS,
//   7 v_cvt_pk_bf16_f32 loop (no MFMA), 8 bf16 MFMAs with the accumulator in architectural VGPRs,
//   9 a GEMM-like loop with all of it
// Prints, per victim form, how many launches wrote other values than the same kernel alone, and the lanes (thread index mod 64) that differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#include <thread>
#include <atomic>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int FORM>
__global__ void __launch_bounds__(256) victim(float *out, int iters, float scale) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float x = 0.25f + (float)(t % 977) * (1.f / 1024.f), y = 0.5f + (float)(t % 313) * (1.f / 512.f), acc = 0.f;
  for (int i = 0; i < iters; ++i) {
    v2f a = {x, y}, b = {y * 0.5f, x * 0.5f};
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
      if (FORM == 0) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(a) : "v"(a), "v"(b));      // src1's HIGH half -> low element
      if (FORM == 1) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(a) : "v"(a), "v"(b));      // src0's instead
      if (FORM == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
      b = b * 0.75f; a = a * 0.5f;
    }
    x = a.x - (float)(int)a.x + 0.25f; y = b.y - (float)(int)b.y + 0.5f; acc = __builtin_fmaf(a.y, b.x, acc * 0.5f);
  }
  out[t] = acc + x + y;
}

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

int main(int argc, char **argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20000, kind = argc > 2 ? atoi(argv[2]) : 1;
  const int n = 4096, per = 8, ring = 200, viters = 60;
  const int agrid = getenv("AGRID") ? atoi(getenv("AGRID")) : 1024;      // workgroups of an aggressor launch
  const bool with_victims = argc > 3 ? atoi(argv[3]) != 0 : true;      // 0: the aggressor launches alone (wall-clock reference for the overlap)
  hipStream_t sa, sv;
  // (STREAMS=k in the environment: create k streams and use the first and the last -- which hardware queue a stream lands on is the runtime's choice)
  const int nstreams = getenv("STREAMS") ? atoi(getenv("STREAMS")) : 2;
  std::vector<hipStream_t> pool(nstreams);
  for (auto &st : pool) CHECK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, 0));
  sa = pool.front(); sv = pool.back();
  float *out, *ref, *sink, *src;
  CHECK(hipMalloc(&src, 64 * 4096 * 4)); CHECK(hipMemset(src, 0, 64 * 4096 * 4));
  CHECK(hipMalloc(&out, (size_t)ring * n * 4)); CHECK(hipMalloc(&ref, n * 4)); CHECK(hipMalloc(&sink, 1024));
  std::vector<float> h((size_t)ring * n), hr(n);
  const char *names[3] = {"v_pk_add_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[1,0]", "v_pk_add_f32 (no op_sel)"};
  const char *agg[10] = {"nothing", "a v_mfma_f32_32x32x16_bf16 loop", "a v_mfma_f32_32x32x2_f32 loop", "an fp32 FMA loop", "an LDS-DMA loop", "an LDS-DMA + bf16 MFMA loop", "4 independent bf16 MFMA chains fed from LDS", "a v_cvt_pk_bf16_f32 loop (no MFMA)", "bf16 MFMAs accumulating in VGPRs", "a GEMM-like loop (LDS-DMA, LDS fragments, cvt_pk, VGPR-accumulating bf16 MFMAs, barriers)"};
  for (int form = 0; form < 3; ++form) {
    auto launch_victim = [&](float *dst) {
      if (form == 0) hipLaunchKernelGGL(victim<0>, dim3(n / 256), dim3(256), 0, sv, dst, viters, 0.7071f);
      if (form == 1) hipLaunchKernelGGL(victim<1>, dim3(n / 256), dim3(256), 0, sv, dst, viters, 0.7071f);
      if (form == 2) hipLaunchKernelGGL(victim<2>, dim3(n / 256), dim3(256), 0, sv, dst, viters, 0.7071f);
    };
    launch_victim(ref); CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(hr.data(), ref, n * 4, hipMemcpyDeviceToHost));
    int bad = 0, done = 0; unsigned long long lanes = 0;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms_total = 0.f;
    const auto w0 = std::chrono::steady_clock::now();
    while (done < launches) {
      CHECK(hipEventRecord(e0, sv));
      std::atomic<bool> stop{false};
      std::thread feeder([&] {                                   // keeps ~4 aggressor launches in flight on its own stream
        CHECK(hipSetDevice(0));
        while (!stop.load()) {
          for (int q = 0; q < 4; ++q) {
            if (kind == 1) hipLaunchKernelGGL(aggressor<1>, dim3(agrid), dim3(256), 0, sa, sink, 400, src);
            if (kind == 2) hipLaunchKernelGGL(aggressor<2>, dim3(agrid), dim3(256), 0, sa, sink, 100, src);
            if (kind == 3) hipLaunchKernelGGL(aggressor<3>, dim3(agrid), dim3(256), 0, sa, sink, 100, src);
            if (kind == 4) hipLaunchKernelGGL(aggressor<4>, dim3(agrid), dim3(256), 0, sa, sink, 200, src);
            if (kind == 5) hipLaunchKernelGGL(aggressor<5>, dim3(agrid), dim3(256), 0, sa, sink, 200, src);
            if (kind == 6) hipLaunchKernelGGL(aggressor<6>, dim3(agrid), dim3(256), 0, sa, sink, 100, src);
            if (kind == 7) hipLaunchKernelGGL(aggressor<7>, dim3(agrid), dim3(256), 0, sa, sink, 100, src);
            if (kind == 8) hipLaunchKernelGGL(aggressor<8>, dim3(agrid), dim3(256), 0, sa, sink, 400, src);
            if (kind == 9) hipLaunchKernelGGL(aggressor<9>, dim3(agrid), dim3(256), 0, sa, sink, 200, src);
          }
          if (kind == 0) break;
          CHECK(hipStreamSynchronize(sa));
        }
      });
      if (with_victims) for (int k = 0; k < ring; ++k) launch_victim(out + (size_t)k * n);
      CHECK(hipEventRecord(e1, sv));
      CHECK(hipStreamSynchronize(sv));
      stop.store(true); feeder.join();
      CHECK(hipDeviceSynchronize());
      { float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms_total += ms; }
      CHECK(hipMemcpy(h.data(), out, (size_t)ring * n * 4, hipMemcpyDeviceToHost));
      for (int l = 0; l < ring; ++l) {
        bool b = false;
        for (int t = 0; t < n; ++t) if (memcmp(&h[(size_t)l * n + t], &hr[t], 4)) { b = true; lanes |= 1ull << (t % 64); }
        bad += b;
      }
      done += ring;
    }
    printf("%-28s beside %s: %d of %d launches wrote other values; lanes that differed: %016llx; victim stream %.1f us per launch; wall %.0f ms\n", names[form], agg[kind], bad, done, lanes, ms_total * 1e3f / done,
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
  }
  return 0;
}
