// TEST INFRASTRUCTURE ONLY (see oracle/README in DESIGN.md §4): lets g++ compile the reference's
// pointops *.cu files, unmodified and where they lie under /root/reference, as HOST code, so that the
// reference's own kernels can be executed on the CPU and the oracle restatements pinned against them.
//
// What it provides: the CUDA spellings those files use (__global__, dim3, blockIdx/threadIdx/blockDim/gridDim,
// __shared__, __syncthreads, atomicAdd, min/max, the cudaError_t boilerplate) and `ref_shim::launch`,
// which runs a grid block after block; the threads of a block are cooperative fibers (ucontext) that are
// resumed round-robin (highest thread id first, see the .cpp), `__syncthreads()` = yield until every live fiber arrived.
// That is a legal CUDA schedule, so barrier kernels (the FPS tree reduction, sampling_cuda_kernel.cu) run
// with their real phase structure.  `kernel<<<grid, block, shmem[, stream]>>>(args);` is the one spelling
// that is not C++: oracle/Makefile.ref rewrites that token sequence (sed, streamed into g++, nothing is written
// to disk) into `ref_shim::launch(ref_shim::cfg(grid, block, shmem[, stream]), [=]{ kernel(args); });`.
// Arithmetic: compiled with -ffp-contract=off, i.e. the source's operations as written (nvcc's default
// -fmad=true may contract a*b+c; which products it fuses is not observable without nvcc).
#pragma once
#include <ucontext.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_shim { unsigned x, y, z; };
extern uint3_shim blockIdx, threadIdx;
extern dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

typedef void *cudaStream_t;
typedef int cudaError_t;
enum { cudaSuccess = 0 };
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline const char *cudaGetErrorString(cudaError_t) { return "host shim"; }

// CUDA's global-namespace overloads
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float atomicAdd(float *p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int *p, int v) { int o = *p; *p = o + v; return o; }

namespace ref_shim {
struct Cfg { dim3 grid, block; };
static inline Cfg cfg(dim3 g, dim3 b, size_t = 0, cudaStream_t = nullptr) { return Cfg{g, b}; }
void launch(const Cfg &c, const std::function<void()> &body);
void sync_threads();
}
#define __syncthreads() ref_shim::sync_threads()
