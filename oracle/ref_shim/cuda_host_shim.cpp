// Fiber scheduler behind oracle/ref_shim/cuda_host_shim.h (TEST INFRASTRUCTURE ONLY).
#include "cuda_host_shim.h"

uint3_shim blockIdx, threadIdx;
dim3 blockDim, gridDim;

namespace ref_shim {
namespace {
constexpr size_t kStack = 64 * 1024;
struct Fiber { ucontext_t ctx; bool done; };
ucontext_t g_main;
Fiber *g_cur = nullptr;
const std::function<void()> *g_body = nullptr;

void trampoline() {
    (*g_body)();
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_main);
}
}  // namespace

void sync_threads() {
    if (g_cur) swapcontext(&g_cur->ctx, &g_main);
}

void launch(const Cfg &c, const std::function<void()> &body) {
    const unsigned nt = c.block.x * c.block.y * c.block.z;
    gridDim = c.grid;
    blockDim = c.block;
    g_body = &body;
    std::vector<Fiber> fibers(nt);
    std::vector<char> stacks(size_t(nt) * kStack);
    for (unsigned bz = 0; bz < c.grid.z; ++bz)
    for (unsigned by = 0; by < c.grid.y; ++by)
    for (unsigned bx = 0; bx < c.grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        for (unsigned t = 0; t < nt; ++t) {
            getcontext(&fibers[t].ctx);
            fibers[t].ctx.uc_stack.ss_sp = stacks.data() + size_t(t) * kStack;
            fibers[t].ctx.uc_stack.ss_size = kStack;
            fibers[t].ctx.uc_link = nullptr;
            fibers[t].done = false;
            makecontext(&fibers[t].ctx, trampoline, 0);
        }
        unsigned live = nt;
        // One pass = one barrier phase.  Threads run from the HIGHEST id to the lowest: the FPS kernels re-read
        // dists_i[0] after their last barrier and thread 0 overwrites it in the next phase without a barrier in
        // between (sampling_cuda_kernel.cu, `old = dists_i[0]` … `dists_i[tid] = besti`); on the GPU every warp's
        // read happens long before thread 0 finishes its distance loop, i.e. thread 0's write is last.
        while (live) {
            for (unsigned t = nt; t-- > 0;) {
                if (fibers[t].done) continue;
                threadIdx = {t % c.block.x, (t / c.block.x) % c.block.y, t / (c.block.x * c.block.y)};
                g_cur = &fibers[t];
                swapcontext(&g_main, &fibers[t].ctx);
                if (fibers[t].done) --live;
            }
        }
        g_cur = nullptr;
    }
}
}  // namespace ref_shim
