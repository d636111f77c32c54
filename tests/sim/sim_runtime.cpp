// sim_runtime.cpp - TEST INFRASTRUCTURE: fiber scheduler of the CPU SIMT executor (see sim_runtime.h).
#include "sim_runtime.h"

#include <algorithm>
#include <atomic>
#include <thread>

namespace sim {

thread_local Block* g_block = nullptr;
thread_local Fiber* g_cur = nullptr;
thread_local ucontext_t g_sched;
thread_local Idx3 g_blockIdx{0, 0, 0};
Idx3 g_gridDim{1, 1, 1}, g_blockDim{1, 1, 1};
thread_local long g_yields = 0;

static const std::function<void()>* g_body = nullptr;

static void trampoline() {
    (*g_body)();
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_sched);
}

// the workgroups of a launch are independent (no kernel of this repository communicates between workgroups inside a launch), so
// they are dealt to a few OS threads; inside a workgroup everything stays on one thread, cooperative and deterministic
static void run_blocks(unsigned first, unsigned stride, unsigned grid, unsigned block, size_t lds_bytes) {
    constexpr size_t kStack = 256 * 1024;
    Block blk;
    blk.nthreads = (int)block;
    blk.fibers.resize(block);
    blk.lds.resize(lds_bytes + 64);
    std::vector<char> stacks((size_t)block * kStack);
    for (unsigned b = first; b < grid; b += stride) {
        g_blockIdx = {b, 0, 0};
        std::memset(blk.lds.data(), 0xFF, blk.lds.size());   // 0xFFFFFFFF = NaN: poison
        blk.block_bar = Barrier{(int)block, 0, 0};
        int nw = (int)((block + kWave - 1) / kWave);
        if (nw > kMaxWaves) { std::fprintf(stderr, "sim: too many waves\n"); std::abort(); }
        for (int w = 0; w < nw; ++w) {
            int n = (int)block - w * kWave;
            blk.wave_bar[w] = Barrier{n > kWave ? kWave : n, 0, 0};
        }
        g_block = &blk;
        for (unsigned t = 0; t < block; ++t) {
            Fiber& f = blk.fibers[t];
            f.done = false;
            f.tid = t;
            f.stack = stacks.data() + (size_t)t * kStack;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = kStack;
            f.ctx.uc_link = &g_sched;
            makecontext(&f.ctx, (void (*)())trampoline, 0);
        }
        unsigned remaining = block;
        long guard = 0;
        while (remaining) {
            for (unsigned t = 0; t < block; ++t) {
                Fiber& f = blk.fibers[t];
                if (f.done) continue;
                g_cur = &f;
                swapcontext(&g_sched, &f.ctx);
                if (f.done) --remaining;
            }
            if (++guard > 50000000L) { std::fprintf(stderr, "sim: deadlock (barrier mismatch?)\n"); std::abort(); }
        }
    }
    g_block = nullptr;
    g_cur = nullptr;
}

void launch(unsigned grid, unsigned block, size_t lds_bytes, const std::function<void()>& body) {
    g_gridDim = {grid, 1, 1};
    g_blockDim = {block, 1, 1};
    g_body = &body;
    unsigned nt = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = std::getenv("VMSIM_THREADS")) nt = (unsigned)std::max(1, std::atoi(e));
    nt = std::min(nt, grid);
    if (nt <= 1) { run_blocks(0, 1, grid, block, lds_bytes); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(run_blocks, t, nt, grid, block, lds_bytes);
    for (auto& x : th) x.join();
}

}  // namespace sim
