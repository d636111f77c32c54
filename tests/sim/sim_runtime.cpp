// sim_runtime.cpp - TEST INFRASTRUCTURE: fiber scheduler of the CPU SIMT executor (see sim_runtime.h).
#include "sim_runtime.h"

namespace sim {

Block* g_block = nullptr;
Fiber* g_cur = nullptr;
ucontext_t g_sched;
Idx3 g_blockIdx{0, 0, 0}, g_gridDim{1, 1, 1}, g_blockDim{1, 1, 1};
long g_yields = 0;

static const std::function<void()>* g_body = nullptr;

static void trampoline() {
    (*g_body)();
    g_cur->done = true;
    swapcontext(&g_cur->ctx, &g_sched);
}

void launch(unsigned grid, unsigned block, size_t lds_bytes, const std::function<void()>& body) {
    constexpr size_t kStack = 256 * 1024;
    Block blk;
    blk.nthreads = (int)block;
    blk.fibers.resize(block);
    blk.lds.resize(lds_bytes + 64);
    std::vector<char> stacks((size_t)block * kStack);
    g_gridDim = {grid, 1, 1};
    g_blockDim = {block, 1, 1};
    g_body = &body;
    for (unsigned b = 0; b < grid; ++b) {
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
            unsigned progressed = 0;
            for (unsigned t = 0; t < block; ++t) {
                Fiber& f = blk.fibers[t];
                if (f.done) continue;
                g_cur = &f;
                swapcontext(&g_sched, &f.ctx);
                if (f.done) { --remaining; ++progressed; }
            }
            if (++guard > 50000000L) { std::fprintf(stderr, "sim: deadlock (barrier mismatch?)\n"); std::abort(); }
        }
    }
    g_block = nullptr;
    g_cur = nullptr;
}

}  // namespace sim
