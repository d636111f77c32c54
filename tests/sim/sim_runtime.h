// sim_runtime.h - TEST INFRASTRUCTURE: a tiny SIMT executor for running the HIP kernel *source* on a CPU.
//
// Workgroups are independent and dealt to a few OS threads; inside a workgroup every work-item is a ucontext fiber
// scheduled round-robin on ONE thread, cooperative (no preemption), fully deterministic.  A wave is 64 consecutive fibers; wave-collective
// operations (the matrix instruction, lane exchanges) and barriers are rendezvous points at which a
// fiber yields until all participants have arrived.  LDS is a per-block byte array filled with
// signalling garbage (NaN patterns) so that reads of unwritten shared memory surface as NaNs.
//
// This models the *interface* documented in vmap_amd/csrc/wave_ops.h (operand/accumulator lane maps of
// v_mfma_f32_32x32x2_f32 from /opt/skills/guides/cdna_hip_programming.md section 3); it does not model timing.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace sim {

struct Idx3 { unsigned x, y, z; };

struct Barrier { int n = 0, count = 0; unsigned gen = 0; };

constexpr int kWave = 64;
constexpr int kMaxWaves = 16;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    unsigned tid = 0;
};

struct Block {
    int nthreads = 0;
    std::vector<Fiber> fibers;
    Barrier block_bar;
    Barrier wave_bar[kMaxWaves];
    float xa[kMaxWaves][kWave];
    float xb[kMaxWaves][kWave];
    unsigned xa4[kMaxWaves][kWave][4];      // bf16 matrix-instruction operands (four dwords per lane)
    unsigned xb4[kMaxWaves][kWave][4];
    const void* xp[kMaxWaves][kWave];       // per-lane addresses of a transposing LDS read
    std::vector<unsigned char> lds;
};

extern thread_local Block* g_block;
extern thread_local Fiber* g_cur;
extern thread_local ucontext_t g_sched;
extern thread_local Idx3 g_blockIdx;
extern Idx3 g_gridDim, g_blockDim;
extern thread_local long g_yields;

inline void yield() {
    ++g_yields;
    swapcontext(&g_cur->ctx, &g_sched);
}

inline void barrier_wait(Barrier& b) {
    unsigned gen = b.gen;
    if (++b.count == b.n) {
        b.count = 0;
        ++b.gen;
    } else {
        while (b.gen == gen) yield();
    }
}

inline unsigned tid() { return g_cur->tid; }
inline int wave_id() { return (int)(g_cur->tid / kWave); }
inline int lane_id() { return (int)(g_cur->tid % kWave); }
inline void wave_barrier() { barrier_wait(g_block->wave_bar[wave_id()]); }

void launch(unsigned grid, unsigned block, size_t lds_bytes, const std::function<void()>& body);

}  // namespace sim
