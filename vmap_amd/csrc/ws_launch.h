// ws_launch.h - the ONE place step_finalize_ws is launched from (k_ws.hip: hidden 64 / 128, k_ws8.hip: hidden 256), so that what depends
// on the launch's LDS size - FinalizeArgs::loss_stage, the loss block's staging capacity - is derived where that size is passed, and the
// choice between the grouped and the one-thread-per-quad form is made by one predicate for every width.
#pragma once
#include "launch.h"
#include "wsplit_kernels.h"

namespace vl {

// many blocks, few rows per object: the form in which one thread per quad walks all row groups (see step_finalize_ws)
inline bool finalize_one_thread_per_quad(const vk::FinalizeArgs& f) {
    return !f.ws_grouped && f.NW <= 16 && (long long)f.n_obj * vk::ws_finalize_blocks(f.PR) >= 512;
}

// step_finalize_ws<NB, Q, PG> on `blocks` row blocks (+ 1: the loss block) of `threads` threads with `lds` bytes
template <int NB, int Q, int PG>
inline int launch_finalize_ws(vk::FinalizeArgs f, const vk::FinalizeHot& h, const int* tab_wt, int blocks, int threads, size_t lds, hipStream_t st) {
    f.loss_stage = vk::loss_stage_cap(lds);          // the row blocks' LDS doubles as the loss block's staging area
    hipLaunchKernelGGL((vk::step_finalize_ws<NB, Q, PG>), dim3(blocks + 1), dim3(threads), lds, st, f, h, tab_wt);
    return launched("step_finalize_ws");
}
// the one-thread-per-quad form
template <int NB>
inline int finalize_wide(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, const int* tab_wt, hipStream_t st) {
    constexpr int Q = vk::kFinQuadsWide;
    return launch_finalize_ws<NB, Q, 1>(f, h, tab_wt, vk::ws_finalize_grid(f.n_obj, f.PR, Q, f.xcd_affine) - 1, Q, (size_t)vk::kFinGroups * Q * 4 * sizeof(float), st);
}
// the grouped form, kFinQuads quads per block
template <int NB>
inline int finalize_grouped(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, const int* tab_wt, hipStream_t st) {
    return launch_finalize_ws<NB, vk::kFinQuads, vk::kFinGroups>(f, h, tab_wt, vk::ws_finalize_grid(f.n_obj, f.PR, vk::kFinQuads, f.xcd_affine) - 1, vk::kFinThreads,
                                                    vk::kFinThreads * 4 * sizeof(float), st);
}

}  // namespace vl
