// launch.h - host-side launchers of the kernel families, one translation unit per family so that hipcc compiles them side
// by side (build(): vmapstep.hip = the C ABI, k_f32.hip, k_s32.hip, k_ws.hip, k_ws8.hip, k_wp.hip, k_misc.hip; no device code crosses
// a unit, so no relocatable device code is needed).  Every function only ENQUEUES on `st` and returns a vmapstep status.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "query_kernels.h"
#include "sample_kernels.h"
#include "step_kernels.h"

namespace vk { struct WsArgs; }

namespace vl {

// vmapstep.hip
int fail(int code, const char* fmt, ...);                                        // sets vmapstep_last_error(), returns code
int ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what);     // hipFuncSetAttribute once per (device, kernel)
int launched(const char* what);                                                  // hipGetLastError() -> status
// Measurement (vmapstep_profile_train_steps): while set on the calling thread, the dominant kernel's launch carries these two
// events - they take the dispatch's own begin / end timestamps (what a rocprofv3 kernel trace reports), not the stream's
struct DispatchEvents { hipEvent_t start, stop; };
extern thread_local const DispatchEvents* g_dispatch_events;
#define VL_LAUNCH_MAIN(kern, grid, block, lds, st, ...)                                                                     \
    do {                                                                                                                    \
        if (const vl::DispatchEvents* de_ = vl::g_dispatch_events)                                                          \
            hipExtLaunchKernelGGL(kern, grid, block, lds, st, de_->start, de_->stop, 0, __VA_ARGS__);                       \
        else                                                                                                                \
            hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                                    \
    } while (0)

// k_f32.hip: the exact-fp32 matrix-instruction kernels (step_main_h32, step_main_gen, step_main_wide<4>) and the
// width-generic prep / finalize kernels
int main_f32(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st);      // by a.hidden / a.wide
int prep_f32(const vk::StepArgs& a, int blocks, hipStream_t st);                 // step_prep: blocks = n_steps + pack blocks
int finalize_generic(const vk::FinalizeArgs& f, int grid, hipStream_t st);       // step_finalize
int finalize_h32(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, int grid, hipStream_t st);

// k_s32.hip: hidden 32 on the bf16 matrix pipe with split operands
int main_s32(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st);
int prep_s32(const vk::StepArgs& a, int n_steps, hipStream_t st);
int finalize_s32(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, int grid, hipStream_t st);

// k_ws.hip / k_wp.hip: hidden 64 / 128 on the bf16 matrix pipe (one wave / two waves per output block)
int main_ws(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st);
int main_wp(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st);
int prep_ws(const vk::StepArgs& a, int n_steps, hipStream_t st);
int finalize_ws(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, const int* tab_wt, hipStream_t st);
// k_ws8.hip: hidden 256 on the same scheme with eight waves (called through main_ws / prep_ws / finalize_ws)
int main_ws8(const vk::StepArgs& a, bool bwd, hipStream_t st);
int prep_ws8(const vk::WsArgs& ga, int n_steps, hipStream_t st);
int finalize_ws8(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, const int* tab_wt, hipStream_t st);

// k_misc.hip: inference query and the frame sampler
int query_points(int hidden, const vk::StepArgs& pack, const vk::QueryArgs& q, long long n_points, hipStream_t st);
int sample_frame(const vs::SampleArgs& a, int n_obj, long long rays_per_object, hipStream_t st);   // a.obj_max != null: the split form (two launches)

}  // namespace vl
