// sim_launch.h - TEST INFRASTRUCTURE: the simulator's launchers, one translation unit per kernel family (mirrors
// vmap_amd/csrc/launch.h) so that the host compiler builds them side by side.
#pragma once
#include "wpair_kernels.h"
#include "wide_kernels.h"
#include "sample_kernels.h"
#include "query_kernels.h"
#include "sim_runtime.h"

namespace sl {
// sim_k_f32.cpp
void prep_f32(const vk::StepArgs& a, int blocks);
int main_f32(const vk::StepArgs& a, int wide, bool bwd, int G);          // step_main_h32 / _gen / _wide<4>
void finalize_generic(const vk::FinalizeArgs& f, int grid);
void finalize_h32(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, int grid);
// sim_k_s32.cpp
void prep_s32(const vk::StepArgs& a);
void main_s32(const vk::StepArgs& a, bool bwd);
void finalize_s32(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, int grid);
// sim_k_ws.cpp / sim_k_wp.cpp
void prep_ws(const vk::WsArgs& wa);
void main_ws(const vk::WsArgs& wa, bool bwd);
void main_wp(const vk::WsArgs& wa, bool bwd);
void finalize_ws(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, const int* tab_wt);
// sim_k_ws8.cpp: hidden 256 (eight waves), called through prep_ws / main_ws / finalize_ws
void prep_ws8(const vk::WsArgs& wa);
void main_ws8(const vk::WsArgs& wa, bool bwd);
void finalize_ws8(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, const int* tab_wt, int grid);
// sim_k_misc.cpp
void sample(const vs::SampleArgs& a, int n_obj, long long rays);
int query(int H, const vk::StepArgs& pack, const vk::QueryArgs& q, int grid);
}  // namespace sl
