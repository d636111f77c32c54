// TEST INFRASTRUCTURE: simulator launchers of the hidden-32 split-bf16 kernels (see sim_launch.h)
#include "sim_launch.h"

namespace sl {
void prep_s32(const vk::StepArgs& a) { sim::launch(a.prep_steps + a.n_obj * vk::kSplitPackBlocks, vk::kWG, 3 * vk::kWG * 4, [&] { vk::step_prep_s32(a); }); }
void main_s32(const vk::StepArgs& a, bool bwd) {
    const bool multi = a.NW < a.NG;
    const int grid = a.xcd_affine ? 8 * ((a.n_obj + 7) / 8) * a.NW : a.n_obj * a.NW;
    const int lb = vk::Img32s::LDS_BYTES;
    if (a.weights_bf16) {
        if (bwd && multi)  sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_s32<true, true, false, false>(a); });
        if (bwd && !multi) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_s32<true, false, false, false>(a); });
        if (!bwd)          sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_s32<false, false, false, false>(a); });
    } else if (bwd && a.bwd6) {               // the six-product backward (tuning.kernel = VMAPSTEP_KERNEL_S32_BWD6)
        if (multi)  sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_s32<true, true, false, true, true>(a); });
        else        sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_s32<true, false, false, true, true>(a); });
    } else {
        if (bwd && multi)  sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_s32<true, true, false, true>(a); });
        if (bwd && !multi) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_s32<true, false, false, true>(a); });
        if (!bwd)          sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_s32<false, false, false, true>(a); });
    }
}
void finalize_s32(const vk::FinalizeArgs& f_in, const vk::FinalizeHot& h, int grid) {
    vk::FinalizeArgs f = f_in;
    const size_t lds = vk::loss_lds_bytes(f.n_obj, f.NW);
    f.loss_stage = vk::loss_stage_cap(lds);
    sim::launch(grid, vk::kWG, (int)lds, [&] { vk::step_finalize_s32(f, h); });
}
}  // namespace sl
