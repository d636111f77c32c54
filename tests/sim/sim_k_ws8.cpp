// TEST INFRASTRUCTURE: simulator launchers of the hidden-256 forms of step_prep_ws / step_main_ws / step_finalize_ws (eight waves)
#include "sim_launch.h"

namespace sl {
void prep_ws8(const vk::WsArgs& wa) {
    sim::launch(vk::ws_prep_grid<8>(wa.s.prep_steps, wa.s.n_obj), vk::kWG, 3 * vk::kWG * 4, [&] { vk::step_prep_ws<8>(wa); });
}
void main_ws8(const vk::WsArgs& wa, bool bwd) {
    using LD = vk::LdsWs<8, 1>;
    const int grid = wa.s.n_obj * wa.s.NW, lb = LD::LDS_BYTES;
    const bool one = wa.s.NG == wa.s.NW;                      // the single-round specialisation, as the library picks it
    if (wa.s.weights_bf16) {
        if (bwd && one) sim::launch(grid, LD::NTH, lb, [&] { vk::step_main_ws<8, true, false, false, 1, true>(wa); });
        else if (bwd) sim::launch(grid, LD::NTH, lb, [&] { vk::step_main_ws<8, true, false, false, 1, false>(wa); });
        else sim::launch(grid, LD::NTH, lb, [&] { vk::step_main_ws<8, false, false, false, 1, false>(wa); });
    } else {
        if (bwd && one) sim::launch(grid, LD::NTH, lb, [&] { vk::step_main_ws<8, true, true, false, 1, true>(wa); });
        else if (bwd) sim::launch(grid, LD::NTH, lb, [&] { vk::step_main_ws<8, true, true, false, 1, false>(wa); });
        else sim::launch(grid, LD::NTH, lb, [&] { vk::step_main_ws<8, false, true, false, 1, false>(wa); });
    }
}
void finalize_ws8(const vk::FinalizeArgs& f_in, const vk::FinalizeHot& h, const int* tab_wt, int grid) {
    vk::FinalizeArgs f = f_in;
    if (!f.ws_grouped) {
        constexpr int Q = vk::kFinQuadsWide;
        const int lds = vk::kFinGroups * Q * 16;
        f.loss_stage = vk::loss_stage_cap(lds);
        sim::launch(vk::ws_finalize_grid(f.n_obj, f.PR, Q, f.xcd_affine), Q, lds, [&] { vk::step_finalize_ws<8, Q, 1>(f, h, tab_wt); });
        return;
    }
    sim::launch(grid, vk::kFinThreads, vk::kFinThreads * 16, [&] { vk::step_finalize_ws<8>(f, h, tab_wt); });
}
}  // namespace sl
