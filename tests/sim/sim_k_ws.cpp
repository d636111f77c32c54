// TEST INFRASTRUCTURE: simulator launchers of step_prep_ws / step_main_ws / step_finalize_ws (see sim_launch.h)
#include "sim_launch.h"

namespace sl {
void prep_ws(const vk::WsArgs& wa) {
    const int n = wa.s.n_obj;
    if (wa.s.hidden == 256) return prep_ws8(wa);
    if (wa.s.hidden == 128) sim::launch(vk::ws_prep_grid<4>(wa.s.prep_steps, n), vk::kWG, 3 * vk::kWG * 4, [&] { vk::step_prep_ws<4>(wa); });
    else sim::launch(vk::ws_prep_grid<2>(wa.s.prep_steps, n), vk::kWG, 3 * vk::kWG * 4, [&] { vk::step_prep_ws<2>(wa); });
}
namespace {
template <int NT>
void main_ws_t(const vk::WsArgs& wa, bool bwd);
}
void main_ws(const vk::WsArgs& wa, bool bwd) {
    if (wa.s.hidden == 256) return main_ws8(wa, bwd);
    if (wa.s.tiles == 3 && wa.s.hidden == 128) {             // three-tile rounds: hidden 128 only
        const int grid = wa.s.n_obj * wa.s.NW, lb = vk::LdsWs<4, 3>::LDS_BYTES;
        const bool one = wa.s.NG == wa.s.NW;                 // the single-round specialisation, as the library picks it
        if (wa.s.weights_bf16) {
            if (bwd && one) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, true, false, false, 3, true>(wa); });
            else if (bwd) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, true, false, false, 3>(wa); });
            else     sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, false, false, false, 3>(wa); });
        } else {
            if (bwd && one) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, true, true, false, 3, true>(wa); });
            else if (bwd) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, true, true, false, 3>(wa); });
            else     sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, false, true, false, 3>(wa); });
        }
        return;
    }
    if (wa.s.tiles == 1) main_ws_t<1>(wa, bwd); else main_ws_t<2>(wa, bwd);
}
namespace {
template <int NT>
void main_ws_t(const vk::WsArgs& wa, bool bwd) {
    const int grid = wa.s.n_obj * wa.s.NW;
    if (wa.s.hidden == 128) {
        const int lb = vk::ImgWs<4>::LDS_BYTES;
        const bool one = wa.s.NG == wa.s.NW;
        if (bwd && one) {
            if (wa.s.weights_bf16) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, true, false, false, NT, true>(wa); });
            else sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, true, true, false, NT, true>(wa); });
            return;
        }
        if (wa.s.weights_bf16) {
            if (bwd) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, true, false, false, NT>(wa); });
            else     sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, false, false, false, NT>(wa); });
        } else {
            if (bwd) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, true, true, false, NT>(wa); });
            else     sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<4, false, true, false, NT>(wa); });
        }
    } else {
        const int lb = vk::ImgWs<2>::LDS_BYTES;
        if (wa.s.weights_bf16) {
            if (bwd) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<2, true, false, false, NT>(wa); });
            else     sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<2, false, false, false, NT>(wa); });
        } else {
            if (bwd) sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<2, true, true, false, NT>(wa); });
            else     sim::launch(grid, vk::kWG, lb, [&] { vk::step_main_ws<2, false, true, false, NT>(wa); });
        }
    }
}
}  // namespace
void finalize_ws(const vk::FinalizeArgs& f_in, const vk::FinalizeHot& h, const int* tab_wt) {
    vk::FinalizeArgs f = f_in;
    f.loss_stage = vk::loss_stage_cap(vk::kFinThreads * 16);
    f.xcd_affine = f.n_obj >= 8 ? 1 : 0;          // as the library's launcher
    const int grid = vk::ws_finalize_grid(f.n_obj, f.PR, vk::kFinQuads, f.xcd_affine);
    if (f.hidden == 256) return finalize_ws8(f, h, tab_wt, grid);
    if (!f.ws_grouped) {            // the form the library's launcher picks for many blocks / few rows (here: on request)
        constexpr int Q = vk::kFinQuadsWide;
        const int lds = vk::kFinGroups * Q * 16, gw = vk::ws_finalize_grid(f.n_obj, f.PR, Q, f.xcd_affine);
        f.loss_stage = vk::loss_stage_cap(lds);
        if (f.hidden == 128) sim::launch(gw, Q, lds, [&] { vk::step_finalize_ws<4, Q, 1>(f, h, tab_wt); });
        else sim::launch(gw, Q, lds, [&] { vk::step_finalize_ws<2, Q, 1>(f, h, tab_wt); });
        return;
    }
    if (f.hidden == 128) sim::launch(grid, vk::kFinThreads, vk::kFinThreads * 16, [&] { vk::step_finalize_ws<4>(f, h, tab_wt); });
    else sim::launch(grid, vk::kFinThreads, vk::kFinThreads * 16, [&] { vk::step_finalize_ws<2>(f, h, tab_wt); });
}
}  // namespace sl
