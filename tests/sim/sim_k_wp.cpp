// TEST INFRASTRUCTURE: simulator launcher of step_main_wp (see sim_launch.h)
#include "sim_launch.h"

namespace sl {
void main_wp(const vk::WsArgs& wa_in, bool bwd) {
    vk::WsArgs wa = wa_in;
    wa.s.xcd_affine = wa.s.n_obj >= 8 ? 1 : 0;             // as the library's launcher (k_wp.hip)
    const int grid = (wa.s.xcd_affine ? 8 * ((wa.s.n_obj + 7) / 8) : wa.s.n_obj) * wa.s.NW;
    if (wa.s.hidden == 128) {
        const int lb = vk::LdsWp<4>::LDS_BYTES;
        if (wa.s.weights_bf16) {
            if (bwd) sim::launch(grid, 512, lb, [&] { vk::step_main_wp<4, true, false>(wa); });
            else     sim::launch(grid, 512, lb, [&] { vk::step_main_wp<4, false, false>(wa); });
        } else {
            if (bwd) sim::launch(grid, 512, lb, [&] { vk::step_main_wp<4, true, true>(wa); });
            else     sim::launch(grid, 512, lb, [&] { vk::step_main_wp<4, false, true>(wa); });
        }
    } else {
        const int lb = vk::LdsWp<2>::LDS_BYTES;
        if (wa.s.weights_bf16) {
            if (bwd) sim::launch(grid, 256, lb, [&] { vk::step_main_wp<2, true, false>(wa); });
            else     sim::launch(grid, 256, lb, [&] { vk::step_main_wp<2, false, false>(wa); });
        } else {
            if (bwd) sim::launch(grid, 256, lb, [&] { vk::step_main_wp<2, true, true>(wa); });
            else     sim::launch(grid, 256, lb, [&] { vk::step_main_wp<2, false, true>(wa); });
        }
    }
}
}  // namespace sl
