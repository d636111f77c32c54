// TEST INFRASTRUCTURE: simulator launchers of the exact-fp32 kernel family (see sim_launch.h)
#include <cmath>
#include <vector>

#include "sim_launch.h"

namespace sl {
void prep_f32(const vk::StepArgs& a, int blocks) { sim::launch(blocks, vk::kWG, 3 * vk::kWG * 4, [&] { vk::step_prep(a); }); }
int main_f32(const vk::StepArgs& a, int wide, bool bwd, int G) {
    const int n = a.n_obj, NW = a.NW, H = a.hidden;
    const bool multi = a.NW < a.NG;
    if (H == 32) {
        const int grid = a.xcd_affine ? 8 * ((n + 7) / 8) * NW : n * NW;
        if (bwd && multi)  sim::launch(grid, vk::kWG, vk::Lds32::BYTES, [&] { vk::step_main_h32<true, true>(a); });
        if (bwd && !multi) sim::launch(grid, vk::kWG, vk::Lds32::BYTES, [&] { vk::step_main_h32<true, false>(a); });
        if (!bwd)          sim::launch(grid, vk::kWG, vk::Lds32::BYTES, [&] { vk::step_main_h32<false, false>(a); });
        return 0;
    }
    const vk::GenLayout GL = vk::gen_layout(H);
    vk::GenArgs ga;
    ga.s = a;
    ga.wave_blocks = vk::gen_wave_blocks(GL.NB);
    std::vector<float> scratch((size_t)n * NW * vk::kWaves * ga.wave_blocks * vk::kBlk, NAN);
    ga.scratch = scratch.data();
    if (wide == 1) {
        if (H % 128 != 0 || G * a.S > vk::kWideTile) return -3;
        ga.s.wide = 1;
        const int lb = vk::LdsWide<4>::bytes(GL.small_n);
        if (bwd) sim::launch(n * NW, 256, lb, [&] { vk::step_main_wide<true, 4>(ga); });
        else     sim::launch(n * NW, 256, lb, [&] { vk::step_main_wide<false, 4>(ga); });
    } else if (bwd) sim::launch(n * NW, vk::kWG, vk::LdsGen::bytes(GL.small_n), [&] { vk::step_main_gen<true>(ga); });
    else     sim::launch(n * NW, vk::kWG, vk::LdsGen::bytes(GL.small_n), [&] { vk::step_main_gen<false>(ga); });
    return 0;
}
void finalize_generic(const vk::FinalizeArgs& f_in, int grid) {
    vk::FinalizeArgs f = f_in;
    const size_t lds = vk::loss_lds_bytes(f.n_obj, f.NW);
    f.loss_stage = vk::loss_stage_cap(lds);
    sim::launch(grid, vk::kWG, (int)lds, [&] { vk::step_finalize(f); });
}
void finalize_h32(const vk::FinalizeArgs& f_in, const vk::FinalizeHot& h, int grid) {
    vk::FinalizeArgs f = f_in;
    const size_t lds = vk::loss_lds_bytes(f.n_obj, f.NW);
    f.loss_stage = vk::loss_stage_cap(lds);
    sim::launch(grid, vk::kWG, (int)lds, [&] { vk::step_finalize_h32(f, h); });
}
}  // namespace sl
