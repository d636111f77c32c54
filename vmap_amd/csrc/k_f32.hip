// k_f32.hip - the exact-fp32 matrix-instruction kernels (v_mfma_f32_32x32x2_f32): step_main_h32 (hidden 32, kept as the
// tuning.kernel = VMAPSTEP_KERNEL_H32_F32 A/B reference of the default split-bf16 kernel), step_main_gen (any hidden = 32 k
// <= 256), step_main_wide<4> (hidden 128 / 256 with few tiles), and the width-generic step_prep / step_finalize /
// step_finalize_h32.  gfx950 only.
#include "launch.h"
#include "wide_kernels.h"

namespace vl {

namespace {
template <bool BWD, bool MULTI, bool STAMPS>
int main_h32(const vk::StepArgs& a, hipStream_t st) {
    auto kern = vk::step_main_h32<BWD, MULTI, STAMPS>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), vk::Lds32::BYTES, "step_main_h32")) return rc;
    const int grid = a.xcd_affine ? 8 * ((a.n_obj + 7) / 8) * a.NW : a.n_obj * a.NW;
    VL_LAUNCH_MAIN(kern, dim3(grid), dim3(vk::kWG), vk::Lds32::BYTES, st, a);
    return launched("step_main_h32");
}
template <bool BWD>
int main_gen(const vk::StepArgs& a, hipStream_t st) {
    auto kern = vk::step_main_gen<BWD>;
    const vk::GenLayout GL = vk::gen_layout(a.hidden);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), vk::LdsGen::bytes(vk::gen_layout(256).small_n), "step_main_gen")) return rc;
    vk::GenArgs ga;
    ga.s = a;
    ga.scratch = a.gen_scratch;
    ga.wave_blocks = vk::gen_wave_blocks(GL.NB);
    VL_LAUNCH_MAIN(kern, dim3(a.n_obj * a.NW), dim3(vk::kWG), vk::LdsGen::bytes(GL.small_n), st, ga);
    return launched("step_main_gen");
}
#ifdef VMAPSTEP_AB
template <bool BWD>
int main_wide(const vk::StepArgs& a, hipStream_t st) {
    using LW = vk::LdsWide<4>;
    auto kern = vk::step_main_wide<BWD, 4>;
    const vk::GenLayout GL = vk::gen_layout(a.hidden);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LW::bytes(vk::gen_layout(256).small_n), "step_main_wide")) return rc;
    vk::GenArgs ga;
    ga.s = a;
    ga.scratch = a.gen_scratch;
    ga.wave_blocks = vk::gen_wave_blocks(GL.NB);
    VL_LAUNCH_MAIN(kern, dim3(a.n_obj * a.NW), dim3(64 * LW::NWAVES), LW::bytes(GL.small_n), st, ga);
    return launched("step_main_wide");
}
#endif
}  // namespace

int main_f32(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st) {
    if (stamps) return fail(-2, "phase stamps exist in the bf16-pipe kernels only (step_main_s32 / _ws / _wp)");
    if (a.hidden != 32) {
#ifdef VMAPSTEP_AB
        if (a.wide == 1) return bwd ? main_wide<true>(a, st) : main_wide<false>(a, st);
#else
        if (a.wide == 1) return fail(-2, "this kernel form ships in the measurement build only (tests/tools/libvmapstep_ab.so: phase stamps and A/B forms no automatic plan launches)");
#endif
        return bwd ? main_gen<true>(a, st) : main_gen<false>(a, st);
    }
    const bool multi = a.NW < a.NG;
    if (bwd) return multi ? main_h32<true, true, false>(a, st) : main_h32<true, false, false>(a, st);
#ifdef VMAPSTEP_AB
    return multi ? main_h32<false, true, false>(a, st) : main_h32<false, false, false>(a, st);
#else
    // forward only (vmapstep_render) on the exact-fp32 kernel: an A/B form (the training forms above back bench.py's value_exact_fp32_kernel)
    return fail(-2, "this kernel form ships in the measurement build only (tests/tools/libvmapstep_ab.so: phase stamps and A/B forms no automatic plan launches)");
#endif
}

int prep_f32(const vk::StepArgs& a, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(vk::step_prep<>, dim3(blocks), dim3(vk::kWG), 3 * vk::kWG * sizeof(int), st, a);
    return launched("step_prep");
}

int finalize_generic(const vk::FinalizeArgs& f, int grid, hipStream_t st) {
    vk::FinalizeArgs g = f;
    const size_t lds = vk::loss_lds_bytes(f.n_obj, f.NW);
    g.loss_stage = vk::loss_stage_cap(lds);
    hipLaunchKernelGGL(vk::step_finalize<>, dim3(grid), dim3(vk::kWG), lds, st, g);
    return launched("step_finalize");
}

int finalize_h32(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, int grid, hipStream_t st) {
    vk::FinalizeArgs g = f;
    const size_t lds = vk::loss_lds_bytes(f.n_obj, f.NW);
    g.loss_stage = vk::loss_stage_cap(lds);
    hipLaunchKernelGGL(vk::step_finalize_h32<>, dim3(grid), dim3(vk::kWG), lds, st, g, h);
    return launched("step_finalize_h32");
}

}  // namespace vl
