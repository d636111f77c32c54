// k_s32.hip - hidden 32 on the bf16 matrix pipe with split operands (split_kernels.h): step_prep_s32, step_main_s32,
// step_finalize_s32.  The default path of BASELINE configs[1] / [3].  gfx950 only.
#include "launch.h"
#include "split_kernels.h"

namespace vl {

namespace {
template <bool BWD, bool MULTI, bool STAMPS, bool W3, bool B6 = false>
int main_v(const vk::StepArgs& a, hipStream_t st) {
    auto kern = vk::step_main_s32<BWD, MULTI, STAMPS, W3, B6>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), vk::Img32s::LDS_BYTES, "step_main_s32")) return rc;
    const int grid = a.xcd_affine ? 8 * ((a.n_obj + 7) / 8) * a.NW : a.n_obj * a.NW;
    VL_LAUNCH_MAIN(kern, dim3(grid), dim3(vk::kWG), vk::Img32s::LDS_BYTES, st, a);
    return launched("step_main_s32");
}
template <bool BWD, bool STAMPS>
int main_bs(const vk::StepArgs& a, hipStream_t st) {
    const bool multi = a.NW < a.NG;
    if (a.weights_bf16) return multi ? main_v<BWD, true, STAMPS, false>(a, st) : main_v<BWD, false, STAMPS, false>(a, st);
    if constexpr (BWD && !STAMPS) {          // the six-product backward (tuning.kernel = VMAPSTEP_KERNEL_S32_BWD6): training instantiations only
        if (a.bwd6) return multi ? main_v<true, true, false, true, true>(a, st) : main_v<true, false, false, true, true>(a, st);
    }
    return multi ? main_v<BWD, true, STAMPS, true>(a, st) : main_v<BWD, false, STAMPS, true>(a, st);
}
}  // namespace

int main_s32(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st) {
#ifdef VMAPSTEP_AB
    if (stamps) return main_bs<true, true>(a, st);
#else
    if (stamps) return fail(-2, "this kernel form ships in the measurement build only (tests/tools/libvmapstep_ab.so: phase stamps and A/B forms no automatic plan launches)");
#endif
    return bwd ? main_bs<true, false>(a, st) : main_bs<false, false>(a, st);
}

int prep_s32(const vk::StepArgs& a, int n_steps, hipStream_t st) {
    hipLaunchKernelGGL(vk::step_prep_s32<>, dim3(n_steps + a.n_obj * vk::kSplitPackBlocks), dim3(vk::kWG), 3 * vk::kWG * sizeof(int), st, a);
    return launched("step_prep_s32");
}

int finalize_s32(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, int grid, hipStream_t st) {
    vk::FinalizeArgs g = f;
    const size_t lds = vk::loss_lds_bytes(f.n_obj, f.NW);
    g.loss_stage = vk::loss_stage_cap(lds);
    hipLaunchKernelGGL(vk::step_finalize_s32<>, dim3(grid), dim3(vk::kWG), lds, st, g, h);
    return launched("step_finalize_s32");
}

}  // namespace vl
