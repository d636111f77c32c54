// k_wp.hip - hidden 64 / 128 on the bf16 matrix pipe, two waves per output block (wpair_kernels.h): step_main_wp.  The
// default at hidden 64 (BASELINE configs[4]).  gfx950 only.
#include "launch.h"
#include "wpair_kernels.h"

namespace vl {

namespace {
template <int NB, bool BWD, bool W3, bool STAMPS>
int main_v(const vk::StepArgs& a, hipStream_t st) {
    using LD = vk::LdsWp<NB>;
    auto kern = vk::step_main_wp<NB, BWD, W3, STAMPS>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LD::LDS_BYTES, "step_main_wp")) return rc;
    vk::WsArgs ga;
    ga.s = a;
    ga.scratch = reinterpret_cast<char*>(a.gen_scratch);
    ga.tab_wt = a.tab_wt;
    // a.xcd_affine (an object's workgroups on one XCD; this family: from eight objects on) is decided once, in fill_step_args
    VL_LAUNCH_MAIN(kern, dim3((a.xcd_affine ? 8 * ((a.n_obj + 7) / 8) : a.n_obj) * a.NW), dim3(LD::NTH), LD::LDS_BYTES, st, ga);
    return launched("step_main_wp");
}
template <int NB>
int main_nb(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st) {
#ifdef VMAPSTEP_AB
    if (stamps) return main_v<NB, true, true, true>(a, st);
#else
    if (stamps) return fail(-2, "this kernel form ships in the measurement build only (tests/tools/libvmapstep_ab.so: phase stamps and A/B forms no automatic plan launches)");
#endif
    if (a.weights_bf16) return bwd ? main_v<NB, true, false, false>(a, st) : main_v<NB, false, false, false>(a, st);
    return bwd ? main_v<NB, true, true, false>(a, st) : main_v<NB, false, true, false>(a, st);
}
}  // namespace

int main_wp(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st) {
#ifdef VMAPSTEP_AB
    if (a.hidden == 128) return main_nb<4>(a, bwd, stamps, st);          // hidden 128 on step_main_wp: A/B reference of step_main_ws<4>
#else
    if (a.hidden == 128) return fail(-2, "this kernel form ships in the measurement build only (tests/tools/libvmapstep_ab.so: phase stamps and A/B forms no automatic plan launches)");
#endif
    return main_nb<2>(a, bwd, stamps, st);
}

}  // namespace vl
