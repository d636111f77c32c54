// k_ws.hip - hidden 64 / 128 on the bf16 matrix pipe, one wave per output block (wsplit_kernels.h): step_prep_ws,
// step_main_ws, step_finalize_ws (the last two also serve step_main_wp).  The background model's path.  gfx950 only.
#include "ws_launch.h"

namespace vl {

namespace {
template <int NB, bool BWD, bool W3, bool STAMPS, int NT, bool ONE>
int main_o(const vk::StepArgs& a, hipStream_t st);
// hidden 128, training: the single-round specialisation when the plan gives every workgroup exactly one round
template <int NB, bool BWD, bool W3, bool STAMPS, int NT>
int main_t(const vk::StepArgs& a, hipStream_t st) {
    if constexpr (NB == 4 && BWD && !STAMPS) {
        if (a.NG == a.NW) return main_o<NB, BWD, W3, STAMPS, NT, true>(a, st);
    }
#ifndef VMAPSTEP_AB
    // three-tile rounds with several rounds per workgroup: no automatic plan launches it (the plan takes three tiles exactly when they
    // give every workgroup ONE round) - measurement build only.  make_plan (vmapstep.hip) refuses that PLAN for every entry point,
    // training and render alike, before anything is launched; this guard only keeps the multi-round TRAINING instantiation out of
    // the product binary - the forward instantiation below stays, it also serves the single-round render of the background shape
    if constexpr (NT == 3 && BWD) return fail(-2, "this kernel form ships in the measurement build only (tests/tools/libvmapstep_ab.so: phase stamps and A/B forms no automatic plan launches)");
    else
#endif
    return main_o<NB, BWD, W3, STAMPS, NT, false>(a, st);
}
template <int NB, bool BWD, bool W3, bool STAMPS, int NT, bool ONE>
int main_o(const vk::StepArgs& a, hipStream_t st) {
    using LD = vk::LdsWs<NB, NT>;
    auto kern = vk::step_main_ws<NB, BWD, W3, STAMPS, NT, ONE>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LD::LDS_BYTES, "step_main_ws")) return rc;
    vk::WsArgs ga;
    ga.s = a;
    ga.scratch = reinterpret_cast<char*>(a.gen_scratch);
    ga.tab_wt = a.tab_wt;
    VL_LAUNCH_MAIN(kern, dim3(a.n_obj * a.NW), dim3(vk::kWG), LD::LDS_BYTES, st, ga);
    return launched("step_main_ws");
}
template <int NB, bool BWD, bool W3, bool STAMPS>
int main_v(const vk::StepArgs& a, hipStream_t st) {
    if constexpr (NB == 4) {
        if (a.tiles == 3) return main_t<NB, BWD, W3, STAMPS, 3>(a, st);      // three-tile rounds: hidden 128 only
    }
    return a.tiles == 1 ? main_t<NB, BWD, W3, STAMPS, 1>(a, st) : main_t<NB, BWD, W3, STAMPS, 2>(a, st);
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

int main_ws(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st) {
    if (a.hidden == 256) return main_ws8(a, bwd, st);                    // k_ws8.hip (eight waves)
    if (a.hidden == 128) return main_nb<4>(a, bwd, stamps, st);
#ifdef VMAPSTEP_AB
    return main_nb<2>(a, bwd, stamps, st);                               // hidden 64 on step_main_ws: A/B reference of step_main_wp<2>
#else
    return fail(-2, "this kernel form ships in the measurement build only (tests/tools/libvmapstep_ab.so: phase stamps and A/B forms no automatic plan launches)");
#endif
}

int prep_ws(const vk::StepArgs& a, int n_steps, hipStream_t st) {
    vk::WsArgs ga;
    ga.s = a;
    ga.s.prep_steps = n_steps;
    ga.scratch = reinterpret_cast<char*>(a.gen_scratch);
    ga.tab_wt = a.tab_wt;
    // parameters without a place in the W^T image (biases, heads, B) keep -1 in its table
    hipError_t e = hipMemsetAsync(a.tab_wt, 0xFF, (size_t)a.PP * sizeof(int), st);
    if (e != hipSuccess) return fail(-4, "hipMemsetAsync(tab_wt): %s", hipGetErrorString(e));
    if (a.hidden == 256) return prep_ws8(ga, n_steps, st);
    if (a.hidden == 128)
        hipLaunchKernelGGL(vk::step_prep_ws<4>, dim3(vk::ws_prep_grid<4>(n_steps, a.n_obj)), dim3(vk::kWG), 3 * vk::kWG * sizeof(int), st, ga);
    else
        hipLaunchKernelGGL(vk::step_prep_ws<2>, dim3(vk::ws_prep_grid<2>(n_steps, a.n_obj)), dim3(vk::kWG), 3 * vk::kWG * sizeof(int), st, ga);
    return launched("step_prep_ws");
}

int finalize_ws(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, const int* tab_wt, hipStream_t st) {
    // (f.xcd_affine - an object's blocks on one XCD, see ws_finalize_grid - is decided once, in fill_finalize_args)
    if (f.hidden == 256) return finalize_ws8(f, h, tab_wt, st);
    if (finalize_one_thread_per_quad(f)) return f.hidden == 128 ? finalize_wide<4>(f, h, tab_wt, st) : finalize_wide<2>(f, h, tab_wt, st);
    if (f.hidden == 128) {
        // the narrow form where it fills the chip with one block per compute unit and the wide one does not (the background step)
        const int narrow = f.n_obj * vk::ws_finalize_blocks(f.PR, vk::kFinQuadsNarrow);
        if (!f.xcd_affine && narrow <= 256 && vk::ws_finalize_grid(f.n_obj, f.PR, vk::kFinQuads, 0) - 1 < narrow) {
            constexpr int T = vk::kFinGroups * vk::kFinQuadsNarrow;
            return launch_finalize_ws<4, vk::kFinQuadsNarrow, vk::kFinGroups>(f, h, tab_wt, narrow, T, T * 4 * sizeof(float), st);
        }
        return finalize_grouped<4>(f, h, tab_wt, st);
    }
    return finalize_grouped<2>(f, h, tab_wt, st);
}

}  // namespace vl
