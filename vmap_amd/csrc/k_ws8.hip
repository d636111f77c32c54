// k_ws8.hip - hidden 256 (the iMAP field, configs/Replica/config_replica_room0_iMAP.json) on the bf16 matrix pipe: step_main_ws<8>
// (wsplit_kernels.h with EIGHT waves, one output block each, single-tile rounds), its prep and finalize.  Its own translation
// unit so that it compiles next to k_ws.hip.  gfx950 only.
#include "ws_launch.h"

namespace vl {

namespace {
template <bool BWD, bool W3, bool ONE>
int main8(const vk::StepArgs& a, hipStream_t st) {
    using LD = vk::LdsWs<8, 1>;
    auto kern = vk::step_main_ws<8, BWD, W3, false, 1, ONE>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), LD::LDS_BYTES, "step_main_ws<8>")) return rc;
    vk::WsArgs ga;
    ga.s = a;
    ga.scratch = reinterpret_cast<char*>(a.gen_scratch);
    ga.tab_wt = a.tab_wt;
    VL_LAUNCH_MAIN(kern, dim3(a.n_obj * a.NW), dim3(LD::NTH), LD::LDS_BYTES, st, ga);
    return launched("step_main_ws<8>");
}
}  // namespace

int main_ws8(const vk::StepArgs& a, bool bwd, hipStream_t st) {
    if (a.tiles != 1) return fail(-3, "step_main_ws at hidden 256 runs single-tile rounds");
    const bool one = a.NG == a.NW;                                       // every workgroup exactly one round
    if (a.weights_bf16) return bwd ? (one ? main8<true, false, true>(a, st) : main8<true, false, false>(a, st)) : main8<false, false, false>(a, st);
    return bwd ? (one ? main8<true, true, true>(a, st) : main8<true, true, false>(a, st)) : main8<false, true, false>(a, st);
}

int prep_ws8(const vk::WsArgs& ga, int n_steps, hipStream_t st) {
    hipLaunchKernelGGL(vk::step_prep_ws<8>, dim3(vk::ws_prep_grid<8>(n_steps, ga.s.n_obj)), dim3(vk::kWG), 3 * vk::kWG * sizeof(int), st, ga);
    return launched("step_prep_ws<8>");
}

int finalize_ws8(const vk::FinalizeArgs& f, const vk::FinalizeHot& h, const int* tab_wt, hipStream_t st) {
    if (finalize_one_thread_per_quad(f)) return finalize_wide<8>(f, h, tab_wt, st);       // the same predicate as the other widths
    return finalize_grouped<8>(f, h, tab_wt, st);
}

}  // namespace vl
