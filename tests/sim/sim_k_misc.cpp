// TEST INFRASTRUCTURE: simulator launchers of the frame sampler and the inference query (see sim_launch.h)
#include "sim_launch.h"
#include "query_split_kernels.h"

namespace sl {
void sample(const vs::SampleArgs& a, int n_obj, long long rays) {
    if (a.obj_max) {
        for (int k = 0; k < n_obj; ++k) a.obj_max[k] = (int)0x80808080u;
        sim::launch(n_obj * a.nsplit, vs::kWG, vs::kWG * 4, [&] { vs::frame_depth_max(a); });
        sim::launch(n_obj * a.nsplit, vs::kWG, vs::kWG * 4, [&] { vs::frame_sample<false>(a); });
        return;
    }
    if (rays <= vs::kMaxStagedRays) sim::launch(n_obj, vs::kWG, (3 * (size_t)rays + vs::kWG) * 4, [&] { vs::frame_sample<true>(a); });
    else sim::launch(n_obj, vs::kWG, vs::kWG * 4, [&] { vs::frame_sample<false>(a); });
}
int query(int H, const vk::StepArgs& pack, const vk::QueryArgs& q, int grid) {
    if (H == 32) {
        sim::launch(vk::kSplitPackBlocks, vk::kWG, 3 * vk::kWG * 4, [&] { vk::step_prep_s32(pack); });
        sim::launch(grid, vk::kWG, vk::kQuerySplitLds, [&] { vk::field_query_s32(q); });
        return 0;
    }
    sim::launch(vk::gen_layout(H).imgp / 1024, vk::kWG, 3 * vk::kWG * 4, [&] { vk::step_prep(pack); });
    switch (H / 32) {
        case 2: sim::launch(grid, vk::kWG, 64, [&] { vk::field_query_gen<2>(q); }); break;
        case 4: sim::launch(grid, vk::kWG, 64, [&] { vk::field_query_gen<4>(q); }); break;
        case 8: sim::launch(grid, vk::kWG, 8 * 1024 * vk::kWaves * 4, [&] { vk::field_query_gen<8>(q); }); break;
        default: return -2;
    }
    return 0;
}
}  // namespace sl
