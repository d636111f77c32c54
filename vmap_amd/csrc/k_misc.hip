// k_misc.hip - the kernels either side of the step: the inference query (query_kernels.h) and the
// batched frame sampler (sample_kernels.h; row 1).  gfx950 only.
#include "launch.h"
#include "query_split_kernels.h"

namespace vl {

int query_points(int hidden, const vk::StepArgs& pack, const vk::QueryArgs& q, long long n_points, hipStream_t st) {
    const long long chunks = (n_points + vk::kMaxPts - 1) / vk::kMaxPts;
    if (hidden == 32) {
        // this object's SPLIT image (step_prep_s32's pack role, zero mask-statistics blocks), then the bf16-pipe query
        hipLaunchKernelGGL(vk::step_prep_s32<>, dim3(vk::kSplitPackBlocks), dim3(vk::kWG), 3 * vk::kWG * sizeof(int), st, pack);
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(vk::field_query_s32<>), vk::kQuerySplitLds, "field_query_s32")) return rc;
        const int grid = (int)(chunks < 512 ? chunks : 512);          // two resident workgroups per CU (2 x 80 KiB of LDS)
        hipLaunchKernelGGL(vk::field_query_s32<>, dim3(grid), dim3(vk::kWG), vk::kQuerySplitLds, st, q);
        return launched("field_query_s32");
    }
    // this object's parameter image: step_prep's pack role with zero mask-statistics blocks
    hipLaunchKernelGGL(vk::step_prep<>, dim3(vk::gen_layout(hidden).imgp / 1024), dim3(vk::kWG), 3 * vk::kWG * sizeof(int), st, pack);
    {
        const int grid = (int)(chunks < 256 ? chunks : 256);
        const int nb = hidden / 32;
        const size_t lds = nb > 4 ? (size_t)nb * 1024 * vk::kWaves * sizeof(float) : 0;   // second activation set (NB > 4)
        if (nb > 4) {
            const void* big[4] = {reinterpret_cast<const void*>(vk::field_query_gen<5>), reinterpret_cast<const void*>(vk::field_query_gen<6>),
                                  reinterpret_cast<const void*>(vk::field_query_gen<7>), reinterpret_cast<const void*>(vk::field_query_gen<8>)};
            if (int rc = ensure_dynamic_lds(big[nb - 5], lds, "field_query_gen")) return rc;
        }
        switch (nb) {
            case 2: hipLaunchKernelGGL(vk::field_query_gen<2>, dim3(grid), dim3(vk::kWG), lds, st, q); break;
            case 3: hipLaunchKernelGGL(vk::field_query_gen<3>, dim3(grid), dim3(vk::kWG), lds, st, q); break;
            case 4: hipLaunchKernelGGL(vk::field_query_gen<4>, dim3(grid), dim3(vk::kWG), lds, st, q); break;
            case 5: hipLaunchKernelGGL(vk::field_query_gen<5>, dim3(grid), dim3(vk::kWG), lds, st, q); break;
            case 6: hipLaunchKernelGGL(vk::field_query_gen<6>, dim3(grid), dim3(vk::kWG), lds, st, q); break;
            case 7: hipLaunchKernelGGL(vk::field_query_gen<7>, dim3(grid), dim3(vk::kWG), lds, st, q); break;
            default: hipLaunchKernelGGL(vk::field_query_gen<8>, dim3(grid), dim3(vk::kWG), lds, st, q); break;
        }
    }
    return launched("field_query");
}

int sample_frame(const vs::SampleArgs& a, int n_obj, long long rays_per_object, hipStream_t st) {
    if (a.obj_max) {
        // split form: nsplit workgroups per object; launch 1 joins the per-slice depth maxima, launch 2 samples
        hipError_t e = hipMemsetAsync(a.obj_max, 0x80, (size_t)n_obj * sizeof(int), st);
        if (e != hipSuccess) return fail(-4, "hipMemsetAsync(obj_max): %s", hipGetErrorString(e));
        hipLaunchKernelGGL(vs::frame_depth_max<>, dim3(n_obj * a.nsplit), dim3(vs::kWG), vs::kWG * sizeof(float), st, a);
        if (int rc = launched("frame_depth_max")) return rc;
        hipLaunchKernelGGL(vs::frame_sample<false>, dim3(n_obj * a.nsplit), dim3(vs::kWG), vs::kWG * sizeof(float), st, a);
        return launched("frame_sample");
    }
    if (rays_per_object <= vs::kMaxStagedRays) {
        const size_t lds = (3 * (size_t)rays_per_object + vs::kWG) * sizeof(float);
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(vs::frame_sample<true>), 160 * 1024, "frame_sample")) return rc;
        hipLaunchKernelGGL(vs::frame_sample<true>, dim3(n_obj), dim3(vs::kWG), lds, st, a);
    } else {
        // more rays per object than the staging area holds (the background model's frame): phase A is evaluated twice
        hipLaunchKernelGGL(vs::frame_sample<false>, dim3(n_obj), dim3(vs::kWG), vs::kWG * sizeof(float), st, a);
    }
    return launched("frame_sample");
}

}  // namespace vl
