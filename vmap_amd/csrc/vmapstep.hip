// vmapstep.hip - C ABI (include/vmapstep.h) over the fused step kernels; gfx950 only.
//
// Host side of the drop-in boundary: validates shapes, lays out the caller-provided workspace, fills the
// kernel argument blocks and enqueues   step_prep -> (step_main_h32 -> step_finalize) x n_steps   on the
// caller's stream.  Never allocates, never synchronises.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/vmapstep.h"
#include "launch.h"
// layouts only (image sizes, LDS / scratch budgets of the plan): no kernel of these headers is instantiated in this unit
#include "wide_kernels.h"
#include "wpair_kernels.h"

namespace {
thread_local char g_err[512] = "";
}

namespace vl {
thread_local const DispatchEvents* g_dispatch_events = nullptr;
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VMAPSTEP_ERR_DEVICE, "%s launch: %s", what, hipGetErrorString(e));
    return VMAPSTEP_OK;
}
}  // namespace vl

namespace {
using vl::fail;
// No tuning state lives in the library: overrides of the automatic plan arrive per call in vmapstep_shape::tuning.
const vmapstep_tuning kAutoTuning = {0, VMAPSTEP_KERNEL_AUTO, 0, 0};
const vmapstep_tuning& tuning_of(const vmapstep_shape* sh) { return (sh && sh->tuning) ? *sh->tuning : kAutoTuning; }

// Every entry point runs on the device that OWNS the caller's stream, whatever device is current on the calling thread (one
// process may drive several GPUs): kernel attributes, CU counts and the launches themselves are per device.  With the NULL
// stream the current device is used as it is.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(void* stream) {
        if (!stream) return;
        int dev = -1, cur = -1;
        if (hipStreamGetDevice(static_cast<hipStream_t>(stream), &dev) != hipSuccess || hipGetDevice(&cur) != hipSuccess) { ok = false; return; }
        if (dev != cur) {
            if (hipSetDevice(dev) != hipSuccess) { ok = false; return; }
            prev = cur;
        }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define VMAPSTEP_ON_STREAM_DEVICE(stream)                                                                   \
    DeviceGuard device_guard_(stream);                                                                      \
    if (!device_guard_.ok) return fail(VMAPSTEP_ERR_DEVICE, "cannot switch to the device of the stream")

constexpr size_t kAlign = 256;
constexpr int kMaxFrameSteps = 256;      // optimisation steps per API call (the flag array has this fixed capacity)
size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

struct Layout {
    int64_t sizes[15];
    int offs[16];
    int P, PP;
};
void make_layout(int H, Layout& L) {
    const vk::GenLayout G = vk::gen_layout(H);
    for (int t = 0; t < 15; ++t) { L.sizes[t] = G.f[t + 1] - G.f[t]; L.offs[t] = G.f[t]; }
    L.offs[15] = G.P;
    L.P = G.P;
    L.PP = G.PP;
}

struct Plan {
    int G, NG, NW;
    int tiles;         // step_main_ws: 32-point tiles per round (2; 1 = single-tile rounds when every tile gets a compute unit of its own; 3: see make_plan)
    size_t off_stats, off_flags, off_ploss, off_imgtab, off_pgrad, off_wimg, off_scratch, total;
    bool generic;      // hidden != 32: step_main_gen (global-memory activations) instead of step_main_h32
    bool split;        // hidden 32 on the bf16 matrix pipe with split operands (step_main_s32; the default at hidden 32)
    bool bwd6;         // ... with the six-product backward (VMAPSTEP_KERNEL_S32_BWD6)
    int PR;            // floats per row of partial gradients: PP (flat order), or RowWs<NB>::PR (step_main_ws / _wp: block-native rows + a row table)
    int wide;          // 0 = step_main_gen, 1 = step_main_wide<4> (hidden 128 / 256: one tile per workgroup, four waves per
                       // tile), 3 = step_main_ws, 4 = step_main_wp (hidden 64 / 128, bf16 matrix pipe)
};

// Per-device facts and one-time per-device function attributes.  A process may drive several GPUs (one
// process, 8 streams): the dynamic-LDS limit of a kernel is a per-device property of the loaded code object, so "set
// once" is keyed by (device, function); lookups take a lock (a handful per API call, next to ~40 kernel launches).
constexpr int kMaxDevices = 64;
std::mutex g_dev_mutex;
int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}
int cu_count() {
    static int cus[kMaxDevices] = {};
    const int dev = current_device();
    if (dev < 0 || dev >= kMaxDevices) return -1;
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    if (!cus[dev] && hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus[dev] = -1;
    return cus[dev];
}
}  // namespace
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel)
int vl::ensure_dynamic_lds(const void* kernel, size_t bytes, const char* what) {
    static std::vector<const void*> done[kMaxDevices];
    const int dev = current_device();
    if (dev < 0 || dev >= kMaxDevices) return fail(VMAPSTEP_ERR_DEVICE, "hipGetDevice failed");
    std::lock_guard<std::mutex> lk(g_dev_mutex);
    for (const void* k : done[dev]) if (k == kernel) return VMAPSTEP_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(VMAPSTEP_ERR_DEVICE, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
    done[dev].push_back(kernel);
    return VMAPSTEP_OK;
}
namespace {

int make_plan(const vmapstep_shape* sh, int max_steps, Plan& pl, const Layout& L) {
    if (!sh) return fail(VMAPSTEP_ERR_ARGUMENT, "shape is null");
    if (sh->n_obj < 1 || sh->rays < 1 || sh->samples < 1 || max_steps < 1)
        return fail(VMAPSTEP_ERR_ARGUMENT, "bad shape n=%d R=%d S=%d steps=%d", sh->n_obj, sh->rays, sh->samples, max_steps);
    if (sh->hidden < 32 || sh->hidden > 256 || sh->hidden % 32 != 0)
        return fail(VMAPSTEP_ERR_UNSUPPORTED, "hidden=%d: supported widths are multiples of 32 up to 256", sh->hidden);
    if (sh->weight_dtype != VMAPSTEP_WEIGHTS_F32 && sh->weight_dtype != VMAPSTEP_WEIGHTS_BF16)
        return fail(VMAPSTEP_ERR_UNSUPPORTED, "weight_dtype=%d", sh->weight_dtype);
    pl.generic = sh->hidden != 32;
    if (sh->samples > vk::kMaxPts)
        return fail(VMAPSTEP_ERR_UNSUPPORTED, "samples=%d > %d", sh->samples, vk::kMaxPts);
    // Wide fields (hidden 128 / 256).  step_main_wide<4>: one 32-point tile per workgroup, four waves split its output
    // blocks - for latency-bound batches where every tile gets its own workgroup (it pays the whole parameter set in
    // partial-gradient traffic per 32 points).  step_main_gen: one wave per tile.
    pl.wide = 0;
    const vmapstep_tuning& tun = tuning_of(sh);
    const int force = tun.kernel;
    if (force < VMAPSTEP_KERNEL_AUTO || (force > VMAPSTEP_KERNEL_WP && force != VMAPSTEP_KERNEL_S32_BWD6)) return fail(VMAPSTEP_ERR_ARGUMENT, "tuning.kernel=%d", force);
    pl.split = !pl.generic && force != VMAPSTEP_KERNEL_H32_F32;
    pl.bwd6 = force == VMAPSTEP_KERNEL_S32_BWD6;
    if (pl.bwd6 && (pl.generic || sh->weight_dtype != VMAPSTEP_WEIGHTS_F32))
        return fail(VMAPSTEP_ERR_UNSUPPORTED, "VMAPSTEP_KERNEL_S32_BWD6: hidden 32 with float32 weights");
    if (pl.generic && sh->hidden % 128 == 0 && force != VMAPSTEP_KERNEL_GEN) {
        if (sh->samples <= vk::kWideTile) {
            const int gw = std::min(vk::kWideTile / sh->samples, sh->rays);
            const long long tiles = (long long)sh->n_obj * ((sh->rays + gw - 1) / gw);
            if (force == VMAPSTEP_KERNEL_WIDE4 || (force == VMAPSTEP_KERNEL_AUTO && tiles <= 256)) pl.wide = 1;
        }
    }
    // hidden 64 / 128: the bf16 matrix pipe with split operands (step_main_ws) unless an exact-fp32 kernel is asked for
    // step_main_ws: one wave per output block; step_main_wp: two (measured: +19 % at hidden 64, where step_main_ws leaves two of its
    // four waves without a block; within 2-3 % at hidden 128 - the automatic choice follows that)
    if (pl.generic && (sh->hidden == 128 || sh->hidden == 64) && sh->samples <= vk::ImgWs<4>::kPts) {
        if (force == VMAPSTEP_KERNEL_AUTO) pl.wide = sh->hidden == 64 ? 4 : 3;
        else if (force == VMAPSTEP_KERNEL_WS1) pl.wide = 3;
        else if (force == VMAPSTEP_KERNEL_WP) pl.wide = 4;
    }
    // hidden 256 (the iMAP field): step_main_ws<8> - eight waves, single-tile rounds.  One round per workgroup while every round
    // gets a compute unit of its own (the 100-ray configuration: 0.232 -> 0.102 ms per step); with more rounds than compute units
    // every further round re-reads and re-writes its 1.4 MB gradient row and the step becomes bound by that traffic - still ahead
    // of the exact-fp32 kernels (the reference's own iMAP batch, 4800 rays: 3.50 -> 2.38 ms, profiles/r04i_*)
    if (pl.generic && sh->hidden == 256 && sh->samples <= 32 && (force == VMAPSTEP_KERNEL_AUTO || force == VMAPSTEP_KERNEL_WS1)) pl.wide = 3;
    if ((force == VMAPSTEP_KERNEL_WS1 || force == VMAPSTEP_KERNEL_WP) && pl.wide < 3)
        return fail(VMAPSTEP_ERR_UNSUPPORTED, "VMAPSTEP_KERNEL_WS1 / _WP: hidden 64 / 128 with at most 64 samples per ray (_WS1 also hidden 256 with at most 32)");
    pl.G = (pl.wide >= 3 ? vk::ImgWs<4>::kPts : pl.wide == 1 ? vk::kWideTile : vk::kMaxPts) / sh->samples;
    pl.tiles = 2;
    if (pl.wide == 3) {
        // step_main_ws, tiles per round.  The kernel's time is the busiest workgroup's rounds, one workgroup per compute unit:
        //  * a mostly idle chip (the ray-sharded background model of a multi-GPU run: 150 rays per rank at 8 ranks): if every
        //    32-point tile can have a compute unit of its own, single-tile rounds (about 0.77 of a two-tile round's time) halve
        //    the points per workgroup; the extra partial-gradient rows cost the finalize ~0.1 us each (profiles/r03j_*);
        //  * more two-tile rounds than compute units (the 1200-ray background batch of ONE GPU: 300 rounds): three-tile rounds
        //    (hidden 128) if they give every workgroup exactly one round (200) - no second round, no read-modify-write of its
        //    gradient row (profiles/r03u_*).
        // tuning.ws_flags: bit 0 = never single-tile rounds, bit 1 = always three-tile rounds (hidden 128; tests), bit 2 = never
        const int g1 = 32 / sh->samples, g2 = pl.G, g3 = 96 / sh->samples;
        const bool autoplan = tun.workgroups_per_object <= 0;
        auto rounds = [&](int g) { return g >= 1 ? (long long)sh->n_obj * ((sh->rays + std::min(g, sh->rays) - 1) / std::min(g, sh->rays)) : (1LL << 40); };
        if (sh->hidden == 256) { pl.G = g1; pl.tiles = 1; }
        else if (sh->hidden == 128 && (tun.ws_flags & 2)) { pl.G = g3; pl.tiles = 3; }
        else if (autoplan && !(tun.ws_flags & 1) && rounds(g1) <= 256) { pl.G = g1; pl.tiles = 1; }
        else if (autoplan && sh->hidden == 128 && !(tun.ws_flags & 4) && rounds(g2) > 256 && rounds(g3) <= 256) { pl.G = g3; pl.tiles = 3; }
    }
    if (pl.G > sh->rays) pl.G = sh->rays;
    pl.NG = (sh->rays + pl.G - 1) / pl.G;
    // workgroup slots of the chip: one per CU, two for step_main_wp at hidden 64 (78 KB of LDS per workgroup)
    const int wg_slots = (pl.wide == 4 && sh->hidden == 64) ? 512 : 256;
    int nw = tun.workgroups_per_object > 0 ? tun.workgroups_per_object : wg_slots / sh->n_obj;
    if (nw < 1) nw = 1;
    if (nw > pl.NG) nw = pl.NG;
    if (pl.wide >= 3 && tun.workgroups_per_object <= 0) {
        // one workgroup per CU: with more rounds than workgroup slots the busiest workgroup sets the kernel time, so spread
        // the rounds evenly (300 rounds on 256 CUs: 150 workgroups x 2 rounds) - fewer partial-gradient rows for the finalize
        const int per = (pl.NG + nw - 1) / nw;
        nw = (pl.NG + per - 1) / per;
    }
    pl.NW = nw;
#ifndef VMAPSTEP_AB
    // The product library carries the kernel forms automatic plans launch (+ the exact-fp32 references step_main_h32 / _gen).  Forms
    // that exist for A/B measurements only - step_main_wide<4>, step_main_ws at hidden 64, step_main_wp at hidden 128, three-tile rounds
    // with several rounds per workgroup - and the phase-stamp instantiations live in the measurement build (tests/tools/libvmapstep_ab.so, built by __graft_entry__.build() with -DVMAPSTEP_AB).
    if (pl.wide == 1 || (pl.wide == 3 && sh->hidden == 64) || (pl.wide == 4 && sh->hidden == 128) || (pl.wide == 3 && pl.tiles == 3 && pl.NW != pl.NG))
        return fail(VMAPSTEP_ERR_UNSUPPORTED, "this kernel form ships in the measurement build only (tests/tools/libvmapstep_ab.so: phase stamps and A/B forms no automatic plan launches)");
#endif
    // buffers that exist once per workgroup: sized for THIS plan's NW (the tuning is part of the shape, so the sizing call
    // and the launches see the same plan; a mismatch is caught by the workspace size check of the call, never silently)
    const size_t nw_cap = (size_t)nw;
    // Every offset is independent of the step count (only the total grows with it): a frame prepared for n steps, a single
    // prepared step of it and the optimiser-only call address the same buffers.  The per-step arrays come last.
    if (max_steps > kMaxFrameSteps) return fail(VMAPSTEP_ERR_UNSUPPORTED, "steps per call %d > %d", max_steps, kMaxFrameSteps);
    size_t o = 0;
    pl.off_ploss = o; o += align_up((size_t)sh->n_obj * nw_cap * 4 * sizeof(float));
    pl.PR = pl.wide >= 3 ? vk::ws_row_floats(sh->hidden) : L.PP;
    // tables: flat parameter -> image position [PP] (+ step_main_ws / _wp: -> W^T image position [PP], row element -> flat parameter [PR])
    pl.off_imgtab = o; o += pl.wide >= 3 ? 2 * align_up((size_t)L.PP * sizeof(int)) + align_up((size_t)pl.PR * sizeof(int)) : pl.generic ? 0 : align_up((size_t)L.PP * sizeof(int));
    pl.off_pgrad = o; o += align_up((size_t)sh->n_obj * nw_cap * pl.PR * sizeof(float));
    const vk::GenLayout GL = vk::gen_layout(sh->hidden);
    pl.off_wimg = o; o += align_up(pl.split ? (size_t)sh->n_obj * vk::Img32s::BYTES : pl.wide >= 3 ? (size_t)sh->n_obj * (sh->hidden == 256 ? vk::ImgWs<8>::BYTES : sh->hidden == 128 ? vk::ImgWs<4>::BYTES : vk::ImgWs<2>::BYTES)
                                                                                   : (size_t)sh->n_obj * GL.imgp * sizeof(float));
    pl.off_scratch = o;
    if (pl.wide >= 3) o += align_up((size_t)sh->n_obj * nw_cap * (pl.wide == 4 ? (sh->hidden == 128 ? vk::LdsWp<4>::WG_SCRATCH : vk::LdsWp<2>::WG_SCRATCH)
                                                                                 : (size_t)vk::kWsScratchMax));
    else if (pl.generic)   // register-image scratch: per wave (step_main_gen) or per workgroup (step_main_wide)
        o += align_up((size_t)sh->n_obj * nw_cap * (pl.wide == 1 ? 1 : vk::kWaves) * vk::gen_wave_blocks(GL.NB) * vk::kBlk * sizeof(float));
    pl.off_flags = o; o += align_up((size_t)kMaxFrameSteps * 4 * sizeof(int));
    pl.off_stats = o; o += align_up((size_t)max_steps * sh->n_obj * 4 * sizeof(float));
    pl.total = o;
    return VMAPSTEP_OK;
}

int check_params(const vmapstep_params* p, const char* what, bool allow_null_entries) {
    if (!p) return fail(VMAPSTEP_ERR_ARGUMENT, "%s is null", what);
    for (int t = 0; t < VMAPSTEP_NUM_FC; ++t)
        if (!p->fc[t].ptr && !allow_null_entries) return fail(VMAPSTEP_ERR_ARGUMENT, "%s.fc[%d] is null", what, t);
    if (!p->pe_B.ptr && !allow_null_entries) return fail(VMAPSTEP_ERR_ARGUMENT, "%s.pe_B is null", what);
    return VMAPSTEP_OK;
}

int check_batch(const vmapstep_batch* b) {
    if (!b) return fail(VMAPSTEP_ERR_ARGUMENT, "batch is null");
    if (!b->z || !b->gt_depth || !b->gt_rgb || !b->sem || !b->depth_mask)
        return fail(VMAPSTEP_ERR_ARGUMENT, "batch has a null tensor");
    // the sample points: either the points tensor, or (ABI v7) the rays they are rebuilt from
    if (!b->pcs && (!b->ray_o || !b->ray_d))
        return fail(VMAPSTEP_ERR_ARGUMENT, "batch has neither pcs nor (ray_o, ray_d)");
    return VMAPSTEP_OK;
}

void fill_step_args(vk::StepArgs& a, const vmapstep_shape* sh, const Plan& pl, const Layout& L,
                    const vmapstep_params* params, const vmapstep_tensor* pe_scale, const vmapstep_batch* b,
                    int64_t ray0, float cw, float ow, char* ws) {
    std::memset(&a, 0, sizeof(a));
    a.n_obj = sh->n_obj; a.R = sh->rays; a.S = sh->samples;
    a.G = pl.G; a.NG = pl.NG; a.NW = pl.NW; a.PP = L.PP; a.tiles = pl.tiles;
    // XCD-affine block map (an object's workgroups on ONE XCD / L2), decided HERE for every kernel family - the launchers, the
    // phase-profile workgroup count and fill_finalize_args read this one value:
    //  * hidden 32 (step_main_s32 / _h32): only while every XCD's share still fits its 32 CUs in one round;
    //  * step_main_wp (hidden 64, two workgroups per CU): from eight objects on (the grid is padded to whole groups of eight objects;
    //    measured: a rank's share of configs[4] 0.2207 -> 0.2112 ms, profiles/round5q_*);
    //  * step_main_ws / _gen / _wide: never (one object, or no per-object L2 reuse to keep).
    a.xcd_affine = pl.wide == 4 ? (sh->n_obj >= 8 ? 1 : 0) : (!pl.generic && ((sh->n_obj + 7) / 8) * pl.NW <= 32) ? 1 : 0;
    for (int t = 0; t < VMAPSTEP_NUM_FC; ++t) a.fc[t] = {params->fc[t].ptr, params->fc[t].obj_stride};
    a.pe_B = {params->pe_B.ptr, params->pe_B.obj_stride};
    a.pe_scale = {pe_scale->ptr, pe_scale->obj_stride};
    if (b->pcs) {
        a.pcs = b->pcs + ray0 * b->pcs_stride[1];
        a.pcs_so = b->pcs_stride[0]; a.pcs_sr = b->pcs_stride[1]; a.pcs_ss = b->pcs_stride[2]; a.pcs_sc = b->pcs_stride[3];
    } else {                                  // ABI v7: the rays the points are rebuilt from (load_point, step_kernels.h)
        a.ray_o = b->ray_o + ray0 * b->ray_o_stride[1];
        a.ro_so = b->ray_o_stride[0]; a.ro_sr = b->ray_o_stride[1]; a.ro_sc = b->ray_o_stride[2];
        a.ray_d = b->ray_d + ray0 * b->ray_d_stride[1];
        a.rd_so = b->ray_d_stride[0]; a.rd_sr = b->ray_d_stride[1]; a.rd_sc = b->ray_d_stride[2];
        a.center = b->center; a.ce_so = b->center_stride;
    }
    a.z = b->z + ray0 * b->z_stride[1];
    a.z_so = b->z_stride[0]; a.z_sr = b->z_stride[1]; a.z_ss = b->z_stride[2];
    a.gt_depth = b->gt_depth + ray0 * b->gt_depth_stride[1];
    a.gd_so = b->gt_depth_stride[0]; a.gd_sr = b->gt_depth_stride[1];
    a.gt_rgb = b->gt_rgb + ray0 * b->gt_rgb_stride[1];
    a.rgb_so = b->gt_rgb_stride[0]; a.rgb_sr = b->gt_rgb_stride[1]; a.rgb_sc = b->gt_rgb_stride[2];
    a.sem = b->sem + ray0 * b->sem_stride[1];
    a.sem_so = b->sem_stride[0]; a.sem_sr = b->sem_stride[1];
    a.dmask = b->depth_mask + ray0 * b->depth_mask_stride[1];
    a.dm_so = b->depth_mask_stride[0]; a.dm_sr = b->depth_mask_stride[1];
    a.color_w = cw; a.opac_w = ow;
    a.hidden = sh->hidden;
    a.weights_bf16 = sh->weight_dtype == VMAPSTEP_WEIGHTS_BF16 ? 1 : 0;
    a.wide = pl.wide;
    a.split = pl.split ? 1 : 0;
    a.bwd6 = pl.bwd6 ? 1 : 0;
    a.stats = reinterpret_cast<float*>(ws + pl.off_stats);
    a.flags = reinterpret_cast<int*>(ws + pl.off_flags);
    a.part_loss = reinterpret_cast<float*>(ws + pl.off_ploss);
    a.img_tab = (!pl.generic || pl.wide >= 3) ? reinterpret_cast<int*>(ws + pl.off_imgtab) : nullptr;   // also read by step_finalize_h32
    a.tab_wt = pl.wide >= 3 ? reinterpret_cast<int*>(ws + pl.off_imgtab + align_up((size_t)L.PP * sizeof(int))) : nullptr;
    a.PR = pl.PR;
    a.row_tab = pl.wide >= 3 ? reinterpret_cast<int*>(ws + pl.off_imgtab + 2 * align_up((size_t)L.PP * sizeof(int))) : nullptr;
    a.part_grad = reinterpret_cast<float*>(ws + pl.off_pgrad);
    a.wimg = reinterpret_cast<float*>(ws + pl.off_wimg);
    a.gen_scratch = reinterpret_cast<float*>(ws + pl.off_scratch);
}

// the dominant kernel of the plan (bwd = false: the forward-only instantiation of vmapstep_render; stamps: vmapstep_profile_phases)
int launch_main(const vk::StepArgs& a, bool bwd, bool stamps, hipStream_t st) {
    if (a.split) return vl::main_s32(a, bwd, stamps, st);
    if (a.wide == 4) return vl::main_wp(a, bwd, stamps, st);
    if (a.wide == 3) return vl::main_ws(a, bwd, stamps, st);
    return vl::main_f32(a, bwd, stamps, st);
}

int launch_prep(const vk::StepArgs& a, int n_steps, hipStream_t st) {
    if (a.wide >= 3) return vl::prep_ws(a, n_steps, st);
    if (a.split) return vl::prep_s32(a, n_steps, st);
    return vl::prep_f32(a, n_steps + a.n_obj * (vk::gen_layout(a.hidden).imgp / 1024), st);
}

void fill_finalize_args(vk::FinalizeArgs& f, const vk::StepArgs& a, const Layout& L, const vmapstep_params* params,
                        const vmapstep_params* grads, const vmapstep_adamw* opt, int step_after, bool have_grad,
                        float* loss_out, int* flags_out, float* terms_out, int step_in_call) {
    std::memset(&f, 0, sizeof(f));
    f.n_obj = a.n_obj; f.NW = a.NW; f.PP = L.PP; f.P = L.P; f.hidden = a.hidden; f.weights_bf16 = a.weights_bf16;
    f.PR = a.PR; f.row_tab = a.row_tab;
    for (int t = 0; t < 16; ++t) f.offs[t] = L.offs[t];
    for (int t = 0; t < 15; ++t) {
        const vmapstep_tensor* pt = t < 14 ? &params->fc[t] : &params->pe_B;
        f.param[t] = {pt->ptr, pt->obj_stride};
        if (grads) {
            const vmapstep_tensor* gt = t < 14 ? &grads->fc[t] : &grads->pe_B;
            f.grad[t] = {gt->ptr, gt->obj_stride};
        }
    }
    f.part_grad = a.part_grad; f.part_loss = a.part_loss; f.wimg = a.wimg;
    f.flags_in = a.flags; f.flags_out = flags_out; f.loss_out = loss_out; f.terms_out = terms_out;
    f.color_w = a.color_w; f.opac_w = a.opac_w;
    f.have_grad = have_grad ? 1 : 0;
    f.do_adam = (opt && have_grad) ? 1 : 0;
    if (f.do_adam) {
        f.m = opt->exp_avg; f.v = opt->exp_avg_sq;
        const double lr = opt->lr, b1 = opt->beta1, b2 = opt->beta2, wd = opt->weight_decay;
        f.decay = (float)(1.0 - lr * wd);
        f.one_minus_beta1 = (float)(1.0 - b1);
        f.beta2 = opt->beta2;
        f.one_minus_beta2 = (float)(1.0 - b2);
        f.eps = opt->eps;
        f.step_size = (float)(lr / (1.0 - std::pow(b1, (double)step_after)));
        f.bias_corr2_sqrt = (float)std::sqrt(1.0 - std::pow(b2, (double)step_after));
        if (opt->bias_table) {           // device-resident step count (graph replay): the two factors above come from the table
            f.adam_tab = opt->bias_table; f.adam_cnt = opt->step_counter; f.adam_i = step_in_call; f.adam_len = opt->table_len;
        }
    }
    // the finalize's own block -> object map: hidden 32 follows the main kernel's; step_finalize_ws (hidden >= 64) deals an object's
    // blocks to one XCD from eight objects on (its scattered 2-byte image stores then merge in one L2: profiles/round5p_*)
    f.xcd_affine = !have_grad ? 0 : a.wide >= 3 ? (a.n_obj >= 8 ? 1 : 0) : a.xcd_affine;
}

// the per-quad fields of a finalize (vk::FinalizeHot) from its FinalizeArgs
void fill_hot(vk::FinalizeHot& h, const vk::FinalizeArgs& f, const vk::StepArgs& a, const Layout& L, const vmapstep_params* params) {
    std::memset(&h, 0, sizeof(h));
    h.m = f.m; h.v = f.v; h.part_grad = f.part_grad; h.wimg = f.wimg; h.img_tab = a.img_tab;
    h.NW = f.NW; h.PP = f.PP; h.PR = f.PR; h.weights_bf16 = f.weights_bf16;
    h.decay = f.decay; h.one_minus_beta1 = f.one_minus_beta1; h.beta2 = f.beta2; h.one_minus_beta2 = f.one_minus_beta2;
    h.eps = f.eps; h.step_size = f.step_size; h.bias_corr2_sqrt = f.bias_corr2_sqrt;
    // parameters that are views of one [n, >= P] slab in flat order (vmap_amd.driver allocates them so): one base
    // pointer instead of a per-element tensor lookup
    h.slab = params->fc[0].ptr; h.slab_stride = params->fc[0].obj_stride;
    for (int t = 1; t < 15 && h.slab; ++t) {
        const vmapstep_tensor* pt = t < 14 ? &params->fc[t] : &params->pe_B;
        if (pt->ptr != params->fc[0].ptr + L.offs[t] || pt->obj_stride != h.slab_stride) h.slab = nullptr;
    }
}

int launch_finalize(const vk::StepArgs& a, const Layout& L, const vmapstep_params* params, const vmapstep_params* grads,
                    const vmapstep_adamw* opt, int step_after, bool have_grad, float* loss_out, int* flags_out,
                    float* terms_out, hipStream_t st, bool generic_finalize, int step_in_call = 0) {
    vk::FinalizeArgs f;
    fill_finalize_args(f, a, L, params, grads, opt, step_after, have_grad, loss_out, flags_out, terms_out, step_in_call);
    const int bpo = (L.PP / 4 + vk::kWG - 1) / vk::kWG;
    // + 1: the loss / flag reduction has a workgroup of its own (it used to ride on block 0 and made it the straggler)
    const int grid = (!have_grad ? 0 : f.xcd_affine ? 8 * ((a.n_obj + 7) / 8) * bpo : a.n_obj * bpo) + 1;
    vk::FinalizeHot h;
    if (a.wide >= 3 && have_grad) {
        // step_main_ws / _wp: one finalize for gradients to the caller and / or AdamW; it is the only writer of the two weight images
        fill_hot(h, f, a, L, params);
        f.ws_grouped = generic_finalize ? 1 : 0;
        return vl::finalize_ws(f, h, a.tab_wt, st);
    }
    if (a.split && f.do_adam) {
        // split image: the table-driven finalize is the only writer of the planes.  A caller that also wants the gradients of
        // this step gets them from a gradient-only pass of the generic kernel first (same ordered sums).
        if (grads) {
            vk::FinalizeArgs fg = f;
            fg.do_adam = 0;
            fg.loss_out = nullptr;             // the loss / flag workgroup runs once, in the second launch
            if (int rc = vl::finalize_generic(fg, grid, st)) return rc;
            std::memset(f.grad, 0, sizeof(f.grad));
        }
        fill_hot(h, f, a, L, params);
        return vl::finalize_s32(f, h, grid, st);
    }
    if (!generic_finalize && a.hidden == 32 && a.img_tab && f.do_adam && !grads) {
        // the common training step at hidden 32: table-driven form (same sums, same update, a third of the instructions)
        fill_hot(h, f, a, L, params);
        return vl::finalize_h32(f, h, grid, st);
    }
    return vl::finalize_generic(f, grid, st);
}

int check_ws(void* ws, size_t bytes, const Plan& pl) {
    if (!ws) return fail(VMAPSTEP_ERR_WORKSPACE, "workspace is null");
    if (reinterpret_cast<uintptr_t>(ws) % kAlign) return fail(VMAPSTEP_ERR_WORKSPACE, "workspace not 256-byte aligned");
    if (bytes < pl.total) return fail(VMAPSTEP_ERR_WORKSPACE, "workspace %zu < required %zu bytes", bytes, pl.total);
    return VMAPSTEP_OK;
}

}  // namespace

extern "C" {

const char* vmapstep_last_error(void) { return g_err; }
int vmapstep_abi_version(void) { return VMAPSTEP_ABI_VERSION; }

int vmapstep_param_layout(int32_t hidden, int64_t sizes[VMAPSTEP_NUM_FC + 1], int64_t* params, int64_t* padded_params) {
    if (hidden < 1) return fail(VMAPSTEP_ERR_ARGUMENT, "hidden=%d", hidden);
    Layout L;
    make_layout(hidden, L);
    if (sizes) for (int t = 0; t < 15; ++t) sizes[t] = L.sizes[t];
    if (params) *params = L.P;
    if (padded_params) *padded_params = L.PP;
    return VMAPSTEP_OK;
}

int vmapstep_workspace_bytes(const vmapstep_shape* shape, int32_t max_steps, size_t* bytes) {
    if (!bytes) return fail(VMAPSTEP_ERR_ARGUMENT, "bytes is null");
    if (!shape) return fail(VMAPSTEP_ERR_ARGUMENT, "shape is null");
    Layout L;
    make_layout(shape->hidden, L);
    Plan pl;
    int rc = make_plan(shape, max_steps, pl, L);
    if (rc) return rc;
    *bytes = pl.total;
    return VMAPSTEP_OK;
}

int vmapstep_describe_plan(const vmapstep_shape* shape, int32_t max_steps, vmapstep_plan_info* info) {
    if (!info) return fail(VMAPSTEP_ERR_ARGUMENT, "info is null");
    if (!shape) return fail(VMAPSTEP_ERR_ARGUMENT, "shape is null");
    Layout L;
    make_layout(shape->hidden, L);
    Plan pl;
    int rc = make_plan(shape, max_steps, pl, L);
    if (rc) return rc;
    std::memset(info, 0, sizeof(*info));
    const int nb = shape->hidden / 32;
    if (pl.split) std::snprintf(info->kernel, sizeof(info->kernel), pl.bwd6 ? "step_main_s32<bwd6>" : "step_main_s32");
    else if (!pl.generic) std::snprintf(info->kernel, sizeof(info->kernel), "step_main_h32");
    else if (pl.wide == 3) std::snprintf(info->kernel, sizeof(info->kernel), "step_main_ws<%d>", nb);
    else if (pl.wide == 4) std::snprintf(info->kernel, sizeof(info->kernel), "step_main_wp<%d>", nb);
    else if (pl.wide == 1) std::snprintf(info->kernel, sizeof(info->kernel), "step_main_wide<4>");
    else std::snprintf(info->kernel, sizeof(info->kernel), "step_main_gen");
    info->rays_per_round = pl.G;
    info->rounds_per_object = pl.NG;
    info->workgroups_per_object = pl.NW;
    info->tiles_per_round = pl.wide == 3 ? pl.tiles : 0;
    info->waves_per_workgroup = pl.wide == 3 ? (nb > 4 ? 8 : 4) : pl.wide == 4 ? 2 * nb : 4;
    info->single_round = pl.NG == pl.NW ? 1 : 0;
    return VMAPSTEP_OK;
}

static int fwd_bwd_impl(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                        const vmapstep_batch* batch, float color_scaling, float opacity_scaling,
                        const vmapstep_params* grads, const vmapstep_outputs* out,
                        void* workspace, size_t workspace_bytes, void* stream, bool do_prep, int step_index = 0) {
    int rc;
    if (!shape) return fail(VMAPSTEP_ERR_ARGUMENT, "shape is null");
    VMAPSTEP_ON_STREAM_DEVICE(stream);
    Layout L;
    make_layout(shape->hidden, L);
    Plan pl;
    if (step_index < 0) return fail(VMAPSTEP_ERR_ARGUMENT, "step_index=%d", step_index);
    // the workspace of a prepared frame holds at least step_index + 1 steps; no offset depends on the step count
    if ((rc = make_plan(shape, step_index + 1, pl, L))) return rc;
    if ((rc = check_params(params, "params", false))) return rc;
    if ((rc = check_params(grads, "grads", true))) return rc;
    if ((rc = check_batch(batch))) return rc;
    if (!pe_scale || !pe_scale->ptr) return fail(VMAPSTEP_ERR_ARGUMENT, "pe_scale is null");
    if (!out || !out->loss || !out->flags) return fail(VMAPSTEP_ERR_ARGUMENT, "outputs.loss / outputs.flags are required");
    if ((rc = check_ws(workspace, workspace_bytes, pl))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    vk::StepArgs a;
    fill_step_args(a, shape, pl, L, params, pe_scale, batch, 0, color_scaling, opacity_scaling, static_cast<char*>(workspace));
    a.prep_steps = 1; a.prep_ray_step = 0;
    a.stats += (size_t)step_index * shape->n_obj * 4;
    a.flags += (size_t)step_index * 4;
    a.dbg_depth = out->render_depth; a.dbg_rgb = out->render_color; a.dbg_opacity = out->opacity; a.dbg_var = out->var;
    if (do_prep && (rc = launch_prep(a, 1, st))) return rc;
    if ((rc = launch_main(a, true, false, st))) return rc;
    return launch_finalize(a, L, params, grads, nullptr, 0, true, out->loss, out->flags, out->loss_terms, st,
                           tuning_of(shape).generic_finalize != 0);
}

int vmapstep_fwd_bwd(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                     const vmapstep_batch* batch, float color_scaling, float opacity_scaling,
                     const vmapstep_params* grads, const vmapstep_outputs* out,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return fwd_bwd_impl(shape, params, pe_scale, batch, color_scaling, opacity_scaling, grads, out, workspace,
                        workspace_bytes, stream, true);
}

int vmapstep_fwd_bwd_prepared(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                              const vmapstep_batch* batch, int32_t step_index, float color_scaling, float opacity_scaling,
                              const vmapstep_params* grads, const vmapstep_outputs* out,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return fwd_bwd_impl(shape, params, pe_scale, batch, color_scaling, opacity_scaling, grads, out, workspace,
                        workspace_bytes, stream, false, step_index);
}

int vmapstep_adamw_apply(const vmapstep_shape* shape, const vmapstep_params* params, const float* grad_slab,
                         int64_t grad_stride, const vmapstep_adamw* opt, const float* loss_terms, int32_t step_index,
                         float color_scaling, float opacity_scaling, const vmapstep_outputs* out,
                         void* workspace, size_t workspace_bytes, void* stream) {
    int rc;
    if (!shape) return fail(VMAPSTEP_ERR_ARGUMENT, "shape is null");
    VMAPSTEP_ON_STREAM_DEVICE(stream);
    Layout L;
    make_layout(shape->hidden, L);
    Plan pl;
    if (step_index < 0) return fail(VMAPSTEP_ERR_ARGUMENT, "step_index=%d", step_index);
    if ((rc = make_plan(shape, step_index + 1, pl, L))) return rc;
    if ((rc = check_params(params, "params", false))) return rc;
    if (!grad_slab || grad_stride != L.PP || reinterpret_cast<uintptr_t>(grad_slab) % 16)
        return fail(VMAPSTEP_ERR_ARGUMENT, "grad_slab: need 16-byte aligned rows of padded_params = %d floats (vmapstep_param_layout)", L.PP);
    if (!opt || !opt->exp_avg || !opt->exp_avg_sq) return fail(VMAPSTEP_ERR_ARGUMENT, "optimiser state is required");
    if (opt->bias_table) return fail(VMAPSTEP_ERR_UNSUPPORTED, "vmapstep_adamw_apply takes the step count from opt->step (bias_table must be NULL)");
    if (loss_terms && (!out || !out->loss || !out->flags)) return fail(VMAPSTEP_ERR_ARGUMENT, "loss_terms given: outputs.loss / outputs.flags are required");
    if (!workspace || reinterpret_cast<uintptr_t>(workspace) % kAlign || workspace_bytes < pl.total)
        return fail(VMAPSTEP_ERR_WORKSPACE, "workspace too small / misaligned for this shape's parameter image");
    // the gradient slab plays the role of ONE row of partial gradients per object (NW = 1, row pitch = grad_stride): the
    // finalize kernels' ordered sum degenerates to a copy, their AdamW update and image rewrite are what is wanted; the
    // reduced loss terms play the role of the one row of loss partials per object
    vk::StepArgs a;
    std::memset(&a, 0, sizeof(a));
    char* ws = static_cast<char*>(workspace);
    a.n_obj = shape->n_obj; a.NW = 1; a.PP = L.PP; a.hidden = shape->hidden;
    a.PR = L.PP; a.row_tab = nullptr;            // the caller's slab is in flat order
    a.weights_bf16 = shape->weight_dtype == VMAPSTEP_WEIGHTS_BF16 ? 1 : 0;
    a.split = pl.split ? 1 : 0;
    a.xcd_affine = 0;
    a.part_grad = const_cast<float*>(grad_slab);
    a.part_loss = const_cast<float*>(loss_terms);
    a.flags = reinterpret_cast<int*>(ws + pl.off_flags) + (size_t)step_index * 4;
    a.color_w = color_scaling; a.opac_w = opacity_scaling;
    a.wimg = reinterpret_cast<float*>(ws + pl.off_wimg);
    a.wide = pl.wide;
    a.img_tab = (!pl.generic || pl.wide >= 3) ? reinterpret_cast<int*>(ws + pl.off_imgtab) : nullptr;
    a.tab_wt = pl.wide >= 3 ? reinterpret_cast<int*>(ws + pl.off_imgtab + align_up((size_t)L.PP * sizeof(int))) : nullptr;
    return launch_finalize(a, L, params, nullptr, opt, opt->step + 1, true, loss_terms ? out->loss : nullptr,
                           loss_terms ? out->flags : nullptr, nullptr, static_cast<hipStream_t>(stream),
                           tuning_of(shape).generic_finalize != 0);
}

int vmapstep_workspace_counts_offset(const vmapstep_shape* shape, int32_t max_steps, size_t* counts_offset) {
    if (!shape || !counts_offset) return fail(VMAPSTEP_ERR_ARGUMENT, "null argument");
    Layout L;
    make_layout(shape->hidden, L);
    Plan pl;
    int rc = make_plan(shape, max_steps, pl, L);
    if (rc) return rc;
    *counts_offset = pl.off_stats;
    return VMAPSTEP_OK;
}

int vmapstep_render(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                    const vmapstep_batch* batch, float color_scaling, float opacity_scaling,
                    const vmapstep_outputs* out, void* workspace, size_t workspace_bytes, void* stream) {
    int rc;
    if (!shape) return fail(VMAPSTEP_ERR_ARGUMENT, "shape is null");
    VMAPSTEP_ON_STREAM_DEVICE(stream);
    Layout L;
    make_layout(shape->hidden, L);
    Plan pl;
    if ((rc = make_plan(shape, 1, pl, L))) return rc;
    if ((rc = check_params(params, "params", false))) return rc;
    if ((rc = check_batch(batch))) return rc;
    if (!pe_scale || !pe_scale->ptr) return fail(VMAPSTEP_ERR_ARGUMENT, "pe_scale is null");
    if (!out || !out->loss || !out->flags) return fail(VMAPSTEP_ERR_ARGUMENT, "outputs.loss / outputs.flags are required");
    if ((rc = check_ws(workspace, workspace_bytes, pl))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    vk::StepArgs a;
    fill_step_args(a, shape, pl, L, params, pe_scale, batch, 0, color_scaling, opacity_scaling, static_cast<char*>(workspace));
    a.prep_steps = 1; a.prep_ray_step = 0;
    a.dbg_depth = out->render_depth; a.dbg_rgb = out->render_color; a.dbg_opacity = out->opacity; a.dbg_var = out->var;
    if ((rc = launch_prep(a, 1, st))) return rc;
    if ((rc = launch_main(a, false, false, st))) return rc;
    return launch_finalize(a, L, params, nullptr, nullptr, 0, false, out->loss, out->flags, out->loss_terms, st,
                           tuning_of(shape).generic_finalize != 0);
}

static int train_steps_impl(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                            const vmapstep_batch* frame, int64_t ray_step, int32_t n_steps,
                            float color_scaling, float opacity_scaling, const vmapstep_adamw* opt,
                            const vmapstep_params* grads, const vmapstep_outputs* out,
                            void* workspace, size_t workspace_bytes, void* stream, bool do_prep, bool do_steps,
                            size_t* flags_offset, float* time_main_ms = nullptr) {
    int rc;
    if (!shape) return fail(VMAPSTEP_ERR_ARGUMENT, "shape is null");
    VMAPSTEP_ON_STREAM_DEVICE(stream);
    if (n_steps < 1) return fail(VMAPSTEP_ERR_ARGUMENT, "n_steps=%d", n_steps);
    Layout L;
    make_layout(shape->hidden, L);
    Plan pl;
    if ((rc = make_plan(shape, n_steps, pl, L))) return rc;
    if ((rc = check_params(params, "params", false))) return rc;
    if (grads && (rc = check_params(grads, "grads", true))) return rc;
    if ((rc = check_batch(frame))) return rc;
    if ((rc = check_ws(workspace, workspace_bytes, pl))) return rc;
    if (flags_offset) *flags_offset = pl.off_flags;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    vmapstep_tensor dummy_scale = {params->fc[0].ptr, 0};
    const vmapstep_tensor* sc = pe_scale ? pe_scale : &dummy_scale;
    vk::StepArgs a;
    const bool device_steps = do_steps && opt && opt->bias_table;
    if (device_steps && (!opt->step_counter || opt->table_len < 1)) return fail(VMAPSTEP_ERR_ARGUMENT, "bias_table given: step_counter and table_len are required");
    if (device_steps && !do_prep) return fail(VMAPSTEP_ERR_UNSUPPORTED, "the device-resident step count is advanced by vmapstep_train_steps' own first launch: not available on the prepared path");
    if (do_prep) {
        fill_step_args(a, shape, pl, L, params, sc, frame, 0, color_scaling, opacity_scaling, ws);
        a.prep_steps = n_steps; a.prep_ray_step = ray_step;
        a.adam_counter = device_steps ? opt->step_counter : nullptr;
        if ((rc = launch_prep(a, n_steps, st))) return rc;
    }
    if (!do_steps) return VMAPSTEP_OK;
    if (!pe_scale || !pe_scale->ptr) return fail(VMAPSTEP_ERR_ARGUMENT, "pe_scale is null");
    if (!opt || !opt->exp_avg || !opt->exp_avg_sq) return fail(VMAPSTEP_ERR_ARGUMENT, "optimiser state is required");
    if (!out || !out->loss || !out->flags) return fail(VMAPSTEP_ERR_ARGUMENT, "outputs.loss / outputs.flags are required");
    // measurement only: the events live in a guard, so that every exit path (a failed launch, a failed record, a failed create
    // part-way through) destroys the ones that exist
    struct Events {
        std::vector<hipEvent_t> v;
        ~Events() { for (hipEvent_t e : v) (void)hipEventDestroy(e); }
        bool empty() const { return v.empty(); }
        hipEvent_t operator[](size_t i) const { return v[i]; }
    } ev;
    if (time_main_ms) {
        ev.v.reserve(4 * (size_t)n_steps);   // per step: stream events in front of / behind the launch, and the launch's own
                                             // dispatch begin / end events (hipExtLaunchKernel)
        for (size_t i = 0; i < 4 * (size_t)n_steps; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return fail(VMAPSTEP_ERR_DEVICE, "hipEventCreate failed");
            ev.v.push_back(e);
        }
    }
    for (int i = 0; i < n_steps; ++i) {
        fill_step_args(a, shape, pl, L, params, pe_scale, frame, (int64_t)i * ray_step, color_scaling, opacity_scaling, ws);
        a.stats += (size_t)i * shape->n_obj * 4;
        a.flags += (size_t)i * 4;
        const bool last = i == n_steps - 1;
        if (last) { a.dbg_depth = out->render_depth; a.dbg_rgb = out->render_color; a.dbg_opacity = out->opacity; a.dbg_var = out->var; }
        if (!ev.empty() && hipEventRecord(ev[4 * i], st) != hipSuccess) return fail(VMAPSTEP_ERR_DEVICE, "hipEventRecord failed");
        vl::DispatchEvents de = {nullptr, nullptr};
        if (!ev.empty()) { de = {ev[4 * i + 2], ev[4 * i + 3]}; vl::g_dispatch_events = &de; }
        rc = launch_main(a, true, false, st);
        vl::g_dispatch_events = nullptr;
        if (rc) return rc;
        if (!ev.empty() && hipEventRecord(ev[4 * i + 1], st) != hipSuccess) return fail(VMAPSTEP_ERR_DEVICE, "hipEventRecord failed");
        if ((rc = launch_finalize(a, L, params, last ? grads : nullptr, opt, opt->step + i + 1, true, out->loss + i, out->flags + 4 * i,
                                  last ? out->loss_terms : nullptr, st, tuning_of(shape).generic_finalize != 0, i))) return rc;
    }
    if (!ev.empty()) {                       // measurement only: the one place this library waits for the device
        bool ok = hipStreamSynchronize(st) == hipSuccess;
        double sum_dispatch = 0.0, sum_pair = 0.0;
        for (int i = 0; i < n_steps && ok; ++i) {
            float ms = 0.0f, pair = 0.0f;
            ok = hipEventElapsedTime(&ms, ev[4 * i + 2], ev[4 * i + 3]) == hipSuccess &&
                 hipEventElapsedTime(&pair, ev[4 * i], ev[4 * i + 1]) == hipSuccess;
            sum_dispatch += ms;
            sum_pair += pair;
        }
        if (!ok) return fail(VMAPSTEP_ERR_DEVICE, "event timing of the step loop failed");
        time_main_ms[0] = (float)(sum_dispatch / n_steps);             // the dispatch's own begin -> end (= a kernel trace's duration)
        time_main_ms[1] = (float)(sum_pair / n_steps);                 // stream events recorded around the launch (includes their own cost)
    }
    return VMAPSTEP_OK;
}

int vmapstep_train_steps(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                         const vmapstep_batch* frame, int64_t ray_step, int32_t n_steps,
                         float color_scaling, float opacity_scaling, const vmapstep_adamw* opt,
                         const vmapstep_params* grads, const vmapstep_outputs* out,
                         void* workspace, size_t workspace_bytes, void* stream) {
    return train_steps_impl(shape, params, pe_scale, frame, ray_step, n_steps, color_scaling, opacity_scaling, opt, grads,
                            out, workspace, workspace_bytes, stream, true, true, nullptr);
}

int vmapstep_prepare(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_batch* frame,
                     int64_t ray_step, int32_t n_steps, void* workspace, size_t workspace_bytes,
                     size_t* flags_offset, void* stream) {
    if (!flags_offset) return fail(VMAPSTEP_ERR_ARGUMENT, "flags_offset is null");
    return train_steps_impl(shape, params, nullptr, frame, ray_step, n_steps, 5.0f, 10.0f, nullptr, nullptr, nullptr,
                            workspace, workspace_bytes, stream, true, false, flags_offset);
}

int vmapstep_train_steps_prepared(const vmapstep_shape* shape, const vmapstep_params* params,
                                  const vmapstep_tensor* pe_scale, const vmapstep_batch* frame, int64_t ray_step,
                                  int32_t n_steps, float color_scaling, float opacity_scaling,
                                  const vmapstep_adamw* opt, const vmapstep_params* grads,
                                  const vmapstep_outputs* out, void* workspace, size_t workspace_bytes, void* stream) {
    return train_steps_impl(shape, params, pe_scale, frame, ray_step, n_steps, color_scaling, opacity_scaling, opt, grads,
                            out, workspace, workspace_bytes, stream, false, true, nullptr);
}

int vmapstep_profile_train_steps(const vmapstep_shape* shape, const vmapstep_params* params, const vmapstep_tensor* pe_scale,
                                 const vmapstep_batch* frame, int64_t ray_step, int32_t n_steps,
                                 float color_scaling, float opacity_scaling, const vmapstep_adamw* opt,
                                 const vmapstep_outputs* out, void* workspace, size_t workspace_bytes, void* stream,
                                 float main_kernel_ms[2]) {
    if (!main_kernel_ms) return fail(VMAPSTEP_ERR_ARGUMENT, "main_kernel_ms is null");
    return train_steps_impl(shape, params, pe_scale, frame, ray_step, n_steps, color_scaling, opacity_scaling, opt,
                            nullptr, out, workspace, workspace_bytes, stream, true, true, nullptr, main_kernel_ms);
}

int vmapstep_profile_main_kernel(const vmapstep_shape* shape, const vmapstep_params* params,
                                 const vmapstep_tensor* pe_scale, const vmapstep_batch* batch, int32_t reps,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    int rc;
    if (!shape) return fail(VMAPSTEP_ERR_ARGUMENT, "shape is null");
    VMAPSTEP_ON_STREAM_DEVICE(stream);
    Layout L;
    make_layout(shape->hidden, L);
    Plan pl;
    if ((rc = make_plan(shape, 1, pl, L))) return rc;
    if ((rc = check_params(params, "params", false))) return rc;
    if ((rc = check_batch(batch))) return rc;
    if (!pe_scale || !pe_scale->ptr) return fail(VMAPSTEP_ERR_ARGUMENT, "pe_scale is null");
    if ((rc = check_ws(workspace, workspace_bytes, pl))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    vk::StepArgs a;
    fill_step_args(a, shape, pl, L, params, pe_scale, batch, 0, 5.0f, 10.0f, static_cast<char*>(workspace));
    a.prep_steps = 1; a.prep_ray_step = 0;
    if ((rc = launch_prep(a, 1, st))) return rc;
    for (int i = 0; i < reps; ++i)
        if ((rc = launch_main(a, true, false, st))) return rc;
    return VMAPSTEP_OK;
}

int vmapstep_profile_phases(const vmapstep_shape* shape, const vmapstep_params* params,
                            const vmapstep_tensor* pe_scale, const vmapstep_batch* batch,
                            uint32_t* timing, size_t timing_elems, int32_t* n_workgroups,
                            void* workspace, size_t workspace_bytes, void* stream) {
    int rc;
    if (!shape) return fail(VMAPSTEP_ERR_ARGUMENT, "shape is null");
    VMAPSTEP_ON_STREAM_DEVICE(stream);
    Layout L;
    make_layout(shape->hidden, L);
    Plan pl;
    if ((rc = make_plan(shape, 1, pl, L))) return rc;
    if ((rc = check_params(params, "params", false))) return rc;
    if ((rc = check_batch(batch))) return rc;
    if (!pe_scale || !pe_scale->ptr) return fail(VMAPSTEP_ERR_ARGUMENT, "pe_scale is null");
    if (!timing || !n_workgroups) return fail(VMAPSTEP_ERR_ARGUMENT, "timing / n_workgroups is null");
    if (pl.generic && (pl.wide < 3 || shape->hidden == 256))
        return fail(VMAPSTEP_ERR_UNSUPPORTED, "phase stamps exist in the hidden=32 kernels and step_main_ws / _wp at hidden 64 / 128 only");
    const size_t need = (size_t)8 * ((shape->n_obj + 7) / 8) * pl.NW * vk::kWaves * vk::kMarks;
    if (timing_elems < need) return fail(VMAPSTEP_ERR_ARGUMENT, "timing buffer %zu < %zu elements", timing_elems, need);
    if ((rc = check_ws(workspace, workspace_bytes, pl))) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    vk::StepArgs a;
    fill_step_args(a, shape, pl, L, params, pe_scale, batch, 0, 5.0f, 10.0f, static_cast<char*>(workspace));
    a.prep_steps = 1; a.prep_ray_step = 0;
    a.timing = timing;
    *n_workgroups = a.xcd_affine ? 8 * ((shape->n_obj + 7) / 8) * pl.NW : shape->n_obj * pl.NW;
    if ((rc = launch_prep(a, 1, st))) return rc;
    return launch_main(a, true, true, st);   // the stamped instantiation
}

int vmapstep_query_workspace_bytes(int32_t hidden, size_t* bytes) {
    if (!bytes) return fail(VMAPSTEP_ERR_ARGUMENT, "bytes is null");
    if (hidden < 32 || hidden > 256 || hidden % 32 != 0)
        return fail(VMAPSTEP_ERR_UNSUPPORTED, "hidden=%d: supported widths are multiples of 32 up to 256", hidden);
    // hidden 32: the split image of field_query_s32 (80 KiB); other widths: the float32 image of field_query_gen
    *bytes = hidden == 32 ? align_up((size_t)vk::Img32s::BYTES) : align_up((size_t)vk::gen_layout(hidden).imgp * sizeof(float));
    return VMAPSTEP_OK;
}

int vmapstep_query_points(int32_t hidden, const vmapstep_params* params, const vmapstep_tensor* pe_scale, int32_t obj_index,
                          const float* points, int64_t n_points, const int64_t points_stride[2],
                          float* occupancy, float* color, void* workspace, size_t workspace_bytes, void* stream) {
    int rc;
    size_t need = 0;
    if ((rc = vmapstep_query_workspace_bytes(hidden, &need))) return rc;
    if ((rc = check_params(params, "params", false))) return rc;
    if (!pe_scale || !pe_scale->ptr || !points || !points_stride || !occupancy || !color || obj_index < 0 || n_points < 0)
        return fail(VMAPSTEP_ERR_ARGUMENT, "null / negative argument");
    if (!workspace || reinterpret_cast<uintptr_t>(workspace) % kAlign || workspace_bytes < need)
        return fail(VMAPSTEP_ERR_WORKSPACE, "workspace must be 256-byte aligned and >= %zu bytes", need);
    if (n_points == 0) return VMAPSTEP_OK;
    VMAPSTEP_ON_STREAM_DEVICE(stream);
    // pack this object's image (step_prep's pack role, zero mask-statistics blocks), then the query kernel
    vk::StepArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_obj = 1; a.hidden = hidden; a.prep_steps = 0;
    for (int t = 0; t < VMAPSTEP_NUM_FC; ++t) a.fc[t] = {params->fc[t].ptr + (long long)obj_index * params->fc[t].obj_stride, 0};
    a.pe_B = {params->pe_B.ptr + (long long)obj_index * params->pe_B.obj_stride, 0};
    a.wimg = static_cast<float*>(workspace);
    vk::QueryArgs q;
    q.wimg = a.wimg;
    q.scale = pe_scale->ptr + (long long)obj_index * pe_scale->obj_stride;
    q.pts = points; q.pts_sn = points_stride[0]; q.pts_sc = points_stride[1];
    q.n_pts = n_points; q.occ = occupancy; q.rgb = color;
    return vl::query_points(hidden, a, q, n_points, static_cast<hipStream_t>(stream));
}

static_assert(sizeof(vmapstep_sample_object) == sizeof(vs::SampleObject), "sample object table layout");

int vmapstep_sample_workspace_bytes(int32_t n_obj, size_t* bytes) {
    if (!bytes || n_obj < 1) return fail(VMAPSTEP_ERR_ARGUMENT, "null / non-positive argument");
    *bytes = align_up((size_t)n_obj * sizeof(int));
    return VMAPSTEP_OK;
}

static int sample_frame_impl(const vmapstep_sample_cfg* cfg, const vmapstep_sample_object* objects_device, int32_t n_obj,
                             float* pcs, float* ray_o, float* ray_d, float* center_out,
                             float* z, float* gt_depth, float* gt_rgb, uint8_t* sem, uint8_t* depth_mask,
                             uint64_t seed, uint32_t frame_counter, const vmapstep_sample_randoms* test_randoms,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!cfg || !objects_device || !z || !gt_depth || !gt_rgb || !sem || !depth_mask)
        return fail(VMAPSTEP_ERR_ARGUMENT, "null argument");
    if (!pcs && !(ray_o && ray_d)) return fail(VMAPSTEP_ERR_ARGUMENT, "neither pcs nor (ray_o, ray_d) given");
    if ((ray_o == nullptr) != (ray_d == nullptr)) return fail(VMAPSTEP_ERR_ARGUMENT, "ray_o and ray_d go together");
    const int S = cfg->n_bins_cam2surface + cfg->n_bins;
    const long long FP = (long long)cfg->frames * cfg->samples_per_frame;
    if (n_obj < 1 || cfg->frames < 1 || cfg->samples_per_frame < 1 || cfg->n_bins_cam2surface < 1 || cfg->n_bins < 1)
        return fail(VMAPSTEP_ERR_ARGUMENT, "bad sampler shape");
    if (S > vs::kMaxS || cfg->n_bins > 16 || cfg->width > 4095 || cfg->height > 4095 || FP > (1 << 24))
        return fail(VMAPSTEP_ERR_UNSUPPORTED, "sampler limits: S<=32, n_bins<=16, W,H<=4095, F*P<=2^24");
    vs::SampleArgs a;
    std::memset(&a, 0, sizeof(a));
    a.objs = reinterpret_cast<const vs::SampleObject*>(objects_device);
    a.n_obj = n_obj; a.W = cfg->width; a.H = cfg->height; a.F = cfg->frames; a.P = cfg->samples_per_frame;
    a.n1 = cfg->n_bins_cam2surface; a.n2 = cfg->n_bins;
    a.fx = cfg->fx; a.fy = cfg->fy; a.cx = cfg->cx; a.cy = cfg->cy;
    a.min_bound = cfg->min_depth; a.eps = cfg->surface_eps; a.stop_eps = cfg->stop_eps;
    a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32); a.frame_counter = frame_counter;
    if (test_randoms) {
        a.rnd.kf_ids = test_randoms->kf_ids; a.rnd.u_w = test_randoms->u_w; a.rnd.u_h = test_randoms->u_h;
        a.rnd.u_z = test_randoms->u_z; a.rnd.g_z = test_randoms->g_z;
    }
    a.pcs = pcs; a.z = z; a.gt_depth = gt_depth; a.gt_rgb = gt_rgb; a.sem = sem; a.depth_mask = depth_mask;
    a.ray_o = ray_o; a.ray_d = ray_d; a.center_out = center_out;
    if (workspace) {
        // split form: as many workgroups per object as fill the chip, at most one ray per thread
        if (reinterpret_cast<uintptr_t>(workspace) % sizeof(int) || workspace_bytes < (size_t)n_obj * sizeof(int))
            return fail(VMAPSTEP_ERR_WORKSPACE, "sampler workspace: need %zu bytes (vmapstep_sample_workspace_bytes)", (size_t)n_obj * sizeof(int));
        const long long per_obj_max = (FP + vs::kWG - 1) / vs::kWG;
        long long ns = std::max(1, 512 / n_obj);
        if (ns > per_obj_max) ns = per_obj_max;
        if (ns > 1) { a.nsplit = (int)ns; a.obj_max = static_cast<int*>(workspace); }
    }
    VMAPSTEP_ON_STREAM_DEVICE(stream);
    return vl::sample_frame(a, n_obj, FP, static_cast<hipStream_t>(stream));
}

int vmapstep_sample_frame(const vmapstep_sample_cfg* cfg, const vmapstep_sample_object* objects_device, int32_t n_obj,
                          float* pcs, float* z, float* gt_depth, float* gt_rgb, uint8_t* sem, uint8_t* depth_mask,
                          uint64_t seed, uint32_t frame_counter, const vmapstep_sample_randoms* test_randoms,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!pcs) return fail(VMAPSTEP_ERR_ARGUMENT, "null argument");
    return sample_frame_impl(cfg, objects_device, n_obj, pcs, nullptr, nullptr, nullptr, z, gt_depth, gt_rgb, sem, depth_mask, seed,
                             frame_counter, test_randoms, workspace, workspace_bytes, stream);
}

int vmapstep_sample_frame_rays(const vmapstep_sample_cfg* cfg, const vmapstep_sample_object* objects_device, int32_t n_obj,
                               float* ray_o, float* ray_d, float* center, float* pcs,
                               float* z, float* gt_depth, float* gt_rgb, uint8_t* sem, uint8_t* depth_mask,
                               uint64_t seed, uint32_t frame_counter, const vmapstep_sample_randoms* test_randoms,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (!ray_o || !ray_d) return fail(VMAPSTEP_ERR_ARGUMENT, "ray_o / ray_d are required");
    return sample_frame_impl(cfg, objects_device, n_obj, pcs, ray_o, ray_d, center, z, gt_depth, gt_rgb, sem, depth_mask, seed,
                             frame_counter, test_randoms, workspace, workspace_bytes, stream);
}

}  // extern "C"
