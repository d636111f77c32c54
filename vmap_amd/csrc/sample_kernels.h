// sample_kernels.h - batched depth-guided ray / sample-point sampler for ALL objects of a frame (gfx950).
//
// Replaces the per-object Python loop of the reference (train.py:208-218 calling vmap.py:319-364
// sceneObject.get_training_samples -> vmap.py:366-459 sample_3d_points, ~30 small ATen launches per object and
// frame, followed by torch.stack at train.py:255-260) with ONE launch that writes the six per-frame tensors directly
// in the layout vmapstep_train_steps consumes ([n, F*P, ...]).  One workgroup per object:
//   A  per ray: keyframe slot of its frame, pixel inside that keyframe's 2-D box, gather RGB+state / depth,
//      pixel ray ((w-cx)/fx, (h-cy)/fy, 1) rotated by the keyframe pose                      (vmap.py:343-362, :31-41)
//   B  max sampled depth of the object (upper bound of the bins of invalid-depth rays)        (vmap.py:391)
//   C  per ray: n1 stratified bins camera->surface, then n2 sorted clipped-normal samples around the surface (this
//      object) or n2 stratified bins in [d-eps, d+stop_eps] (other/unknown), or n1+n2 stratified bins over
//      [min_bound, max depth] for invalid depth; points = origin + dir*z - centre              (vmap.py:395-457)
// HBM-bound byte work (random gathers + ~210 B written per ray): no matrix instructions.
//
// Random numbers: Philox4x32-10, counter = (ray or frame, object, frame counter, stream), key = seed - every value
// is a pure function of its coordinates (reproducible, order-independent).  The reference uses torch's global
// generator, so parity of the random part is statistical; the deterministic part is tested bit-for-bit through the
// test mode in which the per-ray numbers are supplied by the caller (SampleRandoms).
#pragma once
#include <wave_ops.h>

namespace vs {

// Every product and sum of this file is rounded on its own, like the elementwise tensor ops of vmap.py:343-457 it restates -
// and so that the instantiations / launch forms of the sampler (staged, unstaged, split) cannot differ by a fused
// multiply-add (they did, by one ulp of a bin edge, until this pragma).  The build's default (fast) is restored at the end.
#pragma clang fp contract(off)

constexpr int kWG = 256;
constexpr int kMaxS = 32;

struct SampleObject {                 // device-resident table, one entry per object
    const unsigned char* rgbs;        // [K][W][H][4]  RGB + pixel state (0 other, 1 this, 2 unknown), vmap.py:143-156
    const float* depth;               // [K][W][H]
    const float* t_wc;                // [K][4][4]     camera-to-world pose of every keyframe
    const float* bbox;                // [K][4]        u lo, u hi, v lo, v hi
    int n_keyframes;
    int last2[2];                     // the two latest keyframe slots (vmap.py:329-331)
    float center[3];                  // obj_center
    int obj_id;                       // shared-store mode: instance id of this object
    // Shared frame store.  slots == nullptr: rgbs/depth/t_wc are this object's own [K] buffers
    // as in the reference.  slots != nullptr: they are the arrays of ONE store shared by all objects, keyframe k of
    // this object lives in store slot slots[k], the 4th byte of a pixel is unused and the pixel state is derived from
    // the store's instance image (train.py:128-130: inst == obj_id -> 1 this, inst == -1 -> 2 unknown, else 0 other).
    const int* slots;                 // [K] store slot (< 256) of every keyframe of this object, or nullptr
    const int* inst;                  // [C][W][H] instance ids of the store, or nullptr
};

struct SampleRandoms {                // test mode: per-ray numbers supplied by the caller (all may be null)
    const int* kf_ids;                // [n][F]
    const float* u_w; const float* u_h;   // [n][F*P]
    const float* u_z;                 // [n][F*P][S]
    const float* g_z;                 // [n][F*P][n2]
};

struct SampleArgs {
    const SampleObject* objs;
    int n_obj, W, H, F, P, n1, n2;
    float fx, fy, cx, cy, min_bound, eps, stop_eps;
    unsigned seed_lo, seed_hi, frame_counter;
    SampleRandoms rnd;
    float* pcs; float* z; float* gt_depth; float* gt_rgb; unsigned char* sem; unsigned char* depth_mask;   // [n][F*P]...
    // ABI v7 (vmapstep_sample_frame_rays): the hand-off as rays - world-frame origin and direction per ray and the objects' centres; the
    // step kernels rebuild the points from them (load_point, step_kernels.h).  pcs may then be null (not written).
    float* ray_o; float* ray_d;       // [n][F*P][3] or null
    float* center_out;                // [n][3] or null
    // Split form (a workspace was given): nsplit workgroups per object, each on a contiguous slice of the object's rays.  The one
    // quantity that couples an object's rays - its maximum sampled depth (phase B) - comes from a first launch (frame_depth_max:
    // per-slice maxima joined by an order-independent integer atomic max into obj_max[k]).  nsplit = 0 / obj_max = null: one
    // workgroup per object does everything.
    int nsplit; int* obj_max;
};
// float <-> int keys whose signed order is the float order (for the atomic max; any finite float)
__device__ __forceinline__ int depth_key(float f) { const int i = (int)__float_as_uint(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float key_depth(int k) { return __uint_as_float((unsigned)(k >= 0 ? k : k ^ 0x7FFFFFFF)); }

struct U4 { unsigned x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = 0xD2511F53ull * c.x, p1 = 0xCD9E8D57ull * c.z;
        const U4 n = {(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// uniform in [0, 1) with 24 random bits (the granularity of torch.rand for float32)
__device__ __forceinline__ float u01(unsigned r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// torch.linspace(0, 1, nb + 1)[i] for float32: step evaluated from the nearer end
__device__ __forceinline__ float lin01(int i, int nb) {
    const float step = 1.0f / (float)nb;
    return i < (nb + 1) / 2 ? step * (float)i : 1.0f - step * (float)(nb - i);
}

// Phase A for one ray: keyframe slot of its frame, pixel inside that keyframe's box, the two gathers.  A pure function of
// (object, ray, counters): frames with more rays than the LDS staging area holds evaluate it twice instead of staging.
__device__ __forceinline__ void pick_pixel(const SampleArgs& a, const SampleObject& ob, int k, int ray, int FP,
                                           unsigned& pix_code, unsigned& rgba_out, float& d_out) {
    {
        const int f = ray / a.P;
        int kf;
        float uw, uh;
        if (a.rnd.kf_ids) {
            kf = a.rnd.kf_ids[k * a.F + f];
        } else if (ob.n_keyframes > 2 && f >= a.F - 2) {
            kf = ob.last2[f - (a.F - 2)];                                           // vmap.py:329-331
        } else {
            const U4 r = philox4x32_10({(unsigned)f, (unsigned)k, a.frame_counter, 0u}, a.seed_lo, a.seed_hi);
            kf = (int)(u01(r.x) * (float)ob.n_keyframes);
            kf = kf < ob.n_keyframes ? kf : ob.n_keyframes - 1;
        }
        if (a.rnd.u_w) {
            uw = a.rnd.u_w[k * FP + ray];
            uh = a.rnd.u_h[k * FP + ray];
        } else {
            const U4 r = philox4x32_10({(unsigned)ray, (unsigned)k, a.frame_counter, 1u}, a.seed_lo, a.seed_hi);
            uw = u01(r.x);
            uh = u01(r.y);
        }
        const float* bb = ob.bbox + 4 * kf;
        const int iw = (int)(uw * (bb[1] - bb[0]) + bb[0]);                          // vmap.py:347,350 (.long() truncates)
        const int ih = (int)(uh * (bb[3] - bb[2]) + bb[2]);
        const int slot = ob.slots ? ob.slots[kf] : kf;                               // where this keyframe's pixels and pose live
        const long long pix = ((long long)slot * a.W + iw) * a.H + ih;
        unsigned rgba = reinterpret_cast<const unsigned*>(ob.rgbs)[pix];             // vmap.py:353
        if (ob.inst) {
            const int id = ob.inst[pix];
            rgba = (rgba & 0x00FFFFFFu) | (id == ob.obj_id ? (1u << 24) : id == -1 ? (2u << 24) : 0u);
        }
        d_out = ob.depth[pix];                                                       // vmap.py:354
        pix_code = (unsigned)iw | ((unsigned)ih << 12) | ((unsigned)slot << 24);
        rgba_out = rgba;
    }
}

// Split form, launch 1: workgroup (object k, slice) -> max sampled depth of the slice -> atomic max into obj_max[k] (the caller
// memsets obj_max to 0x80 bytes first: a key below every depth).  Same pick_pixel as phase A, so the same pixels as launch 2.
template <int = 0>
__global__ __launch_bounds__(kWG) void frame_depth_max(const SampleArgs a) {
    float* s_red = wv::lds_base();
    const int tid = threadIdx.x, k = blockIdx.x / a.nsplit, part = blockIdx.x - k * a.nsplit;
    const SampleObject ob = a.objs[k];
    const int FP = a.F * a.P, per = (FP + a.nsplit - 1) / a.nsplit, r1 = min(FP, (part + 1) * per);
    float dmax = -3.0e38f;
    for (int ray = part * per + tid; ray < r1; ray += kWG) {
        unsigned px, rgba;
        float d;
        pick_pixel(a, ob, k, ray, FP, px, rgba, d);
        dmax = fmaxf(dmax, d);
    }
    s_red[tid] = dmax;
    __syncthreads();
    for (int w = kWG / 2; w > 0; w >>= 1) {
        if (tid < w) s_red[tid] = fmaxf(s_red[tid], s_red[tid + w]);
        __syncthreads();
    }
    if (tid == 0) atomicMax(a.obj_max + k, depth_key(s_red[0]));
}

// STAGED: phase A's per-ray results are kept in LDS for phase C (3 dwords per ray: F * P <= kMaxStagedRays); otherwise phase C
// evaluates phase A again (the background model's frame: 200 frames x 120 pixels = 24000 rays, train.py:197, cfg.py:67-69).
constexpr int kMaxStagedRays = 12000;
template <bool STAGED>
__global__ __launch_bounds__(kWG) void frame_sample(const SampleArgs a) {
    float* lds = wv::lds_base();
    const bool split = !STAGED && a.obj_max != nullptr;                // launch 2 of the split form: phases A / B were launch 1
    const int tid = threadIdx.x, k = split ? blockIdx.x / a.nsplit : blockIdx.x, part = split ? blockIdx.x - k * a.nsplit : 0;
    const SampleObject ob = a.objs[k];
    const int FP = a.F * a.P, S = a.n1 + a.n2;
    const int per = split ? (FP + a.nsplit - 1) / a.nsplit : FP, ray_begin = part * per, ray_end = min(FP, ray_begin + per);
    const int NST = STAGED ? FP : 0;
    float* s_dep = lds;                                       // [FP]
    unsigned* s_pix = reinterpret_cast<unsigned*>(lds + NST); // [FP]  iw | ih << 12 | kf << 24
    unsigned* s_rgba = s_pix + NST;                           // [FP]
    float* s_red = reinterpret_cast<float*>(s_rgba + NST);    // [kWG]

    // ---- A: pixel choice + gathers ----
    float dmax = -3.0e38f;
    if (!split)
    for (int ray = tid; ray < FP; ray += kWG) {
        unsigned px, rgba;
        float d;
        pick_pixel(a, ob, k, ray, FP, px, rgba, d);
        if (STAGED) {
            s_dep[ray] = d;
            s_pix[ray] = px;
            s_rgba[ray] = rgba;
        }
        dmax = fmaxf(dmax, d);
    }
    // ---- B: max sampled depth of this object (vmap.py:391) ----
    float max_bound;
    if (split) {
        max_bound = key_depth(a.obj_max[k]);
    } else {
        s_red[tid] = dmax;
        __syncthreads();
        for (int w = kWG / 2; w > 0; w >>= 1) {
            if (tid < w) s_red[tid] = fmaxf(s_red[tid], s_red[tid + w]);
            __syncthreads();
        }
        max_bound = s_red[0];
    }

    // ---- C: depth samples and points ----
    for (int ray = ray_begin + tid; ray < ray_end; ray += kWG) {
        unsigned px, rgba;
        float d;
        if (STAGED) { px = s_pix[ray]; rgba = s_rgba[ray]; d = s_dep[ray]; }
        else pick_pixel(a, ob, k, ray, FP, px, rgba, d);
        const int iw = px & 0xFFF, ih = (px >> 12) & 0xFFF, kf = px >> 24;       // kf = slot of the pose
        const unsigned state = rgba >> 24;
        const bool invalid = d <= a.min_bound;                                        // vmap.py:389
        const bool is_obj = state == 1u;
        float uz[kMaxS];
#pragma unroll
        for (int j0 = 0; j0 < kMaxS; j0 += 4) {
            if (j0 < S) {
                if (a.rnd.u_z) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) uz[j0 + j] = j0 + j < S ? a.rnd.u_z[((long long)k * FP + ray) * S + j0 + j] : 0.0f;
                } else {
                    const U4 r = philox4x32_10({(unsigned)ray, (unsigned)k, a.frame_counter, 2u + (unsigned)(j0 >> 2)}, a.seed_lo, a.seed_hi);
                    uz[j0] = u01(r.x); uz[j0 + 1] = u01(r.y); uz[j0 + 2] = u01(r.z); uz[j0 + 3] = u01(r.w);
                }
            }
        }
        float zs[kMaxS];
#pragma unroll
        for (int j = 0; j < kMaxS; ++j) zs[j] = 0.0f;
        if (invalid) {                                                                // vmap.py:395-399
            const float rng = max_bound - a.min_bound, len = rng / (float)S;
#pragma unroll
            for (int j = 0; j < kMaxS; ++j)
                if (j < S) zs[j] = (rng * lin01(j, S) + a.min_bound) + uz[j] * len;
        } else {
            {                                                                         // vmap.py:408-410
                const float rng = (d - a.eps) - a.min_bound, len = rng / (float)a.n1;
#pragma unroll
                for (int j = 0; j < kMaxS; ++j)
                    if (j < a.n1) zs[j] = (rng * lin01(j, a.n1) + a.min_bound) + uz[j] * len;
            }
            if (is_obj) {                                                             // vmap.py:425-430, :75-87
                float g[16];
#pragma unroll
                for (int j0 = 0; j0 < 16; j0 += 4) {
                    if (a.rnd.g_z) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) g[j0 + j] = j0 + j < a.n2 ? a.rnd.g_z[((long long)k * FP + ray) * a.n2 + j0 + j] : 3.0e38f;
                    } else {
                        const U4 r = philox4x32_10({(unsigned)ray, (unsigned)k, a.frame_counter, 16u + (unsigned)(j0 >> 2)}, a.seed_lo, a.seed_hi);
                        const float r0 = sqrtf(-2.0f * logf(1.0f - u01(r.x))), r1 = sqrtf(-2.0f * logf(1.0f - u01(r.z)));
                        const float t0 = 6.28318530718f * u01(r.y), t1 = 6.28318530718f * u01(r.w);
                        g[j0] = r0 * cosf(t0); g[j0 + 1] = r0 * sinf(t0); g[j0 + 2] = r1 * cosf(t1); g[j0 + 3] = r1 * sinf(t1);
                    }
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) g[j] = j < a.n2 ? g[j] * (a.eps / 3.0f) : 3.0e38f;   // normal_(0, delta/3); pad sorts last
#pragma unroll
                for (int pass = 0; pass < 16; ++pass) {                                // odd-even transposition sort
#pragma unroll
                    for (int j = pass & 1; j + 1 < 16; j += 2) {
                        const float lo = fminf(g[j], g[j + 1]), hi = fmaxf(g[j], g[j + 1]);
                        g[j] = lo; g[j + 1] = hi;
                    }
                }
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < a.n2) zs[a.n1 + j] = d + fminf(fmaxf(g[j], -a.eps), a.eps);   // clip, vmap.py:82-83
            } else {                                                                  // vmap.py:441-445
                const float lo = d - a.eps, rng = (d + a.stop_eps) - lo, len = rng / (float)a.n2;
#pragma unroll
                for (int j = 0; j < kMaxS; ++j)
                    if (j >= a.n1 && j < S) zs[j] = (rng * lin01(j - a.n1, a.n2) + lo) + uz[j] * len;
            }
        }
        // ray in the world frame (vmap.py:31-41, :507-516) and the sample points (vmap.py:452-454)
        const float* T = ob.t_wc + 16 * kf;
        const float dx = ((float)iw - a.cx) / a.fx, dy = ((float)ih - a.cy) / a.fy;
        const float wx = T[0] * dx + T[1] * dy + T[2], wy = T[4] * dx + T[5] * dy + T[6], wz = T[8] * dx + T[9] * dy + T[10];
        const float ox = T[3], oy = T[7], oz = T[11];
        const long long row = (long long)k * FP + ray;
        float* pz = a.z + row * S;
#pragma unroll
        for (int j = 0; j < kMaxS; ++j)
            if (j < S) pz[j] = zs[j];
        if (a.pcs) {
            float* pp = a.pcs + row * S * 3;
#pragma unroll
            for (int j = 0; j < kMaxS; ++j) {
                if (j < S) {
                    pp[3 * j + 0] = (ox + wx * zs[j]) - ob.center[0];
                    pp[3 * j + 1] = (oy + wy * zs[j]) - ob.center[1];
                    pp[3 * j + 2] = (oz + wz * zs[j]) - ob.center[2];
                }
            }
        }
        if (a.ray_o) {                                                                // the same three operations happen in the step kernel
            float* po = a.ray_o + row * 3;
            float* pd = a.ray_d + row * 3;
            po[0] = ox; po[1] = oy; po[2] = oz;
            pd[0] = wx; pd[1] = wy; pd[2] = wz;
            if (a.center_out && ray == 0) { a.center_out[3 * k] = ob.center[0]; a.center_out[3 * k + 1] = ob.center[1]; a.center_out[3 * k + 2] = ob.center[2]; }
        }
        a.gt_depth[row] = d;
        a.gt_rgb[row * 3 + 0] = (float)(rgba & 0xFF) / 255.0f;                         // train.py:257 gt_rgb / 255.
        a.gt_rgb[row * 3 + 1] = (float)((rgba >> 8) & 0xFF) / 255.0f;
        a.gt_rgb[row * 3 + 2] = (float)((rgba >> 16) & 0xFF) / 255.0f;
        a.sem[row] = (unsigned char)state;                                            // obj_labels
        a.depth_mask[row] = invalid ? 0 : 1;                                          // valid_depth_mask
    }
}

#pragma clang fp contract(fast)

}  // namespace vs
