// step_kernels.h - fused per-object training step of the vectorised object fields (gfx950 / CDNA4).
//
// Replaces, for one optimisation step, the ATen op stream the reference emits at
//   train.py:293-294  vmap(pe_model) / vmap(fc_model)      (embedding.py:82-91, model.py:54-85)
//   train.py:303-306  loss.step_batch_loss                  (loss.py:5-62, render_rays.py:4-8,26-96)
//   train.py:324-325  batch_loss.backward(); optimiser.step()
// with   step_prep (once per call)  ->  { step_main_h32 -> step_finalize } per step.
//
// step_main_h32 (the dominant kernel): one workgroup = 4 waves = up to 128 sample points (whole rays) of ONE
// object per pass.  The object's parameters live in global memory as a packed "image" that already has the LDS
// layout (built by step_prep, kept current by step_finalize), so staging is 12 asynchronous LDS-DMA loads per
// wave that land while the wave computes the positional encoding of its 32-point tile.  Every wave then runs
//   encoding -> 4 hidden layers + 2 heads -> [workgroup: alpha-compositing, loss, d/d(raw)] -> backward
// entirely out of registers + LDS.  All contractions run on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 in a "points-on-lanes" form that chains layer to layer without transposes:
//
//   P-form of X[point][feature] (32 features):  lane (p = l&31, hi = l>>5), register r  <->  X[p][phi(r,hi)],
//                                               phi(r,hi) = (r&3) + 8*(r>>2) + 4*hi   (the MFMA C/D row map)
//   forward   Y^T = W * X^T      : A = W[j][k] from LDS (lane = j, ds_read_b128), B = X in P-form -> Y in P-form
//   d-prop    dX^T = W^T * dY^T  : A = W[j][k] from LDS (lane = k), B = dY in P-form ->  dX in P-form
//   dW        dW = dY^T * X      : A = dY in F-form, B = X in F-form  (F-form: lane = feature, register r
//                                  <-> point r + 16*hi; obtained from P-form through a wave-private LDS tile)
// The backward is 13 hand-pipelined units of two 16-deep matrix chains each (p_chain / dw_chain_fin below).
//
// Numerics: fp32 throughout, ~1 ulp sin/cos/exp, IEEE division/sqrt; the only reorderings w.r.t. the reference
// are summation orders.  Weight gradients are reduced without atomics: across the 4 waves through staged LDS
// tiles (stage_put / stage_get), across workgroups by plain stores of partials + an ordered sum in step_finalize.
#pragma once
#include <wave_ops.h>   // resolved through -I: csrc/ (device) or tests/sim/ (CPU SIMT executor)

#ifndef VS_ABL                   /* measurement builds only (split_kernels.h) */
#define VS_ABL 0
#endif
namespace vk {

using wv::f32x16;

constexpr int kEmb1 = 87;      // trainer.py:16  xyz + octaves 0..3
constexpr int kEmb2 = 42;      // trainer.py:17  octaves 4..5
constexpr int kDirs = 21;      // embedding.py:51-73
constexpr int kNFc = 14;
constexpr int kWG = 256;
constexpr int kWaves = 4;
constexpr int kMaxPts = 128;   // sample points per workgroup pass (4 tiles of 32)
constexpr int kMarks = 16;     // phase timestamps per wave when StepArgs::timing is set
constexpr float kPi = 3.14159274101257324f;   // float32(np.pi), embedding.py:88

struct TensorRef {
    float* p;
    long long stride;   // elements between consecutive objects
};

// Everything one launch needs.  Passed by value as the kernel argument.
struct StepArgs {
    int n_obj, R, S, G, NG, NW, PP;    // G rays per pass, NG ray groups per object, NW workgroups per object
                                       // (workgroup w takes groups w, w+NW, ...), PP padded params per object
    int prep_steps; long long prep_ray_step;   // step_prep only: block b handles rays [b*ray_step, b*ray_step+R)
    int xcd_affine;                    // 1: block b -> object 8*((b>>3)/NW) + (b&7): an object's workgroups share one XCD/L2
    TensorRef fc[kNFc];                // the 14 field tensors, nn.Module.parameters() order (model.py:28-49)
    TensorRef pe_B;                    // B_layer.weight [n,21,3] (embedding.py:75-76)
    TensorRef pe_scale;                // scale buffer [n] (embedding.py:80)
    float* wimg;                       // [n][Lds32::IMGP] packed parameter image in LDS layout (workspace)
    const float* pcs; long long pcs_so, pcs_sr, pcs_ss, pcs_sc;
    const float* z; long long z_so, z_sr, z_ss;
    const float* gt_depth; long long gd_so, gd_sr;
    const float* gt_rgb; long long rgb_so, rgb_sr, rgb_sc;
    const unsigned char* sem; long long sem_so, sem_sr;
    const unsigned char* dmask; long long dm_so, dm_sr;
    float color_w, opac_w;             // loss.py:6 defaults 5.0 / 10.0
    float* stats;                      // [n][4]   mask counts N_depth&obj, N_obj, N_sem (exact in fp32), unused
    int* flags;                        // [4]      drop_depth, drop_colour, drop_opacity, explode
    float* part_grad;                  // [n][NW][PP]
    float* part_loss;                  // [n][NW][4]
    float* dbg_depth; float* dbg_rgb; float* dbg_opacity; float* dbg_var;   // [n][R](,3) or null
    unsigned* timing;                  // optional [workgroups][kWaves][kMarks] shader-clock stamps (diagnostics)
    int hidden;                        // H (step_prep packs with gen_layout(hidden); step_main_h32 requires 32)
    int weights_bf16;                  // 1: the parameter image holds the masters rounded to bfloat16 (RNE)
    int wide;                          // 1: step_main_wide (tile per workgroup, G*S <= 32) instead of step_main_gen; 3: step_main_ws (wsplit_kernels.h)
    int* img_tab;                      // step_prep: [PP] flat parameter -> image position table (or null)
    int bwd6;                          // 1: step_main_s32 with the six-product (float32-equivalent) backward (tuning.kernel = VMAPSTEP_KERNEL_S32_BWD6)
    int split;                         // 1: hidden 32 on the split-bf16 kernels (split_kernels.h); wimg is then the byte image of Img32s
    float* gen_scratch;                // step_main_gen / step_main_wide: per-wave (per-workgroup) register-image scratch (workspace)
                                       // step_main_ws (wide == 3): per-workgroup cos factors of the encoding
    int* tab_wt;                       // step_prep_ws / step_finalize_ws: [PP] flat parameter -> element of the W^T image planes (or -1)
    int tiles;                         // step_main_ws: 32-point tiles per round, 2 (default, also when 0) or 1 (single-tile rounds: launch plan)
    int* adam_counter;                 // device-resident optimiser step count (vmapstep_adamw::step_counter) or null: the first prep
                                       // block of a training call advances it by the previous call's steps (see prep_stats)
    // ABI v7, pcs == nullptr: the sampler's hand-off as rays - point = (ray_o + ray_d * z) - center (vmap.py:452-454), see load_point
    const float* ray_o; long long ro_so, ro_sr, ro_sc;
    const float* ray_d; long long rd_so, rd_sr, rd_sc;
    const float* center; long long ce_so;          // [n][3] or null (= zeros)
    // the workgroups' rows of partial gradients: PR floats each (= PP: the parameters' flat order; step_main_ws / _wp: RowWs<NB>::PR,
    // block-native order, with row_tab [PR] = the flat parameter behind every row element or -1, written by step_prep_ws)
    int PR; int* row_tab;
};

// Sample point `smp` of ray `ray` of object `obj` in the object frame: read from the points tensor (train.py:272 batch_input_pcs),
// or - pcs == nullptr, ABI v7 - rebuilt from the ray the sampler handed over, (o + d * z) - c with every operation rounded on its
// own: the arithmetic of vmap.py:452-454 and of frame_sample (sample_kernels.h), i.e. the bits the points tensor would hold.
__device__ __forceinline__ void load_point(const StepArgs& a, int obj, int ray, int smp, float& x0, float& x1, float& x2) {
    if (a.pcs) {
        const float* px = a.pcs + obj * a.pcs_so + ray * a.pcs_sr + smp * a.pcs_ss;
        x0 = px[0]; x1 = px[a.pcs_sc]; x2 = px[2 * a.pcs_sc];
        return;
    }
    {
#pragma clang fp contract(off)
        const float zz = a.z[obj * a.z_so + ray * a.z_sr + smp * a.z_ss];
        const float* o = a.ray_o + obj * a.ro_so + ray * a.ro_sr;
        const float* d = a.ray_d + obj * a.rd_so + ray * a.rd_sr;
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        if (a.center) { const float* c = a.center + obj * a.ce_so; c0 = c[0]; c1 = c[1]; c2 = c[2]; }
        x0 = (o[0] + d[0] * zz) - c0;
        x1 = (o[a.ro_sc] + d[a.rd_sc] * zz) - c1;
        x2 = (o[2 * a.ro_sc] + d[2 * a.rd_sc] * zz) - c2;
    }
}

__device__ __forceinline__ constexpr int phi(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---------------------------------------------------------------------------------------------------------
// LDS map for H = 32 (floats).  Weight images are row-major [out][ld]; ld = 4 * odd so that a row starts 16-byte
// aligned and the "lane = row" ds_read_b128 of the forward is bank-conflict free (16 lanes of a group hit 16
// distinct 16-byte slots), while the "lane = column" ds_read_b32 of the d-prop is unit-stride.  Padding columns
// are zero.  The same map, padded to IMGP floats, is the layout of the packed parameter image in global memory.
// ---------------------------------------------------------------------------------------------------------
struct Lds32 {
    static constexpr int H = 32;
    static constexpr int LD_IN = 92;             // 87 data + zero cols 87..91
    static constexpr int LD_M = 36;
    static constexpr int LD_CAT = H + 92;        // h2 part | e1 part (87) | zero cols
    static constexpr int LD_C = H + 52;          // h4 part | e2 part (42) | zero cols
    static constexpr int W_IN = 0;
    static constexpr int W_M1 = W_IN + H * LD_IN;
    static constexpr int W_CAT = W_M1 + H * LD_M;
    static constexpr int W_M2 = W_CAT + H * LD_CAT;
    static constexpr int W_C = W_M2 + H * LD_M;
    static constexpr int B_IN = W_C + H * LD_C;
    static constexpr int B_M1 = B_IN + H;
    static constexpr int B_CAT = B_M1 + H;
    static constexpr int B_M2 = B_CAT + H;
    static constexpr int B_C = B_M2 + H;
    static constexpr int W_A = B_C + H;
    static constexpr int W_OC = W_A + H;          // [3][H]
    static constexpr int B_A = W_OC + 3 * H;      // 1 (+3 pad)
    static constexpr int B_OC = B_A + 4;          // 3 (+1 pad)
    static constexpr int PE_B = B_OC + 4;         // [21][3] (+1 pad)
    static constexpr int IMG = PE_B + 64;         // floats in one parameter image
    static constexpr int IMGP = (IMG + 1023) / 1024 * 1024;   // padded to whole 4 x 1 KiB DMA rounds
    static constexpr int DMA_ROUNDS = IMGP / 1024;            // LDS-DMA instructions per wave
    static constexpr int WGT = 0;                 // weight image
    static constexpr int SMALL0 = B_IN;           // small vectors of the image: biases, heads, B
    static constexpr int SMALL_N = IMG - B_IN;
    static constexpr int TP = 36;                 // pitch of the 32x32 exchange tiles: 4 * odd, so the 16-byte reads of
                                                  // a row by 16 lanes tile the 64 banks exactly (MI355X_MICROARCH.md, LDS)
    static constexpr int SCR = IMGP;              // per-wave transpose scratch: kWaves x 2 x [32][TP]
    static constexpr int SCR_TILE = 32 * TP;
    static constexpr int SCR_WAVE = 2 * SCR_TILE;
    static constexpr int STG_TILE = 32 * TP;      // one staged 32x32 weight-gradient block
    static constexpr int STG = SCR + kWaves * SCR_WAVE;       // 2 buffers x kWaves tiles
    static constexpr int VEC = STG + 2 * kWaves * STG_TILE;   // per-wave private small-vector gradient accumulators
    static constexpr int CB = VEC + kWaves * SMALL_N;         // composite buffer [kMaxPts][8]
    static constexpr int LOSS = CB + kMaxPts * 8;             // per-wave loss partials [kWaves][4]
    static constexpr int TOTAL = LOSS + kWaves * 4;
    static constexpr int BYTES = TOTAL * 4;
};
static_assert(Lds32::BYTES <= 160 * 1024, "LDS budget");
static_assert(Lds32::W_M1 % 4 == 0 && Lds32::W_CAT % 4 == 0 && Lds32::W_M2 % 4 == 0 && Lds32::W_C % 4 == 0, "16-byte rows");

// start offsets of the tensors in the natural flat per-object order (14 field tensors, then B)
struct Flat32 {
    static constexpr int H = 32;
    static constexpr int W_IN = 0, B_IN = W_IN + H * kEmb1, W_M1 = B_IN + H, B_M1 = W_M1 + H * H;
    static constexpr int W_CAT = B_M1 + H, B_CAT = W_CAT + H * (H + kEmb1), W_M2 = B_CAT + H, B_M2 = W_M2 + H * H;
    static constexpr int W_A = B_M2 + H, B_A = W_A + H, W_C = B_A + 1, B_C = W_C + H * (H + kEmb2);
    static constexpr int W_OC = B_C + H, B_OC = W_OC + 3 * H, PE_B = B_OC + 3, P = PE_B + 63;
};

// position in the parameter image of element o of tensor t (t = 0..13 field tensors, 14 = B_layer.weight)
__device__ __forceinline__ int image_index(int t, int o) {
    using L = Lds32;
    switch (t) {
        case 0: { const int r = o / kEmb1; return L::W_IN + r * L::LD_IN + (o - r * kEmb1); }
        case 1: return L::B_IN + o;
        case 2: { const int r = o / 32; return L::W_M1 + r * L::LD_M + (o - r * 32); }
        case 3: return L::B_M1 + o;
        case 4: { const int r = o / (32 + kEmb1); return L::W_CAT + r * L::LD_CAT + (o - r * (32 + kEmb1)); }
        case 5: return L::B_CAT + o;
        case 6: { const int r = o / 32; return L::W_M2 + r * L::LD_M + (o - r * 32); }
        case 7: return L::B_M2 + o;
        case 8: return L::W_A + o;
        case 9: return L::B_A + o;
        case 10: { const int r = o / (32 + kEmb2); return L::W_C + r * L::LD_C + (o - r * (32 + kEmb2)); }
        case 11: return L::B_C + o;
        case 12: return L::W_OC + o;
        case 13: return L::B_OC + o;
        default: return L::PE_B + o;
    }
}


// ---------------------------------------------------------------------------------------------------------
// Runtime layout for any hidden width H (multiple of 32): the same image map as Lds32 with H-dependent pitches
// (all pitches and offsets multiples of 4 floats = 16 bytes) and the natural flat parameter order.  For H = 32 it
// reproduces Lds32 / Flat32 exactly (checked by static_asserts below).
// ---------------------------------------------------------------------------------------------------------
struct GenLayout {
    int H, NB;
    int ld_in, ld_m, ld_cat, ld_c;
    int w_in, w_m1, w_cat, w_m2, w_c, b_in, b_m1, b_cat, b_m2, b_c, w_a, w_oc, b_a, b_oc, pe_b, img, imgp;
    int small_n;                         // floats from b_in to the end of the image
    int f[16];                           // flat start offsets of the 15 tensors, f[15] = P
    int P, PP;
};
__host__ __device__ constexpr GenLayout gen_layout(int H) {
    GenLayout L{};
    L.H = H; L.NB = H / 32;
    L.ld_in = 92; L.ld_m = H + 4; L.ld_cat = H + 92; L.ld_c = H + 52;
    L.w_in = 0;
    L.w_m1 = L.w_in + H * L.ld_in;
    L.w_cat = L.w_m1 + H * L.ld_m;
    L.w_m2 = L.w_cat + H * L.ld_cat;
    L.w_c = L.w_m2 + H * L.ld_m;
    L.b_in = L.w_c + H * L.ld_c;
    L.b_m1 = L.b_in + H; L.b_cat = L.b_m1 + H; L.b_m2 = L.b_cat + H; L.b_c = L.b_m2 + H;
    L.w_a = L.b_c + H; L.w_oc = L.w_a + H; L.b_a = L.w_oc + 3 * H; L.b_oc = L.b_a + 4; L.pe_b = L.b_oc + 4;
    L.img = L.pe_b + 64;
    L.imgp = (L.img + 1023) / 1024 * 1024;
    L.small_n = L.img - L.b_in;
    const int sz[15] = {H * kEmb1, H, H * H, H, H * (H + kEmb1), H, H * H, H, H, 1, H * (H + kEmb2), H, 3 * H, 3, 63};
    int o = 0;
    for (int t = 0; t < 15; ++t) { L.f[t] = o; o += sz[t]; }
    L.f[15] = o;
    L.P = o;
    L.PP = (o + 63) / 64 * 64;
    return L;
}
static_assert(gen_layout(32).imgp == Lds32::IMGP && gen_layout(32).w_c == Lds32::W_C && gen_layout(32).pe_b == Lds32::PE_B &&
              gen_layout(32).ld_cat == Lds32::LD_CAT && gen_layout(32).ld_c == Lds32::LD_C && gen_layout(32).ld_m == Lds32::LD_M,
              "generic layout must reproduce the H = 32 LDS map");
static_assert(gen_layout(32).f[10] == Flat32::W_C && gen_layout(32).f[14] == Flat32::PE_B && gen_layout(32).P == Flat32::P, "flat order");

// position in the parameter image of element o of tensor t, any width
__device__ __forceinline__ int gen_image_index(const GenLayout& L, int t, int o) {
    const int H = L.H;
    switch (t) {
        case 0: { const int r = o / kEmb1; return L.w_in + r * L.ld_in + (o - r * kEmb1); }
        case 1: return L.b_in + o;
        case 2: { const int r = o / H; return L.w_m1 + r * L.ld_m + (o - r * H); }
        case 3: return L.b_m1 + o;
        case 4: { const int r = o / (H + kEmb1); return L.w_cat + r * L.ld_cat + (o - r * (H + kEmb1)); }
        case 5: return L.b_cat + o;
        case 6: { const int r = o / H; return L.w_m2 + r * L.ld_m + (o - r * H); }
        case 7: return L.b_m2 + o;
        case 8: return L.w_a + o;
        case 9: return L.b_a + o;
        case 10: { const int r = o / (H + kEmb2); return L.w_c + r * L.ld_c + (o - r * (H + kEmb2)); }
        case 11: return L.b_c + o;
        case 12: return L.w_oc + o;
        case 13: return L.b_oc + o;
        default: return L.pe_b + o;
    }
}

// inverse: which tensor element sits at image position x (false = zero padding)
__device__ __forceinline__ bool gen_image_source(const GenLayout& L, int x, int& t, int& o) {
    const int H = L.H;
    if (x < L.b_in) {
        int base, ld, nc;
        if (x < L.w_m1) { t = 0; base = L.w_in; ld = L.ld_in; nc = kEmb1; }
        else if (x < L.w_cat) { t = 2; base = L.w_m1; ld = L.ld_m; nc = H; }
        else if (x < L.w_m2) { t = 4; base = L.w_cat; ld = L.ld_cat; nc = H + kEmb1; }
        else if (x < L.w_c) { t = 6; base = L.w_m2; ld = L.ld_m; nc = H; }
        else { t = 10; base = L.w_c; ld = L.ld_c; nc = H + kEmb2; }
        const int r = (x - base) / ld, c = (x - base) - r * ld;
        o = r * nc + c;
        return c < nc;
    }
    if (x < L.b_m1) { t = 1; o = x - L.b_in; return true; }
    if (x < L.b_cat) { t = 3; o = x - L.b_m1; return true; }
    if (x < L.b_m2) { t = 5; o = x - L.b_cat; return true; }
    if (x < L.b_c) { t = 7; o = x - L.b_m2; return true; }
    if (x < L.w_a) { t = 11; o = x - L.b_c; return true; }
    if (x < L.w_oc) { t = 8; o = x - L.w_a; return true; }
    if (x < L.b_a) { t = 12; o = x - L.w_oc; return true; }
    if (x < L.b_oc) { t = 9; o = x - L.b_a; return o < 1; }
    if (x < L.pe_b) { t = 13; o = x - L.b_oc; return o < 3; }
    t = 14; o = x - L.pe_b;
    return o < 63;
}

// float32 -> nearest bfloat16 (ties to even), returned as float32
__device__ __forceinline__ float round_bf16(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}

__device__ __forceinline__ void load_bias(f32x16& acc, const float* b, int hi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = b[phi(r, hi)];
}
__device__ __forceinline__ void relu_to(float (&h)[16], const f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) h[r] = wv::relu(acc[r]);
}
__device__ __forceinline__ void zero_acc(f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
}

// forward: acc[p][j] += sum_k W[j][k0 + phi(r,hi)] * x[p][phi(r,hi)], r < 4*NQ;  wrow = &W[j = lane&31][k0 + 4*hi]
// (16-byte aligned): one ds_read_b128 feeds four matrix instructions (r = 4q..4q+3 <-> columns 8q..8q+3)
template <int NQ>
__device__ __forceinline__ void fwd_mm(f32x16& acc, const float* wrow, const float (&x)[16]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const wv::f32x4 w = *reinterpret_cast<const wv::f32x4*>(wrow + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = wv::mfma32(w[i], x[4 * q + i], acc);
    }
}
// dW: acc[j][k] += sum_q dyF[q][j] * xF[q][k]
__device__ __forceinline__ void dw_mm(f32x16& acc, const float (&dyF)[16], const float (&xF)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = wv::mfma32(dyF[r], xF[r], acc);
}
// bias gradient: sum over the 32 points of a tile of dY (F-form), lane = feature; accumulated in the wave's
// private small-vector area (single owner lane per address -> plain read-modify-write)
__device__ __forceinline__ void add_db(float* gb, const float (&dyF)[16], int p31, int hi) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += dyF[r];
    s += wv::swap_half(s);
    if (hi == 0) gb[p31] += s;
}
// the same sum kept in a register (written to the small-vector area once, after the last unit: an LDS
// read-modify-write between two chains is a fully exposed round trip)
__device__ __forceinline__ float db_sum(const float (&dyF)[16]) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += dyF[r];
    return s + wv::swap_half(s);
}
// this wave's quarter of a reduced block -> natural row-major tensor (row length K), columns col0 .. col0+ncols-1
template <int K>
__device__ __forceinline__ void store_quarter(float* out, const float (&q)[4], int col0, int ncols, int wave, int p31, int hi) {
    if (p31 < ncols) {
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(q[i], &out[(8 * wave + 4 * hi + i) * K + col0 + p31]);
    }
}
// ---- split-phase forms (the backward is software-pipelined by hand: one wave per SIMD has nobody else to hide an
// LDS round trip or a barrier, so every round trip is started BEFORE a 16-deep matrix-instruction chain and consumed
// after it) ----
template <int LD>
__device__ __forceinline__ void bwd_w(float (&w)[16], const float* wcol) {
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = wcol[((r & 3) + 8 * (r >> 2)) * LD];
}
__device__ __forceinline__ void bwd_fma(f32x16& acc, const float (&w)[16], const float (&dy)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = wv::mfma32(w[r], dy[r], acc);
}
// Exchange tiles.  One wave with nobody to share its SIMD gets a fifth of the LDS rate on 4-byte reads but the full
// rate on 16-byte reads and on stores, so every tile is laid out for 16-byte reads:
//   transpose tile [feature][point]: P-form lanes store single dwords (lane = point: consecutive addresses), F-form
//     lanes (lane = feature) read their 16 points as four 16-byte loads;
//   staging tile [column][row]: a lane's registers 4j..4j+3 are rows 8j+4hi..+3 of its column: four 16-byte stores;
//     the reducing wave reads its rows 8w+4hi..+3 of the four waves' tiles as four 16-byte loads.
__device__ __forceinline__ void toF_put(float* scr, const float (&P)[16], int p31, int hi) {
    wv::wave_lds_fence();   // earlier reads of this tile are ordered before the overwrite
#pragma unroll
    for (int r = 0; r < 16; ++r) scr[phi(r, hi) * Lds32::TP + p31] = P[r];
    wv::wave_lds_fence();
}
__device__ __forceinline__ void toF_get(float (&F)[16], const float* scr, int p31, int hi) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const wv::f32x4 v = *reinterpret_cast<const wv::f32x4*>(scr + p31 * Lds32::TP + 16 * hi + 4 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i) F[4 * j + i] = v[i];
    }
}
__device__ __forceinline__ void stage_put(float* stage, const f32x16& acc, int wave, int p31, int hi) {
    float* mine = stage + wave * Lds32::STG_TILE + p31 * Lds32::TP + 4 * hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wv::f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[4 * j + i];
        *reinterpret_cast<wv::f32x4*>(mine + 8 * j) = v;
    }
}
// after the workgroup barrier that follows stage_put: this wave's quarter (rows 8w..8w+7) of the four staged tiles
__device__ __forceinline__ void stage_get(float (&q)[4], const float* stage, int wave, int p31, int hi) {
    const float* rd = stage + p31 * Lds32::TP + 8 * wave + 4 * hi;
    const wv::f32x4 t0 = *reinterpret_cast<const wv::f32x4*>(rd);
    const wv::f32x4 t1 = *reinterpret_cast<const wv::f32x4*>(rd + Lds32::STG_TILE);
    const wv::f32x4 t2 = *reinterpret_cast<const wv::f32x4*>(rd + 2 * Lds32::STG_TILE);
    const wv::f32x4 t3 = *reinterpret_cast<const wv::f32x4*>(rd + 3 * Lds32::STG_TILE);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] += (t0[i] + t1[i]) + (t2[i] + t3[i]);
}
// d-prop chain of one unit (16 matrix instructions, acc += W^T dy) with the unit's LDS traffic threaded through it, a
// few DS instructions after every matrix instruction: staging of the previous weight-gradient tile, the P->F
// transposes of this unit's delta (HAS_D) and input block (HAS_X).  Issued as one burst the DS queue back-pressures
// instruction issue and the matrix pipe idles; spread over the chain most of it is hidden.
template <bool HAS_D, bool HAS_X, bool HAS_STAGE>
__device__ __forceinline__ void p_chain(f32x16& accP, const float (&w)[16], const float (&dy)[16],
                                        float* scrD, const float (&dP)[16], float (&dF)[16],
                                        float* scrX, const float (&xP)[16], float (&xF)[16],
                                        float* stage, const f32x16& accPrev, int wave, int p31, int hi) {
    constexpr int N_ST = HAS_STAGE ? 4 : 0, N_D = HAS_D ? 16 : 0, N_X = HAS_X ? 16 : 0;
    constexpr int N_GD = HAS_D ? 4 : 0, N_GX = HAS_X ? 4 : 0;
    constexpr int PUT0 = N_ST, GET0 = PUT0 + N_D + N_X, TOTAL = GET0 + N_GD + N_GX;
    constexpr int PER = (TOTAL + 2) / 3;             // DS work after the 4th, 8th and 12th matrix instruction; the
                                                     // last four cover its latency before the barrier that follows.
                                                     // Matrix instructions stay in back-to-back groups of four: a
                                                     // dependent one issued right behind its producer forwards the
                                                     // accumulator, any instruction in between costs ~40 cycles.
    float* mine = stage + wave * Lds32::STG_TILE + p31 * Lds32::TP + 4 * hi;
    wv::wave_lds_fence();
    wv::sched_fence();          // the VALU work before the chain stays before it
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int r = 4 * g; r < 4 * g + 4; ++r) accP = wv::mfma32(w[r], dy[r], accP);
        wv::sched_fence();
#pragma unroll
        for (int i = g * PER; i < (g + 1) * PER; ++i) {
            if (g == 3 || i >= TOTAL) continue;
            if (i == GET0) wv::wave_lds_fence();
            if (i < PUT0) {
                wv::f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accPrev[4 * i + e];
                *reinterpret_cast<wv::f32x4*>(mine + 8 * i) = v;
            } else if (i < PUT0 + N_D) {
                const int k = i - PUT0;
                scrD[phi(k, hi) * Lds32::TP + p31] = dP[k];
            } else if (i < GET0) {
                const int k = i - PUT0 - N_D;
                scrX[phi(k, hi) * Lds32::TP + p31] = xP[k];
            } else if (i < GET0 + N_GD) {
                const int j = i - GET0;
                const wv::f32x4 v = *reinterpret_cast<const wv::f32x4*>(scrD + p31 * Lds32::TP + 16 * hi + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) dF[4 * j + e] = v[e];
            } else {
                const int j = i - GET0 - N_GD;
                const wv::f32x4 v = *reinterpret_cast<const wv::f32x4*>(scrX + p31 * Lds32::TP + 16 * hi + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) xF[4 * j + e] = v[e];
            }
        }
        wv::sched_fence();
    }
}
// weight-gradient chain of one unit (acc += dyF^T xF) that also fetches the weight column operands of the NEXT unit's
// d-prop chain (LDN = row pitch of that matrix, 0 = nothing to fetch)
template <int LDN>
__device__ __forceinline__ void dw_chain(f32x16& acc, const float (&dyF)[16], const float (&xF)[16],
                                         float (&wn)[16], const float* wcol_next) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int r = 4 * g; r < 4 * g + 4; ++r) acc = wv::mfma32(dyF[r], xF[r], acc);
        wv::sched_fence();
        if (LDN > 0 && g < 2) {
#pragma unroll
            for (int k = 8 * g; k < 8 * g + 8; ++k) wn[k] = wcol_next[((k & 3) + 8 * (k >> 2)) * LDN];
        }
        wv::sched_fence();
    }
}
// the same chain that also finishes the PREVIOUS block (after the workgroup barrier that follows its staging): the
// four staged tiles are read after the first group of matrix instructions and summed / stored after the third, so
// neither the LDS latency nor the store address arithmetic sits between two chains
template <int LDN, bool MULTI, int K>
__device__ __forceinline__ void dw_chain_fin(f32x16& acc, const float (&dyF)[16], const float (&xF)[16],
                                             float (&wn)[16], const float* wcol_next,
                                             float (&qp)[4], const float* stage, float* out, int col0, int ncols,
                                             int wave, int p31, int hi) {
    wv::f32x4 t0, t1, t2, t3;
    const float* rd = stage + p31 * Lds32::TP + 8 * wave + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int r = 4 * g; r < 4 * g + 4; ++r) acc = wv::mfma32(dyF[r], xF[r], acc);
        wv::sched_fence();
        if (g == 0) {
            t0 = *reinterpret_cast<const wv::f32x4*>(rd);
            t1 = *reinterpret_cast<const wv::f32x4*>(rd + Lds32::STG_TILE);
            t2 = *reinterpret_cast<const wv::f32x4*>(rd + 2 * Lds32::STG_TILE);
            t3 = *reinterpret_cast<const wv::f32x4*>(rd + 3 * Lds32::STG_TILE);
        }
        if (LDN > 0 && g < 2) {
#pragma unroll
            for (int k = 8 * g; k < 8 * g + 8; ++k) wn[k] = wcol_next[((k & 3) + 8 * (k >> 2)) * LDN];
        }
        if (g == 2) {
            if (MULTI) {
#pragma unroll
                for (int i = 0; i < 4; ++i) qp[i] += (t0[i] + t1[i]) + (t2[i] + t3[i]);
            } else {
                float q[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = 0.0f + ((t0[i] + t1[i]) + (t2[i] + t3[i]));
                store_quarter<K>(out, q, col0, ncols, wave, p31, hi);
            }
        }
        wv::sched_fence();
    }
}
template <bool MULTI, int K>
__device__ __forceinline__ void finish_block(float (&qp)[4], const float* stage, float* out, int col0, int ncols,
                                             int wave, int p31, int hi) {
    if (MULTI) {
        stage_get(qp, stage, wave, p31, hi);
    } else {
        float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        stage_get(q, stage, wave, p31, hi);
        store_quarter<K>(out, q, col0, ncols, wave, p31, hi);
    }
}

// sin and cos of a float32 argument, ~1.5 ulp, branch-free; valid for |x| < 2^20 (the encoding's arguments are
// 2^f * pi * proj with |proj| of a few units).  Cody-Waite reduction by pi/2 in three float32 pieces with FMAs,
// then the classic degree-7/8 minimax polynomials on [-pi/4, pi/4].  Replaces the library sincosf, whose
// inlined Payne-Hanek path costs ~100 instructions per call site; tiles that do hold a larger argument take
// the library path instead (decided once per wave, see kSinCosFastLimit).
constexpr float kSinCosFastLimit = 1048576.0f;
__device__ __forceinline__ void sincos_f32(float x, float& s, float& c) {
    const float n = rintf(x * 0.636619772367581343f);              // x * 2/pi
    float r = fmaf(-n, 1.57079637050628662109375f, x);             // pi/2 = C1 + C2 + C3
    r = fmaf(-n, -4.37113900018624283e-8f, r);
    r = fmaf(-n, -1.71512449026012451e-15f, r);
    const int q = (int)n & 3;
    const float r2 = r * r;
    const float sp = fmaf(r * r2, fmaf(r2, fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f),
                                        4.166664568298827e-2f), fmaf(r2, -0.5f, 1.0f));
    const float ss = (q & 1) ? cp : sp;
    const float cc = (q & 1) ? sp : cp;
    s = __uint_as_float(__float_as_uint(ss) ^ (((unsigned)q << 30) & 0x80000000u));          // (q & 2) ? -ss : ss
    c = __uint_as_float(__float_as_uint(cc) ^ ((((unsigned)q + 1u) << 30) & 0x80000000u));   // ((q + 1) & 2) ? -cc : cc
}

// The same for N independent arguments in lockstep: every step is applied to all N before the next one, so the
// N dependency chains interleave in program order (one wave per SIMD has no other wave to hide VALU latency).
template <int N>
__device__ __forceinline__ void sincos_f32xN(const float (&x)[N], float (&s)[N], float (&c)[N]) {
    float n[N], r[N], r2[N], sp[N], cp[N];
    int q[N];
#pragma unroll
    for (int k = 0; k < N; ++k) n[k] = rintf(x[k] * 0.636619772367581343f);
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = fmaf(-n[k], 1.57079637050628662109375f, x[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = fmaf(-n[k], -4.37113900018624283e-8f, r[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) r[k] = fmaf(-n[k], -1.71512449026012451e-15f, r[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) { q[k] = (int)n[k] & 3; r2[k] = r[k] * r[k]; }
#pragma unroll
    for (int k = 0; k < N; ++k) { sp[k] = fmaf(r2[k], -1.9515295891e-4f, 8.3321608736e-3f); cp[k] = fmaf(r2[k], 2.443315711809948e-5f, -1.388731625493765e-3f); }
#pragma unroll
    for (int k = 0; k < N; ++k) { sp[k] = fmaf(r2[k], sp[k], -1.6666654611e-1f); cp[k] = fmaf(r2[k], cp[k], 4.166664568298827e-2f); }
#pragma unroll
    for (int k = 0; k < N; ++k) { sp[k] = fmaf(r[k] * r2[k], sp[k], r[k]); cp[k] = fmaf(r2[k] * r2[k], cp[k], fmaf(r2[k], -0.5f, 1.0f)); }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float ss = (q[k] & 1) ? cp[k] : sp[k];
        const float cc = (q[k] & 1) ? sp[k] : cp[k];
        // quadrant signs as sign-bit flips (the same values as `(q & 2) ? -ss : ss` and `((q + 1) & 2) ? -cc : cc`, zeros
        // included): bit 1 of q / of q + 1 shifted into bit 31 and xor-ed in - a shift and one three-operand bit
        // instruction each instead of and + compare + select
        const unsigned qu = (unsigned)q[k];
        s[k] = __uint_as_float(__float_as_uint(ss) ^ ((qu << 30) & 0x80000000u));
        c[k] = __uint_as_float(__float_as_uint(cc) ^ (((qu + 1u) << 30) & 0x80000000u));
    }
}

// One embedding slot.  c = index into the 129-wide encoding (embedding.py:85-89: 3 + f*21 + d), or -1 = padding.
__device__ __forceinline__ void pe_slot(int c, const float (&t)[3], const float (&proj)[kDirs],
                                        float& pre, float& tv, float& fac) {
    pre = 0.0f; tv = 0.0f; fac = 0.0f;
    if (c >= 0 && c < 3) {
        tv = t[c];
    } else if (c >= 3) {
        const int f = (c - 3) / kDirs, d = (c - 3) % kDirs;
        const float band = (float)(1 << f);
        fac = kPi * band;              // d sin(x*pi*band)/dx = cos(.) * pi * band; exact: band is a power of two
        pre = proj[d] * fac;           // == fl32(fl32(proj * band) * fl32(pi)) of embedding.py:85,88 (scaling by 2^f commutes
                                       //    with rounding), one multiply instead of two
    }
}
// One 32-feature block of the encoding in P-form (+ cos * pi * 2^f for the backward).  base = first encoding
// index of this group (0 for e1, 87 for e2), limit = width of the group, kb = block within the group.
template <int NSTEPS, bool BIG>
__device__ __forceinline__ void pe_block(float (&e)[16], float (&cf)[16], int base, int limit, int kb,
                                         const float (&t)[3], const float (&proj)[kDirs], int hi) {
    // pass 1: arguments (or xyz values / padding) of the 16 slots; pass 2: sin/cos four slots at a time
    float arg[16], fac[16], tv[16];
    bool is_sin[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        arg[r] = 0.0f; fac[r] = 0.0f; tv[r] = 0.0f; is_sin[r] = false;
        if (r < NSTEPS) {
            const int l0 = 32 * kb + phi(r, 0), l1 = 32 * kb + phi(r, 1);
            const int c0 = l0 < limit ? base + l0 : -1, c1 = l1 < limit ? base + l1 : -1;
            float pre0, tv0, fac0, pre1, tv1, fac1;
            pe_slot(c0, t, proj, pre0, tv0, fac0);
            pe_slot(c1, t, proj, pre1, tv1, fac1);
            arg[r] = hi ? pre1 : pre0;                         // fl32(xb * fl32(pi)), embedding.py:88
            fac[r] = hi ? fac1 : fac0;                          // 0 for xyz / padding slots
            tv[r] = hi ? tv1 : tv0;
            is_sin[r] = hi ? (c1 >= 3) : (c0 >= 3);
        }
    }
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {
        float a4[4] = {arg[r0], arg[r0 + 1], arg[r0 + 2], arg[r0 + 3]}, s4[4], c4[4];
        if (r0 < NSTEPS) {
            if (BIG) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sincosf(a4[k], &s4[k], &c4[k]);
            } else {
                sincos_f32xN<4>(a4, s4, c4);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + k;
            e[r] = r < NSTEPS ? (is_sin[r] ? s4[k] : tv[r]) : 0.0f;
            cf[r] = r < NSTEPS ? c4[k] * fac[r] : 0.0f;
        }
    }
}
// gradient w.r.t. the 21 projections from one block of d(encoding)
template <int NSTEPS>
__device__ __forceinline__ void pe_block_bwd(float (&dproj)[kDirs], const f32x16& de, const float (&cf)[16],
                                             int base, int limit, int kb, int hi) {
#pragma unroll
    for (int r = 0; r < NSTEPS; ++r) {
        const int l0 = 32 * kb + phi(r, 0), l1 = 32 * kb + phi(r, 1);
        const int c0 = l0 < limit ? base + l0 : -1, c1 = l1 < limit ? base + l1 : -1;
        const float g = de[r] * cf[r];
        if (c0 >= 3) dproj[(c0 - 3) % kDirs] += hi ? 0.0f : g;
        if (c1 >= 3) dproj[(c1 - 3) % kDirs] += hi ? g : 0.0f;
    }
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float sgnf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// ---------------------------------------------------------------------------------------------------------
// Workgroup phase between forward and backward: per-ray compositing, loss and d loss / d raw
// (loss.py:24-60, render_rays.py:26-96).  cb = composite buffer [kMaxPts][8]: in  row[0..3] = occupancy, colour,
// row[6] = z ; out row[0..3] = d/d(raw alpha), d/d(raw colour).  loss_cells = per-wave loss partials [kWaves][4].
// ---------------------------------------------------------------------------------------------------------
// per-ray ground truth / masks / normalisers of the ray a lane composites (loaded early so the latency is hidden)
struct RayMeta {
    unsigned char sem, dm;
    float gtd, q0, q1, q2;
    float inv_dd, inv_o, inv_s;
};
// the same in two halves: the per-ray vector loads (issued early, e.g. in front of the MLP forward) ...
__device__ __forceinline__ RayMeta load_ray_meta_rays(const StepArgs& a, int obj, int rr) {
    RayMeta m;
    m.sem = a.sem[obj * a.sem_so + rr * a.sem_sr];
    m.dm = a.dmask[obj * a.dm_so + rr * a.dm_sr];
    m.gtd = a.gt_depth[obj * a.gd_so + rr * a.gd_sr];
    const float* rgb = a.gt_rgb + obj * a.rgb_so + rr * a.rgb_sr;
    m.q0 = rgb[0]; m.q1 = rgb[a.rgb_sc]; m.q2 = rgb[2 * a.rgb_sc];
    m.inv_dd = m.inv_o = m.inv_s = 0.0f;
    return m;
}
// ... and the per-object scalar part (switches and normalisers), where the compositing starts
__device__ __forceinline__ RayMeta finish_ray_meta(const StepArgs& a, int obj, RayMeta m) {
    m.inv_dd = a.flags[0] ? 0.0f : 1.0f / (a.stats[obj * 4 + 0] + 1e-10f);            // render_rays.py:68-73,87
    m.inv_o = a.flags[1] ? 0.0f : 1.0f / (a.stats[obj * 4 + 1] + 1e-10f);
    m.inv_s = a.flags[2] ? 0.0f : 1.0f / (a.stats[obj * 4 + 2] + 1e-10f);
    return m;
}
__device__ __forceinline__ RayMeta load_ray_meta(const StepArgs& a, int obj, int rr) {
    RayMeta m;
    m.sem = a.sem[obj * a.sem_so + rr * a.sem_sr];
    m.dm = a.dmask[obj * a.dm_so + rr * a.dm_sr];
    m.gtd = a.gt_depth[obj * a.gd_so + rr * a.gd_sr];
    const float* rgb = a.gt_rgb + obj * a.rgb_so + rr * a.rgb_sr;
    m.q0 = rgb[0]; m.q1 = rgb[a.rgb_sc]; m.q2 = rgb[2 * a.rgb_sc];
    m.inv_dd = a.flags[0] ? 0.0f : 1.0f / (a.stats[obj * 4 + 0] + 1e-10f);            // render_rays.py:68-73,87
    m.inv_o = a.flags[1] ? 0.0f : 1.0f / (a.stats[obj * 4 + 1] + 1e-10f);
    m.inv_s = a.flags[2] ? 0.0f : 1.0f / (a.stats[obj * 4 + 2] + 1e-10f);
    return m;
}

template <bool BWD, int NWAVES = kWaves>
__device__ __forceinline__ void composite_phase(const StepArgs& a, float* cb, float* loss_cells, int obj, int ray0, int nrays,
                                                int wave, int lane, int tid, const RayMeta& pre) {
    // ---- per-ray compositing, loss and d loss / d raw  (loss.py:24-60, render_rays.py:26-96) ----
    if (__builtin_expect(a.S <= 16, 1)) {
        // 16 lanes per ray: lane i of a group holds sample i; products/sums are scans and butterflies
        for (int g0 = 4 * wave; g0 < nrays; g0 += 4 * NWAVES) {      // wave-uniform trip count
            const int g = g0 + (lane >> 4), i = lane & 15;
            const bool on = g < nrays && i < a.S;
            const int rr = ray0 + min(g, nrays - 1);
            float* row = cb + (min(g, nrays - 1) * a.S + min(i, a.S - 1)) * 8;
            const float o = on ? row[0] : 0.0f, c0 = on ? row[1] : 0.0f, c1 = on ? row[2] : 0.0f, c2 = on ? row[3] : 0.0f;
            const float zi = on ? row[6] : 0.0f;
            const float f = on ? (1.0f - o) + 1e-10f : 1.0f;         // render_rays.py:29
            // Transmittance, depth and variance in SEQUENTIAL sample order (lane broadcasts): on saturated rays the
            // variance is rounding noise and 1/(sqrt(V)+1e-4) amplifies it, so the reference's order is mirrored
            // (cumprod, then product tensor, then sum; render_rays.py:31-32,47-51) instead of a tree.
            // Lanes past the ray's last sample hold f = 1 and w = 0, so running every loop over all 16 row lanes adds
            // exact ones / zeros and leaves the sequential result over the S samples unchanged.
            float T = 1.0f;
#define VK_T(J) { const float fj = wv::row_bcast<J>(f); T = J < i ? T * fj : T; }
            if (!(VS_ABL & 8192)) {
            VK_T(0) VK_T(1) VK_T(2) VK_T(3) VK_T(4) VK_T(5) VK_T(6) VK_T(7)
            VK_T(8) VK_T(9) VK_T(10) VK_T(11) VK_T(12) VK_T(13) VK_T(14)
            }
#undef VK_T
            const float w = o * T;                                     // render_rays.py:32
            const float wz = w * zi;
            float D = 0.0f;
#define VK_D(J) D += wv::row_bcast<J>(wz);
            if (VS_ABL & 8192) D = wz; else {
            VK_D(0) VK_D(1) VK_D(2) VK_D(3) VK_D(4) VK_D(5) VK_D(6) VK_D(7)
            VK_D(8) VK_D(9) VK_D(10) VK_D(11) VK_D(12) VK_D(13) VK_D(14) VK_D(15)           // loss.py:27
            }
#undef VK_D
            const float dz = zi - D;
            const float wd = w * (dz * dz);
            float V = 0.0f;
#define VK_V(J) V += wv::row_bcast<J>(wd);
            if (VS_ABL & 8192) V = wd; else {
            VK_V(0) VK_V(1) VK_V(2) VK_V(3) VK_V(4) VK_V(5) VK_V(6) VK_V(7)
            VK_V(8) VK_V(9) VK_V(10) VK_V(11) VK_V(12) VK_V(13) VK_V(14) VK_V(15)           // loss.py:28-29 (detached)
            }
#undef VK_V
            const float O = wv::row_sum16(w);                           // loss.py:31
            const float C0 = wv::row_sum16(w * c0), C1 = wv::row_sum16(w * c1), C2 = wv::row_sum16(w * c2);   // loss.py:30
            const RayMeta mt = g0 == 4 * wave ? pre : load_ray_meta(a, obj, rr);    // first round: prefetched by the caller
            const unsigned char s = mt.sem, dm = mt.dm;
            const float m_o = s != 0 ? 1.0f : 0.0f, m_s = s != 2 ? 1.0f : 0.0f;      // loss.py:16-19
            const float m_dd = (dm != 0 && s != 0) ? 1.0f : 0.0f;                     // loss.py:37
            const float gtd = mt.gtd, q0 = mt.q0, q1 = mt.q1, q2 = mt.q2;
            const float inv_dd = mt.inv_dd, inv_o = mt.inv_o, inv_s = mt.inv_s;
            const float info = (VS_ABL & 16384) ? V + 1e-4f : 1.0f / (sqrtf(V) + 1e-4f);   // render_rays.py:75-79
            const float rd = D - gtd, rc0 = C0 - q0, rc1 = C1 - q1, rc2 = C2 - q2, ro = O - m_o;
            const bool lead = g < nrays && i == 0;
            float ld = lead ? fabsf(rd) * m_dd * info * inv_dd : 0.0f;
            float lc = lead ? (fabsf(rc0) + fabsf(rc1) + fabsf(rc2)) * m_o * inv_o : 0.0f;
            float lo = lead ? fabsf(ro) * m_s * inv_s : 0.0f;
            if (!(VS_ABL & 32768)) {
            ld += wv::shfl(ld, lane ^ 16); lc += wv::shfl(lc, lane ^ 16); lo += wv::shfl(lo, lane ^ 16);
            ld += wv::swap_half(ld); lc += wv::swap_half(lc); lo += wv::swap_half(lo);
            }
            if (lane == 0 && !(VS_ABL & 32768)) {                      // this wave's private loss partials
                loss_cells[wave * 4 + 0] += ld;
                loss_cells[wave * 4 + 1] += lc;
                loss_cells[wave * 4 + 2] += lo;
            }
            if (lead) {
                if (a.dbg_depth) a.dbg_depth[obj * a.R + rr] = D;
                if (a.dbg_opacity) a.dbg_opacity[obj * a.R + rr] = O;
                if (a.dbg_var) a.dbg_var[obj * a.R + rr] = V;
                if (a.dbg_rgb) {
                    float* q = a.dbg_rgb + (obj * a.R + rr) * 3;
                    q[0] = C0; q[1] = C1; q[2] = C2;
                }
            }
            if (BWD) {
                const float gD = sgnf(rd) * m_dd * info * inv_dd;
                const float gO = a.opac_w * sgnf(ro) * m_s * inv_s;
                const float k_c = a.color_w * m_o * inv_o;
                const float gC0 = k_c * sgnf(rc0), gC1 = k_c * sgnf(rc1), gC2 = k_c * sgnf(rc2);
                const float gw = gD * zi + gC0 * c0 + gC1 * c1 + gC2 * c2 + gO;
                const float gww = on ? gw * w : 0.0f;
                // sum_{k>i} g_w_k * w_k, accumulated directly from the last sample down (no total-minus-prefix:
                // that difference cancels catastrophically and is then divided by f, which can be 1e-10)
                float suffix = 0.0f;
#define VK_S(J) { const float gj = wv::row_bcast<J>(gww); suffix = J > i ? suffix + gj : suffix; }
                if (!(VS_ABL & 8192)) {
                VK_S(15) VK_S(14) VK_S(13) VK_S(12) VK_S(11) VK_S(10) VK_S(9) VK_S(8)
                VK_S(7) VK_S(6) VK_S(5) VK_S(4) VK_S(3) VK_S(2) VK_S(1)
                }
#undef VK_S
                const float d_occ = (VS_ABL & 16384) ? gw * T - suffix * f : gw * T - suffix / f;               // cumprod backward: reverse-cumsum / input
                if (on) {
                    row[0] = 10.0f * (d_occ * o * (1.0f - o));         // through sigmoid and the *10 (model.py:77)
                    row[1] = w * gC0 * c0 * (1.0f - c0);               // through the colour sigmoid (model.py:83)
                    row[2] = w * gC1 * c1 * (1.0f - c1);
                    row[3] = w * gC2 * c2 * (1.0f - c2);
                }
            }
        }
    } else if (tid < nrays) {
        // long rays (S > 16): one lane per ray, sequential over the samples
        const int g = tid, rr = ray0 + g;
        float* rows = cb + g * a.S * 8;
        float T = 1.0f, D = 0.0f, O = 0.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
        for (int i = 0; i < a.S; ++i) {
            float* row = rows + i * 8;
            const float o = row[0];
            const float w = o * T;
            const float zi = row[6];
            row[4] = T; row[5] = w;
            D += w * zi; O += w;
            C0 += w * row[1]; C1 += w * row[2]; C2 += w * row[3];
            T *= (1.0f - o) + 1e-10f;
        }
        float V = 0.0f;
        for (int i = 0; i < a.S; ++i) {
            const float* row = rows + i * 8;
            const float d = row[6] - D;
            V += row[5] * (d * d);
        }
        const unsigned char s = a.sem[obj * a.sem_so + rr * a.sem_sr];
        const unsigned char dm = a.dmask[obj * a.dm_so + rr * a.dm_sr];
        const float m_o = s != 0 ? 1.0f : 0.0f, m_s = s != 2 ? 1.0f : 0.0f;
        const float m_dd = (dm != 0 && s != 0) ? 1.0f : 0.0f;
        const float gtd = a.gt_depth[obj * a.gd_so + rr * a.gd_sr];
        const float* rgb = a.gt_rgb + obj * a.rgb_so + rr * a.rgb_sr;
        const float q0 = rgb[0], q1 = rgb[a.rgb_sc], q2 = rgb[2 * a.rgb_sc];
        const float inv_dd = a.flags[0] ? 0.0f : 1.0f / (a.stats[obj * 4 + 0] + 1e-10f);
        const float inv_o = a.flags[1] ? 0.0f : 1.0f / (a.stats[obj * 4 + 1] + 1e-10f);
        const float inv_s = a.flags[2] ? 0.0f : 1.0f / (a.stats[obj * 4 + 2] + 1e-10f);
        const float info = 1.0f / (sqrtf(V) + 1e-4f);
        const float rd = D - gtd, rc0 = C0 - q0, rc1 = C1 - q1, rc2 = C2 - q2, ro = O - m_o;
        wv::lds_add(loss_cells + 0, fabsf(rd) * m_dd * info * inv_dd);
        wv::lds_add(loss_cells + 1, (fabsf(rc0) + fabsf(rc1) + fabsf(rc2)) * m_o * inv_o);
        wv::lds_add(loss_cells + 2, fabsf(ro) * m_s * inv_s);
        if (a.dbg_depth) a.dbg_depth[obj * a.R + rr] = D;
        if (a.dbg_opacity) a.dbg_opacity[obj * a.R + rr] = O;
        if (a.dbg_var) a.dbg_var[obj * a.R + rr] = V;
        if (a.dbg_rgb) {
            float* q = a.dbg_rgb + (obj * a.R + rr) * 3;
            q[0] = C0; q[1] = C1; q[2] = C2;
        }
        if (BWD) {
            const float gD = sgnf(rd) * m_dd * info * inv_dd;
            const float gO = a.opac_w * sgnf(ro) * m_s * inv_s;
            const float k_c = a.color_w * m_o * inv_o;
            const float gC0 = k_c * sgnf(rc0), gC1 = k_c * sgnf(rc1), gC2 = k_c * sgnf(rc2);
            float suffix = 0.0f;
            for (int i = a.S - 1; i >= 0; --i) {
                float* row = rows + i * 8;
                const float o = row[0], c0 = row[1], c1 = row[2], c2 = row[3];
                const float Ti = row[4], w = row[5], zi = row[6];
                const float gw = gD * zi + gC0 * c0 + gC1 * c1 + gC2 * c2 + gO;
                const float f = (1.0f - o) + 1e-10f;
                const float d_occ = gw * Ti - suffix / f;
                suffix += gw * w;
                row[0] = 10.0f * (d_occ * o * (1.0f - o));
                row[1] = w * gC0 * c0 * (1.0f - c0);
                row[2] = w * gC1 * c1 * (1.0f - c1);
                row[3] = w * gC2 * c2 * (1.0f - c2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// step_prep, one launch per API call:
//   blocks [0, prep_steps)            per-object mask counts of step b and the batch-wide "any object has an
//                                     empty mask" switches (loss.py:16-19,38,46,56; render_rays.py:68-73)
//   blocks [prep_steps, +n_obj*imgp/1024)  pack the objects' 15 tensors into their LDS-layout parameter images
// ---------------------------------------------------------------------------------------------------------
// mask statistics of optimisation step `step` of the frame (one workgroup): per-object counts + the batch-wide switches
__device__ __forceinline__ void prep_stats(const StepArgs& a, int step, int table_len_P, int table_len_PP) {
    const int tid = threadIdx.x;
    // one wave per object (objects w, w+4, ...): per-lane counts over the rays, a shuffle tree, no workgroup barrier
    // per object (the old form - one block reduction per object, 20 in sequence - took 25 us per frame)
    float* lds = wv::lds_base();
    int* dropw = reinterpret_cast<int*>(lds);          // [kWaves]
    const int lane = tid & 63, wave = tid >> 6;
    // Device-resident step count of the fused AdamW (a frame call replayed as a graph has frozen arguments, so the bias
    // corrections of its steps cannot come from the host): [0] = updates applied before THIS call, [1] = steps of this call,
    // folded into [0] by the next call's first launch.  Single writer, stream-ordered against every reader (the finalize
    // kernels of this call read [0] only).
    if (step == 0 && tid == 0 && a.adam_counter) { a.adam_counter[0] += a.adam_counter[1]; a.adam_counter[1] = a.prep_steps; }
    if (step == 0 && a.img_tab) {                      // padding entries; the real ones are written by the pack blocks of object 0
        for (int i = table_len_P + tid; i < table_len_PP; i += kWG) a.img_tab[i] = 0;
    }
    const unsigned char* sem = a.sem + step * a.prep_ray_step * a.sem_sr;
    const unsigned char* dmask = a.dmask + step * a.prep_ray_step * a.dm_sr;
    float* stats = a.stats + (long long)step * a.n_obj * 4;
    int* flags = a.flags + step * 4;
    int drop = 0;                                      // bit 0 depth, bit 1 colour, bit 2 opacity
    for (int k = wave; k < a.n_obj; k += kWaves) {
        float nd = 0.0f, no = 0.0f, ns = 0.0f;         // counts < 2^24: exact in float32
        for (int r = lane; r < a.R; r += 64) {
            const unsigned char s = sem[k * a.sem_so + r * a.sem_sr];
            const unsigned char dm = dmask[k * a.dm_so + r * a.dm_sr];
            const bool mo = s != 0, ms = s != 2;
            nd += ((dm != 0) && mo) ? 1.0f : 0.0f;
            no += mo ? 1.0f : 0.0f;
            ns += ms ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            nd += wv::shfl(nd, lane ^ m); no += wv::shfl(no, lane ^ m); ns += wv::shfl(ns, lane ^ m);
        }
        if (lane == 0) {
            stats[k * 4 + 0] = nd;          // raw counts: a ray-sharded caller (shared background model) sums them over
            stats[k * 4 + 1] = no;          // ranks before the step loop; the normaliser 1/(N+1e-10) is formed in step_main
            stats[k * 4 + 2] = ns;
            stats[k * 4 + 3] = 0.0f;
        }
        drop |= (nd == 0.0f ? 1 : 0) | (no == 0.0f ? 2 : 0) | (ns == 0.0f ? 4 : 0);
    }
    if (lane == 0) dropw[wave] = drop;
    __syncthreads();
    if (tid == 0) {
        int d = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) d |= dropw[w];
        flags[0] = d & 1; flags[1] = (d >> 1) & 1; flags[2] = (d >> 2) & 1; flags[3] = 0;
    }
}

template <int = 0>
__global__ __launch_bounds__(kWG) void step_prep(const StepArgs a) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= a.prep_steps) {
        // pack: imgp / 1024 workgroups per object, one 16-byte image slot per thread (destination-major: the zero
        // padding is written by the same pass, no barrier, coalesced stores)
        const GenLayout L = gen_layout(a.hidden);
        const int per_obj = L.imgp / 1024;
        const int b = blockIdx.x - a.prep_steps;
        const int k = b / per_obj;
        const int x0 = (b - k * per_obj) * 1024 + 4 * tid;
        wv::f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int t, o;
            float f = 0.0f;
            if (gen_image_source(L, x0 + e, t, o)) {
                f = t < kNFc ? a.fc[t].p[k * a.fc[t].stride + o] : a.pe_B.p[k * a.pe_B.stride + o];
                if (k == 0 && a.img_tab) a.img_tab[L.f[t] + o] = x0 + e;     // flat parameter -> image position (the inverse map, for free)
            }
            v[e] = a.weights_bf16 ? round_bf16(f) : f;
        }
        *reinterpret_cast<wv::f32x4*>(a.wimg + (long long)k * L.imgp + x0) = v;
        return;
    }
    const GenLayout LL = gen_layout(a.hidden);
    prep_stats(a, blockIdx.x, LL.P, LL.PP);
}

// ---------------------------------------------------------------------------------------------------------
// step_finalize: ordered sum of the per-workgroup partials -> gradients (and, fused, torch.optim.AdamW's
// single-tensor update: decoupled decay, lerp first moment, bias-corrected denominator; the updated value is
// written to the parameter tensor AND to the packed image the next step's step_main stages from), the scalar
// loss (loss.py:59-60) and the "loss explode" flag (render_rays.py:88-90).
// The two halves are device functions (finalize_quad: four consecutive flat parameters of one object;
// finalize_loss: the loss / flag reduction of one workgroup) shared by the finalize kernels of all widths.
// ---------------------------------------------------------------------------------------------------------
struct FinalizeArgs {
    int n_obj, NW, PP, P;              // NW partials per object; P = real parameter count per object (flat order: 14 field tensors, then B)
    int offs[kNFc + 2];                // flat start offset of each tensor, offs[15] = P
    TensorRef param[kNFc + 1];         // parameters (updated in place when do_adam)
    TensorRef grad[kNFc + 1];          // gradient outputs (p may be null: skip)
    float* m; float* v;                // Adam moments, [n][PP] slabs (when do_adam)
    float* wimg;                       // [n][imgp] packed parameter image (updated when do_adam)
    int hidden;                        // H: image layout = gen_layout(hidden)
    int weights_bf16;                  // 1: image values are rounded to bfloat16
    const float* part_grad; const float* part_loss;
    const int* flags_in; int* flags_out;
    float* loss_out;                   // [1]
    float* terms_out;                  // optional [n_obj][4]: per-object depth / colour / opacity terms (unweighted) and l_batch (loss.py:59)
    float color_w, opac_w;
    int do_adam;
    int have_grad;                     // 0: forward-only call, skip the gradient/optimiser part
    int xcd_affine;                    // block -> object map that keeps an object on the XCD step_main used for it
    // AdamW constants, evaluated by the host in double and rounded once (as torch's Python-side scalars are)
    float decay, one_minus_beta1, beta2, one_minus_beta2, eps, step_size, bias_corr2_sqrt;
    // ... or, for graph replay, the two step-dependent ones from a host-built table indexed by the device-resident step count:
    // adam_tab[2 t] = lr / (1 - beta1^(t+1)), adam_tab[2 t + 1] = sqrt(1 - beta2^(t+1)), t = adam_cnt[0] + adam_i (clamped)
    const float* adam_tab; const int* adam_cnt; int adam_i, adam_len;
    int ws_grouped;                    // launcher hint (tuning.generic_finalize on the step_main_ws / _wp path): step_finalize_ws always with a
                                       // thread per quad AND row group, never the one-thread-per-quad form the launcher picks for many blocks / few rows
    int loss_stage;                    // loss partials (16 B each) the launch's LDS has room for behind the loss block's reduction scratch
                                       // (set by the launcher, loss_stage_cap); 0: the loss block reads them from memory one by one
    int PR; const int* row_tab;        // floats per row of part_grad; step_finalize_ws: row element -> flat parameter (or -1), null = the rows are in flat order
};
// LDS of a finalize launch as the loss block sees it: kWG floats + kWG ints of reduction scratch, then the staging area of the partials
constexpr int kLossRedBytes = 2 * kWG * 4;
constexpr int kLossStageMax = 1024;    // what the 256-thread finalize kernels make room for (16 KB)
__host__ __device__ inline int loss_stage_cap(size_t lds_bytes) { return lds_bytes > (size_t)kLossRedBytes ? (int)((lds_bytes - kLossRedBytes) / 16) : 0; }
// LDS bytes of a 256-thread finalize launch (step_finalize, _h32, _s32) and the capacity to put into FinalizeArgs::loss_stage
__host__ __device__ inline size_t loss_lds_bytes(int n_obj, int NW) {
    const long long total = (long long)n_obj * NW;
    return (size_t)kLossRedBytes + (total <= kLossStageMax ? (size_t)total * 16 : 0);
}
// the two step-dependent AdamW constants of this launch (wave-uniform scalar loads in table mode)
template <class Consts>
__device__ __forceinline__ void adam_step_consts(const FinalizeArgs& f, const Consts& c, float& step_size, float& bias_corr2_sqrt) {
    step_size = c.step_size; bias_corr2_sqrt = c.bias_corr2_sqrt;
    if (f.adam_tab) {
        const int t = min(f.adam_cnt[0] + f.adam_i, f.adam_len - 1);
        step_size = f.adam_tab[2 * t]; bias_corr2_sqrt = f.adam_tab[2 * t + 1];
    }
}

// torch.optim.AdamW, single-tensor form, one element
template <class Consts>      // FinalizeArgs or FinalizeHot: same field names
__device__ __forceinline__ void adamw_elem(const Consts& a, float step_size, float bias_corr2_sqrt, float ge, float& p, float& m, float& v) {
    // every operation rounded on its own, like the eager tensor ops it restates - and so that the two places this is
    // inlined into (step_finalize, the carried finalize) cannot end up with different fused forms
#pragma clang fp contract(off)
    p = p * a.decay;                                          // param.mul_(1 - lr * wd)
    m = m + (ge - m) * a.one_minus_beta1;                     // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (ge * ge) * a.one_minus_beta2;          // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    const float denom = sqrtf(v) / bias_corr2_sqrt + a.eps;
    p = p - step_size * (m / denom);                        // param.addcdiv_(exp_avg, denom, -lr / bc1)
}

// flat parameters 4*q4 .. 4*q4+3 of object obj.  The partial rows and the moment slabs are PP-pitched (PP % 64 == 0,
// 256-byte aligned) so they move as 16-byte vectors; parameter / gradient tensors are scattered.  WT: the image is
// written with write-through stores (read by other workgroups of the same launch).
template <bool WT>
__device__ __forceinline__ void finalize_quad(const FinalizeArgs& a, const GenLayout& GL, int obj, int q4) {
    const int i0 = 4 * q4;
    const wv::f32x4* pg = reinterpret_cast<const wv::f32x4*>(a.part_grad + (long long)obj * a.NW * a.PP + i0);
    // element addresses and the parameter values first: their loads travel with the partials' instead of forming a chain
    // of four round trips behind them (a parameter load cannot be moved above the previous element's stores by the compiler)
    float* pp[4]; float* gp[4]; float pv[4]; int img[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = min(i0 + e, a.P - 1);
        int t = 0;
#pragma unroll
        for (int k = 1; k <= kNFc; ++k) t += i >= a.offs[k];
        const int o = i - a.offs[t];
        pp[e] = a.param[t].p + obj * a.param[t].stride + o;
        gp[e] = a.grad[t].p ? a.grad[t].p + obj * a.grad[t].stride + o : nullptr;
        img[e] = gen_image_index(GL, t, o);
        pv[e] = a.do_adam ? *pp[e] : 0.0f;
    }
    wv::f32x4 g = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        // ordered sum over the NW partials; eight independent 16-byte loads in flight per lane (the kernel is
        // latency-bound: 240 workgroups of one wave per SIMD, 10 MB to read)
        const long long qs = a.PP / 4;
        int q = 0;
        for (; q + 8 <= a.NW; q += 8) {
            wv::f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = pg[(q + u) * qs];
#pragma unroll
            for (int u = 0; u < 8; ++u) g += t[u];
        }
        wv::f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = q + u < a.NW ? pg[(q + u) * qs] : wv::f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (q + u < a.NW) g += t[u];
    }
    float ss, bc;
    adam_step_consts(a, a, ss, bc);
    const long long s = (long long)obj * a.PP + i0;
    wv::f32x4 m4 = {0.0f, 0.0f, 0.0f, 0.0f}, v4 = m4;
    if (a.do_adam) {
        m4 = *reinterpret_cast<const wv::f32x4*>(a.m + s);
        v4 = *reinterpret_cast<const wv::f32x4*>(a.v + s);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = i0 + e;
        if (i < a.P) {
            const float ge = g[e];
            if (gp[e]) *gp[e] = ge;
            if (a.do_adam) {
                float p = pv[e], m = m4[e], v = v4[e];
                adamw_elem(a, ss, bc, ge, p, m, v);
                *pp[e] = p; m4[e] = m; v4[e] = v;
                float* ip = a.wimg + (long long)obj * GL.imgp + img[e];
                const float pw = a.weights_bf16 ? round_bf16(p) : p;
                if (WT) wv::store_wt(ip, pw); else *ip = pw;
            }
        }
    }
    if (a.do_adam) {
        *reinterpret_cast<wv::f32x4*>(a.m + s) = m4;
        *reinterpret_cast<wv::f32x4*>(a.v + s) = v4;
    }
}

// per-object loss terms: thread q sums object q, q+256, ... ; then a block reduction (loss.py:59-60).  One whole
// workgroup; uses the first 2 KiB of its LDS.  flags_in[3] != 0 (a carried finalize that timed out) is passed on as bit 1.
__device__ __forceinline__ void finalize_loss(const FinalizeArgs& a) {
    if (!a.loss_out) return;           // optimiser-only call (vmapstep_adamw_apply): no loss partials to reduce (uniform exit)
    float* red = wv::lds_base();       // kWG floats + kWG ints
    int* redi = reinterpret_cast<int*>(red + kWG);
    // The partials are summed in workgroup order by ONE thread per object (the order is part of the result).  Read from memory in that
    // loop they are NW dependent round trips - 200 for the one-object background step: the loss block, not the row reads, bounded that
    // launch (round 5: ~23 us against 14 for the rows).  So the whole block first copies them to LDS, one round trip, same order after.
    const int total = a.n_obj * a.NW;
    const bool staged = total <= a.loss_stage;
    wv::f32x4* stage = reinterpret_cast<wv::f32x4*>(redi + kWG);
    if (staged) {
        for (int i = threadIdx.x; i < total; i += blockDim.x) stage[i] = *reinterpret_cast<const wv::f32x4*>(a.part_loss + 4ll * i);
        __syncthreads();
    }
    float loss = 0.0f;
    int explode = 0;
    const bool act = threadIdx.x < kWG;  // blocks wider than kWG threads (step_finalize_ws): the extra waves only take part in the barriers
    for (int k = act ? (int)threadIdx.x : a.n_obj; k < a.n_obj; k += kWG) {
        float ld = 0.0f, lc = 0.0f, lo = 0.0f;
        for (int q = 0; q < a.NW; ++q) {
            const long long i = (long long)k * a.NW + q;
            wv::f32x4 pl;
            if (staged) pl = stage[i];
            else pl = *reinterpret_cast<const wv::f32x4*>(a.part_loss + i * 4);
            ld += pl[0]; lc += pl[1]; lo += pl[2];
        }
        explode |= (ld > 100000.0f) || (lc > 100000.0f) || (lo > 100000.0f);   // render_rays.py:88
        const float lb = ld + lc * a.color_w + lo * a.opac_w;                   // loss.py:59
        loss += lb;
        if (a.terms_out) { a.terms_out[4 * k] = ld; a.terms_out[4 * k + 1] = lc; a.terms_out[4 * k + 2] = lo; a.terms_out[4 * k + 3] = lb; }
    }
    if (act) {
        red[threadIdx.x] = loss;
        redi[threadIdx.x] = explode;
    }
    __syncthreads();
    for (int w = kWG / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[threadIdx.x] += red[threadIdx.x + w];
            redi[threadIdx.x] |= redi[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.loss_out[0] = red[0];                                                 // loss.py:60
        const int err = a.flags_in[3] ? 2 : 0;
        a.flags_out[0] = a.flags_in[0]; a.flags_out[1] = a.flags_in[1]; a.flags_out[2] = a.flags_in[2];
        a.flags_out[3] = redi[0] | err;
    }
}

// The fields the table-driven finalize kernels read per quad, side by side in the kernel argument segment (the by-field
// accesses of FinalizeArgs cost a dozen dependent scalar round trips in front of the first vector load).
struct FinalizeHot {
    float* m; float* v; const float* part_grad; float* wimg;
    const int* img_tab;                // [PP] image position of every flat parameter (step_prep)
    float* slab; long long slab_stride;   // non-null: the 15 parameter tensors are views of one [n, >= P] slab in flat order
                                       // (flat parameter i of object k at slab[k * slab_stride + i]): no per-element lookup
    int NW, PP, weights_bf16;
    float decay, one_minus_beta1, beta2, one_minus_beta2, eps, step_size, bias_corr2_sqrt;
    int PR;                            // floats per row of part_grad (step_finalize_ws; = PP elsewhere)
};
// tensor index and offset inside it of flat parameter i (hidden 32, compile-time offsets)
__device__ __forceinline__ void flat32_tensor_of(int i, int& t, int& o) {
    using F = Flat32;
    constexpr int offs[16] = {F::W_IN, F::B_IN, F::W_M1, F::B_M1, F::W_CAT, F::B_CAT, F::W_M2, F::B_M2,
                              F::W_A, F::B_A, F::W_C, F::B_C, F::W_OC, F::B_OC, F::PE_B, F::P};
    t = 0;
    int base = 0;
#pragma unroll
    for (int k = 1; k <= kNFc; ++k) {
        const bool ge = i >= offs[k];
        t += ge ? 1 : 0;
        base = ge ? offs[k] : base;
    }
    o = i - base;
}

// ---------------------------------------------------------------------------------------------------------
// step_main_h32
// MULTI = a workgroup covers several ray groups (NW < NG): reduced gradient quarters persist in registers over
// the passes and are stored once at the end; otherwise (one pass per workgroup) each quarter is stored as soon
// as it is reduced and no accumulator registers are held.
// ---------------------------------------------------------------------------------------------------------
template <bool BWD, bool MULTI, bool STAMPS>
__device__ __forceinline__ void step_main_body(const StepArgs& a) {
    using L = Lds32;
    using F = Flat32;
    constexpr int H = 32;
    float* lds = wv::lds_base();
    float* W = lds + L::WGT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p31 = lane & 31, hi = lane >> 5;
    float* Gv = lds + L::VEC + wave * L::SMALL_N - L::SMALL0;   // this wave's private small-vector gradients
    // Block -> (object, workgroup-of-object).  The dispatcher places block b on XCD b % 8 (measured: tests/tools/xcd_probe.hip,
    // profiles/r01m_xcd_probe.jsonl); with the
    // affine map all workgroups of an object sit on one XCD, so its parameter image is fetched into that L2 once
    // instead of once per workgroup.  Pure speed/traffic choice: any placement is correct.
    int obj, wgo;
    if (a.xcd_affine) {
        const int slot = blockIdx.x >> 3;
        const int og = slot / a.NW;
        obj = og * 8 + (blockIdx.x & 7);
        wgo = slot - og * a.NW;
        if (obj >= a.n_obj) return;
    } else {
        obj = blockIdx.x / a.NW;
        wgo = blockIdx.x - obj * a.NW;
    }
    unsigned* tmark = STAMPS && a.timing ? a.timing + ((long long)blockIdx.x * kWaves + wave) * kMarks : nullptr;
#define VK_MARK(i) do { if constexpr (STAMPS) { if (tmark && lane == 0) tmark[i] = wv::clock32(); } } while (0)
    VK_MARK(0);
    if (BWD) {
        for (int i = tid; i < kWaves * L::SMALL_N; i += kWG) lds[L::VEC + i] = 0.0f;
    }
    if (tid < kWaves * 4) lds[L::LOSS + tid] = 0.0f;
    float* out = a.part_grad + ((long long)(obj * a.NW + wgo)) * a.PP;   // this workgroup's partial gradients
    float qacc[13][4];      // MULTI only: this wave's quarter of every reduced weight-gradient block
#pragma unroll
    for (int b = 0; b < 13; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) qacc[b][i] = 0.0f;
    }
    float* stg0 = lds + L::STG;
    float* stg1 = stg0 + kWaves * L::STG_TILE;
    float* scrX = lds + L::SCR + wave * L::SCR_WAVE;
    float* scrD = scrX + L::SCR_TILE;
    float* cb = lds + L::CB;
    const float* cbw = cb + wave * 32 * 8;       // this wave's 32 rows of the composite buffer
    const float scale = a.pe_scale.p[obj * a.pe_scale.stride];
    const float* Bg = a.wimg + (long long)obj * L::IMGP + L::PE_B;   // B_layer.weight, read from the (global) image

    const int tid_k = tid;
    for (int grp = wgo; grp < a.NG; grp += a.NW) {   // ---- one pass = up to kMaxPts points (whole rays) ----
    // lane coordinates of this pass (shadowing the kernel-scope ones): opaque per iteration in multi-pass mode
    const int tid = MULTI ? wv::opaque_iter(tid_k) : tid_k;
    const int lane = tid & 63, wave = tid >> 6, p31 = lane & 31, hi = lane >> 5;
    __syncthreads();                                 // previous pass finished reading the composite buffer
    for (int i = tid; i < kMaxPts * 8; i += kWG) cb[i] = 0.0f;   // padding rows must read as zero

    // ---- this lane's sample point ----
    const int ray0 = grp * a.G;
    const int nrays = min(a.G, a.R - ray0);
    const int npts = nrays * a.S;
    const int pt = wave * 32 + p31;
    const bool valid = pt < npts;
    const int lray = valid ? pt / a.S : 0;
    const int smp = valid ? pt - lray * a.S : 0;
    const int ray = ray0 + lray;
    float t[3] = {0.0f, 0.0f, 0.0f};
    if (valid) {
        float x0, x1, x2;
        load_point(a, obj, ray, smp, x0, x1, x2);
        t[0] = x0 / scale;                 // embedding.py:83  x / self.scale
        t[1] = x1 / scale;
        t[2] = x2 / scale;
    }
    float proj[kDirs];
#pragma unroll
    for (int d = 0; d < kDirs; ++d)        // embedding.py:84 B_layer(tensor); B straight from the global image (wave-uniform)
        proj[d] = fmaf(t[2], Bg[3 * d + 2], fmaf(t[1], Bg[3 * d + 1], t[0] * Bg[3 * d]));

    // ---- first pass: start the asynchronous copy of the parameter image into LDS (lands during the encoding) ----
    if (grp == wgo) {
        const float* src = a.wimg + (long long)obj * L::IMGP + wave * 256 + lane * 4;
#pragma unroll
        for (int c = 0; c < L::DMA_ROUNDS; ++c) wv::glds16(src + c * 1024, W + c * 1024 + wave * 256);
    }
    VK_MARK(1);

    // ---- encoding (embedding.py:82-91), P-form blocks ----
    float e1a[16], e1b[16], e1c[16], e2a[16], e2b[16];          // sin / xyz values
    float c1a[16], c1b[16], c1c[16], c2a[16], c2b[16];          // cos * pi * 2^f
    {
        float amax = 0.0f;
#pragma unroll
        for (int d = 0; d < kDirs; ++d) amax = fmaxf(amax, fabsf(proj[d]));
        if (__builtin_expect(!wv::wave_any(!(amax * (32.0f * kPi) < kSinCosFastLimit)), 1)) {   // the library path is laid out after the hot code
            pe_block<16, false>(e1a, c1a, 0, kEmb1, 0, t, proj, hi);
            pe_block<16, false>(e1b, c1b, 0, kEmb1, 1, t, proj, hi);
            pe_block<12, false>(e1c, c1c, 0, kEmb1, 2, t, proj, hi);
            pe_block<16, false>(e2a, c2a, kEmb1, kEmb2, 0, t, proj, hi);
            pe_block<6, false>(e2b, c2b, kEmb1, kEmb2, 1, t, proj, hi);
        } else {   // cold: some point of this tile is absurdly far from its object (or not finite)
            pe_block<16, true>(e1a, c1a, 0, kEmb1, 0, t, proj, hi);
            pe_block<16, true>(e1b, c1b, 0, kEmb1, 1, t, proj, hi);
            pe_block<12, true>(e1c, c1c, 0, kEmb1, 2, t, proj, hi);
            pe_block<16, true>(e2a, c2a, kEmb1, kEmb2, 0, t, proj, hi);
            pe_block<6, true>(e2b, c2b, kEmb1, kEmb2, 1, t, proj, hi);
        }
    }
    VK_MARK(2);
    __syncthreads();        // parameter image landed (the barrier drains the LDS-DMA), composite buffer zeroed

    // ---- field MLP forward (model.py:59-83) ----
    float h1[16], h2[16], h3[16], h4[16], hc[16];
    f32x16 acc;
    {
        const float* w = W + L::W_IN + p31 * L::LD_IN + 4 * hi;
        load_bias(acc, W + L::B_IN, hi);
        fwd_mm<4>(acc, w, e1a);
        fwd_mm<4>(acc, w + 32, e1b);
        fwd_mm<3>(acc, w + 64, e1c);
        relu_to(h1, acc);                                        // :59 in_layer
    }
    {
        load_bias(acc, W + L::B_M1, hi);
        fwd_mm<4>(acc, W + L::W_M1 + p31 * L::LD_M + 4 * hi, h1);
        relu_to(h2, acc);                                        // :60 mid1
    }
    {
        const float* w = W + L::W_CAT + p31 * L::LD_CAT + 4 * hi;
        load_bias(acc, W + L::B_CAT, hi);
        fwd_mm<4>(acc, w, h2);                                   // :63 cat((fc2, x[:emb1]))
        fwd_mm<4>(acc, w + H, e1a);
        fwd_mm<4>(acc, w + H + 32, e1b);
        fwd_mm<3>(acc, w + H + 64, e1c);
        relu_to(h3, acc);                                        // :64 cat_layer
    }
    {
        load_bias(acc, W + L::B_M2, hi);
        fwd_mm<4>(acc, W + L::W_M2 + p31 * L::LD_M + 4 * hi, h3);
        relu_to(h4, acc);                                        // :67 mid2
    }
    {
        const float* w = W + L::W_C + p31 * L::LD_C + 4 * hi;
        load_bias(acc, W + L::B_C, hi);
        fwd_mm<4>(acc, w, h4);                                   // :81 cat((fc4, x[emb1:]))
        fwd_mm<4>(acc, w + H, e2a);
        fwd_mm<2>(acc, w + H + 32, e2b);
        relu_to(hc, acc);                                        // :81 color_linear
    }
    VK_MARK(3);
    {
        float ra = 0.0f, r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = phi(r, hi);
            ra = fmaf(W[L::W_A + j], h4[r], ra);                 // :71 out_alpha
            r0 = fmaf(W[L::W_OC + j], hc[r], r0);                // :82 out_color
            r1 = fmaf(W[L::W_OC + H + j], hc[r], r1);
            r2 = fmaf(W[L::W_OC + 2 * H + j], hc[r], r2);
        }
        ra += wv::swap_half(ra); r0 += wv::swap_half(r0); r1 += wv::swap_half(r1); r2 += wv::swap_half(r2);
        ra += W[L::B_A]; r0 += W[L::B_OC]; r1 += W[L::B_OC + 1]; r2 += W[L::B_OC + 2];
        if (valid && hi == 0) {
            float* row = cb + pt * 8;
            row[6] = a.z[obj * a.z_so + ray * a.z_sr + smp * a.z_ss];
            row[0] = sigmoidf_acc(ra * 10.0f);                   // :77 raw*10 ; render_rays.py:6 sigmoid
            row[1] = sigmoidf_acc(r0);                           // :83 sigmoid(raw_color)
            row[2] = sigmoidf_acc(r1);
            row[3] = sigmoidf_acc(r2);
        }
    }
    VK_MARK(4);
    __syncthreads();
    VK_MARK(5);

    {
        const StepArgs& al = wv::kernarg_late(a);      // batch pointers, strides and loss weights: fetched here
        composite_phase<BWD>(al, cb, lds + L::LOSS, obj, ray0, nrays, wave, lane, tid,
                             load_ray_meta(al, obj, ray0 + min(4 * wave + (lane >> 4), nrays - 1)));
    }
    __syncthreads();
    VK_MARK(6);
    if (BWD) {
    // ---- backward ----
    float d_raw = 0.0f, d_c0 = 0.0f, d_c1 = 0.0f, d_c2 = 0.0f;
    {
        const float* row = cb + pt * 8;          // pt < kMaxPts always; padding rows hold zeros
        d_raw = row[0]; d_c0 = row[1]; d_c1 = row[2]; d_c2 = row[3];
    }
    float xF[16], dF[16];
    float dproj[kDirs];
#pragma unroll
    for (int d = 0; d < kDirs; ++d) dproj[d] = 0.0f;

    // heads: out_alpha / out_color weight + bias gradients (lane = hidden feature)
    float gb_c = 0.0f;   // bias gradients of the five hidden layers (lane = feature), stored after the last unit
    float w[16];     // weight-column operands of the next d-prop chain (fetched during the previous dW chain)
    float dcp[16];   // d hc (pre-activation), P-form
    float d4[16];    // d h4 (pre-activation), P-form
    {
        float h4F[16];
        toF_put(scrX, h4, p31, hi);
        toF_put(scrD, hc, p31, hi);
        toF_get(h4F, scrX, p31, hi);
        toF_get(xF, scrD, p31, hi);              // hcF
        float ga = 0.0f, g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, sa = 0.0f, s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* row = cbw + (r + 16 * hi) * 8;
            const float da = row[0], q0 = row[1], q1 = row[2], q2 = row[3];
            ga = fmaf(da, h4F[r], ga);
            g0 = fmaf(q0, xF[r], g0); g1 = fmaf(q1, xF[r], g1); g2 = fmaf(q2, xF[r], g2);
            sa += da; s0 += q0; s1 += q1; s2 += q2;
        }
        ga += wv::swap_half(ga); g0 += wv::swap_half(g0); g1 += wv::swap_half(g1); g2 += wv::swap_half(g2);
        sa += wv::swap_half(sa); s0 += wv::swap_half(s0); s1 += wv::swap_half(s1); s2 += wv::swap_half(s2);
        if (hi == 0) {
            Gv[L::W_A + p31] += ga;
            Gv[L::W_OC + p31] += g0;
            Gv[L::W_OC + H + p31] += g1;
            Gv[L::W_OC + 2 * H + p31] += g2;
            if (p31 == 0) {
                Gv[L::B_A] += sa;
                Gv[L::B_OC + 0] += s0;
                Gv[L::B_OC + 1] += s1;
                Gv[L::B_OC + 2] += s2;
            }
        }
        // d hc = W_oc^T d rawc, through the ReLU
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = phi(r, hi);
            const float v = W[L::W_OC + j] * d_c0 + W[L::W_OC + H + j] * d_c1 + W[L::W_OC + 2 * H + j] * d_c2;
            dcp[r] = wv::opaque(hc[r]) > 0.0f ? v : 0.0f;
        }
        // ---- 13 units, one 32x32 weight-gradient block each.  A unit = d-prop chain (16 matrix instructions carrying
        // the unit's transposes and the staging of the previous block), workgroup barrier, reduction + store of the
        // previous block, the VALU work that needs the chain's result, then the dW chain (16 matrix instructions
        // carrying the weight-column loads of the next unit). ----
        f32x16 acc2;
        // unit 0: color_linear, x = h4 (already transposed for the heads)
        bwd_w<L::LD_C>(w, W + L::W_C + 4 * hi * L::LD_C + p31);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = W[L::W_A + phi(r, hi)] * d_raw;       // d h4 = W_a d raw + W_c[:, :H]^T d hc
        p_chain<true, false, false>(acc2, w, dcp, scrD, dcp, dF, scrX, dcp, xF, stg0, acc, wave, p31, hi);
#pragma unroll
        for (int r = 0; r < 16; ++r) d4[r] = wv::opaque(h4[r]) > 0.0f ? acc2[r] : 0.0f;
        gb_c = db_sum(dF);
        zero_acc(acc);
        dw_chain<L::LD_C>(acc, dF, h4F, w, W + L::W_C + 4 * hi * L::LD_C + H + p31);
        // unit 1: x = e2 block 0
        zero_acc(acc2);
        p_chain<false, true, true>(acc2, w, dcp, scrD, dcp, dF, scrX, e2a, xF, stg0, acc, wave, p31, hi);   // d e2 (block 0)
        __syncthreads();
        pe_block_bwd<16>(dproj, acc2, c2a, kEmb1, kEmb2, 0, hi);
        zero_acc(acc);
        dw_chain_fin<L::LD_C, MULTI, H + kEmb2>(acc, dF, xF, w, W + L::W_C + 4 * hi * L::LD_C + H + min(32 + p31, 46),
                     qacc[0], stg0, out + F::W_C, 0, 32, wave, p31, hi);
        // unit 2: x = e2 block 1
        zero_acc(acc2);
        p_chain<false, true, true>(acc2, w, dcp, scrD, dcp, dF, scrX, e2b, xF, stg1, acc, wave, p31, hi);
        __syncthreads();
        pe_block_bwd<6>(dproj, acc2, c2b, kEmb1, kEmb2, 1, hi);
        zero_acc(acc);
        dw_chain_fin<L::LD_M, MULTI, H + kEmb2>(acc, dF, xF, w, W + L::W_M2 + 4 * hi * L::LD_M + p31,
                     qacc[1], stg1, out + F::W_C, H, 32, wave, p31, hi);
    }
    VK_MARK(7);
    VK_MARK(8);
    float e1aF[16];
    {
        float d3[16], dn[16], dF3[16];
        f32x16 acc2, de;
        const float* wc = W + L::W_CAT + 4 * hi * L::LD_CAT;
        const float* wi = W + L::W_IN + 4 * hi * L::LD_IN;
        // unit 3: mid2, delta = d4, x = h3
        zero_acc(acc2);
        p_chain<true, true, true>(acc2, w, d4, scrD, d4, dF, scrX, h3, xF, stg0, acc, wave, p31, hi);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) d3[r] = wv::opaque(h3[r]) > 0.0f ? acc2[r] : 0.0f;
        const float gb_m2 = db_sum(dF);
        zero_acc(acc);
        dw_chain_fin<L::LD_CAT, MULTI, H + kEmb2>(acc, dF, xF, w, wc + p31,
                     qacc[2], stg0, out + F::W_C, H + 32, kEmb2 - 32, wave, p31, hi);
        VK_MARK(9);
        // unit 4: cat_layer, delta = d3 (its transpose dF3 is kept for the three e1 blocks), x = h2
        zero_acc(acc2);
        p_chain<true, true, true>(acc2, w, d3, scrD, d3, dF3, scrX, h2, xF, stg1, acc, wave, p31, hi);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) dn[r] = wv::opaque(h2[r]) > 0.0f ? acc2[r] : 0.0f;          // d2
        const float gb_cat = db_sum(dF3);
        zero_acc(acc);
        dw_chain_fin<L::LD_M, MULTI, H>(acc, dF3, xF, w, W + L::W_M1 + 4 * hi * L::LD_M + p31,
                     qacc[3], stg1, out + F::W_M2, 0, 32, wave, p31, hi);
        VK_MARK(10);
        // unit 5: mid1, delta = d2, x = h1
        zero_acc(acc2);
        p_chain<true, true, true>(acc2, w, dn, scrD, dn, dF, scrX, h1, xF, stg0, acc, wave, p31, hi);
        __syncthreads();
        const float gb_m1 = db_sum(dF);
        zero_acc(acc);
        dw_chain_fin<L::LD_CAT, MULTI, H + kEmb1>(acc, dF, xF, w, wc + H + p31,
                     qacc[4], stg0, out + F::W_CAT, 0, 32, wave, p31, hi);
#pragma unroll
        for (int r = 0; r < 16; ++r) dn[r] = wv::opaque(h1[r]) > 0.0f ? acc2[r] : 0.0f;          // d1 (d2 is dead: its chain is done)
        VK_MARK(11);
        // units 6..11: the three blocks of e1 feed both cat_layer (delta d3) and in_layer (delta d1): one transpose of
        // the block, two d-prop chains into the same accumulator, two weight-gradient blocks
        // unit 6: cat_layer x e1 block 0
        zero_acc(de);
        p_chain<false, true, true>(de, w, d3, scrD, d3, dF, scrX, e1a, e1aF, stg1, acc, wave, p31, hi);
        __syncthreads();
        zero_acc(acc);
        dw_chain_fin<L::LD_IN, MULTI, H>(acc, dF3, e1aF, w, wi + p31,
                     qacc[8], stg1, out + F::W_M1, 0, 32, wave, p31, hi);
        // unit 7: in_layer x e1 block 0 (transposes delta d1 once)
        p_chain<true, false, true>(de, w, dn, scrD, dn, dF, scrX, dn, xF, stg0, acc, wave, p31, hi);
        __syncthreads();
        const float gb_in = db_sum(dF);
        pe_block_bwd<16>(dproj, de, c1a, 0, kEmb1, 0, hi);
        zero_acc(acc);
        dw_chain_fin<L::LD_CAT, MULTI, H + kEmb1>(acc, dF, e1aF, w, wc + H + 32 + p31,
                     qacc[5], stg0, out + F::W_CAT, H, 32, wave, p31, hi);
        // unit 8: cat_layer x e1 block 1
        zero_acc(de);
        p_chain<false, true, true>(de, w, d3, scrD, d3, dF, scrX, e1b, xF, stg1, acc, wave, p31, hi);
        __syncthreads();
        zero_acc(acc);
        dw_chain_fin<L::LD_IN, MULTI, kEmb1>(acc, dF3, xF, w, wi + 32 + p31,
                     qacc[9], stg1, out + F::W_IN, 0, 32, wave, p31, hi);
        // unit 9: in_layer x e1 block 1
        p_chain<false, false, true>(de, w, dn, scrD, dn, dF, scrX, dn, xF, stg0, acc, wave, p31, hi);
        __syncthreads();
        pe_block_bwd<16>(dproj, de, c1b, 0, kEmb1, 1, hi);
        zero_acc(acc);
        dw_chain_fin<L::LD_CAT, MULTI, H + kEmb1>(acc, dF, xF, w, wc + H + min(64 + p31, 88),
                     qacc[6], stg0, out + F::W_CAT, H + 32, 32, wave, p31, hi);
        // unit 10: cat_layer x e1 block 2
        zero_acc(de);
        p_chain<false, true, true>(de, w, d3, scrD, d3, dF, scrX, e1c, xF, stg1, acc, wave, p31, hi);
        __syncthreads();
        zero_acc(acc);
        dw_chain_fin<L::LD_IN, MULTI, kEmb1>(acc, dF3, xF, w, wi + min(64 + p31, 88),
                     qacc[10], stg1, out + F::W_IN, 32, 32, wave, p31, hi);
        // unit 11: in_layer x e1 block 2
        p_chain<false, false, true>(de, w, dn, scrD, dn, dF, scrX, dn, xF, stg0, acc, wave, p31, hi);
        __syncthreads();
        pe_block_bwd<12>(dproj, de, c1c, 0, kEmb1, 2, hi);
        zero_acc(acc);
        dw_chain_fin<0, MULTI, H + kEmb1>(acc, dF, xF, w, wi,
                     qacc[7], stg0, out + F::W_CAT, H + 64, kEmb1 - 64, wave, p31, hi);
        if (hi == 0) {
            Gv[L::B_C + p31] += gb_c;
            Gv[L::B_M2 + p31] += gb_m2;
            Gv[L::B_CAT + p31] += gb_cat;
            Gv[L::B_M1 + p31] += gb_m1;
            Gv[L::B_IN + p31] += gb_in;
        }
    }
    VK_MARK(12);
    // unit 12: B_layer.weight gradient: dB[d][j] = sum_points dproj[d] * t[j]  (t = encoding columns 0..2)
    {
        float dpP[16];
#pragma unroll
        for (int d = 0; d < kDirs; ++d) dproj[d] += wv::swap_half(dproj[d]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f0 = phi(r, 0), f1 = phi(r, 1);
            const float v0 = f0 < kDirs ? dproj[f0 < kDirs ? f0 : 0] : 0.0f;
            const float v1 = f1 < kDirs ? dproj[f1 < kDirs ? f1 : 0] : 0.0f;
            dpP[r] = hi ? v1 : v0;
        }
        toF_put(scrD, dpP, p31, hi);
        toF_get(dF, scrD, p31, hi);
        stage_put(stg1, acc, wave, p31, hi);
        __syncthreads();
        zero_acc(acc);                              // the dB chain carries the finish of block 11 like every other unit
        dw_chain_fin<0, MULTI, kEmb1>(acc, dF, e1aF, w, nullptr, qacc[11], stg1, out + F::W_IN, 64, kEmb1 - 64, wave, p31, hi);
        stage_put(stg0, acc, wave, p31, hi);
        __syncthreads();
        if (MULTI) {
            stage_get(qacc[12], stg0, wave, p31, hi);           // rows = direction d (21 valid), cols 0..2 = xyz
        } else {
            float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            stage_get(q, stg0, wave, p31, hi);
            if (p31 < 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = 8 * wave + 4 * hi + i;
                    if (d < kDirs) out[F::PE_B + 3 * d + p31] = q[i];
                }
            }
        }
    }
    VK_MARK(13);
    }   // BWD
    if (!MULTI) break;      // one pass per workgroup: no back edge, so nothing of the body is hoisted to the entry
    }   // pass loop
    __syncthreads();
    VK_MARK(14);
    if (tid == 0) {
        float* pl = a.part_loss + (obj * a.NW + wgo) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            pl[k] = (lds[L::LOSS + k] + lds[L::LOSS + 4 + k]) + (lds[L::LOSS + 8 + k] + lds[L::LOSS + 12 + k]);
        pl[3] = 0.0f;
    }
    if (!BWD) return;

    // ---- write this workgroup's partial gradients in the natural flat order ----
    if (MULTI) {
        store_quarter<H + kEmb2>(out + F::W_C, qacc[0], 0, 32, wave, p31, hi);
        store_quarter<H + kEmb2>(out + F::W_C, qacc[1], H, 32, wave, p31, hi);
        store_quarter<H + kEmb2>(out + F::W_C, qacc[2], H + 32, kEmb2 - 32, wave, p31, hi);
        store_quarter<H>(out + F::W_M2, qacc[3], 0, 32, wave, p31, hi);
        store_quarter<H + kEmb1>(out + F::W_CAT, qacc[4], 0, 32, wave, p31, hi);
        store_quarter<H + kEmb1>(out + F::W_CAT, qacc[5], H, 32, wave, p31, hi);
        store_quarter<H + kEmb1>(out + F::W_CAT, qacc[6], H + 32, 32, wave, p31, hi);
        store_quarter<H + kEmb1>(out + F::W_CAT, qacc[7], H + 64, kEmb1 - 64, wave, p31, hi);
        store_quarter<H>(out + F::W_M1, qacc[8], 0, 32, wave, p31, hi);
        store_quarter<kEmb1>(out + F::W_IN, qacc[9], 0, 32, wave, p31, hi);
        store_quarter<kEmb1>(out + F::W_IN, qacc[10], 32, 32, wave, p31, hi);
        store_quarter<kEmb1>(out + F::W_IN, qacc[11], 64, kEmb1 - 64, wave, p31, hi);
        if (p31 < 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = 8 * wave + 4 * hi + i;
                if (d < kDirs) out[F::PE_B + 3 * d + p31] = qacc[12][i];
            }
        }
    }
    // small vectors: sum of the four waves' private accumulators
    for (int sv = tid; sv < L::SMALL_N; sv += kWG) {
        const float* v = lds + L::VEC + sv;
        const float g = (v[0] + v[L::SMALL_N]) + (v[2 * L::SMALL_N] + v[3 * L::SMALL_N]);
        const int i = L::SMALL0 + sv;         // position in the LDS image map
        int o = -1;
        if (i < L::B_M1) o = F::B_IN + (i - L::B_IN);
        else if (i < L::B_CAT) o = F::B_M1 + (i - L::B_M1);
        else if (i < L::B_M2) o = F::B_CAT + (i - L::B_CAT);
        else if (i < L::B_C) o = F::B_M2 + (i - L::B_M2);
        else if (i < L::W_A) o = F::B_C + (i - L::B_C);
        else if (i < L::W_OC) o = F::W_A + (i - L::W_A);
        else if (i < L::B_A) o = F::W_OC + (i - L::W_OC);
        else if (i == L::B_A) o = F::B_A;
        else if (i >= L::B_OC && i < L::B_OC + 3) o = F::B_OC + (i - L::B_OC);
        if (o >= 0) out[o] = g;
    }
    VK_MARK(15);
#undef VK_MARK
}

template <bool BWD, bool MULTI, bool STAMPS = false>      // STAMPS: the phase-clock instantiation (vmapstep_profile_phases)
__global__ __launch_bounds__(kWG, 1) void step_main_h32(const StepArgs a) {
    step_main_body<BWD, MULTI, STAMPS>(a);
}
// finalize_quad for hidden 32 in the common case (AdamW on, no gradient output): the image position of a flat parameter
// comes from the table step_prep wrote (img_tab) and the parameter address from compile-time offsets - or from one slab
// base - instead of fifteen runtime comparisons and the integer divisions of gen_image_index per element.  Same sums in
// the same order, same adamw_elem: bit-identical to finalize_quad (measured: step_finalize spent its 6.4 us issuing
// ~1000 instructions per thread on 3.5 waves per SIMD, not waiting for memory).
template <bool SLAB>
__device__ __forceinline__ void finalize_quad_h32(const FinalizeArgs& f, const FinalizeHot& a, int obj, int q) {
    float ss, bc;
    adam_step_consts(f, a, ss, bc);
    using L = Lds32;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const long long s = (long long)obj * a.PP + 4 * q;
    const wv::f32x4* pg = reinterpret_cast<const wv::f32x4*>(a.part_grad + (long long)obj * a.NW * a.PP + 4 * q);
    wv::f32x4 m4 = *reinterpret_cast<const wv::f32x4*>(a.m + s);
    wv::f32x4 v4 = *reinterpret_cast<const wv::f32x4*>(a.v + s);
    const i32x4 img = *reinterpret_cast<const i32x4*>(a.img_tab + 4 * q);
    float* pp[4]; float pv[4];
    int tq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = min(4 * q + e, Flat32::P - 1);          // the one padding lane (index P) re-reads the last parameter
        if (SLAB) {
            pp[e] = a.slab + obj * a.slab_stride + i;
            tq[e] = 0;
        } else {
            int o;
            flat32_tensor_of(i, tq[e], o);
            pp[e] = f.param[tq[e]].p + obj * f.param[tq[e]].stride + o;
        }
    }
    const bool pvec = 4 * q + 3 < Flat32::P && tq[0] == tq[3];      // the quad lies inside ONE tensor: one 16-byte access, in and out
    if (pvec) {
        const wv::f32x4 p4 = *reinterpret_cast<const wv::f32x4u*>(pp[0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = p4[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = *pp[e];
    }
    wv::f32x4 g = {0.0f, 0.0f, 0.0f, 0.0f};
    {
        const long long qs = a.PP / 4;
        int u0 = 0;
        for (; u0 + 8 <= a.NW; u0 += 8) {
            wv::f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = pg[(u0 + u) * qs];
#pragma unroll
            for (int u = 0; u < 8; ++u) g += t[u];
        }
        wv::f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = u0 + u < a.NW ? pg[(u0 + u) * qs] : wv::f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u0 + u < a.NW) g += t[u];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (4 * q + e < Flat32::P) {
            float p = pv[e], m = m4[e], v = v4[e];
            adamw_elem(a, ss, bc, g[e], p, m, v);
            if (!pvec) *pp[e] = p;
            pv[e] = p; m4[e] = m; v4[e] = v;
            a.wimg[(long long)obj * L::IMGP + img[e]] = a.weights_bf16 ? round_bf16(p) : p;
        }
    }
    if (pvec) *reinterpret_cast<wv::f32x4u*>(pp[0]) = wv::f32x4{pv[0], pv[1], pv[2], pv[3]};
    *reinterpret_cast<wv::f32x4*>(a.m + s) = m4;
    *reinterpret_cast<wv::f32x4*>(a.v + s) = v4;
}

// step_finalize for hidden 32, AdamW on, gradients not wanted by the caller (every training step but the last of a call
// that asks for them): same grid, same block -> object map, same loss workgroup
template <int = 0>
__global__ __launch_bounds__(kWG) void step_finalize_h32(const FinalizeArgs a, const FinalizeHot hh) {
    const int quads = a.PP / 4;
    const int blocks_per_obj = (quads + kWG - 1) / kWG;
    if (blockIdx.x == gridDim.x - 1) {
        finalize_loss(a);
        return;
    }
    int obj, part;
    if (a.xcd_affine) {
        const int slot = blockIdx.x >> 3;
        const int og = slot / blocks_per_obj;
        obj = og * 8 + (blockIdx.x & 7);
        part = slot - og * blocks_per_obj;
    } else {
        obj = blockIdx.x / blocks_per_obj;
        part = blockIdx.x - obj * blocks_per_obj;
    }
    const int q4 = part * kWG + threadIdx.x;
    if (obj < a.n_obj && q4 < quads && 4 * q4 < a.P) {
        if (hh.slab) finalize_quad_h32<true>(a, hh, obj, q4);
        else finalize_quad_h32<false>(a, hh, obj, q4);
    }
}

template <int = 0>
__global__ __launch_bounds__(kWG) void step_finalize(const FinalizeArgs a) {
    const GenLayout GL = gen_layout(a.hidden);
    // one thread per 4 consecutive flat parameters
    const int quads = a.PP / 4;
    const int blocks_per_obj = (quads + kWG - 1) / kWG;
    int obj, part;
    if (a.xcd_affine) {
        // same object -> XCD placement as step_main (block b runs on XCD b % 8): the partials this block sums were written
        // through this XCD's L2 a few microseconds ago, and the image slice it writes is read there by the next step
        const int slot = blockIdx.x >> 3;
        const int og = slot / blocks_per_obj;
        obj = og * 8 + (blockIdx.x & 7);
        part = slot - og * blocks_per_obj;
    } else {
        obj = blockIdx.x / blocks_per_obj;
        part = blockIdx.x - obj * blocks_per_obj;
    }
    const int q4 = part * kWG + threadIdx.x;
    if (a.have_grad && blockIdx.x != gridDim.x - 1 && obj < a.n_obj && q4 < quads && 4 * q4 < a.P)
        finalize_quad<false>(a, GL, obj, q4);
    if (blockIdx.x == gridDim.x - 1)       // a workgroup of its own (the launch has one more than the parameter blocks)
        finalize_loss(a);
}

}  // namespace vk
