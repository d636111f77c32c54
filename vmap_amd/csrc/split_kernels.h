// split_kernels.h - the fused per-object training step for hidden = 32 on the bf16 matrix pipe (gfx950 / CDNA4).
//
// Same path, phases, reductions and ABI as step_main_h32 (step_kernels.h; train.py:293-326), but every contraction runs
// on v_mfma_f32_32x32x16_bf16 with SPLIT operands instead of the exact-fp32 matrix instruction:
//
//   a float32 x is carried as up to three bfloat16 planes  x = hi + mid + lo  (hi = rne(x), mid = rne(x - hi),
//   lo = rne(x - hi - mid): 24 significant bits, i.e. exact), a product of two such values as the sum of bf16 x bf16
//   products (each exact in float32), accumulated in float32 by the matrix instruction:
//     forward   W.x  = hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi     (6 products: float32-equivalent; the
//                       dropped terms are <= 2^-24 relative) - needed: a ReLU kink or a saturated occupancy amplifies
//                       anything coarser beyond the 1e-4 parity bar (measured on the reference fixtures)
//     backward  d-prop and weight gradients = hi.hi + hi.mid + mid.hi          (3 products, 2^-16 per product:
//                       gradients stay within 1e-5 of the reference fixtures, the same as the exact-fp32 kernel)
//   Why: the exact-fp32 matrix instruction runs at the VALU rate ON the VALU lanes (64 clocks per 32x32x2, nothing else
//   issues meanwhile); the bf16 form does 8x the K per instruction in 32 clocks on the matrix pipe proper and the wave's
//   VALU work issues next to it (profiles/r02a_bf16_probe.jsonl).  One 32-point tile: 574 fp32 instructions = 36.7 k
//   clocks  ->  288 bf16 instructions = 9.2 k clocks, mostly hidden behind the tile's VALU work.
//
// What else differs from step_main_h32:
//   * parameter image = three bf16 planes of the five weight matrices, rows in the K-order the matrix instruction
//     consumes (one ds_read_b128 per plane and 16-deep step), + the small float32 vectors; written by step_prep_s32,
//     kept current by step_finalize_s32 (three 2-byte stores per parameter);
//   * d-prop needs W^T as the A operand: read from the SAME row-major planes with ds_read_b64_tr_b16 (4x4 transpose in
//     the LDS read path) - no transposed copy; the P->F transposes of the weight-gradient operands use it too (packed
//     P-form quads stored with ds_write_b64, F-form read back transposed): 8 + 8 DS instructions per operand instead of 20;
//   * the encoding's K order is free (the weight columns are permuted to match when the image is built), so it is laid out
//     by OWNER LANE: the lane pair (p, hi) of a point splits the 21 directions 11 / 10 and each lane evaluates ONE sincos
//     per direction and five double-angle steps (2^f scaling is exact in float32: sin(2^f a) from (sin a, cos a) is the
//     reference's embedding.py:85-88 value up to the recurrence's rounding) - 11 sincos per lane instead of 62;
//   * the biases of in / cat / colour layers ride in a weight column that meets a constant-1 slot of the encoding:
//     no bias preload in the forward, and the bias gradient is a column of the weight-gradient block.
#pragma once
#ifndef VS_ABL                   /* measurement builds only: see below */
#define VS_ABL 0
#endif
#include "step_kernels.h"

// Measurement builds only (tests/tools/build_variant.py ... -DVS_ABL=<mask>; results WRONG on purpose - the product never defines it):
// bit 7: step_finalize_s32 does not rewrite the parameter image; bit 4: only the partial-gradient global stores are skipped; bit 5: the finishing (staged reads, sums, stores) is skipped, the staging writes stay;
// bit 3: only the staging / finishing of the blocks is skipped (the products stay);
// bit 2: only the P->F transposes of the weight-gradient operands are skipped (tile_put / tile_get);
// bit 0: the backward without its weight-gradient products (no matrix instructions in mm_dw_il, no P->F transposes, no staging /
// finishing of blocks) = the d-prop critical chain alone; bit 1: without the backward's workgroup barriers.
// The front half (round 6d): bit 8: no LDS-DMA of the parameter image (LDS holds whatever it held); bit 9: the sample point is a function of the lane
// (no loads of pcs / z); bit 10: the per-ray ground truth and the per-object normalisers are constants (no loads); bit 11: no compositing
// (barriers kept); bit 12: the encoding's sincos + octave recurrence replaced by a copy; inside the compositing (step_kernels.h):
// bit 13: no sequential scans (T, D, V, suffix); bit 14: no square root / divisions; bit 15: no loss sums; bit 16: no workgroup barriers
// around the compositing (what a wave-local compositing - whole rays per wave - could save at most).
#ifndef VS_ABL
#define VS_ABL 0
#endif
#if VS_ABL & 2
#define VS_BWD_BARRIER() do { } while (0)
#else
#define VS_BWD_BARRIER() __syncthreads()
#endif

namespace vk {

using wv::u32x2;
using wv::u32x4;

// ---------------------------------------------------------------------------------------------------------
// Image layout (bytes; the global image of an object and its LDS copy are identical).
// Plane = [W_in | W_m1 | W_cat | W_m2 | W_c], each [32 rows][steps][hi][t = 0..7] bf16: element (row j, step s, hi, t)
// is W[j][k(s, hi, t)], k(s, hi, t) = 16 s + (t & 3) + 4 hi + 8 (t >> 2) for hidden-layer inputs (the P-form register map
// of wave_ops.h) and the slot tables below for encoding inputs.  Row pitch = 16 bytes x odd: the "lane = row"
// ds_read_b128 of the forward is bank-conflict free.
// ---------------------------------------------------------------------------------------------------------
struct Img32s {
    static constexpr int PIT_IN = 208, PIT_M = 80, PIT_CAT = 272, PIT_C = 176;      // 6 / 2 / 8 / 5 steps of 32 bytes (+16)
    static constexpr int O_IN = 0, O_M1 = O_IN + 32 * PIT_IN, O_CAT = O_M1 + 32 * PIT_M, O_M2 = O_CAT + 32 * PIT_CAT,
                         O_C = O_M2 + 32 * PIT_M, PLANE = O_C + 32 * PIT_C;          // 26112 bytes per plane
    static constexpr int P_HI = 0, P_MID = PLANE, SMALL = 2 * PLANE, SMALL_BYTES = 2048, P_LO = SMALL + SMALL_BYTES,
                         END = P_LO + PLANE, BYTES = (END + 4095) / 4096 * 4096;     // 81920: 20 rounds of 4 x 1 KiB LDS-DMA
    static constexpr int ROUNDS = BYTES / 4096;
    static constexpr int ELEMS = PLANE / 2;                                          // bf16 elements per plane
    // small float32 vectors (index in floats from SMALL)
    static constexpr int B_M1 = 0, B_M2 = 32, W_A = 64, W_OC = 96, B_A = 192, B_OC = 196, PE_B = 200, SMALL_N = 208;
    // per-wave gradient accumulators (LDS region VEC): the SMALL_N slots above + the B_layer.weight partials [half][3 i + j] (round 6)
    static constexpr int G_PEB = SMALL_N, VEC_N = SMALL_N + 66 + 6;
    // ---- LDS map of step_main_s32 (bytes) ----
    static constexpr int TILE = 32 * Lds32::TP * 4;          // 4608: one float32 32x32 exchange tile = two bf16 planes [32][72 B]
    static constexpr int TPL = TILE / 2;                     // one bf16 plane of a transpose tile
    static constexpr int TPIT = 72;                          // bytes per point row of a bf16 transpose plane (8 x odd)
    // Round 6: the weight-gradient operands are transposed on the matrix pipe (toF_mm), so the per-wave tiles only carry the two
    // float32 head transposes (written at the end of the forward, read at the start of the backward) - the staging tiles of the
    // backward's cross-wave sums ALIAS them (first staged in unit 1, behind unit 0's barrier) instead of overlaying the lo planes:
    // the whole image stays valid to the end of the pass (no lo-plane re-copy in multi-pass launches), 36 KB less LDS.
    static constexpr int SCR = BYTES;                        // per-wave float32 transpose tiles: kWaves x 2
    static constexpr int STG = SCR;                          // staging tiles: the same bytes, later in the pass
    static constexpr int STG_BYTES = 2 * kWaves * TILE, kWavesTiles = kWaves * 2 * TILE;
    static constexpr int VEC = SCR + kWaves * 2 * TILE;      // per-wave small-vector gradient accumulators
    static constexpr int CB = VEC + kWaves * VEC_N * 4;
    static constexpr int LOSS = CB + kMaxPts * 8 * 4;
    static constexpr int LDS_BYTES = LOSS + kWaves * 4 * 4;
};
static_assert(Img32s::PLANE == 26112 && Img32s::BYTES == 81920, "image size");
static_assert(Img32s::STG >= Img32s::BYTES && Img32s::STG_BYTES == Img32s::kWavesTiles, "staging tiles = the transpose tiles' bytes, behind the image");
static_assert(Img32s::LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(Img32s::PLANE % 16 == 0 && Img32s::SMALL % 16 == 0 && Img32s::P_LO % 16 == 0, "16-byte planes");

constexpr int kSlotOne = -2, kSlotPad = -1;
// Encoding slots, by OWNER LANE.  A lane (point p, half hi) holds 48 registers of the 87-wide first group (R = 0..47,
// consumed 8 per 16-deep step) and 24 of the 42-wide second group.  Returns the index inside the group (embedding.py:85-89
// order: xyz, then 3 + 21 f + d), kSlotOne for the constant-1 slot that meets the bias column, kSlotPad for zero padding.
__host__ __device__ constexpr int e1_slot(int R, int hi) {
    if (hi == 0) return R < 44 ? 3 + 21 * (R & 3) + (R >> 2) : (R < 47 ? R - 44 : kSlotOne);     // directions 0..10, xyz, one
    return R < 40 ? 3 + 21 * (R & 3) + 11 + (R >> 2) : kSlotPad;                                   // directions 11..20
}
__host__ __device__ constexpr int e2_slot(int R, int hi) {
    if (hi == 0) return R < 22 ? 21 * (R & 1) + (R >> 1) : (R == 22 ? kSlotOne : kSlotPad);        // octaves 4, 5 of directions 0..10
    return R < 20 ? 21 * (R & 1) + 11 + (R >> 1) : kSlotPad;
}
__host__ __device__ constexpr int hidden_k(int s, int hi, int t) { return 16 * s + (t & 3) + 4 * hi + 8 * (t >> 2); }

// Which parameter sits at bf16 element x of a plane: tensor t (0..13 field tensors in nn.Module.parameters() order) and
// offset o inside it; false = zero padding.
__host__ __device__ inline bool split_image_source(int x, int& t, int& o) {
    using I = Img32s;
    const int byte = 2 * x;
    int base, pit, steps, kind;          // kind: 0 in, 1 m1, 2 cat, 3 m2, 4 c
    if (byte < I::O_M1) { base = I::O_IN; pit = I::PIT_IN; steps = 6; kind = 0; }
    else if (byte < I::O_CAT) { base = I::O_M1; pit = I::PIT_M; steps = 2; kind = 1; }
    else if (byte < I::O_M2) { base = I::O_CAT; pit = I::PIT_CAT; steps = 8; kind = 2; }
    else if (byte < I::O_C) { base = I::O_M2; pit = I::PIT_M; steps = 2; kind = 3; }
    else { base = I::O_C; pit = I::PIT_C; steps = 5; kind = 4; }
    const int j = (byte - base) / pit, rb = (byte - base) - j * pit;
    const int s = rb >> 5, hi = (rb >> 4) & 1, tt = (rb >> 1) & 7;
    if (s >= steps) return false;
    const int k = hidden_k(s, hi, tt);
    switch (kind) {
        case 0: {
            const int c = e1_slot(8 * s + tt, hi);
            if (c == kSlotPad) return false;
            if (c == kSlotOne) { t = 1; o = j; } else { t = 0; o = j * kEmb1 + c; }
            return true;
        }
        case 1: t = 2; o = j * 32 + k; return true;
        case 2: {
            if (s < 2) { t = 4; o = j * (32 + kEmb1) + k; return true; }
            const int c = e1_slot(8 * (s - 2) + tt, hi);
            if (c == kSlotPad) return false;
            if (c == kSlotOne) { t = 5; o = j; } else { t = 4; o = j * (32 + kEmb1) + 32 + c; }
            return true;
        }
        case 3: t = 6; o = j * 32 + k; return true;
        default: {
            if (s < 2) { t = 10; o = j * (32 + kEmb2) + k; return true; }
            const int c = e2_slot(8 * (s - 2) + tt, hi);
            if (c == kSlotPad) return false;
            if (c == kSlotOne) { t = 11; o = j; } else { t = 10; o = j * (32 + kEmb2) + 32 + c; }
            return true;
        }
    }
}
// Which parameter sits at float i of the small-vector region (false = padding)
__host__ __device__ inline bool split_small_source(int i, int& t, int& o) {
    using I = Img32s;
    if (i < I::B_M2) { t = 3; o = i - I::B_M1; return true; }
    if (i < I::W_A) { t = 7; o = i - I::B_M2; return true; }
    if (i < I::W_OC) { t = 8; o = i - I::W_A; return true; }
    if (i < I::B_A) { t = 12; o = i - I::W_OC; return true; }
    if (i < I::B_OC) { t = 9; o = i - I::B_A; return o < 1; }
    if (i < I::PE_B) { t = 13; o = i - I::B_OC; return o < 3; }
    t = 14; o = i - I::PE_B;
    return o < 63;
}

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

// one float32 -> its three bfloat16 planes (as 16-bit patterns)
__device__ __forceinline__ void split3_scalar(float p, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned ph = wv::pack_bf16(p, 0.0f) & 0xFFFFu;
    const float r1 = p - bf_lo(ph);
    const unsigned pm = wv::pack_bf16(r1, 0.0f) & 0xFFFFu;
    const float r2 = r1 - bf_lo(pm);
    h = ph; m = pm; l = wv::pack_bf16(r2, 0.0f) & 0xFFFFu;
}
// parameter value p -> its place in the image.  loc >= 0: bf16 element of the planes; loc < 0: float (loc & 0x7fffffff)
// of the small-vector region.  bf16 weights (BASELINE configs[3]/[4]): the image holds rne_bf16(p) - one plane.
__device__ __forceinline__ void split_image_store(char* img, int loc, float p, int weights_bf16) {
    using I = Img32s;
    if (loc < 0) {
        reinterpret_cast<float*>(img + I::SMALL)[loc & 0x7FFFFFFF] = weights_bf16 ? round_bf16(p) : p;
        return;
    }
    unsigned h, m, l;
    split3_scalar(p, h, m, l);
    if (weights_bf16) { m = 0u; l = 0u; }
    reinterpret_cast<unsigned short*>(img + I::P_HI)[loc] = (unsigned short)h;
    reinterpret_cast<unsigned short*>(img + I::P_MID)[loc] = (unsigned short)m;
    reinterpret_cast<unsigned short*>(img + I::P_LO)[loc] = (unsigned short)l;
}

// ---------------------------------------------------------------------------------------------------------
// step_prep_s32: blocks [0, prep_steps) mask statistics (as step_prep); then 13 blocks per object that build the
// object's split image (one thread per 4 consecutive bf16 elements = 8 bytes of each plane; the last 64 threads of an
// object's last block fill the small-vector region) and, for object 0, the flat parameter -> image location table.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSplitPackBlocks = (Img32s::ELEMS / 4 + kWG - 1) / kWG;       // 13
static_assert(kSplitPackBlocks * kWG - Img32s::ELEMS / 4 >= 64, "room for the small-vector threads");

template <int = 0>
__global__ __launch_bounds__(kWG) void step_prep_s32(const StepArgs a) {
    using I = Img32s;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < a.prep_steps) {
        prep_stats(a, blockIdx.x, Flat32::P, (Flat32::P + 63) / 64 * 64);
        return;
    }
    const int b = blockIdx.x - a.prep_steps;
    const int k = b / kSplitPackBlocks;
    const int q = (b - k * kSplitPackBlocks) * kWG + tid;           // quad of elements
    char* img = reinterpret_cast<char*>(a.wimg) + (long long)k * I::BYTES;
    constexpr int offs[16] = {Flat32::W_IN, Flat32::B_IN, Flat32::W_M1, Flat32::B_M1, Flat32::W_CAT, Flat32::B_CAT, Flat32::W_M2, Flat32::B_M2,
                              Flat32::W_A, Flat32::B_A, Flat32::W_C, Flat32::B_C, Flat32::W_OC, Flat32::B_OC, Flat32::PE_B, Flat32::P};
    if (q < I::ELEMS / 4) {
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int t, o;
            float f = 0.0f;
            if (split_image_source(4 * q + e, t, o)) {
                f = a.fc[t].p[k * a.fc[t].stride + o];
                if (k == 0 && a.img_tab) a.img_tab[offs[t] + o] = 4 * q + e;
            }
            split3_scalar(f, h[e], m[e], l[e]);
            if (a.weights_bf16) { m[e] = 0u; l[e] = 0u; }
        }
        *reinterpret_cast<u32x2*>(img + I::P_HI + 8 * q) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<u32x2*>(img + I::P_MID + 8 * q) = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
        *reinterpret_cast<u32x2*>(img + I::P_LO + 8 * q) = u32x2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    } else {
        // small float32 vectors (+ zero padding of the region and of the image tail): 64 threads x 8 floats = 2048 bytes
        const int s0 = (q - I::ELEMS / 4) * 8;
        if (s0 < I::SMALL_BYTES / 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int t, o;
                float f = 0.0f;
                if (split_small_source(s0 + e, t, o)) {
                    f = t < kNFc ? a.fc[t].p[k * a.fc[t].stride + o] : a.pe_B.p[k * a.pe_B.stride + o];
                    if (k == 0 && a.img_tab) a.img_tab[offs[t] + o] = (int)(0x80000000u | (unsigned)(s0 + e));
                }
                reinterpret_cast<float*>(img + I::SMALL)[s0 + e] = a.weights_bf16 ? round_bf16(f) : f;
            }
            // the unused tail of the image (read by the LDS-DMA, never used) is zeroed once
            for (int i = I::END / 4 + (q - I::ELEMS / 4); i < I::BYTES / 4; i += 64) reinterpret_cast<float*>(img)[i] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// step_finalize_s32: step_finalize_h32 (same sums in the same order, same adamw_elem) writing the split image
// ---------------------------------------------------------------------------------------------------------
template <bool SLAB>
__device__ __forceinline__ void finalize_quad_s32(const FinalizeArgs& f, const FinalizeHot& a, int obj, int q) {
    float ss, bc;
    adam_step_consts(f, a, ss, bc);
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
        const int i = min(4 * q + e, Flat32::P - 1);
        if (SLAB) {
            pp[e] = a.slab + obj * a.slab_stride + i;
            tq[e] = 0;
        } else {
            int o;
            flat32_tensor_of(i, tq[e], o);
            pp[e] = f.param[tq[e]].p + obj * f.param[tq[e]].stride + o;
        }
    }
    // a quad that lies inside ONE parameter tensor (all but the few that straddle two) is one 16-byte access at a 4-byte boundary
    // instead of four 4-byte ones, in and out
    const bool pvec = 4 * q + 3 < Flat32::P && tq[0] == tq[3];
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
    char* image = reinterpret_cast<char*>(a.wimg) + (long long)obj * Img32s::BYTES;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (4 * q + e < Flat32::P) {
            float p = pv[e], m = m4[e], v = v4[e];
            adamw_elem(a, ss, bc, g[e], p, m, v);
            if (!pvec) *pp[e] = p;
            pv[e] = p; m4[e] = m; v4[e] = v;
            if (!(VS_ABL & 128)) split_image_store(image, img[e], p, a.weights_bf16);      // (bit 7: the finalize leaves the image alone - what its rewrite costs the NEXT main kernel)
        }
    }
    if (pvec) *reinterpret_cast<wv::f32x4u*>(pp[0]) = wv::f32x4{pv[0], pv[1], pv[2], pv[3]};
    *reinterpret_cast<wv::f32x4*>(a.m + s) = m4;
    *reinterpret_cast<wv::f32x4*>(a.v + s) = v4;
}

template <int = 0>
__global__ __launch_bounds__(kWG) void step_finalize_s32(const FinalizeArgs a, const FinalizeHot hh) {
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
        if (hh.slab) finalize_quad_s32<true>(a, hh, obj, q4);
        else finalize_quad_s32<false>(a, hh, obj, q4);
    }
}

// ---------------------------------------------------------------------------------------------------------
// device helpers of step_main_s32
// ---------------------------------------------------------------------------------------------------------
// N float32 registers (P-form, consecutive register pairs) -> NPL packed bf16 planes of N / 2 dwords each
template <int N, int NPL>
__device__ __forceinline__ void split_planes(const float (&x)[N], unsigned (&h)[N / 2], unsigned (&m)[N / 2], unsigned (&l)[N / 2]) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned ph = wv::pack_bf16(a, b);
        h[i] = ph;
        if (NPL >= 2) {
            const float ra = a - bf_lo(ph), rb = b - bf_hi(ph);
            const unsigned pm = wv::pack_bf16(ra, rb);
            m[i] = pm;
            if (NPL >= 3) l[i] = wv::pack_bf16(ra - bf_lo(pm), rb - bf_hi(pm));
        }
    }
}
template <int N>
__device__ __forceinline__ u32x4 opnd(const unsigned (&u)[N], int s) { return u32x4{u[4 * s], u[4 * s + 1], u[4 * s + 2], u[4 * s + 3]}; }

// forward, one 16-deep step: acc += W[:, step] . x[step]; wrow = this lane's row of the hi plane at the step (+16 hi)
template <bool W3>
__device__ __forceinline__ void fwd_step(f32x16& acc, const char* wrow, u32x4 xh, u32x4 xm, u32x4 xl) {
    using I = Img32s;
    const u32x4 wh = *reinterpret_cast<const u32x4*>(wrow + I::P_HI);
    if (W3) {
        const u32x4 wm = *reinterpret_cast<const u32x4*>(wrow + I::P_MID);
        const u32x4 wl = *reinterpret_cast<const u32x4*>(wrow + I::P_LO);
        acc = wv::mfma_bf16(wh, xl, acc);        // smallest terms first
        acc = wv::mfma_bf16(wl, xh, acc);
        acc = wv::mfma_bf16(wm, xm, acc);
        acc = wv::mfma_bf16(wh, xm, acc);
        acc = wv::mfma_bf16(wm, xh, acc);
        acc = wv::mfma_bf16(wh, xh, acc);
    } else {                                     // bf16 weights: one plane x the three planes of the activations
        acc = wv::mfma_bf16(wh, xl, acc);
        acc = wv::mfma_bf16(wh, xm, acc);
        acc = wv::mfma_bf16(wh, xh, acc);
    }
}

// NS consecutive 16-deep steps of a layer with the weight rows of step s + 1 requested before the matrix instructions of
// step s are issued (one wave per SIMD: a read issued right in front of its use is a fully exposed LDS round trip)
template <bool W3, int NS>
__device__ __forceinline__ void fwd_chain(f32x16& acc, const char* wrow, const unsigned* xh, const unsigned* xm, const unsigned* xl) {
    using I = Img32s;
    u32x4 wh = *reinterpret_cast<const u32x4*>(wrow + I::P_HI), wm = wh, wl = wh;
    if (W3) {
        wm = *reinterpret_cast<const u32x4*>(wrow + I::P_MID);
        wl = *reinterpret_cast<const u32x4*>(wrow + I::P_LO);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        u32x4 nh = wh, nm = wm, nl = wl;
        if (s + 1 < NS) {
            nh = *reinterpret_cast<const u32x4*>(wrow + 32 * (s + 1) + I::P_HI);
            if (W3) {
                nm = *reinterpret_cast<const u32x4*>(wrow + 32 * (s + 1) + I::P_MID);
                nl = *reinterpret_cast<const u32x4*>(wrow + 32 * (s + 1) + I::P_LO);
            }
        }
        const u32x4 bh = u32x4{xh[4 * s], xh[4 * s + 1], xh[4 * s + 2], xh[4 * s + 3]};
        const u32x4 bm = u32x4{xm[4 * s], xm[4 * s + 1], xm[4 * s + 2], xm[4 * s + 3]};
        const u32x4 bl = u32x4{xl[4 * s], xl[4 * s + 1], xl[4 * s + 2], xl[4 * s + 3]};
        if (W3) {
            acc = wv::mfma_bf16(wh, bl, acc);        // smallest terms first
            acc = wv::mfma_bf16(wl, bh, acc);
            acc = wv::mfma_bf16(wm, bm, acc);
            acc = wv::mfma_bf16(wh, bm, acc);
            acc = wv::mfma_bf16(wm, bh, acc);
            acc = wv::mfma_bf16(wh, bh, acc);
        } else {
            acc = wv::mfma_bf16(wh, bl, acc);
            acc = wv::mfma_bf16(wh, bm, acc);
            acc = wv::mfma_bf16(wh, bh, acc);
        }
        wh = nh; wm = nm; wl = nl;
    }
}

// lane coordinates of the transposing reads: 16-lane group G = lane >> 4 (half = G & 1: which 16 of the operand's 32 rows,
// hi = G >> 1), c = lane & 15, source-lane role jj = c >> 2 (row of the 4 x 4 block), q = c & 3 (its quad of 4 elements)
struct TrLane { int half, hi, jj, q; };
__device__ __forceinline__ TrLane tr_lane(int lane) { return TrLane{(lane >> 4) & 1, lane >> 5, (lane & 15) >> 2, lane & 3}; }

// A operand of a d-prop chain: W^T for the 32 input columns that start at 16-deep step `step0` of matrix rows at `wmat`
// (hi plane, row pitch PIT): w[pl * 8 + s * 4 + u * 2 + {0, 1}], pl = 0 hi / 1 mid plane, s = 16-deep step over the 32
// OUTPUT rows j, u = which four of the lane's eight j: element t = 4 u + jj' <-> j = 16 s + 8 u + 4 hi + jj'.
// NP = planes fetched (1: bf16 weights; 2: hi, mid; 3: + lo for the six-product backward): w[pl * 8 + ...]
template <int PIT, int NP>
__device__ __forceinline__ void wt_get(unsigned (&w)[24], const char* wmat, int step0, const TrLane& L) {
    using I = Img32s;
    const char* base = wmat + (4 * L.hi + L.jj) * PIT + (step0 + L.half) * 32 + ((L.q & 1) * 8 + (L.q >> 1) * 4) * 2;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32x2 v = wv::lds_tr16(base + (pl == 0 ? I::P_HI : pl == 1 ? I::P_MID : I::P_LO) + (16 * s + 8 * u) * PIT);
                w[pl * 8 + s * 4 + u * 2] = v[0];
                w[pl * 8 + s * 4 + u * 2 + 1] = v[1];
            }
}
// P-form planes (hi, mid; NQ quads of four features per lane) -> wave-private transpose tile [plane][point][feature]
template <int NQ>
__device__ __forceinline__ void tile_put(char* tile, const unsigned* h, const unsigned* m, int p31, int hi) {
    using I = Img32s;
    if (VS_ABL & 5) return;
    wv::wave_lds_fence();   // earlier reads of this tile are ordered before the overwrite
    char* row = tile + p31 * I::TPIT + hi * 8;
#pragma unroll
    for (int mq = 0; mq < NQ; ++mq) {
        *reinterpret_cast<u32x2*>(row + 16 * mq) = u32x2{h[2 * mq], h[2 * mq + 1]};
        *reinterpret_cast<u32x2*>(row + I::TPL + 16 * mq) = u32x2{m[2 * mq], m[2 * mq + 1]};
    }
    wv::wave_lds_fence();
}
// F-form operand (lane = feature, elements = 16 points per half): f[pl * 8 + s * 4 + u * 2 + {0, 1}]; element t = 4 u + jj'
// of step s <-> point 16 s + 8 hi + t
__device__ __forceinline__ void tile_get(unsigned (&f)[16], const char* tile, const TrLane& L) {
    using I = Img32s;
    if (VS_ABL & 5) { for (int i = 0; i < 16; ++i) f[i] = 0x3F803F80u + (unsigned)((size_t)tile & 0xFFu); return; }
    const char* base = tile + (8 * L.hi + L.jj) * I::TPIT + (16 * L.half + 4 * L.q) * 2;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const u32x2 v = wv::lds_tr16(base + pl * I::TPL + (16 * s + 4 * u) * I::TPIT);
                f[pl * 8 + s * 4 + u * 2] = v[0];
                f[pl * 8 + s * 4 + u * 2 + 1] = v[1];
            }
}
// ---- P-form -> F-form ON THE MATRIX PIPE (round 6).  The weight-gradient products contract over POINTS, so both operands must hold
// 8 consecutive points per lane (lane = feature): a lane <-> register transpose of the P-form planes.  Through LDS (tile_put + tile_get:
// 8 ds_write_b64 + 16 ds_read_b64_tr_b16 per block and wave, four waves at once) that cost 3.2 us of the kernel's 23.2 (measurement
// builds, profiles/round6_ablation_step_main_s32.jsonl).  Here the plane itself is the A operand (lane = point i, k = feature) of a
// product with a SELECTOR B[k][n] = (feature(k) == n): D[i][n] = X[i][n] exactly (bf16 x 1.0, zeros), and D comes back in the
// accumulator map - lane = n = feature, register r <-> point (r & 3) + 8 (r >> 2) + 4 hi - which IS an F-form: registers 8 s .. 8 s + 7
// are the eight k of 16-deep step s.  Both operands of a weight-gradient product are built this way, so their point <-> k maps agree.
// 2 matrix instructions + 8 v_cvt_pk_bf16_f32 per plane, no LDS.
struct SelOps { u32x4 s0, s1; };
// lane (n = p31, hi'): element t of step s is 1.0 iff hidden_k(s, hi', t) == n
__device__ __forceinline__ SelOps sel_ops(int p31, int hi) {
    const int q = p31 & 15, t = (q & 3) + 4 * (q >> 3);
    const bool mine = ((q >> 2) & 1) == hi;
    const unsigned one = (t & 1) ? 0x3F800000u : 0x00003F80u;
    u32x4 w = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (mine && (t >> 1) == j) ? one : 0u;
    const u32x4 z = {0u, 0u, 0u, 0u};
    return SelOps{p31 < 16 ? w : z, p31 < 16 ? z : w};
}
// NQ = 4: a whole 32-feature block (8 dwords per plane); NQ = 2: its first 16-deep step only (features 16..31 of the result are zero)
template <int NQ>
__device__ __forceinline__ void toF_plane(unsigned* f, const unsigned* h, const SelOps& S) {
    f32x16 t;
    zero_acc(t);
    t = wv::mfma_bf16(u32x4{h[0], h[1], h[2], h[3]}, S.s0, t);
    if (NQ == 4) t = wv::mfma_bf16(u32x4{h[4], h[5], h[6], h[7]}, S.s1, t);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = wv::pack_bf16(t[2 * j], t[2 * j + 1]);
}
template <int NQ>
__device__ __forceinline__ void toF_mm(unsigned (&f)[16], const unsigned* h, const unsigned* m, const SelOps& S) {
    toF_plane<NQ>(f, h, S);
    toF_plane<NQ>(f + 8, m, S);
}
// the same for NPL planes into f[pl * 8 + ...] (NPL = 3: the six-product backward also carries the lo plane)
template <int NQ, int NPL>
__device__ __forceinline__ void toF_mm3(unsigned (&f)[24], const unsigned* h, const unsigned* m, const unsigned* l, const SelOps& S) {
    toF_plane<NQ>(f, h, S);
    toF_plane<NQ>(f + 8, m, S);
    if (NPL == 3) toF_plane<NQ>(f + 16, l, S);
}
// weight gradient: acc[j][k] += sum_p dY[p][j] X[p][k], both operands in F-form planes: hi.mid + mid.hi + hi.hi
__device__ __forceinline__ void dw_mm_s(f32x16& acc, const unsigned (&dyF)[16], const unsigned (&xF)[16]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const u32x4 ah = u32x4{dyF[4 * s], dyF[4 * s + 1], dyF[4 * s + 2], dyF[4 * s + 3]};
        const u32x4 am = u32x4{dyF[8 + 4 * s], dyF[8 + 4 * s + 1], dyF[8 + 4 * s + 2], dyF[8 + 4 * s + 3]};
        const u32x4 bh = u32x4{xF[4 * s], xF[4 * s + 1], xF[4 * s + 2], xF[4 * s + 3]};
        const u32x4 bm = u32x4{xF[8 + 4 * s], xF[8 + 4 * s + 1], xF[8 + 4 * s + 2], xF[8 + 4 * s + 3]};
        acc = wv::mfma_bf16(ah, bm, acc);
        acc = wv::mfma_bf16(am, bh, acc);
        acc = wv::mfma_bf16(ah, bh, acc);
    }
}
// bias gradient of a layer whose input has no constant-1 slot (mid1, mid2): dY^T . ones -> every column of the result holds
// sum_p dY[p][j]; lanes 0 and 32 add their 16 rows to the wave's small-vector accumulators
__device__ __forceinline__ void db_ones(float* gb, const unsigned (&dyF)[16], int p31, int hi) {
    const u32x4 ones = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    f32x16 acc;
    zero_acc(acc);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        acc = wv::mfma_bf16(u32x4{dyF[8 + 4 * s], dyF[8 + 4 * s + 1], dyF[8 + 4 * s + 2], dyF[8 + 4 * s + 3]}, ones, acc);
        acc = wv::mfma_bf16(u32x4{dyF[4 * s], dyF[4 * s + 1], dyF[4 * s + 2], dyF[4 * s + 3]}, ones, acc);
    }
    if (p31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gb[phi(r, hi)] += acc[r];
    }
}
// ReLU mask from the packed hi plane of the activation: d[r] = h[r] > 0 ? v[r] : 0 (h >= 0: nonzero <=> positive)
__device__ __forceinline__ void relu_mask(float (&d)[16], const f32x16& v, const unsigned (&hh)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned u = wv::opaque_u(hh[i]);
        d[2 * i] = (u & 0xFFFFu) != 0u ? v[2 * i] : 0.0f;
        d[2 * i + 1] = u > 0xFFFFu ? v[2 * i + 1] : 0.0f;
    }
}

// where column k (= lane) of a 32-column weight-gradient block goes.  KIND 0: natural (column k of the tensor row),
// 1: first-group encoding block blk (0..2), 2: second-group block blk (0..1).  col = column inside the tensor row (after
// the hidden part) or -1; bias = the column is the layer's bias gradient.
template <int KIND>
__host__ __device__ __forceinline__ void col_target(int blk, int k, int& col, bool& bias) {
    bias = false;
    if (KIND == 0) { col = k; return; }
    const int hs = (k >> 2) & 1, r = (k & 3) + 4 * (k >> 3), R = 16 * blk + r;
    int c;
    if (KIND == 1) c = e1_slot(R, hs);
    else c = R < 24 ? e2_slot(R, hs) : kSlotPad;
    bias = c == kSlotOne;
    col = c >= 0 ? c : -1;
}
// this wave's quarter (rows 8 wave + 4 hi + i) of a reduced block -> the workgroup's partial gradients
#if VS_ABL & 16
#define VS_PARTIAL_STORE(v, p) asm volatile("" ::"v"(v), "v"(p))
#elif defined(VS_PARTIAL_PLAIN)          /* A/B: the rows as ordinary stores */
#define VS_PARTIAL_STORE(v, p) (*(p) = (v))
#else
#define VS_PARTIAL_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif
template <int K>
__device__ __forceinline__ void store_quarter_map(float* out_w, float* out_b, const float (&q)[4], int col, bool bias, int ncols,
                                                  int wave, int hi) {
    if (col >= 0 && col < ncols) {
#pragma unroll
        for (int i = 0; i < 4; ++i) VS_PARTIAL_STORE(q[i], &out_w[(8 * wave + 4 * hi + i) * K + col]);
    } else if (bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) VS_PARTIAL_STORE(q[i], &out_b[8 * wave + 4 * hi + i]);
    }
}

// after the workgroup barrier that follows a block's staging: reduce this wave's quarter over the four staged tiles and
// store it (single pass) or keep it (MULTI)
template <int KIND, int K, bool MULTI>
__device__ __forceinline__ void finish_block_s(float (&qp)[4], const float* stage, float* out_w, float* out_b, int blk, int ncols,
                                               int wave, int p31, int hi) {
    if (VS_ABL & (9 | 32)) return;
    int col; bool bias;
    col_target<KIND>(blk, p31, col, bias);
    if (MULTI) {
        stage_get(qp, stage, wave, p31, hi);
    } else {
        float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        stage_get(q, stage, wave, p31, hi);
        store_quarter_map<K>(out_w, out_b, q, col, bias, ncols, wave, hi);
    }
}


// ---- matrix chains with VALU work interleaved BY HAND.  One wave per SIMD issues in order: a matrix instruction that waits
// for the pipe blocks everything behind it, so independent VALU work hides behind matrix instructions only when the two are
// interleaved in the instruction stream (<= 6 VALU per bf16 matrix instruction are free: profiles/r02a_bf16_probe.jsonl).
// Every helper takes a callable vc(i) that is invoked right after matrix instruction i, between scheduling fences. ----
// MODE 0: bf16 weights (one weight plane x delta hi, mid); 1: float32 weights, three products (hi.mid + mid.hi + hi.hi, ~2^-16);
// 2: float32 weights, SIX products (+ hi.lo + lo.hi + mid.mid, ~2^-24: the float32-equivalent backward, tuning.bwd_products = 6)
constexpr int kDpropPS(int mode) { return mode == 0 ? 2 : mode == 1 ? 3 : 6; }
constexpr int kDpropMM(int mode) { return 2 * kDpropPS(mode); }
template <int MODE, class VC>
__device__ __forceinline__ void mm_dprop_il(f32x16& acc, const unsigned (&w)[24], const unsigned (&dh)[8], const unsigned (&dm)[8], const unsigned (&dl)[8], VC&& vc) {
    constexpr int PS = kDpropPS(MODE);
    wv::sched_fence();
#pragma unroll
    for (int i = 0; i < 2 * PS; ++i) {
        const int s = i / PS, k = i % PS;
        const u32x4 wh = u32x4{w[4 * s], w[4 * s + 1], w[4 * s + 2], w[4 * s + 3]};
        const u32x4 wm = u32x4{w[8 + 4 * s], w[8 + 4 * s + 1], w[8 + 4 * s + 2], w[8 + 4 * s + 3]};
        const u32x4 wl = u32x4{w[16 + 4 * s], w[16 + 4 * s + 1], w[16 + 4 * s + 2], w[16 + 4 * s + 3]};
        if (MODE == 2) {                               // smallest terms first
            acc = k == 0 ? wv::mfma_bf16(wh, opnd(dl, s), acc) : k == 1 ? wv::mfma_bf16(wl, opnd(dh, s), acc) : k == 2 ? wv::mfma_bf16(wm, opnd(dm, s), acc)
                : k == 3 ? wv::mfma_bf16(wh, opnd(dm, s), acc) : k == 4 ? wv::mfma_bf16(wm, opnd(dh, s), acc) : wv::mfma_bf16(wh, opnd(dh, s), acc);
        } else if (k == 0) acc = wv::mfma_bf16(wh, opnd(dm, s), acc);
        else if (MODE == 1 && k == 1) acc = wv::mfma_bf16(wm, opnd(dh, s), acc);
        else acc = wv::mfma_bf16(wh, opnd(dh, s), acc);
        wv::sched_fence();
        vc(i);
        wv::sched_fence();
    }
}
// weight-gradient chain (6 matrix instructions), optionally followed by the 4 of the ones-column bias gradient (DB)
// B6: six products per 16-deep step (+ hi.lo + lo.hi + mid.mid) and the lo plane in the bias sums
constexpr int kDwMM(bool b6) { return b6 ? 12 : 6; }
constexpr int kDwDbMM(bool b6) { return b6 ? 18 : 10; }
template <bool DB, bool B6, class VC>
__device__ __forceinline__ void mm_dw_il(f32x16& acc, f32x16& accb, const unsigned (&dyF)[24], const unsigned (&xF)[24], VC&& vc) {
    const u32x4 ones = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    constexpr int PS = B6 ? 6 : 3, NDW = 2 * PS, PB = B6 ? 3 : 2;
    wv::sched_fence();
#pragma unroll
    for (int i = 0; i < (DB ? NDW + 2 * PB : NDW); ++i) {
        if (VS_ABL & 1) {
        } else if (i < NDW) {
            const int s = i / PS, k = i % PS;
            const u32x4 ah = u32x4{dyF[4 * s], dyF[4 * s + 1], dyF[4 * s + 2], dyF[4 * s + 3]};
            const u32x4 am = u32x4{dyF[8 + 4 * s], dyF[8 + 4 * s + 1], dyF[8 + 4 * s + 2], dyF[8 + 4 * s + 3]};
            const u32x4 al = u32x4{dyF[16 + 4 * s], dyF[16 + 4 * s + 1], dyF[16 + 4 * s + 2], dyF[16 + 4 * s + 3]};
            const u32x4 bh = u32x4{xF[4 * s], xF[4 * s + 1], xF[4 * s + 2], xF[4 * s + 3]};
            const u32x4 bm = u32x4{xF[8 + 4 * s], xF[8 + 4 * s + 1], xF[8 + 4 * s + 2], xF[8 + 4 * s + 3]};
            const u32x4 bl = u32x4{xF[16 + 4 * s], xF[16 + 4 * s + 1], xF[16 + 4 * s + 2], xF[16 + 4 * s + 3]};
            if (B6) {                                  // smallest terms first
                acc = k == 0 ? wv::mfma_bf16(ah, bl, acc) : k == 1 ? wv::mfma_bf16(al, bh, acc) : k == 2 ? wv::mfma_bf16(am, bm, acc)
                    : k == 3 ? wv::mfma_bf16(ah, bm, acc) : k == 4 ? wv::mfma_bf16(am, bh, acc) : wv::mfma_bf16(ah, bh, acc);
            } else
                acc = k == 0 ? wv::mfma_bf16(ah, bm, acc) : k == 1 ? wv::mfma_bf16(am, bh, acc) : wv::mfma_bf16(ah, bh, acc);
        } else {
            const int s = (i - NDW) / PB, k = (i - NDW) % PB, pl = PB - 1 - k;      // lo (B6), mid, hi
            const u32x4 a = u32x4{dyF[8 * pl + 4 * s], dyF[8 * pl + 4 * s + 1], dyF[8 * pl + 4 * s + 2], dyF[8 * pl + 4 * s + 3]};
            accb = wv::mfma_bf16(a, ones, accb);
        }
        wv::sched_fence();
        vc(i);
        wv::sched_fence();
    }
}
// ReLU mask + two-plane split of register pair j of a d-prop result (the VALU work of a hidden unit, one chunk per pair)
template <int NPD = 2>
__device__ __forceinline__ void mask_split_pair(int j, const f32x16& v, const unsigned (&hh)[8], unsigned (&dh)[8], unsigned (&dm)[8], unsigned (&dl)[8], float dep) {
    const unsigned u = wv::opaque_u(hh[j]);
    const float va = wv::after(v[2 * j], dep);           // (the tie sits outside the select: inside it the select becomes a branch)
    const float a = (u & 0xFFFFu) != 0u ? va : 0.0f;
    const float b = u > 0xFFFFu ? v[2 * j + 1] : 0.0f;
    const unsigned ph = wv::pack_bf16(a, b);
    dh[j] = ph;
    const float ra = a - bf_lo(ph), rb = b - bf_hi(ph);
    const unsigned pm = wv::pack_bf16(ra, rb);
    dm[j] = pm;
    if (NPD == 3) dl[j] = wv::pack_bf16(ra - bf_lo(pm), rb - bf_hi(pm));
}
// finishing a staged block in two chunks: 0 = the four 16-byte reads of this wave's quarter, 1 = sum + store (or keep, MULTI)
struct FinState { wv::f32x4 t0, t1, t2, t3; };
template <int KIND, int K, bool MULTI>
__device__ __forceinline__ void fin_chunk(int j, FinState& st, float (&qp)[4], const float* stage, float* out_w, float* out_b, int blk,
                                          int ncols, int wave, int p31, int hi) {
    if (VS_ABL & (9 | 32)) return;
    if (j == 0) {
        const float* rd = stage + p31 * Lds32::TP + 8 * wave + 4 * hi;
        st.t0 = *reinterpret_cast<const wv::f32x4*>(rd);
        st.t1 = *reinterpret_cast<const wv::f32x4*>(rd + Lds32::STG_TILE);
        st.t2 = *reinterpret_cast<const wv::f32x4*>(rd + 2 * Lds32::STG_TILE);
        st.t3 = *reinterpret_cast<const wv::f32x4*>(rd + 3 * Lds32::STG_TILE);
    } else {
        if (MULTI) {
#pragma unroll
            for (int i = 0; i < 4; ++i) qp[i] += (st.t0[i] + st.t1[i]) + (st.t2[i] + st.t3[i]);
        } else {
            float q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = 0.0f + ((st.t0[i] + st.t1[i]) + (st.t2[i] + st.t3[i]));
            int col; bool bias;
            col_target<KIND>(blk, p31, col, bias);
            store_quarter_map<K>(out_w, out_b, q, col, bias, ncols, wave, hi);
        }
    }
}

// One "gap" of the forward: the ReLU + three-plane split of a layer's 16 outputs (16 half-pair chunks of 6-7 VALU
// instructions) interleaved with NS 16-deep steps of an encoding half that does not depend on them (6 / 3 matrix
// instructions per step).  wrowB = the lane's weight row of the hi plane at the first of those steps.
template <bool W3, int NS>
__device__ __forceinline__ void gap_fill(f32x16& accB, const char* wrowB, const unsigned* xh, const unsigned* xm, const unsigned* xl,
                                         const f32x16& accA, float (&hf)[16], unsigned (&hh)[8], unsigned (&hm)[8], unsigned (&hl)[8]) {
    using I = Img32s;
    constexpr int PS = W3 ? 6 : 3, NM = NS * PS;
    u32x4 wh[NS], wm[NS], wl[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        wh[s] = *reinterpret_cast<const u32x4*>(wrowB + 32 * s + I::P_HI);
        if (W3) {
            wm[s] = *reinterpret_cast<const u32x4*>(wrowB + 32 * s + I::P_MID);
            wl[s] = *reinterpret_cast<const u32x4*>(wrowB + 32 * s + I::P_LO);
        }
    }
    float ra[8], rb[8];
    wv::sched_fence();
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int s = m / PS, k = m % PS;
        const u32x4 bh = u32x4{xh[4 * s], xh[4 * s + 1], xh[4 * s + 2], xh[4 * s + 3]};
        const u32x4 bm = u32x4{xm[4 * s], xm[4 * s + 1], xm[4 * s + 2], xm[4 * s + 3]};
        const u32x4 bl = u32x4{xl[4 * s], xl[4 * s + 1], xl[4 * s + 2], xl[4 * s + 3]};
        if (W3) {
            accB = k == 0 ? wv::mfma_bf16(wh[s], bl, accB) : k == 1 ? wv::mfma_bf16(wl[s], bh, accB) : k == 2 ? wv::mfma_bf16(wm[s], bm, accB)
                 : k == 3 ? wv::mfma_bf16(wh[s], bm, accB) : k == 4 ? wv::mfma_bf16(wm[s], bh, accB) : wv::mfma_bf16(wh[s], bh, accB);
        } else {
            accB = k == 0 ? wv::mfma_bf16(wh[s], bl, accB) : k == 1 ? wv::mfma_bf16(wh[s], bm, accB) : wv::mfma_bf16(wh[s], bh, accB);
        }
        wv::sched_fence();
#pragma unroll
        for (int c = m * 16 / NM; c < (m + 1) * 16 / NM; ++c) {
            const int i = c >> 1;
            if ((c & 1) == 0) {
                const float a = wv::relu(wv::after(accA[2 * i], accB[0])), b = wv::relu(accA[2 * i + 1]);   // behind matrix instruction m
                hf[2 * i] = a; hf[2 * i + 1] = b;
                const unsigned ph = wv::pack_bf16(a, b);
                hh[i] = ph;
                ra[i] = a - bf_lo(ph); rb[i] = b - bf_hi(ph);
            } else {
                const unsigned pm = wv::pack_bf16(wv::after(ra[i], accB[0]), rb[i]);
                hm[i] = pm;
                hl[i] = wv::pack_bf16(ra[i] - bf_lo(pm), rb[i] - bf_hi(pm));
            }
        }
        wv::sched_fence();
    }
}

// sin / cos of the six octaves 2^f * a of one angle: one accurate sincos + five double-angle steps.  2^f * a is exact in
// float32, so this IS sin(fl32(2^f * proj * pi)) of embedding.py:85-88 up to the recurrence's rounding (~2^f ulp).
template <bool BIG>
__device__ __forceinline__ void octave_sincos(float a0, float (&s)[6], float (&c)[6]) {
    if (BIG) sincosf(a0, &s[0], &c[0]);
    else sincos_f32(a0, s[0], c[0]);
#pragma unroll
    for (int f = 1; f < 6; ++f) {
        const float t = s[f - 1] + s[f - 1];
        s[f] = t * c[f - 1];
        c[f] = fmaf(-t, s[f - 1], 1.0f);
    }
}

__device__ __forceinline__ void stage_put_s(float* stage, const f32x16& acc, int wave, int p31, int hi) {
    if (!(VS_ABL & 9)) stage_put(stage, acc, wave, p31, hi);
    else if (VS_ABL & 8) {                      // the products stay: their result is "used" (no instruction)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[r]));
    }
}
__device__ __forceinline__ void stage_get_s(float (&q)[4], const float* stage, int wave, int p31, int hi) {
    if (!(VS_ABL & 9)) stage_get(q, stage, wave, p31, hi);
}

// ---------------------------------------------------------------------------------------------------------
// step_main_s32<BWD, MULTI, STAMPS, W3>:  W3 = float32 weights as three planes (false: bf16 weights, one plane)
// ---------------------------------------------------------------------------------------------------------
template <bool BWD, bool MULTI, bool STAMPS, bool W3, bool B6 = false>
__device__ __forceinline__ void step_main_s32_body(const StepArgs& a) {
    static_assert(!B6 || (W3 && BWD), "six-product backward: float32 weights, training instantiations");
    using I = Img32s;
    using F = Flat32;
    constexpr int H = 32;
    char* lds = reinterpret_cast<char*>(wv::lds_base());
    const char* W = lds;                                       // the image
    const float* SM = reinterpret_cast<const float*>(lds + I::SMALL);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p31 = lane & 31, hi = lane >> 5;
    float* Gv = reinterpret_cast<float*>(lds + I::VEC) + wave * I::VEC_N;        // this wave's private small-vector gradients
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
#define VS_MARK(i) do { if constexpr (STAMPS) { if (tmark && lane == 0) tmark[i] = wv::clock32(); } } while (0)
    VS_MARK(0);
    if (BWD) {
        for (int i = tid; i < kWaves * I::VEC_N; i += kWG) reinterpret_cast<float*>(lds + I::VEC)[i] = 0.0f;
    }
    float* loss_cells = reinterpret_cast<float*>(lds + I::LOSS);
    if (tid < kWaves * 4) loss_cells[tid] = 0.0f;
    float* out = a.part_grad + ((long long)(obj * a.NW + wgo)) * a.PP;   // this workgroup's partial gradients
    float qacc[12][4];      // MULTI only: this wave's quarter of every reduced weight-gradient block
#pragma unroll
    for (int b = 0; b < 12; ++b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) qacc[b][i] = 0.0f;
    }
    float* stg0 = reinterpret_cast<float*>(lds + I::STG);
    float* stg1 = stg0 + kWaves * Lds32::STG_TILE;
    char* scrX = lds + I::SCR + wave * 2 * I::TILE;
    char* scrD = scrX + I::TILE;
    float* cb = reinterpret_cast<float*>(lds + I::CB);
    const float* cbw = cb + wave * 32 * 8;
    const float scale = a.pe_scale.p[obj * a.pe_scale.stride];
    const char* gimg = reinterpret_cast<const char*>(a.wimg) + (long long)obj * I::BYTES;
    const float* Bg = reinterpret_cast<const float*>(gimg + I::SMALL) + I::PE_B;     // B_layer.weight, from the global image

    const int tid_k = tid;
    for (int grp = wgo; grp < a.NG; grp += a.NW) {   // ---- one pass = up to kMaxPts points (whole rays) ----
    const int tid = MULTI ? wv::opaque_iter(tid_k) : tid_k;
    const int lane = tid & 63, wave = tid >> 6, p31 = lane & 31, hi = lane >> 5;
    const TrLane TL = tr_lane(lane);
    __syncthreads();                                 // previous pass finished with the composite buffer and the staging tiles
    for (int i = tid; i < kMaxPts * 8; i += kWG) cb[i] = 0.0f;

    // ---- this lane's sample point ----
    const int ray0 = grp * a.G;
    const int nrays = min(a.G, a.R - ray0);
    const int npts = nrays * a.S;
    const int pt = wave * 32 + p31;
    const bool valid = pt < npts;
    const int lray = valid ? pt / a.S : 0;
    const int smp = valid ? pt - lray * a.S : 0;
    const int ray = ray0 + lray;
    float px3[3] = {0.0f, 0.0f, 0.0f};
    if (VS_ABL & 512) { px3[0] = 0.01f * (float)lane; px3[1] = 0.02f * (float)wave; px3[2] = 0.003f * (float)(lane + wave); }
    else if (valid) load_point(a, obj, ray, smp, px3[0], px3[1], px3[2]);
    // ---- asynchronous copy of the parameter image into LDS (issued behind the loads of the sample point, whose latency its 20 instructions cover; lands during the encoding) ----
    if (grp == wgo && !(VS_ABL & 256)) {
        const char* src = gimg + wave * 1024 + lane * 16;
#pragma unroll
        for (int c = 0; c < I::ROUNDS; ++c)
            wv::glds16(reinterpret_cast<const float*>(src + c * 4096), reinterpret_cast<float*>(lds + c * 4096 + wave * 1024));
    }
    const float t[3] = {px3[0] / scale, px3[1] / scale, px3[2] / scale};          // embedding.py:83  x / self.scale (0 for padding lanes)
    // this lane's directions: hi = 0 -> 0..10, hi = 1 -> 11..20 (+ one dummy)
    float proj[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int d0 = i, d1 = i < 10 ? 11 + i : 20;
        const float b0 = hi ? Bg[3 * d1] : Bg[3 * d0], b1 = hi ? Bg[3 * d1 + 1] : Bg[3 * d0 + 1], b2 = hi ? Bg[3 * d1 + 2] : Bg[3 * d0 + 2];
        proj[i] = fmaf(t[2], b2, fmaf(t[1], b1, t[0] * b0));          // embedding.py:84 B_layer(tensor)
    }

    VS_MARK(1);

    // ---- encoding (embedding.py:82-91): own directions, octaves by double-angle recurrence ----
    float e1[48], e2[24];          // values of this lane's slots (P-form registers of the two groups)
    float cfac[11][6];             // cos * pi * 2^f of this lane's directions (backward)
    {
        float amax = 0.0f;
#pragma unroll
        for (int i = 0; i < 11; ++i) amax = fmaxf(amax, fabsf(proj[i]));
        const bool fast = !wv::wave_any(!(amax * (32.0f * kPi) < kSinCosFastLimit));
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            float s[6], c[6];
            const float a0 = proj[i] * kPi;            // fl32(proj * fl32(pi)); the octaves 2^f * a0 are exact
            if (VS_ABL & 4096) { for (int f = 0; f < 6; ++f) { s[f] = a0; c[f] = proj[i]; } }
            else if (__builtin_expect(fast, 1)) octave_sincos<false>(a0, s, c);
            else octave_sincos<true>(a0, s, c);
            const bool own = i < 10 || hi == 0;        // the eleventh direction of the hi = 1 lanes is a dummy
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                const float sv = own ? s[f] : 0.0f;
                cfac[i][f] = own ? c[f] * (kPi * (float)(1 << f)) : 0.0f;
                if (f < 4) e1[4 * i + f] = sv;
                else e2[2 * i + (f - 4)] = sv;
            }
        }
        // slots behind the directions: hi = 0: xyz + the constant one (bias column); hi = 1: zero padding
        e1[44] = hi ? 0.0f : t[0]; e1[45] = hi ? 0.0f : t[1]; e1[46] = hi ? 0.0f : t[2]; e1[47] = hi ? 0.0f : 1.0f;
        e2[22] = hi ? 0.0f : 1.0f; e2[23] = 0.0f;
    }
    unsigned e1h[24], e1m[24], e1l[24], e2h[12], e2m[12], e2l[12];
    split_planes<48, 3>(e1, e1h, e1m, e1l);
    split_planes<24, 3>(e2, e2h, e2m, e2l);
    VS_MARK(2);
    __syncthreads();        // parameter image landed (the barrier drains the LDS-DMA), composite buffer zeroed
    // ground truth of the ray this lane composites: its six vector loads fly during the MLP forward (at the compositing they
    // cost a full memory round trip of an otherwise idle workgroup)
    RayMeta rays_meta;
    if (VS_ABL & 1024) { rays_meta.sem = 1; rays_meta.dm = 1; rays_meta.gtd = 1.5f; rays_meta.q0 = 0.2f; rays_meta.q1 = 0.3f; rays_meta.q2 = 0.4f; rays_meta.inv_dd = rays_meta.inv_o = rays_meta.inv_s = 0.01f; }
    else rays_meta = load_ray_meta_rays(a, obj, ray0 + min(4 * wave + (lane >> 4), nrays - 1));
    wv::sched_fence();

    // ---- field MLP forward (model.py:59-83) ----
    unsigned h1h[8], h1m[8], h1l[8], h2h[8], h2m[8], h2l[8], h3h[8], h3m[8], h3l[8], h4h[8], h4m[8], h4l[8];   // (the lo planes live on only in the six-product backward)
    float h4[16], hc[16];
    f32x16 acc;
    {
        // The layer-to-layer chain (matrix chain -> ReLU -> split into planes -> next matrix chain) would leave the matrix pipe
        // idle while the ~100 VALU instructions of a split run; the encoding halves of cat_layer and color_linear do not
        // depend on any hidden layer, so their 54 matrix instructions are interleaved with the four splits (gap_fill).
        float hf[16];
        f32x16 accE, accC;
        const char* w = W + I::O_IN + p31 * I::PIT_IN + 16 * hi;
        const char* wcat = W + I::O_CAT + p31 * I::PIT_CAT + 16 * hi;
        const char* wc = W + I::O_C + p31 * I::PIT_C + 16 * hi;
        zero_acc(acc);                                            // the bias rides in the column of the constant-1 slot
        fwd_chain<W3, 6>(acc, w, e1h, e1m, e1l);
        zero_acc(accE);
        gap_fill<W3, 3>(accE, wcat + 32 * 2, e1h, e1m, e1l, acc, hf, h1h, h1m, h1l);                 // :59 in_layer -> h1 | :63 x[:emb1] half, steps 0..2
        w = W + I::O_M1 + p31 * I::PIT_M + 16 * hi;
        load_bias(acc, SM + I::B_M1, hi);
        fwd_chain<W3, 2>(acc, w, h1h, h1m, h1l);
        gap_fill<W3, 3>(accE, wcat + 32 * 5, e1h + 12, e1m + 12, e1l + 12, acc, hf, h2h, h2m, h2l);  // :60 mid1 -> h2 | steps 3..5
        fwd_chain<W3, 2>(accE, wcat, h2h, h2m, h2l);                                                  // :63 cat((fc2, x[:emb1]))
        zero_acc(accC);
        gap_fill<W3, 2>(accC, wc + 32 * 2, e2h, e2m, e2l, accE, hf, h3h, h3m, h3l);                  // :64 cat_layer -> h3 | :81 x[emb1:] half, steps 0, 1
        w = W + I::O_M2 + p31 * I::PIT_M + 16 * hi;
        load_bias(acc, SM + I::B_M2, hi);
        fwd_chain<W3, 2>(acc, w, h3h, h3m, h3l);
        gap_fill<W3, 1>(accC, wc + 32 * 4, e2h + 8, e2m + 8, e2l + 8, acc, h4, h4h, h4m, h4l);       // :67 mid2 -> h4 | step 2
        fwd_chain<W3, 2>(accC, wc, h4h, h4m, h4l);                                                    // :81 cat((fc4, x[emb1:]))
        relu_to(hc, accC);                                        // :81 color_linear
    }
    VS_MARK(3);
    {
        float ra = 0.0f, r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = phi(r, hi);
            ra = fmaf(SM[I::W_A + j], h4[r], ra);                 // :71 out_alpha
            r0 = fmaf(SM[I::W_OC + j], hc[r], r0);                // :82 out_color
            r1 = fmaf(SM[I::W_OC + H + j], hc[r], r1);
            r2 = fmaf(SM[I::W_OC + 2 * H + j], hc[r], r2);
        }
        ra += wv::swap_half(ra); r0 += wv::swap_half(r0); r1 += wv::swap_half(r1); r2 += wv::swap_half(r2);
        ra += SM[I::B_A]; r0 += SM[I::B_OC]; r1 += SM[I::B_OC + 1]; r2 += SM[I::B_OC + 2];
        if (valid && hi == 0) {
            float* row = cb + pt * 8;
            row[6] = (VS_ABL & 512) ? 1.0f + 0.1f * (float)smp : a.z[obj * a.z_so + ray * a.z_sr + smp * a.z_ss];
            row[0] = sigmoidf_acc(ra * 10.0f);                   // :77 raw*10 ; render_rays.py:6 sigmoid
            row[1] = sigmoidf_acc(r0);                           // :83 sigmoid(raw_color)
            row[2] = sigmoidf_acc(r1);
            row[3] = sigmoidf_acc(r2);
        }
        if (BWD) {
            // float32 transposes of h4 / hc for the head gradients (consumed after the compositing; the tiles are idle until then)
            toF_put(reinterpret_cast<float*>(scrX), h4, p31, hi);
            toF_put(reinterpret_cast<float*>(scrD), hc, p31, hi);
        }
    }
    VS_MARK(4);
    if (!(VS_ABL & 65536)) __syncthreads();
    VS_MARK(5);
    {
        const StepArgs& al = wv::kernarg_late(a);
        if (!(VS_ABL & 2048))
        composite_phase<BWD>(al, cb, loss_cells, obj, ray0, nrays, wave, lane, tid,
                             (VS_ABL & 1024) ? rays_meta : finish_ray_meta(al, obj, rays_meta));
    }
    if (!(VS_ABL & 65536)) __syncthreads();
    VS_MARK(6);
    if (BWD) {
    // ---- backward ----
    float d_raw, d_c0, d_c1, d_c2;
    {
        const float* row = cb + pt * 8;          // pt < kMaxPts always; padding rows hold zeros
        d_raw = row[0]; d_c0 = row[1]; d_c1 = row[2]; d_c2 = row[3];
    }
    constexpr int DM = !W3 ? 0 : B6 ? 2 : 1, NA = kDpropMM(DM), NPW = !W3 ? 1 : B6 ? 3 : 2, NPB = B6 ? 3 : 2, NDW = kDwMM(B6), NDB = kDwDbMM(B6);
    unsigned dF[24];
    float dproj[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) dproj[i] = 0.0f;
    unsigned dch[8], dcm[8], dcl[8], d4h[8], d4m[8], d4l[8];
    {
        // heads: out_alpha / out_color weight + bias gradients (lane = hidden feature), float32 as in step_main_h32
        float h4F[16], hcF[16];
        toF_get(h4F, reinterpret_cast<const float*>(scrX), p31, hi);
        toF_get(hcF, reinterpret_cast<const float*>(scrD), p31, hi);
        float ga = 0.0f, g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, sa = 0.0f, s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* row = cbw + (r + 16 * hi) * 8;
            const float da = row[0], q0 = row[1], q1 = row[2], q2 = row[3];
            ga = fmaf(da, h4F[r], ga);
            g0 = fmaf(q0, hcF[r], g0); g1 = fmaf(q1, hcF[r], g1); g2 = fmaf(q2, hcF[r], g2);
            sa += da; s0 += q0; s1 += q1; s2 += q2;
        }
        ga += wv::swap_half(ga); g0 += wv::swap_half(g0); g1 += wv::swap_half(g1); g2 += wv::swap_half(g2);
        sa += wv::swap_half(sa); s0 += wv::swap_half(s0); s1 += wv::swap_half(s1); s2 += wv::swap_half(s2);
        if (hi == 0) {
            Gv[I::W_A + p31] += ga;
            Gv[I::W_OC + p31] += g0;
            Gv[I::W_OC + H + p31] += g1;
            Gv[I::W_OC + 2 * H + p31] += g2;
            if (p31 == 0) {
                Gv[I::B_A] += sa;
                Gv[I::B_OC + 0] += s0;
                Gv[I::B_OC + 1] += s1;
                Gv[I::B_OC + 2] += s2;
            }
        }
        // d hc = W_oc^T d rawc, through the ReLU
        float dcp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = phi(r, hi);
            const float v = SM[I::W_OC + j] * d_c0 + SM[I::W_OC + H + j] * d_c1 + SM[I::W_OC + 2 * H + j] * d_c2;
            dcp[r] = wv::opaque(hc[r]) > 0.0f ? v : 0.0f;
        }
        split_planes<16, NPB>(dcp, dch, dcm, dcl);
    }
    // ---- 13 units, one 32x32 weight-gradient block each.  Per unit two matrix chains, each with VALU work interleaved by
    // hand (one wave per SIMD issues in order):
    //   A: d-prop chain  W^T . delta          next to  finishing the block staged two units ago + staging the last block
    //   B: weight-gradient chain  delta^T . x  next to  the VALU work on A's result (ReLU mask + split of the next delta,
    //                                                   or the encoding's d(projection) sums)
    // and the operands are requested ONE UNIT AHEAD: transposed weight columns (wA / wB), the F-form of the next input block
    // (xA / xB; the tile is refilled as soon as its previous content has been read); a new delta is transposed behind B. ----
    f32x16 acc2, accb;
    FinState fs;
    unsigned wA[24], wB[24], xA[24], xB[24];
    unsigned d3h[8], d3m[8], d3l[8], d2h[8], d2m[8], d2l[8], d1h[8], d1m[8], d1l[8], dF3[24], dF1[24];
    const auto nothing = [](int) {};
#define VS_FIN(KIND, K, QI, STG, OW, OB, BLK, NC)                                                                             \
    [&](int i) {                                                                                                              \
        if (i == 0) fin_chunk<KIND, K, MULTI>(0, fs, qacc[QI], STG, OW, OB, BLK, NC, wave, p31, hi);                          \
        if (i == NA - 3) fin_chunk<KIND, K, MULTI>(1, fs, qacc[QI], STG, OW, OB, BLK, NC, wave, p31, hi);                     \
    }
    const SelOps SEL = sel_ops(p31, hi);
    wt_get<I::PIT_C, NPW>(wA, W + I::O_C, 0, TL);
    toF_mm3<4, NPB>(dF, dch, dcm, dcl, SEL);                                 // F(d hc): the delta of units 0..2
    toF_mm3<4, NPB>(xA, h4h, h4m, h4l, SEL);                                 // F(h4)
    wt_get<I::PIT_C, NPW>(wB, W + I::O_C, 2, TL);
    // unit 0: colour layer x h4;  d h4 = W_a d raw + W_c[:, :H]^T d hc
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = SM[I::W_A + phi(r, hi)] * d_raw;
    mm_dprop_il<DM>(acc2, wA, dch, dcm, dcl, nothing);
    toF_mm3<4, NPB>(xB, e2h, e2m, e2l, SEL);                                 // F(second-group slots 0..15)
    zero_acc(acc);
    mm_dw_il<false, B6>(acc, accb, dF, xA, [&](int i) {
#pragma unroll
        for (int j = i * 8 / NDW; j < (i + 1) * 8 / NDW; ++j) mask_split_pair<NPB>(j, acc2, h4h, d4h, d4m, d4l, acc[0]);
    });
    wt_get<I::PIT_C, NPW>(wA, W + I::O_C, 4, TL);
    VS_BWD_BARRIER();
    // unit 1: x = second-group slots 0..15
    zero_acc(acc2);
    mm_dprop_il<DM>(acc2, wB, dch, dcm, dcl, [&](int i) { if (i == NA - 1) stage_put_s(stg0, acc, wave, p31, hi); });       // block 0
    toF_mm3<2, NPB>(xA, e2h + 8, e2m + 8, e2l + 8, SEL);                         // F(second-group slots 16..23): half a block
    wt_get<I::PIT_M, NPW>(wB, W + I::O_M2, 0, TL);
    zero_acc(acc);
    mm_dw_il<false, B6>(acc, accb, dF, xB, [&](int i) {
        if (i < 4) {
#pragma unroll
            for (int r = 4 * i; r < 4 * i + 4; ++r) dproj[r >> 1] += wv::after(acc2[r], acc[0]) * cfac[r >> 1][4 + (r & 1)];     // slot R = r: direction r >> 1, octave 4 + (r & 1)
        }
    });
    VS_BWD_BARRIER();
    // unit 2: x = second-group slots 16..23
    zero_acc(acc2);
    {
        auto fin = VS_FIN(0, H + kEmb2, 0, stg0, out + F::W_C, nullptr, 0, 32);
        mm_dprop_il<DM>(acc2, wA, dch, dcm, dcl, [&](int i) { fin(i); if (i == NA - 1) stage_put_s(stg1, acc, wave, p31, hi); });   // block 1
    }
    toF_mm3<4, NPB>(xB, h3h, h3m, h3l, SEL);                                 // F(h3)
    wt_get<I::PIT_CAT, NPW>(wA, W + I::O_CAT, 0, TL);
    zero_acc(acc);
    mm_dw_il<false, B6>(acc, accb, dF, xA, [&](int i) {
        if (i < 3) {
#pragma unroll
            for (int r = 2 * i; r < 2 * i + 2; ++r) dproj[8 + (r >> 1)] += wv::after(acc2[r], acc[0]) * cfac[8 + (r >> 1)][4 + (r & 1)];   // slots 16..21: directions 8..10
        }
    });
    toF_mm3<4, NPB>(dF, d4h, d4m, d4l, SEL);                                 // F(d4) (F(d hc) was the delta of units 0..2)
    VS_BWD_BARRIER();
    VS_MARK(7);
    // unit 3: mid2, delta = d4, x = h3
    zero_acc(acc2);
    {
        auto fin = VS_FIN(2, H + kEmb2, 1, stg1, out + F::W_C + H, out + F::B_C, 0, kEmb2);
        mm_dprop_il<DM>(acc2, wB, d4h, d4m, d4l, [&](int i) { fin(i); if (i == NA - 1) stage_put_s(stg0, acc, wave, p31, hi); });   // block 2
    }
    toF_mm3<4, NPB>(xA, h2h, h2m, h2l, SEL);                                 // F(h2)
    wt_get<I::PIT_M, NPW>(wB, W + I::O_M1, 0, TL);
    zero_acc(acc);
    zero_acc(accb);
    mm_dw_il<true, B6>(acc, accb, dF, xB, [&](int i) {
#pragma unroll
        for (int j = i * 8 / NDB; j < (i + 1) * 8 / NDB; ++j) mask_split_pair<NPB>(j, acc2, h3h, d3h, d3m, d3l, i < NDW ? acc[0] : accb[0]);
    });
    if (p31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Gv[I::B_M2 + phi(r, hi)] += accb[r];
    }
    toF_mm3<4, NPB>(dF3, d3h, d3m, d3l, SEL);                                // F(d3): kept for the three first-group blocks
    VS_BWD_BARRIER();
    VS_MARK(8);
    // unit 4: cat_layer, delta = d3, x = h2
    zero_acc(acc2);
    {
        auto fin = VS_FIN(2, H + kEmb2, 2, stg0, out + F::W_C + H, out + F::B_C, 1, kEmb2);
        mm_dprop_il<DM>(acc2, wA, d3h, d3m, d3l, [&](int i) { fin(i); if (i == NA - 1) stage_put_s(stg1, acc, wave, p31, hi); });   // block 3
    }
    toF_mm3<4, NPB>(xB, h1h, h1m, h1l, SEL);                                 // F(h1)
    wt_get<I::PIT_CAT, NPW>(wA, W + I::O_CAT, 2, TL);
    zero_acc(acc);
    mm_dw_il<false, B6>(acc, accb, dF3, xA, [&](int i) {
#pragma unroll
        for (int j = i * 8 / NDW; j < (i + 1) * 8 / NDW; ++j) mask_split_pair<NPB>(j, acc2, h2h, d2h, d2m, d2l, acc[0]);
    });
    toF_mm3<4, NPB>(dF, d2h, d2m, d2l, SEL);                                 // F(d2)
    VS_BWD_BARRIER();
    VS_MARK(9);
    // unit 5: mid1, delta = d2, x = h1
    zero_acc(acc2);
    {
        auto fin = VS_FIN(0, H, 3, stg1, out + F::W_M2, nullptr, 0, 32);
        mm_dprop_il<DM>(acc2, wB, d2h, d2m, d2l, [&](int i) { fin(i); if (i == NA - 1) stage_put_s(stg0, acc, wave, p31, hi); });   // block 4
    }
    toF_mm3<4, NPB>(xA, e1h, e1m, e1l, SEL);                                 // F(first-group block 0)
    wt_get<I::PIT_IN, NPW>(wB, W + I::O_IN, 0, TL);
    zero_acc(acc);
    zero_acc(accb);
    mm_dw_il<true, B6>(acc, accb, dF, xB, [&](int i) {
#pragma unroll
        for (int j = i * 8 / NDB; j < (i + 1) * 8 / NDB; ++j) mask_split_pair<NPB>(j, acc2, h1h, d1h, d1m, d1l, i < NDW ? acc[0] : accb[0]);
    });
    if (p31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Gv[I::B_M1 + phi(r, hi)] += accb[r];
    }
    toF_mm3<4, NPB>(dF1, d1h, d1m, d1l, SEL);                                // F(d1): kept for the three in_layer blocks
    VS_BWD_BARRIER();
    VS_MARK(10);
    // units 6..11: the three first-group blocks feed cat_layer (delta d3, weights wA) and in_layer (delta d1, weights wB);
    // the block's F-form alternates between xA and xB
    f32x16 de;
#pragma unroll
    for (int blk = 0; blk < 3; ++blk) {
        unsigned (&xc)[24] = (blk & 1) ? xB : xA;
        unsigned (&xn)[24] = (blk & 1) ? xA : xB;
        // cat_layer x block: finishes block 4 (blk 0) or the cat block of the previous round; stages the last in / mid1 block
        zero_acc(de);
        if (blk == 0) {
            auto fin = VS_FIN(0, H + kEmb1, 4, stg0, out + F::W_CAT, nullptr, 0, 32);
            mm_dprop_il<DM>(de, wA, d3h, d3m, d3l, [&](int i) { fin(i); if (i == NA - 1) stage_put_s(stg1, acc, wave, p31, hi); });   // block 5
        } else {
            auto fin = VS_FIN(1, H + kEmb1, 6 + blk - 1, stg0, out + F::W_CAT + H, out + F::B_CAT, blk - 1, kEmb1);
            mm_dprop_il<DM>(de, wA, d3h, d3m, d3l, [&](int i) { fin(i); if (i == NA - 1) stage_put_s(stg1, acc, wave, p31, hi); });   // block 9 + blk - 1
        }
        if (blk < 2) {
            wt_get<I::PIT_CAT, NPW>(wA, W + I::O_CAT, 2 + 2 * (blk + 1), TL);
            toF_mm3<4, NPB>(xn, e1h + 8 * (blk + 1), e1m + 8 * (blk + 1), e1l + 8 * (blk + 1), SEL);
        }
        zero_acc(acc);
        mm_dw_il<false, B6>(acc, accb, dF3, xc, nothing);
        VS_BWD_BARRIER();
        // in_layer x block: finishes the mid1 block (blk 0) or the previous round's in block; stages this round's cat block
        if (blk == 0) {
            auto fin = VS_FIN(0, H, 5, stg1, out + F::W_M1, nullptr, 0, 32);
            mm_dprop_il<DM>(de, wB, d1h, d1m, d1l, [&](int i) { fin(i); if (i == NA - 1) stage_put_s(stg0, acc, wave, p31, hi); });   // block 6 + blk
        } else {
            auto fin = VS_FIN(1, kEmb1, 9 + blk - 1, stg1, out + F::W_IN, out + F::B_IN, blk - 1, kEmb1);
            mm_dprop_il<DM>(de, wB, d1h, d1m, d1l, [&](int i) { fin(i); if (i == NA - 1) stage_put_s(stg0, acc, wave, p31, hi); });   // block 6 + blk
        }
        if (blk < 2) wt_get<I::PIT_IN, NPW>(wB, W + I::O_IN, 2 * (blk + 1), TL);
        zero_acc(acc);
        mm_dw_il<false, B6>(acc, accb, dF1, xc, [&](int i) {
            // d(first-group slot R = 16 blk + r) -> direction R >> 2, octave R & 3 (slots 44..47: xyz / one / padding: no gradient)
            if (i < 4) {
#pragma unroll
                for (int r = 4 * i; r < 4 * i + 4; ++r) {
                    const int R = 16 * blk + r;
                    if (R < 44) dproj[R >> 2] += wv::after(de[r], acc[0]) * cfac[R >> 2][R & 3];
                }
            }
        });
        VS_BWD_BARRIER();
    }
    VS_MARK(11);
    VS_MARK(12);
    // ---- unit 12: B_layer.weight gradient dB[d][j] = sum_points dproj[d] * t[j] (embedding.py:84) - 21 x 3 numbers.  Round 6: float32
    //      VALU products + a DPP reduction over the wave's 32 points into the wave's private accumulators (summed over the waves
    //      with the other small vectors behind the pass loop) instead of a 32 x 32 matrix block through split / transpose /
    //      products / staging and two workgroup barriers (3.9 k of the pass's 50 k clocks, profiles/round6a_phase_clocks_*): one
    //      barrier, no matrix instruction, and the exact float32 products of step_main_h32 ----
    {
        // finish the last cat block (staged in stg0), stage the last in block
        fin_chunk<1, H + kEmb1, MULTI>(0, fs, qacc[8], stg0, out + F::W_CAT + H, out + F::B_CAT, 2, kEmb1, wave, p31, hi);
        stage_put_s(stg1, acc, wave, p31, hi);                      // block 11
        fin_chunk<1, H + kEmb1, MULTI>(1, fs, qacc[8], stg0, out + F::W_CAT + H, out + F::B_CAT, 2, kEmb1, wave, p31, hi);
        float keep[3] = {0.0f, 0.0f, 0.0f};                         // value v = 3 i + j lives on lane (lane & 15) == (v & 15) of the half's second row, slot v >> 4
#pragma unroll
        for (int i = 0; i < 11; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int v = 3 * i + j;
                const float sum = wv::half_sum32_hi_row(dproj[i] * t[j]);      // this lane's direction: hi ? 11 + i : i (the eleventh of the hi = 1 half is a dummy: dproj = 0)
                keep[v >> 4] = (lane & 15) == (v & 15) ? sum : keep[v >> 4];
            }
        }
        if (lane & 16) {
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                const int v = 16 * sl + (lane & 15);
                if (v < 33) Gv[I::G_PEB + 36 * hi + v] += keep[sl];
            }
        }
        VS_BWD_BARRIER();
        finish_block_s<1, kEmb1, MULTI>(qacc[11], stg1, out + F::W_IN, out + F::B_IN, 2, kEmb1, wave, p31, hi);
    }
#undef VS_FIN
    VS_MARK(13);
    }   // BWD
    if (!MULTI) break;      // one pass per workgroup: no back edge
    }   // pass loop
    __syncthreads();
    VS_MARK(14);
    if (tid == 0) {
        float* pl = a.part_loss + (obj * a.NW + wgo) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            pl[k] = (loss_cells[k] + loss_cells[4 + k]) + (loss_cells[8 + k] + loss_cells[12 + k]);
        pl[3] = 0.0f;
    }
    if (!BWD) return;

    // ---- write this workgroup's partial gradients in the natural flat order ----
    if (MULTI) {
        int col; bool bias;
        col_target<0>(0, p31, col, bias);
        store_quarter_map<H + kEmb2>(out + F::W_C, nullptr, qacc[0], col, false, 32, wave, hi);
        store_quarter_map<H>(out + F::W_M2, nullptr, qacc[3], col, false, 32, wave, hi);
        store_quarter_map<H + kEmb1>(out + F::W_CAT, nullptr, qacc[4], col, false, 32, wave, hi);
        store_quarter_map<H>(out + F::W_M1, nullptr, qacc[5], col, false, 32, wave, hi);
        col_target<2>(0, p31, col, bias);
        store_quarter_map<H + kEmb2>(out + F::W_C + H, out + F::B_C, qacc[1], col, bias, kEmb2, wave, hi);
        col_target<2>(1, p31, col, bias);
        store_quarter_map<H + kEmb2>(out + F::W_C + H, out + F::B_C, qacc[2], col, bias, kEmb2, wave, hi);
#pragma unroll
        for (int blk = 0; blk < 3; ++blk) {
            col_target<1>(blk, p31, col, bias);
            store_quarter_map<H + kEmb1>(out + F::W_CAT + H, out + F::B_CAT, qacc[6 + blk], col, bias, kEmb1, wave, hi);
            store_quarter_map<kEmb1>(out + F::W_IN, out + F::B_IN, qacc[9 + blk], col, bias, kEmb1, wave, hi);
        }
    }
    // small vectors: sum of the four waves' private accumulators (biases of mid1 / mid2, the heads, B_layer.weight)
    for (int sv = tid; sv < I::VEC_N; sv += kWG) {
        const float* v = reinterpret_cast<const float*>(lds + I::VEC) + sv;
        const float g = (v[0] + v[I::VEC_N]) + (v[2 * I::VEC_N] + v[3 * I::VEC_N]);
        int o = -1;
        if (sv >= I::G_PEB) {                                        // [half][3 i + j]: direction half ? 11 + i : i
            const int x = sv - I::G_PEB, hf = x >= 36 ? 1 : 0, vv = x - 36 * hf, i = vv / 3;
            if (vv < 33 && i < (hf ? 10 : 11)) o = F::PE_B + 3 * (hf ? 11 + i : i) + (vv - 3 * i);
        } else
        if (sv < I::B_M2) o = F::B_M1 + (sv - I::B_M1);
        else if (sv < I::W_A) o = F::B_M2 + (sv - I::B_M2);
        else if (sv < I::W_OC) o = F::W_A + (sv - I::W_A);
        else if (sv < I::B_A) o = F::W_OC + (sv - I::W_OC);
        else if (sv == I::B_A) o = F::B_A;
        else if (sv >= I::B_OC && sv < I::B_OC + 3) o = F::B_OC + (sv - I::B_OC);
        if (o >= 0) out[o] = g;
    }
    VS_MARK(15);
#undef VS_MARK
}

template <bool BWD, bool MULTI, bool STAMPS, bool W3, bool B6 = false>
__global__ __launch_bounds__(kWG, 1) void step_main_s32(const StepArgs a) {
    step_main_s32_body<BWD, MULTI, STAMPS, W3, B6>(a);
}

}  // namespace vk
