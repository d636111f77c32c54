// wsplit_kernels.h - the fused training step for a WIDE field (hidden = 128: the background model of train.py:308-316) on the
// bf16 matrix pipe with split operands (gfx950 / CDNA4).  Same numerics scheme as split_kernels.h (float32 = hi + mid + lo
// bfloat16 planes; six products forward, three backward), different shape - at hidden 128 the weights (3 x 94 340 bf16) do
// not fit LDS and the step is bound by OPERAND DELIVERY, not by the matrix pipe: every 32x32x16 instruction eats 2 KiB of
// operands in 32 clocks (64 B/clk per SIMD), the CU delivers 64 B/clk from L1 and 128 B/clk from LDS.  So:
//
//   * a ROUND = two 32-point tiles belongs to the WORKGROUP; wave w owns output block w (32 features) of every layer and runs
//     it for both tiles at once: one weight operand (read from the object's image in global memory / L2 as ready-made 1 KiB
//     matrix operand chunks [64 lanes][8 bf16]) feeds two accumulators - per 12 matrix instructions a wave reads 3 KiB from L1 and 6 KiB from LDS (8 and 16 B/clk per SIMD);
//   * layer inputs travel between the waves through LDS as lane-contiguous P-form operand images (all four waves hold the
//     same points on the same lanes, so what one wave stores is directly another wave's B operand: ds_write_b128 /
//     ds_read_b128, no transposes); the encoding is evaluated once per round, split by (tile, direction half) over the waves;
//   * one workgroup per CU, one wave per SIMD: nothing hides a load but the wave's own matrix instructions, so every operand
//     stream runs through explicit prefetch rings (weights two steps ahead, LDS operands one) pinned by scheduling fences;
//     the hi / mid planes of the five activations (ReLU masks and weight-gradient operands of the backward pass) wait in
//     an L2-resident scratch area of the workgroup;
//   * backward, per layer: the wave's delta block -> P-form image (d-prop operand of everybody) + F-form registers (its own
//     weight-gradient operand); its block of the layer input -> F-form image in LDS (weight-gradient operand of everybody);
//     weight-gradient blocks accumulate over the round's two tiles in registers and go straight to the workgroup's row of partial
//     gradients, each into its own 4 KiB slot as it stands (RowWs; stored by the first round, added by later ones); W^T comes from a
//     second, transposed image.
#pragma once
#include "split_kernels.h"
#ifndef VS_ABLW                  /* measurement builds only, see "device helpers of step_main_ws" */
#define VS_ABLW 0
#endif

namespace vk {

// Cos factors of the encoding (cos(2^f a) pi 2^f: what the backward multiplies d(sin) by) wait in the workgroup's scratch; step_main_wp: per tile and
// direction slot 0..10 as [64 lanes][4 floats] (octaves 0..3 = the first encoding group) followed by [64 lanes][2 floats] (octaves 4, 5 =
// the second group): a lane's factors of one direction are ONE 16-byte and ONE 8-byte access, and a wave's access covers 1 KiB / 512 B
// without a gap.  Round 6d: as [66][64 lanes] floats they were 6 four-byte stores per direction and 16 four-byte loads per encoding
// block and tile; as [lane][8 floats] (32-byte lane pitch, half of every line touched) the kernels got SLOWER (background 70.3 -> 74.5 us,
// hidden 64 +12 %: profiles/round6d_cos_factor_layout_ab.jsonl) - what the memory system wants is whole lines per wave access.  Kept for
// step_main_wp (a rank's share of configs[4] 167.4 -> 164.5 us); step_main_ws stays with [66][64 lanes] (same tile size), see enc_fetch.
constexpr int kCfDir = 64 * 6;                                            // floats per direction slot
constexpr int kCfTile = 11 * kCfDir;                                      // floats per tile
__device__ __forceinline__ void cf_store(float* cf_tile, int i, int lane, const float (&v)[6]) {
    float* q = cf_tile + i * kCfDir;
    *reinterpret_cast<wv::f32x4*>(q + lane * 4) = wv::f32x4{v[0], v[1], v[2], v[3]};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<f32x2*>(q + 256 + lane * 2) = f32x2{v[4], v[5]};
}
// the 16 factors a lane needs for encoding block blk of group 1 (R = 16 blk + r <-> direction R >> 2, octave R & 3; R >= 44: none) or
// group 2 (direction R >> 1, octave 4 + (R & 1); R >= 22: none)
__device__ __forceinline__ void cf_load(float (&cf)[16], const float* cf_tile, int group, int blk, int lane) {
    if (group == 1) {
        const float* q = cf_tile + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * blk + j;
            wv::f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (i < 11) v = *reinterpret_cast<const wv::f32x4*>(q + i * kCfDir);
            cf[4 * j] = v[0]; cf[4 * j + 1] = v[1]; cf[4 * j + 2] = v[2]; cf[4 * j + 3] = v[3];
        }
    } else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const float* q = cf_tile + 256 + lane * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = 8 * blk + j;
            f32x2 v = {0.0f, 0.0f};
            if (i < 11) v = *reinterpret_cast<const f32x2*>(q + i * kCfDir);
            cf[2 * j] = v[0]; cf[2 * j + 1] = v[1];
        }
    }
}

template <int NB>
struct ImgWs {
    static constexpr int H = 32 * NB;
    static constexpr int JS = 2 * NB;                                    // 16-deep steps over a layer's H outputs / hidden inputs
    // forward image W: chunks of 1 KiB per plane, index = base(layer) + ob * steps(layer) + s; planes hi, mid, lo interleaved
    static constexpr int KS_IN = 6, KS_M = JS, KS_CAT = JS + 6, KS_C = JS + 3;
    static constexpr int CW_IN = 0, CW_M1 = CW_IN + NB * KS_IN, CW_CAT = CW_M1 + NB * KS_M, CW_M2 = CW_CAT + NB * KS_CAT,
                         CW_C = CW_M2 + NB * KS_M, CW_N = CW_C + NB * KS_C;
    // transposed image W^T: index = base(layer) + kb * JS + s'; kb = input block (hidden blocks first, then encoding blocks)
    static constexpr int IB_IN = 3, IB_M = NB, IB_CAT = NB + 3, IB_C = NB + 2;
    static constexpr int CT_IN = 0, CT_M1 = CT_IN + IB_IN * JS, CT_CAT = CT_M1 + IB_M * JS, CT_M2 = CT_CAT + IB_CAT * JS,
                         CT_C = CT_M2 + IB_M * JS, CT_N = CT_C + IB_C * JS;
    static constexpr long long W_BYTES = (long long)CW_N * 3 * 1024, WT_OFF = W_BYTES, WT_BYTES = (long long)CT_N * 2 * 1024;
    static constexpr long long SMALL_OFF = WT_OFF + WT_BYTES;
    // small float32 vectors
    static constexpr int B_M1 = 0, B_M2 = H, W_A = 2 * H, W_OC = 3 * H, B_A = 6 * H, B_OC = 6 * H + 4, PE_B = 6 * H + 8, SMALL_N = 6 * H + 72;
    static constexpr long long BYTES = (SMALL_OFF + SMALL_N * 4 + 4095) / 4096 * 4096;
    static constexpr int W_ELEMS = CW_N * 512, WT_ELEMS = CT_N * 512;     // bf16 elements per plane
    // scratch per workgroup (global memory, L2-resident): cos factors of the encoding [tile][11 directions][384] (cf_store); then the hi / mid planes of
    // the five activations of the wave's block [layer][tile][plane][step][256 threads][16 B] (ReLU masks and weight-gradient
    // operands of the backward pass: 160 registers a thread cannot afford next to the operand prefetch rings)
    static constexpr int CF_BYTES = 2 * kCfTile * 4;
    static constexpr int ACTS_OFF = (CF_BYTES + 4095) / 4096 * 4096;
    static constexpr int WG_SCRATCH = ACTS_OFF + 5 * 2 * 2 * 2 * 4096;
    // ---- LDS map (bytes) ----
    static constexpr int XCH = 3072;                                       // one 16-deep step of a P-form operand image: 3 planes x 1 KiB
    static constexpr int DCH = 2048;                                       // ... of a delta image: 2 planes
    static constexpr int ACT = 0, ACT_ST = NB * 2 * XCH, ACT_BYTES = 2 * ACT_ST;      // forward: layer input images [tile][block][step][plane]
    static constexpr int EF = 0, EF_ST = 5 * 4096;                         // backward (over ACT): F-form images of the 5 encoding blocks [tile][5]
    static constexpr int R0_BYTES = ACT_BYTES > 2 * EF_ST ? ACT_BYTES : 2 * EF_ST;
    static constexpr int EIM = R0_BYTES, E_ST = 9 * XCH, E2_OFF = 6 * XCH;             // forward: encoding images [tile][6 + 3 steps][plane]
    static constexpr int DLT = EIM, DLT_ST = NB * 2 * DCH;                 // backward (over EIM): delta images [tile][block][step][plane]
    static constexpr int XF = DLT + 2 * DLT_ST, XF_ST = NB * 4096;         // backward: F-form images of the layer input [tile][block]
    static constexpr int R1_BYTES = 2 * E_ST > 2 * DLT_ST + 2 * XF_ST ? 2 * E_ST : 2 * DLT_ST + 2 * XF_ST;
    static constexpr int SCRT = EIM + R1_BYTES;                            // one transpose tile per wave
    static constexpr int HP = SCRT + kWaves * Img32s::TILE;                // head partial sums [wave][tile][32 points][4]
    static constexpr int CBO = HP + kWaves * 2 * 32 * 4 * 4;               // composite buffer [64 points][8]
    static constexpr int LOSS = CBO + 64 * 8 * 4;
    static constexpr int LDS_BYTES = LOSS + kWaves * 4 * 4;
    static constexpr int kPts = 64;                                        // sample points per round
};
static_assert(ImgWs<4>::LDS_BYTES <= 160 * 1024 && ImgWs<2>::LDS_BYTES <= 160 * 1024, "LDS budget");

// LDS and scratch maps of step_main_ws<NB, ., ., ., NT>.  NT = 1, 2: the two-tile map of ImgWs<NB> (single-tile rounds leave the
// second tile's slots unused).  NT = 3 (hidden 128 only): three tiles = 96 points per round, for batches whose two-tile rounds
// outnumber the compute units (the 1200-ray background batch of ONE GPU: 300 two-tile rounds on 256 compute units = two rounds
// for the busiest workgroups and a read-modify-write of the 377 KB gradient row; 200 three-tile rounds = one round each).  The
// forward images of three tiles (72 + 81 KB) leave 7 KB: the heads' partial sums overlay the layer-input images (one more
// barrier behind color_linear), and in the backward pass the F-form images of the SECOND encoding group (used once, by
// color_linear's weight gradients) wait in the workgroup's scratch instead of LDS.
// Hidden 256 (NB = 8; round 3): EIGHT waves, one output block each, two per SIMD; single-tile rounds only (the two-tile images of
// eight blocks would need 200 KB), one-tile maps.
template <int NB> __host__ __device__ constexpr int ws_waves() { return NB > 4 ? 8 : 4; }
template <int NB, int NT>
struct LdsWs {
    using I = ImgWs<NB>;
    static_assert(NT >= 1 && NT <= 3, "tiles per round");
    static_assert(NT < 3 || NB == 4, "three-tile rounds: hidden 128");
    static_assert(NB <= 4 || NT == 1, "hidden 256: single-tile rounds");
    static constexpr int NWV = ws_waves<NB>(), NTH = 64 * NWV;             // waves / threads of a workgroup
    static constexpr int TT = NB > 4 ? 1 : NT < 2 ? 2 : NT;                // tile slots of the maps
    static constexpr bool WIDE3 = NT == 3;
    static constexpr bool HP_ALIAS = WIDE3, EF2_GLOBAL = WIDE3;
    static constexpr int XCH = I::XCH, DCH = I::DCH;
    static constexpr int ACT_ST = I::ACT_ST, E_ST = I::E_ST, E2_OFF = I::E2_OFF, DLT_ST = I::DLT_ST, XF_ST = I::XF_ST;
    static constexpr int EF_BLOCKS = EF2_GLOBAL ? 3 : 5, EF_ST = EF_BLOCKS * 4096;  // encoding F-form blocks per tile held in LDS
    static constexpr int TILES_BYTES = NWV * Img32s::TILE;                // one transpose tile per wave
    static constexpr int HP_BYTES = NWV * TT * 32 * 4 * 4, CB_BYTES = 32 * TT * 8 * 4;
    // two regions that the forward and the backward images share (all forms but three-tile rounds)
    static constexpr int R0 = TT * ACT_ST > TT * EF_ST ? TT * ACT_ST : TT * EF_ST;
    static constexpr int R1 = TT * E_ST > TT * (DLT_ST + XF_ST) ? TT * E_ST : TT * (DLT_ST + XF_ST);
    // forward
    static constexpr int ACT = 0;
    static constexpr int EIM = WIDE3 ? TT * ACT_ST : R0;
    // backward
    static constexpr int EF = 0;
    static constexpr int SCRT = WIDE3 ? EF + TT * EF_ST : EIM + R1;
    static constexpr int DLT = WIDE3 ? SCRT + TILES_BYTES : EIM;
    static constexpr int XF = DLT + TT * DLT_ST;
    static constexpr int HP = WIDE3 ? ACT : SCRT + TILES_BYTES;
    static constexpr int CBO = WIDE3 ? EIM + TT * E_ST : HP + HP_BYTES;
    static constexpr int LOSS = CBO + CB_BYTES;
    static constexpr int LDS_BYTES = LOSS + NWV * 4 * 4;
    static_assert(!WIDE3 || XF + TT * XF_ST <= CBO, "backward images end in front of the composite buffer");
    static_assert(WIDE3 || NB > 4 || (EIM == I::EIM && SCRT == I::SCRT && LOSS == I::LOSS && LDS_BYTES == I::LDS_BYTES), "NT <= 2: the map of ImgWs");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static constexpr int kPts = 32 * TT;                                   // sample-point slots of a round's composite buffer
    // scratch per workgroup: cos factors [tile][11][384] (cf_store); activation planes [layer][tile][plane][step][threads][16 B];
    // (three-tile rounds) F-form images of the second encoding group [tile][2][4 KiB]
    static constexpr int CHUNK = NTH * 16;                                 // one 16-deep step of an activation plane, all threads
    static constexpr int CF_BYTES = TT * kCfTile * 4;
    static constexpr int ACTS_OFF = (CF_BYTES + 4095) / 4096 * 4096;
    static constexpr int EF2_OFF = ACTS_OFF + 5 * TT * 2 * 2 * CHUNK;
    static constexpr int WG_SCRATCH = EF2_OFF + (EF2_GLOBAL ? TT * 2 * 4096 : 0);
    static_assert(WIDE3 || NB > 4 || (ACTS_OFF == I::ACTS_OFF && WG_SCRATCH == I::WG_SCRATCH), "NT <= 2: the scratch map of ImgWs");
};
constexpr int kWsScratchMax = LdsWs<4, 3>::WG_SCRATCH;                  // the largest of the forms (host-side sizing)
static_assert(kWsScratchMax >= ImgWs<4>::WG_SCRATCH && kWsScratchMax >= ImgWs<2>::WG_SCRATCH && kWsScratchMax >= LdsWs<8, 1>::WG_SCRATCH, "scratch sizing");

struct WsArgs {
    StepArgs s;
    char* scratch;                 // [workgroups][WG_SCRATCH]
    int* tab_wt;                   // [PP] flat parameter -> element of the W^T planes (or -1); s.img_tab is the W / small-vector table
};

// ---- which parameter sits where -------------------------------------------------------------------------------------
// element x of a W plane -> (tensor t, offset o); false = zero padding
template <int NB>
__host__ __device__ inline bool ws_w_source(int x, int& t, int& o) {
    using I = ImgWs<NB>;
    constexpr int H = I::H;
    const int chunk = x >> 9, lane = (x >> 3) & 63, tt = x & 7, j = lane & 31, hi = lane >> 5;
    int base, ks, kind;
    if (chunk < I::CW_M1) { base = I::CW_IN; ks = I::KS_IN; kind = 0; }
    else if (chunk < I::CW_CAT) { base = I::CW_M1; ks = I::KS_M; kind = 1; }
    else if (chunk < I::CW_M2) { base = I::CW_CAT; ks = I::KS_CAT; kind = 2; }
    else if (chunk < I::CW_C) { base = I::CW_M2; ks = I::KS_M; kind = 3; }
    else { base = I::CW_C; ks = I::KS_C; kind = 4; }
    const int ob = (chunk - base) / ks, s = (chunk - base) - ob * ks, row = 32 * ob + j;
    const int k = hidden_k(s, hi, tt);
    switch (kind) {
        case 0: {
            const int c = e1_slot(8 * s + tt, hi);
            if (c == kSlotPad) return false;
            if (c == kSlotOne) { t = 1; o = row; } else { t = 0; o = row * kEmb1 + c; }
            return true;
        }
        case 1: t = 2; o = row * H + k; return true;
        case 2: {
            if (s < I::JS) { t = 4; o = row * (H + kEmb1) + k; return true; }
            const int c = e1_slot(8 * (s - I::JS) + tt, hi);
            if (c == kSlotPad) return false;
            if (c == kSlotOne) { t = 5; o = row; } else { t = 4; o = row * (H + kEmb1) + H + c; }
            return true;
        }
        case 3: t = 6; o = row * H + k; return true;
        default: {
            if (s < I::JS) { t = 10; o = row * (H + kEmb2) + k; return true; }
            const int c = e2_slot(8 * (s - I::JS) + tt, hi);
            if (c == kSlotPad) return false;
            if (c == kSlotOne) { t = 11; o = row; } else { t = 10; o = row * (H + kEmb2) + H + c; }
            return true;
        }
    }
}
// element x of a W^T plane -> (tensor, offset); lane = input column inside its 32-block, elements = 8 output rows
template <int NB>
__host__ __device__ inline bool ws_wt_source(int x, int& t, int& o) {
    using I = ImgWs<NB>;
    constexpr int H = I::H;
    const int chunk = x >> 9, lane = (x >> 3) & 63, tt = x & 7, kl = lane & 31, hi = lane >> 5;
    int base, kind;
    if (chunk < I::CT_M1) { base = I::CT_IN; kind = 0; }
    else if (chunk < I::CT_CAT) { base = I::CT_M1; kind = 1; }
    else if (chunk < I::CT_M2) { base = I::CT_CAT; kind = 2; }
    else if (chunk < I::CT_C) { base = I::CT_M2; kind = 3; }
    else { base = I::CT_C; kind = 4; }
    const int kb = (chunk - base) / I::JS, sp = (chunk - base) - kb * I::JS;
    const int j = hidden_k(sp, hi, tt);                                  // output row
    const int hs = (kl >> 2) & 1, r = (kl & 3) + 4 * (kl >> 3);          // encoding blocks: lane kl = phi(r, hs)
    switch (kind) {
        case 0: {
            const int c = e1_slot(16 * kb + r, hs);
            if (c < 0) return false;
            t = 0; o = j * kEmb1 + c; return true;
        }
        case 1: t = 2; o = j * H + 32 * kb + kl; return true;
        case 2: {
            if (kb < NB) { t = 4; o = j * (H + kEmb1) + 32 * kb + kl; return true; }
            const int c = e1_slot(16 * (kb - NB) + r, hs);
            if (c < 0) return false;
            t = 4; o = j * (H + kEmb1) + H + c; return true;
        }
        case 3: t = 6; o = j * H + 32 * kb + kl; return true;
        default: {
            if (kb < NB) { t = 10; o = j * (H + kEmb2) + 32 * kb + kl; return true; }
            const int R = 16 * (kb - NB) + r;
            const int c = R < 24 ? e2_slot(R, hs) : kSlotPad;
            if (c < 0) return false;
            t = 10; o = j * (H + kEmb2) + H + c; return true;
        }
    }
}
template <int NB>
__host__ __device__ inline bool ws_small_source(int i, int& t, int& o) {
    using I = ImgWs<NB>;
    if (i < I::B_M2) { t = 3; o = i - I::B_M1; return true; }
    if (i < I::W_A) { t = 7; o = i - I::B_M2; return true; }
    if (i < I::W_OC) { t = 8; o = i - I::W_A; return true; }
    if (i < I::B_A) { t = 12; o = i - I::W_OC; return true; }
    if (i < I::B_OC) { t = 9; o = i - I::B_A; return o < 1; }
    if (i < I::PE_B) { t = 13; o = i - I::B_OC; return o < 3; }
    t = 14; o = i - I::PE_B;
    return o < 63;
}

// ---- the workgroup's row of partial gradients, BLOCK-NATIVE (round 6) ------------------------------------------------------
// A wave's 32 x 32 weight-gradient block leaves the matrix pipe with 16 values per lane (lane = column, register r <-> row phi(r, hi)).
// Stored in the parameters' flat order that is sixteen 4-byte stores per lane, each a 128-byte run; stored AS IT STANDS - register
// group g of lane l at float g * 256 + l * 4 of the block's 4 KiB slot - it is four 16-byte stores that cover 1 KiB each without a
// gap (measured with a deliberately wrong layout first: step_main_ws<4> 73.1 -> 67.6 us on the background step,
// profiles/round6d_*).  So the row is a sequence of block slots - per output block ob the blocks of color_linear, mid2, cat_layer,
// mid1, in_layer in the order the backward produces them - followed by the small vectors; step_finalize_ws walks the ROW and finds
// every element's parameter through a table (row position -> flat parameter, -1 for the padding columns of the encoding blocks)
// that step_prep_ws writes.  The sums, their order and the update do not change: only where a number waits in between.
#ifndef VK_ROW_T                 /* measurement, 1: the weight-gradient blocks of dw_layer leave TRANSPOSED (dw_mm_pair<., true>: a lane's four
                                    registers = four consecutive parameters, the finalize's update on 16-byte accesses) - slower: background
                                    step 0.0876 -> 0.0889 ms, a rank's share of configs[4] 0.196 -> 0.201 ms (profiles/round6d_rows_transposed_ab.jsonl) */
#define VK_ROW_T 0
#endif
constexpr bool kRowT = VK_ROW_T != 0;
template <int NB>
struct RowWs {
    static constexpr int H = 32 * NB;
    static constexpr int NKB_C = NB + 2, NKB_M2 = NB, NKB_CAT = NB + 3, NKB_M1 = NB, NKB_IN = 3;
    static constexpr int S_C = 0, S_M2 = S_C + NKB_C, S_CAT = S_M2 + NKB_M2, S_M1 = S_CAT + NKB_CAT, S_IN = S_M1 + NKB_M1, PER_OB = S_IN + NKB_IN;
    static constexpr int BLOCKS = NB * PER_OB;
    // the small vectors behind the blocks (floats): mid1 bias, mid2 bias, out_alpha weight, out_color weight [3][H], out_alpha bias, out_color bias, B_layer
    static constexpr int SMALL = BLOCKS * 1024;
    static constexpr int B_M1 = SMALL, B_M2 = B_M1 + H, W_A = B_M2 + H, W_OC = W_A + H, B_A = W_OC + 3 * H, B_OC = B_A + 1, PE_B = B_OC + 3, END = PE_B + 63;
    static constexpr int PR = (END + 63) / 64 * 64;                        // floats per row
    __host__ __device__ static constexpr int slot(int layer0, int ob, int kb) { return (ob * PER_OB + layer0 + kb) * 1024; }
};
__host__ __device__ inline int ws_row_floats(int hidden) { return hidden == 256 ? RowWs<8>::PR : hidden == 128 ? RowWs<4>::PR : RowWs<2>::PR; }
// element r of a row -> (tensor t, offset o); false = padding
template <int NB>
__host__ __device__ inline bool ws_row_source(int r, int& t, int& o) {
    using RW = RowWs<NB>;
    constexpr int H = RW::H;
    if (r >= RW::SMALL) {
        if (r < RW::B_M2) { t = 3; o = r - RW::B_M1; return true; }
        if (r < RW::W_A) { t = 7; o = r - RW::B_M2; return true; }
        if (r < RW::W_OC) { t = 8; o = r - RW::W_A; return true; }
        if (r < RW::B_A) { t = 12; o = r - RW::W_OC; return true; }
        if (r < RW::B_OC) { t = 9; o = 0; return true; }
        if (r < RW::PE_B) { t = 13; o = r - RW::B_OC; return true; }
        if (r < RW::END) { t = 14; o = r - RW::PE_B; return true; }
        return false;
    }
    const int sl = r >> 10, w = r & 1023, g = w >> 8, lane = (w >> 2) & 63, j = w & 3, p31 = lane & 31, hi = lane >> 5;
    const int ob = sl / RW::PER_OB, ls = sl - ob * RW::PER_OB;
    const int br = 8 * g + 4 * hi + j;                                     // the block's row = phi(4 g + j, hi); its column = p31
    const int row = 32 * ob + (kRowT ? p31 : br), bc = kRowT ? br : p31;   // output row of the layer; input column inside block kb
    int tw, tb, K, kb, nh, kind, ncols;                                     // weight / bias tensor, its row length, block, hidden blocks, encoding kind
    if (ls < RW::S_M2) { tw = 10; tb = 11; K = H + kEmb2; kb = ls - RW::S_C; nh = NB; kind = 2; ncols = kEmb2; }
    else if (ls < RW::S_CAT) { tw = 6; tb = -1; K = H; kb = ls - RW::S_M2; nh = NB; kind = 0; ncols = 0; }
    else if (ls < RW::S_M1) { tw = 4; tb = 5; K = H + kEmb1; kb = ls - RW::S_CAT; nh = NB; kind = 1; ncols = kEmb1; }
    else if (ls < RW::S_IN) { tw = 2; tb = -1; K = H; kb = ls - RW::S_M1; nh = NB; kind = 0; ncols = 0; }
    else { tw = 0; tb = 1; K = kEmb1; kb = ls - RW::S_IN; nh = 0; kind = 1; ncols = kEmb1; }
    if (kb < nh) { t = tw; o = row * K + 32 * kb + bc; return true; }
    int col; bool bias;
    if (kind == 1) col_target<1>(kb - nh, bc, col, bias); else col_target<2>(kb - nh, bc, col, bias);
    if (col >= 0 && col < ncols) { t = tw; o = row * K + (K - ncols) + col; return true; }
    if (bias) { t = tb; o = row; return true; }
    return false;
}

// ---- step_prep_ws: mask statistics (blocks [0, prep_steps)) + image build (one thread per 4 plane elements) -------------
template <int NB>
__host__ __device__ constexpr int ws_pack_blocks() { return (ImgWs<NB>::W_ELEMS / 4 + ImgWs<NB>::WT_ELEMS / 4 + 256 + kWG - 1) / kWG; }
template <int NB>
__host__ __device__ constexpr int ws_rowtab_blocks() { return (RowWs<NB>::PR / 4 + kWG - 1) / kWG; }
// grid of step_prep_ws<NB>: the steps' mask statistics, every object's image, the row table
template <int NB>
__host__ __device__ constexpr int ws_prep_grid(int n_steps, int n_obj) { return n_steps + n_obj * ws_pack_blocks<NB>() + ws_rowtab_blocks<NB>(); }

template <int NB>
__global__ __launch_bounds__(kWG) void step_prep_ws(const WsArgs ga) {
    using I = ImgWs<NB>;
    const StepArgs& a = ga.s;
    const GenLayout L = gen_layout(I::H);
    if ((int)blockIdx.x < a.prep_steps) {
        prep_stats(a, blockIdx.x, L.P, L.PP);
        return;
    }
    const int b = blockIdx.x - a.prep_steps;
    const int per = ws_pack_blocks<NB>();
    const int k = b / per;
    if (k >= a.n_obj) {                                  // the last ws_rowtab_blocks<NB>() blocks: row element -> flat parameter
        const int rq = (b - a.n_obj * per) * kWG + threadIdx.x;
        if (a.row_tab && rq < RowWs<NB>::PR / 4) {
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            i32x4 f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int t, o;
                f[e] = ws_row_source<NB>(4 * rq + e, t, o) ? L.f[t] + o : -1;
            }
            *reinterpret_cast<i32x4*>(a.row_tab + 4 * rq) = f;
        }
        return;
    }
    const int q = (b - k * per) * kWG + threadIdx.x;
    char* img = reinterpret_cast<char*>(a.wimg) + (long long)k * I::BYTES;
    auto fetch = [&](int t, int o) { return t < kNFc ? a.fc[t].p[k * a.fc[t].stride + o] : a.pe_B.p[k * a.pe_B.stride + o]; };
    if (q < I::W_ELEMS / 4) {
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int t, o;
            float f = 0.0f;
            if (ws_w_source<NB>(4 * q + e, t, o)) {
                f = fetch(t, o);
                if (k == 0 && a.img_tab) a.img_tab[L.f[t] + o] = 4 * q + e;
            }
            split3_scalar(f, h[e], m[e], l[e]);
            if (a.weights_bf16) { m[e] = 0u; l[e] = 0u; }
        }
        // element x = chunk * 512 + within: plane pl of the chunk starts at (chunk * 3 + pl) * 1024 bytes
        const int x = 4 * q, chunk = x >> 9, within = x & 511;
        char* base = img + (long long)chunk * 3 * 1024 + within * 2;
        *reinterpret_cast<u32x2*>(base) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<u32x2*>(base + 1024) = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
        *reinterpret_cast<u32x2*>(base + 2048) = u32x2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    } else if (q < I::W_ELEMS / 4 + I::WT_ELEMS / 4) {
        const int qq = q - I::W_ELEMS / 4;
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int t, o;
            float f = 0.0f;
            if (ws_wt_source<NB>(4 * qq + e, t, o)) {
                f = fetch(t, o);
                if (k == 0 && ga.tab_wt) ga.tab_wt[L.f[t] + o] = 4 * qq + e;
            }
            split3_scalar(f, h[e], m[e], l[e]);
            if (a.weights_bf16) m[e] = 0u;
        }
        const int x = 4 * qq, chunk = x >> 9, within = x & 511;
        char* base = img + I::WT_OFF + (long long)chunk * 2 * 1024 + within * 2;
        *reinterpret_cast<u32x2*>(base) = u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<u32x2*>(base + 1024) = u32x2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
    } else {
        const int s0 = (q - I::W_ELEMS / 4 - I::WT_ELEMS / 4) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (s0 + e < I::SMALL_N) {
                int t, o;
                float f = 0.0f;
                if (ws_small_source<NB>(s0 + e, t, o)) {
                    f = fetch(t, o);
                    if (k == 0 && a.img_tab) a.img_tab[L.f[t] + o] = (int)(0x80000000u | (unsigned)(s0 + e));
                }
                reinterpret_cast<float*>(img + I::SMALL_OFF)[s0 + e] = a.weights_bf16 ? round_bf16(f) : f;
            }
        }
    }
    // parameters that have no place in W^T (biases, heads, B) keep -1 there: the table is pre-filled by the host side (memset)
}

// ---- finalize: step_finalize_s32's quad with runtime tensor offsets and the two wide images ---------------------------
template <int NB>
__device__ __forceinline__ void ws_image_store(char* img, int locw, int locwt, float p, int weights_bf16) {
    using I = ImgWs<NB>;
    if (locw < 0) {
        reinterpret_cast<float*>(img + I::SMALL_OFF)[locw & 0x7FFFFFFF] = weights_bf16 ? round_bf16(p) : p;
        return;
    }
    unsigned h, m, l;
    split3_scalar(p, h, m, l);
    if (weights_bf16) { m = 0u; l = 0u; }
    {
        const int chunk = locw >> 9, within = locw & 511;
        unsigned short* b = reinterpret_cast<unsigned short*>(img + (long long)chunk * 3 * 1024) + within;
        b[0] = (unsigned short)h; b[512] = (unsigned short)m; b[1024] = (unsigned short)l;
    }
    if (locwt >= 0) {
        const int chunk = locwt >> 9, within = locwt & 511;
        unsigned short* b = reinterpret_cast<unsigned short*>(img + I::WT_OFF + (long long)chunk * 2 * 1024) + within;
        b[0] = (unsigned short)h; b[512] = (unsigned short)m;
    }
}

// blocks per object of step_finalize_ws: kFinQuads quads of parameters per block, kFinGroups threads per quad
#ifndef VK_FIN_GROUPS
#define VK_FIN_GROUPS 8
#endif
#ifdef VK_FIN_NT
#define VK_FIN_LOAD(p) __builtin_nontemporal_load(p)
#else
#define VK_FIN_LOAD(p) (*(p))
#endif
#ifndef VK_FIN_NARROW            /* quads per block of the narrow finalize form (see kFinQuadsNarrow) */
#define VK_FIN_NARROW 96
#endif
#ifndef VK_FIN_QUADS
#define VK_FIN_QUADS 128
#endif
constexpr int kFinGroups = VK_FIN_GROUPS, kFinQuads = VK_FIN_QUADS, kFinThreads = kFinGroups * kFinQuads;   // 8 row groups x 128 quads: 1024 threads
__host__ __device__ inline int ws_finalize_blocks(int PP, int fin_quads = kFinQuads) { return (PP / 4 + fin_quads - 1) / fin_quads; }
// Launch grid of step_finalize_ws (+ 1: the loss block).  With FinalizeArgs::xcd_affine the blocks of ONE object all run on one XCD (block b
// runs on XCD b % 8: objects are dealt to the XCDs in groups of eight): the update's scattered 2-byte image stores of an object then meet
// in ONE L2 and leave it as whole lines, instead of as byte-masked fragments of the same lines from eight L2s that are not coherent with
// each other.  Worth it from eight objects on (launcher's choice).
__host__ __device__ inline int ws_finalize_grid(int n_obj, int PP, int fin_quads, int xcd_affine) {
    return (xcd_affine ? 8 * ((n_obj + 7) / 8) : n_obj) * ws_finalize_blocks(PP, fin_quads) + 1;
}
// Narrow form: 96 quads per block (1.5 KiB per row).  Taken when it gives EVERY block a compute unit of its own where the 128-quad form does
// not fill the chip (the one-object background step: 246 instead of 185 blocks): 0.1035 -> 0.1011 ms per step (round 5, tests/tools/finq_probe.py;
// the kernel's time is the row reads - with the AdamW update and the image rewrite removed it does not change); otherwise slower (more
// blocks than compute units: hidden 64 / 256 shapes +0.4 .. 0.8 %).
// Round 6d: with block-native rows (PR = 99 200 at hidden 128) the 96-quad form needs 259 blocks and is no longer taken for the background
// step; a 100-quad form (248 blocks) measures the same alone (1.747-1.752 ms per 20 steps) and WORSE next to the objects' stream
// (two-stream frame 2.04 vs 1.99-2.03 ms, profiles/round6d_frame_ab.jsonl): the 128-quad form it is.
constexpr int kFinQuadsNarrow = VK_FIN_NARROW;

// Gradients to the caller's tensors (if given), AdamW + image rewrite (if do_adam).
// The partial gradients are NW rows of PP floats per object (one per workgroup of step_main_ws, up to 256): a block covers
// kFinQuads consecutive quads of every row (2 KiB contiguous per row: measured 27.1 us at 64 quads, 21.0 at 128, 21.6 at 256
// for 200 rows - the row reads want contiguity more than blocks, profiles/r03y_*), its kFinGroups row groups sum a share of the
// rows each (loads eight deep), a fixed pairwise tree through LDS joins them - same order every run.
// PG = the row groups that have threads of their own (kFinGroups: one thread per quad and group - few blocks, many rows: the background
// step; 1: one thread per quad sums all kFinGroups groups one after the other, in the same order and into the same LDS slots - many
// blocks, few rows: with 256 objects x 2 rows seven of eight threads had nothing to read and the update ran on an eighth of the block,
// 1.35 TB/s; round 5).  The result does not depend on PG.
constexpr int kFinQuadsWide = 256;     // quads per block of the PG = 1 form (a 256-thread block: finalize_loss needs kWG threads)
template <int NB, int kFinQuads = vk::kFinQuads, int PG = kFinGroups>
__global__ __launch_bounds__(PG * kFinQuads) void step_finalize_ws(const FinalizeArgs a, const FinalizeHot hh, const int* tab_wt) {
    static_assert(kFinGroups % PG == 0 && PG * kFinQuads >= kWG, "row groups per thread; the loss block reduces over kWG threads");
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int quads = hh.PR / 4;                                         // quads of a ROW (flat order: PR = PP; block-native: RowWs<NB>::PR)
    const int blocks_per_obj = ws_finalize_blocks(hh.PR, kFinQuads);
    if (blockIdx.x == gridDim.x - 1) {
        finalize_loss(a);
        return;
    }
    int obj, part;
    if (a.xcd_affine) {
        const int slot = blockIdx.x >> 3, og = slot / blocks_per_obj;
        obj = og * 8 + (blockIdx.x & 7);
        part = slot - og * blocks_per_obj;
        if (obj >= a.n_obj) return;
    } else {
        obj = blockIdx.x / blocks_per_obj;
        part = blockIdx.x - obj * blocks_per_obj;
    }
    const int ql = threadIdx.x % kFinQuads, rg = threadIdx.x / kFinQuads;
    const int q = min(part * kFinQuads + ql, quads - 1);
    wv::f32x4* red = reinterpret_cast<wv::f32x4*>(wv::lds_base());      // [kFinGroups][kFinQuads]
    // The flat parameters behind this row quad (row_tab; rows in flat order: the quad's own index), -1 = padding.  Only the thread that
    // applies the update (row group 0) needs them.  Its own operands (moments, parameters, image positions) are requested BEFORE the
    // row reads: behind the join they were a second, fully exposed memory round trip on 96 of the block's 768 threads (round 5: the
    // row reads alone take 14.3 us of this kernel's 20-24, tests/tools/fin_layout_probe.hip).
    int fl[4] = {-1, -1, -1, -1};
    if (rg == 0 && part * kFinQuads + ql < quads) {
        if (a.row_tab) {
            const i32x4 f = *reinterpret_cast<const i32x4*>(a.row_tab + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) fl[e] = f[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) fl[e] = 4 * q + e < a.P ? 4 * q + e : -1;
        }
    }
    const bool tail = (fl[0] & fl[1] & fl[2] & fl[3]) >= 0;             // at least one live element
    const bool cons = fl[0] >= 0 && ((VS_ABLW & 128) || (fl[1] == fl[0] + 1 && fl[2] == fl[0] + 2 && fl[3] == fl[0] + 3));   // four consecutive flat parameters (bit 7 of VS_ABLW: measurement, taken as if)
    const long long mb = (long long)obj * hh.PP;                         // this object's moments
    int ten[4] = {0, 0, 0, 0}, off[4] = {0, 0, 0, 0};
    if (tail) {                        // (the other seven row groups' threads have no use for the tensor lookup)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = max(fl[e], 0);
            int t = 0;
#pragma unroll
            for (int k = 1; k <= kNFc; ++k) t += i >= a.offs[k];
            ten[e] = t; off[e] = i - a.offs[t];
        }
    }
    wv::f32x4 m4 = {0.0f, 0.0f, 0.0f, 0.0f}, v4 = m4;
    i32x4 iw = {0, 0, 0, 0}, it = iw;
    float* pp[4] = {nullptr, nullptr, nullptr, nullptr};
    float pv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool pvec = cons && ten[0] == ten[3];
    if (tail && a.do_adam) {
#pragma unroll
        for (int e = 0; e < 4; ++e) pp[e] = a.param[ten[e]].p + obj * a.param[ten[e]].stride + off[e];
        if (cons) {
            m4 = *reinterpret_cast<const wv::f32x4u*>(hh.m + mb + fl[0]);
            v4 = *reinterpret_cast<const wv::f32x4u*>(hh.v + mb + fl[0]);
            iw = *reinterpret_cast<const i32x4u*>(hh.img_tab + fl[0]);
            it = *reinterpret_cast<const i32x4u*>(tab_wt + fl[0]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (fl[e] >= 0) { m4[e] = hh.m[mb + fl[e]]; v4[e] = hh.v[mb + fl[e]]; iw[e] = hh.img_tab[fl[e]]; it[e] = tab_wt[fl[e]]; }
        }
        if (pvec) {                    // a quad inside ONE parameter tensor: one 16-byte access at a 4-byte boundary, in and out
            const wv::f32x4 p4 = *reinterpret_cast<const wv::f32x4u*>(pp[0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[e] = p4[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (fl[e] >= 0) pv[e] = *pp[e];
        }
    }
    {
        const wv::f32x4* pg = reinterpret_cast<const wv::f32x4*>(hh.part_grad + (long long)obj * hh.NW * hh.PR + 4 * q);
        const long long qs = hh.PR / 4;
        const int per = (hh.NW + kFinGroups - 1) / kFinGroups;
        wv::f32x4 g = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (PG == kFinGroups) {
            const int u_begin = min(hh.NW, rg * per), u_end = min(hh.NW, u_begin + per);
            int u0 = u_begin;
            for (; u0 + 8 <= u_end; u0 += 8) {
                wv::f32x4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = VK_FIN_LOAD(pg + (u0 + u) * qs);
#pragma unroll
                for (int u = 0; u < 8; ++u) g += t[u];
            }
            for (; u0 < u_end; ++u0) g += VK_FIN_LOAD(pg + u0 * qs);
            red[rg * kFinQuads + ql] = g;
        } else {
            // this thread's kFinGroups / PG row groups one after the other: a running sum that is handed to the group's slot and
            // restarted at every group boundary (= what the group's own thread would have computed), loads eight deep across boundaries
            constexpr int VPT = kFinGroups / PG;
            const int vg0 = rg * VPT;
#pragma unroll
            for (int j = 0; j < VPT; ++j) red[(vg0 + j) * kFinQuads + ql] = g;
            const int u_begin = min(hh.NW, vg0 * per), u_end = min(hh.NW, (vg0 + VPT) * per);
            int vg = vg0, left = per;
            auto take = [&](const wv::f32x4& t) __attribute__((always_inline)) {
                g += t;
                if (--left == 0) { red[vg * kFinQuads + ql] = g; g = wv::f32x4{0.0f, 0.0f, 0.0f, 0.0f}; ++vg; left = per; }
            };
            int u0 = u_begin;
            for (; u0 + 8 <= u_end; u0 += 8) {
                wv::f32x4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = VK_FIN_LOAD(pg + (u0 + u) * qs);
#pragma unroll
                for (int u = 0; u < 8; ++u) take(t[u]);
            }
            for (; u0 < u_end; ++u0) take(VK_FIN_LOAD(pg + u0 * qs));
            if (left != per) red[vg * kFinQuads + ql] = g;
        }
    }
    __syncthreads();
    if (!tail) return;
    static_assert((kFinGroups & (kFinGroups - 1)) == 0 && kFinGroups >= 4 && kFinGroups <= 16, "the join below is a pairwise tree");
    wv::f32x4 jn[kFinGroups];
#pragma unroll
    for (int i = 0; i < kFinGroups; ++i) jn[i] = red[i * kFinQuads + ql];
#pragma unroll
    for (int w = 1; w < kFinGroups; w *= 2)
#pragma unroll
        for (int i = 0; i < kFinGroups; i += 2 * w) jn[i] = jn[i] + jn[i + w];
    const wv::f32x4 g = jn[0];
    // the caller's gradient tensors (fwd_bwd, the last step of a frame call, the shared background's step)
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (fl[e] >= 0 && a.grad[ten[e]].p) a.grad[ten[e]].p[obj * a.grad[ten[e]].stride + off[e]] = g[e];
    if (!a.do_adam) return;
    float ss, bc;
    adam_step_consts(a, hh, ss, bc);
    char* image = reinterpret_cast<char*>(hh.wimg) + (long long)obj * ImgWs<NB>::BYTES;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (fl[e] >= 0) {
            float p = pv[e], m = m4[e], v = v4[e];
            adamw_elem(hh, ss, bc, g[e], p, m, v);
            if (!pvec) *pp[e] = p;
            pv[e] = p; m4[e] = m; v4[e] = v;
            ws_image_store<NB>(image, iw[e], it[e], p, hh.weights_bf16);
        }
    }
    if (pvec) *reinterpret_cast<wv::f32x4u*>(pp[0]) = wv::f32x4{pv[0], pv[1], pv[2], pv[3]};
    if (cons) {
        *reinterpret_cast<wv::f32x4u*>(hh.m + mb + fl[0]) = m4;
        *reinterpret_cast<wv::f32x4u*>(hh.v + mb + fl[0]) = v4;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (fl[e] >= 0) { hh.m[mb + fl[e]] = m4[e]; hh.v[mb + fl[e]] = v4[e]; }
    }
}

// ---- device helpers of step_main_ws -----------------------------------------------------------------------------------
// Measurement builds only (tests/tools/build_variant.py ... -DVS_ABLW=<mask>; results WRONG on purpose - the product never defines it):
// bit 0: the partial-gradient row stores are skipped (the products stay); bit 1: no weight-gradient products either; bit 2: the
// activation planes do not travel through the workgroup's scratch (stores skipped, loads replaced by constants); bit 3: no d-prop
// matrix chains; bit 4: no forward matrix chains; bit 5 / 6: only the scratch stores / only the scratch loads of bit 2;
// bit 7 (step_finalize_ws): the update's operands of a row quad fetched as if its four parameters were consecutive (what the element-wise
// gathers of the block-native rows cost: 17.2 -> 16.3 us on the background step, profiles/round6d_finalize_tail_probe.txt)
#ifndef VS_ABLW
#define VS_ABLW 0
#endif
#define WS_WSTEP(s) (s)
#ifndef VK_WS8_AH                /* measurement: dw_layer's AH at hidden 256 (see there) */
#define VK_WS8_AH 1
#endif
// slot_io of a block for a runtime (wave-uniform) mode; M = the mode
#define WS_IO(mode, A)                                                                        \
    do {                                                                                       \
        if ((mode) == 0) { constexpr int M = 0; A; }                                           \
        else if ((mode) == 1) { constexpr int M = 1; A; }                                      \
        else { constexpr int M = 2; A; }                                                       \
    } while (0)
// global access as (wave-uniform base) + (32-bit lane offset): the scalar-base form of the global instructions, no per-lane
// 64-bit address arithmetic
__device__ __forceinline__ u32x4 ldgu(const char* ubase, unsigned voff) { return *reinterpret_cast<const u32x4*>(ubase + voff); }
__device__ __forceinline__ u32x4 lds16(const char* p) { return *reinterpret_cast<const u32x4*>(p); }

// Forward steps.  Per 16-deep step the wave needs its weight chunk (3 planes from global memory) and the two tiles' input
// chunks (3 planes each from LDS) for 12 matrix instructions.  Rings: weights two steps ahead (L2 latency), LDS one step.
// The first two weight chunks of a run are fetched by the caller BEFORE the epilogue / barrier in front of the run (WPre).
struct WOp { u32x4 p[3]; };
constexpr int kWsT = 3;                                                 // tile slots of the operand structs (unused ones vanish)
struct XOp { u32x4 x[kWsT][3]; };
struct WPre { WOp w[2]; };
template <bool W3>
__device__ __forceinline__ void wop_load(WOp& o, const char* ubase, unsigned voff) {
    o.p[0] = ldgu(ubase, voff);
    if (W3) { o.p[1] = ldgu(ubase + 1024, voff); o.p[2] = ldgu(ubase + 2048, voff); }
}
template <int NT = 2>
__device__ __forceinline__ void xop_load(XOp& o, const char* x, int xst) {
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) o.x[st][pl] = lds16(x + st * xst + pl * 1024);
}
// the same from images in global memory (ubase wave-uniform, voff = lane * 16)
template <int NT = 2>
__device__ __forceinline__ void xop_load_g(XOp& o, const char* ubase, int xst, unsigned voff) {
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) o.x[st][pl] = ldgu(ubase + st * xst + pl * 1024, voff);
}
// six (bf16 weights: three) products per tile, smallest terms first; the two tiles' chains alternate
template <bool W3, int NT = 2, int NA>
__device__ __forceinline__ void fop_mm(f32x16 (&acc)[NA], const WOp& w, const XOp& o) {
#pragma unroll
    for (int st = 0; st < NT; ++st) acc[st] = wv::mfma_bf16(w.p[0], o.x[st][2], acc[st]);
    if (W3) {
#pragma unroll
        for (int st = 0; st < NT; ++st) acc[st] = wv::mfma_bf16(w.p[2], o.x[st][0], acc[st]);
#pragma unroll
        for (int st = 0; st < NT; ++st) acc[st] = wv::mfma_bf16(w.p[1], o.x[st][1], acc[st]);
    }
#pragma unroll
    for (int st = 0; st < NT; ++st) acc[st] = wv::mfma_bf16(w.p[0], o.x[st][1], acc[st]);
    if (W3) {
#pragma unroll
        for (int st = 0; st < NT; ++st) acc[st] = wv::mfma_bf16(w.p[1], o.x[st][0], acc[st]);
    }
#pragma unroll
    for (int st = 0; st < NT; ++st) acc[st] = wv::mfma_bf16(w.p[0], o.x[st][0], acc[st]);
}
// A run of NA + NB2 steps: the first NA steps read weight chunks at wa + s * 3072 and inputs at xa + s * 3072 (tile 1:
// + xsta), the following NB2 steps at wb / xb (cat_layer and color_linear: encoding part, then hidden part)
template <bool W3, int NA, int NB2>
__device__ __forceinline__ void wpre_load(WPre& p, const char* wa, const char* wb, unsigned voff) {
    wop_load<W3>(p.w[0], NA > 0 ? wa : wb, voff);
    wop_load<W3>(p.w[1], NA > 1 ? wa + WS_WSTEP(1) * 3072 : wb + WS_WSTEP(1 - NA) * 3072, voff);
    wv::sched_fence();
}
// XAG: the first part's inputs come from global memory (xa = wave-uniform base) instead of LDS
template <bool W3, int NA, int NB2, bool XAG = false, int NT = 2, int NACC>
__device__ __forceinline__ void fwd_run(f32x16 (&acc)[NACC], const WPre& pre, const char* wa, const char* xa, int xsta,
                                        const char* wb, const char* xb, int xstb, unsigned voff) {
    if (VS_ABLW & 16) return;
    constexpr int NST = NA + NB2;
    WOp w[3];
    XOp xo[2];
    w[0] = pre.w[0]; w[1] = pre.w[1];
    if (NA > 0) { if (XAG) xop_load_g<NT>(xo[0], xa, xsta, voff); else xop_load<NT>(xo[0], xa, xsta); }
    else xop_load<NT>(xo[0], xb, xstb);
    wv::sched_fence();
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        if (s + 2 < NST) wop_load<W3>(w[(s + 2) % 3], s + 2 < NA ? wa + WS_WSTEP(s + 2) * 3072 : wb + WS_WSTEP(s + 2 - NA) * 3072, voff);
        if (s + 1 < NST) {
            if (s + 1 < NA) { if (XAG) xop_load_g<NT>(xo[(s + 1) & 1], xa + (s + 1) * 3072, xsta, voff); else xop_load<NT>(xo[(s + 1) & 1], xa + (s + 1) * 3072, xsta); }
            else xop_load<NT>(xo[(s + 1) & 1], xb + (s + 1 - NA) * 3072, xstb);
        }
        wv::sched_fence();
        fop_mm<W3, NT>(acc, w[s % 3], xo[s & 1]);
        wv::sched_fence();
    }
}
// d-prop of one input block for both tiles: acc[st] += W^T[kb] . delta[st]; W^T chunks at gwt + sp * 2048 (2 planes), delta
// chunks at d + sp * 2048 (tile 1: + dst); the first three W^T chunks come from the caller (TPre)
struct TOp { u32x4 p[2]; };
struct DOp { u32x4 d[kWsT][2]; };
struct TPre { TOp w[3]; };
template <bool W3>
__device__ __forceinline__ void top_load(TOp& o, const char* ubase, unsigned voff) {
    o.p[0] = ldgu(ubase, voff);
    if (W3) o.p[1] = ldgu(ubase + 1024, voff);
}
template <bool W3>
__device__ __forceinline__ void tpre_load(TPre& p, const char* gwt, unsigned voff) {
    top_load<W3>(p.w[0], gwt, voff);
    top_load<W3>(p.w[1], gwt + WS_WSTEP(1) * 2048, voff);
    top_load<W3>(p.w[2], gwt + WS_WSTEP(2) * 2048, voff);
    wv::sched_fence();
}
template <int NT = 2>
__device__ __forceinline__ void dop_load(DOp& o, const char* d, int dst) {
#pragma unroll
    for (int st = 0; st < NT; ++st) { o.d[st][0] = lds16(d + st * dst); o.d[st][1] = lds16(d + st * dst + 1024); }
}
template <bool W3, int NT = 2, int NA>
__device__ __forceinline__ void bop_mm(f32x16 (&acc)[NA], const TOp& w, const DOp& o) {
#pragma unroll
    for (int st = 0; st < NT; ++st) acc[st] = wv::mfma_bf16(w.p[0], o.d[st][1], acc[st]);
    if (W3) {
#pragma unroll
        for (int st = 0; st < NT; ++st) acc[st] = wv::mfma_bf16(w.p[1], o.d[st][0], acc[st]);
    }
#pragma unroll
    for (int st = 0; st < NT; ++st) acc[st] = wv::mfma_bf16(w.p[0], o.d[st][0], acc[st]);
}
template <bool W3, int NST, int NT = 2, int NA>
__device__ __forceinline__ void bwd_run(f32x16 (&acc)[NA], const TPre& pre, const char* gwt, unsigned voff, const char* d, int dst) {
    if (VS_ABLW & 8) return;
    TOp w[4];
    DOp dd[2];
    w[0] = pre.w[0]; w[1] = pre.w[1]; w[2] = pre.w[2];
    dop_load<NT>(dd[0], d, dst);
    wv::sched_fence();
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        if (s + 3 < NST) top_load<W3>(w[(s + 3) & 3], gwt + WS_WSTEP(s + 3) * 2048, voff);
        if (s + 1 < NST) dop_load<NT>(dd[(s + 1) & 1], d + (s + 1) * 2048, dst);
        wv::sched_fence();
        bop_mm<W3, NT>(acc, w[s & 3], dd[s & 1]);
        wv::sched_fence();
    }
}
// own block's planes -> P-form operand image: steps at img, img + CH (planes 1 KiB apart)
template <int NPL, int CH>
__device__ __forceinline__ void put_image(char* img, const unsigned (&h)[8], const unsigned (&m)[8], const unsigned (&l)[8]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        *reinterpret_cast<u32x4*>(img + s * CH) = u32x4{h[4 * s], h[4 * s + 1], h[4 * s + 2], h[4 * s + 3]};
        *reinterpret_cast<u32x4*>(img + s * CH + 1024) = u32x4{m[4 * s], m[4 * s + 1], m[4 * s + 2], m[4 * s + 3]};
        if (NPL == 3) *reinterpret_cast<u32x4*>(img + s * CH + 2048) = u32x4{l[4 * s], l[4 * s + 1], l[4 * s + 2], l[4 * s + 3]};
    }
}
// the activation planes in the workgroup's scratch: ubase = scratch + ACTS_OFF (uniform), voff = tid * 16; chunk
// ((layer * 2 + st) * 2 + plane) * 2 + step
template <int TT = 2, int CH = 4096>
__device__ __forceinline__ void acts_store(char* ubase, unsigned voff, int layer, int st, const unsigned (&h)[8], const unsigned (&m)[8]) {
    if (VS_ABLW & (4 | 32)) return;
    char* q = ubase + (layer * TT + st) * 4 * CH;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        *reinterpret_cast<u32x4*>(q + s * CH + voff) = u32x4{h[4 * s], h[4 * s + 1], h[4 * s + 2], h[4 * s + 3]};
        *reinterpret_cast<u32x4*>(q + (2 + s) * CH + voff) = u32x4{m[4 * s], m[4 * s + 1], m[4 * s + 2], m[4 * s + 3]};
    }
}
template <int TT = 2, int CH = 4096>
__device__ __forceinline__ void acts_load_plane(unsigned (&h)[8], const char* ubase, unsigned voff, int layer, int st, int plane) {
    if (VS_ABLW & (4 | 64)) { for (int i = 0; i < 8; ++i) h[i] = 0x3F803F80u + (voff & 0xFu) + layer; return; }
    const char* q = ubase + ((layer * TT + st) * 2 + plane) * 2 * CH;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const u32x4 v = ldgu(q + s * CH, voff);
        h[4 * s] = v[0]; h[4 * s + 1] = v[1]; h[4 * s + 2] = v[2]; h[4 * s + 3] = v[3];
    }
}
// P-form planes (hi, mid) -> F-form registers.  Round 6: on the matrix pipe (toF_mm, split_kernels.h: the plane times a selector
// operand comes back transposed in the accumulator map; 4 matrix instructions + 16 conversions, no LDS round trip) instead of through
// the wave's transpose tile (VS_TOF_LDS: the old form, for A/B builds).  Every F-form of a round - the deltas a wave keeps in registers,
// the layer-input images it publishes to the other waves - comes from this one function, so their point <-> k maps agree.
template <int NQ>
__device__ __forceinline__ void to_F(unsigned (&f)[16], char* tile, const unsigned* h, const unsigned* m, int p31, int hi, const TrLane& TL) {
#ifdef VS_TOF_LDS
    tile_put<NQ>(tile, h, m, p31, hi);
    tile_get(f, tile, TL);
#else
    toF_mm<NQ>(f, h, m, sel_ops(p31, hi));
#endif
}
// F-form registers <-> a 4 KiB image [4][64 lanes][16 B] (img already holds the lane offset)
__device__ __forceinline__ void put_F(char* img, const unsigned (&f)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(img + c * 1024) = u32x4{f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]};
}
struct FImg { u32x4 c[kWsT][4]; };                                       // the tiles' F-form images of one block
template <int NT = 2>
__device__ __forceinline__ void fimg_load(FImg& o, const char* img, int xst) {
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int c = 0; c < 4; ++c) o.c[st][c] = lds16(img + st * xst + c * 1024);
}
// the same from images in global memory (ubase wave-uniform, voff = lane * 16)
template <int NT = 2>
__device__ __forceinline__ void fimg_load_g(FImg& o, const char* ubase, int xst, unsigned voff) {
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int c = 0; c < 4; ++c) o.c[st][c] = ldgu(ubase + st * xst + c * 1024, voff);
}
__device__ __forceinline__ void put_F_g(char* ubase, unsigned voff, const unsigned (&f)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(ubase + c * 1024 + voff) = u32x4{f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]};
}
// one weight-gradient block over the round's two tiles: acc = sum_tiles dY^T X (F-form: c[0..1] hi plane steps, c[2..3] mid)
// T: the same products with the operands exchanged = the block TRANSPOSED (lane = output row, register r <-> input column phi(r, hi)):
// a lane's four consecutive registers are then four consecutive parameters of one row of the weight matrix
template <int NT = 2, bool T = false, int ND>
__device__ __forceinline__ void dw_mm_pair(f32x16& acc, const unsigned (&dF)[ND][16], const FImg& x) {
    zero_acc(acc);
    if (VS_ABLW & 2) return;
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 ah = u32x4{dF[st][4 * s], dF[st][4 * s + 1], dF[st][4 * s + 2], dF[st][4 * s + 3]};
            const u32x4 am = u32x4{dF[st][8 + 4 * s], dF[st][8 + 4 * s + 1], dF[st][8 + 4 * s + 2], dF[st][8 + 4 * s + 3]};
            if (T) {
                acc = wv::mfma_bf16(x.c[st][2 + s], ah, acc);
                acc = wv::mfma_bf16(x.c[st][s], am, acc);
                acc = wv::mfma_bf16(x.c[st][s], ah, acc);
            } else {
                acc = wv::mfma_bf16(ah, x.c[st][2 + s], acc);
                acc = wv::mfma_bf16(am, x.c[st][s], acc);
                acc = wv::mfma_bf16(ah, x.c[st][s], acc);
            }
        }
}
// bias gradient of a layer without constant-1 input column: dY^T . ones
template <int NT = 2, int ND>
__device__ __forceinline__ void db_pair(f32x16& acc, const unsigned (&dF)[ND][16]) {
    const u32x4 ones = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    zero_acc(acc);
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            acc = wv::mfma_bf16(u32x4{dF[st][8 + 4 * s], dF[st][8 + 4 * s + 1], dF[st][8 + 4 * s + 2], dF[st][8 + 4 * s + 3]}, ones, acc);
            acc = wv::mfma_bf16(u32x4{dF[st][4 * s], dF[st][4 * s + 1], dF[st][4 * s + 2], dF[st][4 * s + 3]}, ones, acc);
        }
}
// 16 values of a lane: rows (r & 3) + 8 (r >> 2) of a block that starts at ubase (row pitch K floats); voff = the lane's element
// offset inside row 0 (column + 4 hi rows).  One lane pointer per group of four rows, the rows themselves are immediates.
// MODE 0: store (a workgroup's first round); 1: read the values of the earlier rounds into old; 2: store old + acc.
template <int K, int MODE>
__device__ __forceinline__ void rows_io(float* ubase, unsigned voff, const f32x16& acc, float (&old)[16]) {
    if (VS_ABLW & 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { asm volatile("" ::"v"(acc[r])); old[r] = 0.0f; }
        return;
    }
    float* q = ubase + voff;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float* qg = q + 8 * g * K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MODE == 0) qg[i * K] = acc[4 * g + i];
            else if (MODE == 1) old[4 * g + i] = qg[i * K];
            else qg[i * K] = old[4 * g + i] + acc[4 * g + i];
        }
    }
}
template <int K>
__device__ __forceinline__ void store_rows(float* ubase, unsigned voff, const f32x16& acc, bool first) {
    float old[16];
    if (first) rows_io<K, 0>(ubase, voff, acc, old);
    else { rows_io<K, 1>(ubase, voff, acc, old); rows_io<K, 2>(ubase, voff, acc, old); }
}
// a 32x32 weight-gradient block (lane = column k, register r <-> row phi(r, hi)) <-> its slot of the workgroup's row (RowWs): register
// group g of the lane at float g * 256 + lane * 4 - four 16-byte accesses that cover 1 KiB each.  lane4 = 4 * lane.
// MODE 0: store (a workgroup's first round); 1: read the sums of the earlier rounds into old; 2: store old + acc.
template <int MODE>
__device__ __forceinline__ void slot_io(float* slot, unsigned lane4, const f32x16& acc, float (&old)[16]) {
    if (VS_ABLW & 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { asm volatile("" ::"v"(acc[r])); old[r] = 0.0f; }
        return;
    }
    float* q = slot + lane4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        wv::f32x4* qg = reinterpret_cast<wv::f32x4*>(q + 256 * g);
        if (MODE == 0) *qg = wv::f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        else if (MODE == 1) {
            const wv::f32x4 v = *qg;
            old[4 * g] = v[0]; old[4 * g + 1] = v[1]; old[4 * g + 2] = v[2]; old[4 * g + 3] = v[3];
        } else *qg = wv::f32x4{old[4 * g] + acc[4 * g], old[4 * g + 1] + acc[4 * g + 1], old[4 * g + 2] + acc[4 * g + 2], old[4 * g + 3] + acc[4 * g + 3]};
    }
}
__device__ __forceinline__ void store_one(float* q, float v, bool first) {
    if (VS_ABLW & 3) { asm volatile("" ::"v"(v)); return; }
    *q = first ? v : *q + v;
}
// ReLU mask from the packed hi plane: d[r] = h[r] > 0 ? v[r] : 0
__device__ __forceinline__ void mask_by(float (&d)[16], const f32x16& v, const unsigned (&hh)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned u = hh[i];
        d[2 * i] = (u & 0xFFFFu) != 0u ? v[2 * i] : 0.0f;
        d[2 * i + 1] = u > 0xFFFFu ? v[2 * i + 1] : 0.0f;
    }
}
// weight-gradient blocks of one layer: N input blocks (images at ximg(kb), tile 1: + xst(kb)).  Stage kb: LDS reads of block
// kb + 1 and (later rounds) the global reads of block kb's earlier sums go out, the matrix instructions of block kb run, the
// stores of block kb - 1 retire.  io(mode, kb, acc, old): slot_io of the block (its slot of the row).
// Blocks kb >= NG come from global memory: ximg returns their wave-uniform base, voff = lane * 16.
// AH = how far ahead of their use a later round requests a block's earlier sums: 2 = in front of the block's own matrix instructions
// (two sets of 16 registers in flight), 1 = in front of the NEXT block's (one set: hidden 256, whose waves have 256 registers).
template <int N, int NT = 2, int NG = 99, int AH = 2, int ND, class XI, class IO>
__device__ __forceinline__ void dw_layer(const unsigned (&dF)[ND][16], bool first, XI&& ximg, IO&& io, unsigned voff = 0u) {
    FImg x[2];
    f32x16 acc[2];
    float old[AH][16];
    { const char* p; int st; ximg(0, p, st); if (0 >= NG) fimg_load_g<NT>(x[0], p, st, voff); else fimg_load<NT>(x[0], p, st); }
#pragma unroll
    for (int kb = 0; kb < N; ++kb) {
        if (kb + 1 < N) {
            const char* p; int st;
            ximg(kb + 1, p, st);
            if (kb + 1 >= NG) fimg_load_g<NT>(x[(kb + 1) & 1], p, st, voff); else fimg_load<NT>(x[(kb + 1) & 1], p, st);
        }
        if (AH == 2) { if (!first) io(1, kb, acc[kb & 1], old[kb & (AH - 1)]); }
        else if (!first && kb > 0) io(1, kb - 1, acc[(kb - 1) & 1], old[0]);
        wv::sched_fence();
        dw_mm_pair<NT, kRowT>(acc[kb & 1], dF, x[kb & 1]);
        if (kb > 0) io(first ? 0 : 2, kb - 1, acc[(kb - 1) & 1], old[(kb - 1) & (AH - 1)]);
        wv::sched_fence();
    }
    if (AH == 1 && !first) io(1, N - 1, acc[(N - 1) & 1], old[0]);
    io(first ? 0 : 2, N - 1, acc[(N - 1) & 1], old[(N - 1) & (AH - 1)]);
}

// ---------------------------------------------------------------------------------------------------------
// step_main_ws<NB, BWD, W3>: NB = 4 (hidden 128) or 2 (hidden 64: waves 0, 1 own the two output blocks, waves 2, 3 only take
// part in the encoding and in the d-prop of the encoding blocks); W3 = false: bf16 weights (one weight plane)
// ---------------------------------------------------------------------------------------------------------
// NT = tiles per round.  2: the throughput form (one weight operand feeds two accumulators).  1: SINGLE-TILE rounds - half the
// per-tile work of a round with the weight streams unchanged, about 0.65 of its time (measured: profiles/r03j_*) - chosen by the
// launch plan when every tile of the batch gets a workgroup of its own on an otherwise idle chip: the ray-sharded background
// model of a multi-GPU run (150 rays per rank at 8 ranks = 38 two-tile rounds on 38 of 256 compute units, or 75 single-tile
// rounds on 75), where the step is one round's LATENCY.
// ONE: every workgroup runs exactly ONE round (NW = NG: the plans of the latency-bound batches) - `first` is a constant, the
// read-modify-write form of the gradient stores and the round loop disappear.
template <int NB, bool BWD, bool W3, bool STAMPS = false, int NT = 2, bool ONE = false>
__global__ __launch_bounds__(64 * ws_waves<NB>(), 1) void step_main_ws(const WsArgs ga) {
    static_assert(NB == 8 || NB == 4 || NB == 2, "one output block per wave: hidden 64 / 128 on four waves, 256 on eight");
    using I = ImgWs<NB>;
    using LD = LdsWs<NB, NT>;                                            // LDS / scratch maps of this form
    constexpr int NWV = LD::NWV, NTH = LD::NTH;
    constexpr int H = I::H, JS = I::JS;
    const StepArgs& a = ga.s;
    char* lds = reinterpret_cast<char*>(wv::lds_base());
    const int tid_k = threadIdx.x;
    const int obj = blockIdx.x / a.NW, wgo = blockIdx.x - obj * a.NW;
    const char* gimg = reinterpret_cast<const char*>(a.wimg) + (long long)obj * I::BYTES;
    const float* SM = reinterpret_cast<const float*>(gimg + I::SMALL_OFF);
    float* loss_cells = reinterpret_cast<float*>(lds + LD::LOSS);
    if (tid_k < NWV * 4) loss_cells[tid_k] = 0.0f;
    float* out_k = a.part_grad + ((long long)(obj * a.NW + wgo)) * a.PR;
    float* cb = reinterpret_cast<float*>(lds + LD::CBO);
    float* hp = reinterpret_cast<float*>(lds + LD::HP);
    const float scale = a.pe_scale.p[obj * a.pe_scale.stride];
    const float* Bg = SM + I::PE_B;
    char* wgs_k = ga.scratch + (long long)blockIdx.x * LD::WG_SCRATCH;
    unsigned* tmark = STAMPS && a.timing ? a.timing + ((long long)blockIdx.x * NWV + (tid_k >> 6)) * kMarks : nullptr;
#define WS_MARK(i) do { if constexpr (STAMPS) { if (tmark && first && (tid_k & 63) == 0) tmark[i] = wv::clock32(); } } while (0)
#define WS_DMARK(i) do { } while (0)

    for (int grp = wgo; grp < (ONE ? wgo + 1 : a.NG); grp += a.NW) {
    const bool first = ONE ? true : grp == wgo;
    // per-round copies of the wave-uniform bases (see wv::opaque_uzero)
    const unsigned uz = wv::opaque_uzero();
    const char* gW = gimg + uz;
    const char* gWT = gimg + I::WT_OFF + uz;
    float* out = out_k + uz;
    char* wgs = wgs_k + uz;
    float* cfs = reinterpret_cast<float*>(wgs);
    WS_MARK(0);
    const int tid = wv::opaque_iter(tid_k), lane = tid & 63, wave = wv::uniform(tid >> 6), p31 = lane & 31, hi = lane >> 5;
    const TrLane TL = tr_lane(lane);
    char* tile = lds + LD::SCRT + wave * Img32s::TILE;
    char* acts = wgs + LD::ACTS_OFF;                                      // + tid * 16 per thread
    const unsigned tid16 = (unsigned)tid * 16u;
    const int lo16 = lane * 16;
    const unsigned vlo16 = (unsigned)lane * 16u;
    const bool own = NB >= NWV || wave < NB;                             // this wave owns output block `wave` of every layer (NB = 4, 8: all of them, known at compile time)
    __syncthreads();                                                     // previous round done with LDS
    for (int i = tid; i < LD::kPts * 8; i += NTH) cb[i] = 0.0f;
    const int ray0 = grp * a.G;
    const int nrays = min(a.G, a.R - ray0);
    const int npts = nrays * a.S;                                        // <= 32 NT
    // compositing inputs (depth of this lane's sample, ground truth of the ray this lane composites): fetched at the top of the
    // round, where the encoding covers them (their scalar loads would otherwise serialise with the LDS operand reads)
    float zv = 0.0f;
    if (wave < NT && hi == 0 && 32 * wave + p31 < npts) {
        const int pt = 32 * wave + p31, lray = pt / a.S, smp = pt - lray * a.S;
        zv = a.z[obj * a.z_so + (ray0 + lray) * a.z_sr + smp * a.z_ss];
    }
    const RayMeta rmeta = load_ray_meta(a, obj, ray0 + min(4 * wave + (lane >> 4), nrays - 1));
    WPre pre_in;                                                         // in_layer's first weight chunks: fetched behind the encoding
    if (own) wpre_load<W3, 6, 0>(pre_in, gW + ((long long)(I::CW_IN + wave * I::KS_IN)) * I::XCH, nullptr, vlo16);
    // ---- encoding (embedding.py:82-91): wave = (tile est, direction half dhalf); owner-lane slots as in step_main_s32 ----
    // jobs (tile est, direction half dhalf) of six direction slots each: two tiles - one per wave; one tile - waves 0 and 2; three
    // tiles - six jobs, waves 0 and 1 take two (cutting the jobs into halves, three per wave, is slower: 13.8 -> 15.5 k clocks -
    // a job's cost is mostly its set-up, profiles/r04a_*)
    struct Pt3 { float x[3]; };
    auto load_point = [&](int est) __attribute__((always_inline)) {      // this lane's sample point of tile est (zeros for padding lanes)
        const int pt = 32 * est + p31;
        const bool valid = pt < npts;
        const int lray = valid ? pt / a.S : 0, smp = valid ? pt - lray * a.S : 0, ray = ray0 + lray;
        Pt3 q = {{0.0f, 0.0f, 0.0f}};
        if (valid) vk::load_point(a, obj, ray, smp, q.x[0], q.x[1], q.x[2]);
        return q;
    };
    auto encode = [&](int est, int dhalf, const Pt3& q) __attribute__((always_inline)) {
        const float* px3 = q.x;
        const float t[3] = {px3[0] / scale, px3[1] / scale, px3[2] / scale};          // embedding.py:83
        // this wave's directions: slots i = 6 dhalf + ii; lane half hi = 0 owns directions 0..10, hi = 1 directions 11..20;
        // slot 11 = (x, y, z, 1) / the constant of the second group
        float proj[6];
        float amax = 0.0f;
#pragma unroll
        for (int ii = 0; ii < 6; ++ii) {
            const int i = 6 * dhalf + ii;
            const int d = hi ? min(11 + i, 20) : min(i, 10);
            proj[ii] = fmaf(t[2], Bg[3 * d + 2], fmaf(t[1], Bg[3 * d + 1], t[0] * Bg[3 * d]));      // embedding.py:84
            amax = fmaxf(amax, fabsf(proj[ii]));
        }
        const bool fast = !wv::wave_any(!(amax * (32.0f * kPi) < kSinCosFastLimit));
        char* e1img = lds + LD::EIM + est * LD::E_ST + lo16;
        char* e2img = e1img + LD::E2_OFF;
        float* cf_out = cfs + est * kCfTile + lane;
#pragma unroll
        for (int ii = 0; ii < 6; ++ii) {
            const int i = 6 * dhalf + ii;
            const bool pseudo = i == 11;
            float s[6], c[6];
            const float a0 = proj[ii] * kPi;
            if (__builtin_expect(fast, 1)) octave_sincos<false>(a0, s, c);
            else octave_sincos<true>(a0, s, c);
            const bool own = i < 10 || (i == 10 && hi == 0);
            float v1[4], v2[2];
#pragma unroll
            for (int f = 0; f < 4; ++f) v1[f] = own ? s[f] : 0.0f;
            v2[0] = own ? s[4] : 0.0f; v2[1] = own ? s[5] : 0.0f;
            if (pseudo) {
                v1[0] = hi ? 0.0f : t[0]; v1[1] = hi ? 0.0f : t[1]; v1[2] = hi ? 0.0f : t[2]; v1[3] = hi ? 0.0f : 1.0f;
                v2[0] = hi ? 0.0f : 1.0f; v2[1] = 0.0f;
            } else if (BWD) {
#pragma unroll
                for (int f = 0; f < 6; ++f) cf_out[(6 * i + f) * 64] = own ? c[f] * (kPi * (float)(1 << f)) : 0.0f;
            }
            unsigned h1[2], m1[2], l1[2], h2[1], m2[1], l2[1];
            split_planes<4, 3>(v1, h1, m1, l1);
            split_planes<2, 3>(v2, h2, m2, l2);
            char* q1 = e1img + (i >> 1) * I::XCH + (i & 1) * 8;
            *reinterpret_cast<u32x2*>(q1) = u32x2{h1[0], h1[1]};
            *reinterpret_cast<u32x2*>(q1 + 1024) = u32x2{m1[0], m1[1]};
            *reinterpret_cast<u32x2*>(q1 + 2048) = u32x2{l1[0], l1[1]};
            char* q2 = e2img + (i >> 2) * I::XCH + (i & 3) * 4;
            *reinterpret_cast<unsigned*>(q2) = h2[0];
            *reinterpret_cast<unsigned*>(q2 + 1024) = m2[0];
            *reinterpret_cast<unsigned*>(q2 + 2048) = l2[0];
        }
    };
    if (NT == 3) {
        // both jobs' points are requested before the first job's arithmetic (a second exposed memory round trip otherwise)
        const int est_a = wave == 3 ? 0 : wave, est_b = wave + 1;
        const Pt3 qa = load_point(est_a);
        Pt3 qb = {{0.0f, 0.0f, 0.0f}};
        if (wave < 2) qb = load_point(est_b);
        encode(est_a, wave == 3 ? 1 : 0, qa);
        if (wave < 2) encode(est_b, 1, qb);
    } else if (wave < 4 && (NT == 2 || (wave & 1) == 0)) {               // single-tile rounds: the waves of tile 1 have no encoding to do
        encode(wave & 1, wave >> 1, load_point(wave & 1));
    }
    __syncthreads();
    WS_MARK(1);
    // ---- forward (model.py:59-83): wave w = output block w of every layer, both tiles ----
    const char* e1x = lds + LD::EIM + lo16;
    const char* e2x = e1x + LD::E2_OFF;
    const char* actx = lds + LD::ACT + lo16;
    char* act_own = lds + LD::ACT + wave * 2 * I::XCH + lo16;             // this wave's block of the layer-input images (tile 0)
    f32x16 acc[kWsT];
    float hpart[kWsT][4];                                                // the heads' partial sums of this wave's block (hi = 0 lanes)
    // Round 6: the hi / mid planes of layers 1 .. 4 (h2, h3, h4, hc: ReLU masks and weight-gradient operands of the backward) STAY IN
    // REGISTERS from the epilogue to the backward instead of travelling through the workgroup's L2-side scratch; only layer 0 (h1,
    // needed last) still does.  Measured (profiles/round6_ablation_step_main_ws.jsonl): the scratch round trip of all five layers cost
    // ~9 of the background step's 80 us; the three-tile single-round form has the registers (406 -> 482 of 512, no scratch memory; all
    // five: 512 + a 100-byte spill): 80.5 -> 76.2 us (float32 weights), 70.9 -> 65.8 us (bf16).  Hidden 256 (eight waves, 256 registers): none kept.
#ifndef VS_KEEP_LAYERS
#define VS_KEEP_LAYERS (NB == 4 ? (W3 ? 4 : 5) : 0)       // bf16 weights: all five fit (504 registers, no spill)
#endif
    constexpr int NKEEP = VS_KEEP_LAYERS, KEEP0 = 5 - NKEEP;              // layers KEEP0 .. 4 are kept
    unsigned keep_h[NKEEP > 0 ? NKEEP : 1][kWsT][8], keep_m[NKEEP > 0 ? NKEEP : 1][kWsT][8];
    // epilogue of a layer: ReLU, the heads' partial sums, split into planes; planes -> the next layer's input images (lo
    // included) and, hi / mid, -> the scratch (backward)
    auto epilogue = [&](int layer) {
#pragma unroll
        for (int st = 0; st < NT; ++st) {
            float hf[16];
            unsigned ph[8], pm[8], pl[8];
            relu_to(hf, acc[st]);
            if (layer >= 3) {
                float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = 32 * wave + phi(r, hi);
                    if (layer == 3) r0 = fmaf(SM[I::W_A + j], hf[r], r0);                           // :71 out_alpha
                    else {                                                                          // :82 out_color
                        r0 = fmaf(SM[I::W_OC + j], hf[r], r0);
                        r1 = fmaf(SM[I::W_OC + H + j], hf[r], r1);
                        r2 = fmaf(SM[I::W_OC + 2 * H + j], hf[r], r2);
                    }
                }
                r0 += wv::swap_half(r0);
                if (layer == 4) { r1 += wv::swap_half(r1); r2 += wv::swap_half(r2); }
                if (LD::HP_ALIAS) {                                      // the cells overlay images still in use: kept until flush_heads()
                    // opaque: evaluated HERE (otherwise the sums sink into the guarded flush and keep hf alive over a whole layer)
                    if (layer == 3) hpart[st][0] = wv::opaque(r0);
                    else { hpart[st][1] = wv::opaque(r0); hpart[st][2] = wv::opaque(r1); hpart[st][3] = wv::opaque(r2); }
                } else if (hi == 0) {
                    float* cell = hp + ((wave * LD::TT + st) * 32 + p31) * 4;
                    if (layer == 3) cell[0] = r0;
                    else { cell[1] = r0; cell[2] = r1; cell[3] = r2; }
                }
            }
            split_planes<16, 3>(hf, ph, pm, pl);
            if (layer < 4) put_image<3, I::XCH>(act_own + st * LD::ACT_ST, ph, pm, pl);
            if (BWD) {
                if (layer >= KEEP0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { keep_h[layer - KEEP0 < 0 ? 0 : layer - KEEP0][st][i] = ph[i]; keep_m[layer - KEEP0 < 0 ? 0 : layer - KEEP0][st][i] = pm[i]; }
                } else acts_store<LD::TT, LD::CHUNK>(acts, tid16, layer, st, ph, pm);
            }
        }
    };
    auto wchunk = [&](int base, int ks, int s) { return gW + ((long long)(base + wave * ks + s)) * I::XCH; };     // wave-uniform
    WPre pre;
    if (own) {
#pragma unroll
        for (int st = 0; st < NT; ++st) zero_acc(acc[st]);               // :59 in_layer (bias rides in the constant-1 column)
        fwd_run<W3, 6, 0, false, NT>(acc, pre_in, wchunk(I::CW_IN, I::KS_IN, 0), e1x, LD::E_ST, nullptr, nullptr, 0, vlo16);
        wpre_load<W3, 0, JS>(pre, nullptr, wchunk(I::CW_M1, I::KS_M, 0), vlo16);
        epilogue(0);
    }
    __syncthreads();
    WS_MARK(2);
    if (own) {
        load_bias(acc[0], SM + I::B_M1 + 32 * wave, hi);                 // :60 mid1
#pragma unroll
        for (int st = 1; st < NT; ++st) acc[st] = acc[0];
        fwd_run<W3, 0, JS, false, NT>(acc, pre, nullptr, nullptr, 0, wchunk(I::CW_M1, I::KS_M, 0), actx, LD::ACT_ST, vlo16);
        wpre_load<W3, 6, JS>(pre, wchunk(I::CW_CAT, I::KS_CAT, JS), wchunk(I::CW_CAT, I::KS_CAT, 0), vlo16);
    }
    __syncthreads();                                                     // everybody has read h1
    if (own) epilogue(1);
    __syncthreads();
    WS_MARK(3);
    if (own) {
#pragma unroll
        for (int st = 0; st < NT; ++st) zero_acc(acc[st]);               // :63-64 cat_layer: encoding part, then h2
        fwd_run<W3, 6, JS, false, NT>(acc, pre, wchunk(I::CW_CAT, I::KS_CAT, JS), e1x, LD::E_ST, wchunk(I::CW_CAT, I::KS_CAT, 0), actx, LD::ACT_ST, vlo16);
        wpre_load<W3, 0, JS>(pre, nullptr, wchunk(I::CW_M2, I::KS_M, 0), vlo16);
    }
    __syncthreads();
    if (own) epilogue(2);
    __syncthreads();
    WS_MARK(4);
    if (own) {
        load_bias(acc[0], SM + I::B_M2 + 32 * wave, hi);                 // :67 mid2
#pragma unroll
        for (int st = 1; st < NT; ++st) acc[st] = acc[0];
        fwd_run<W3, 0, JS, false, NT>(acc, pre, nullptr, nullptr, 0, wchunk(I::CW_M2, I::KS_M, 0), actx, LD::ACT_ST, vlo16);
        wpre_load<W3, 3, JS>(pre, wchunk(I::CW_C, I::KS_C, JS), wchunk(I::CW_C, I::KS_C, 0), vlo16);
    }
    __syncthreads();
    if (own) epilogue(3);
    __syncthreads();
    WS_MARK(5);
    if (own) {
#pragma unroll
        for (int st = 0; st < NT; ++st) zero_acc(acc[st]);               // :81 color_linear: second encoding group, then h4
        fwd_run<W3, 3, JS, false, NT>(acc, pre, wchunk(I::CW_C, I::KS_C, JS), e2x, LD::E_ST, wchunk(I::CW_C, I::KS_C, 0), actx, LD::ACT_ST, vlo16);
        epilogue(4);
    }
    if (LD::HP_ALIAS) {
        __syncthreads();                                                 // everybody has read h4: its images make room for the heads' cells
        if (own && hi == 0) {
#pragma unroll
            for (int st = 0; st < NT; ++st)
                *reinterpret_cast<wv::f32x4*>(hp + ((wave * LD::TT + st) * 32 + p31) * 4) = wv::f32x4{hpart[st][0], hpart[st][1], hpart[st][2], hpart[st][3]};
        }
    }
    __syncthreads();
    WS_MARK(6);
    if (wave < NT && hi == 0) {                                          // heads of tile `wave`: sum of the four waves' partials
        const int pt = 32 * wave + p31;
        if (pt < npts) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = hp[((0 * LD::TT + wave) * 32 + p31) * 4 + c] + hp[((1 * LD::TT + wave) * 32 + p31) * 4 + c];
                if (NB >= 4) v[c] += hp[((2 * LD::TT + wave) * 32 + p31) * 4 + c] + hp[((3 * LD::TT + wave) * 32 + p31) * 4 + c];
                if (NB == 8) v[c] += (hp[((4 * LD::TT + wave) * 32 + p31) * 4 + c] + hp[((5 * LD::TT + wave) * 32 + p31) * 4 + c]) +
                                     (hp[((6 * LD::TT + wave) * 32 + p31) * 4 + c] + hp[((7 * LD::TT + wave) * 32 + p31) * 4 + c]);
            }
            float* row = cb + pt * 8;
            row[6] = zv;
            row[0] = sigmoidf_acc((v[0] + SM[I::B_A]) * 10.0f);           // :77 raw*10 ; render_rays.py:6
            row[1] = sigmoidf_acc(v[1] + SM[I::B_OC]);                    // :83
            row[2] = sigmoidf_acc(v[2] + SM[I::B_OC + 1]);
            row[3] = sigmoidf_acc(v[3] + SM[I::B_OC + 2]);
        }
    }
    __syncthreads();
    {
        const StepArgs& al = wv::kernarg_late(ga).s;
        composite_phase<BWD, NWV>(al, cb, loss_cells, obj, ray0, nrays, wave, lane, tid, rmeta);
    }
    __syncthreads();
    WS_MARK(7);
    if (BWD) {
    // ---- backward ----
    unsigned ah[kWsT][8], am[kWsT][8];                                         // planes of an activation block coming back from the scratch
    auto fetch = [&](int layer) {
        if (layer >= KEEP0) {
#pragma unroll
            for (int st = 0; st < NT; ++st)
#pragma unroll
                for (int i = 0; i < 8; ++i) { ah[st][i] = keep_h[layer - KEEP0 < 0 ? 0 : layer - KEEP0][st][i]; am[st][i] = keep_m[layer - KEEP0 < 0 ? 0 : layer - KEEP0][st][i]; }
            return;
        }
#pragma unroll
        for (int st = 0; st < NT; ++st) { acts_load_plane<LD::TT, LD::CHUNK>(ah[st], acts, tid16, layer, st, 0); acts_load_plane<LD::TT, LD::CHUNK>(am[st], acts, tid16, layer, st, 1); }
        wv::sched_fence();
    };
    if (own) fetch(3);                                                   // h4: lands during the encoding transposes
    float d_raw[kWsT], d_c0[kWsT], d_c1[kWsT], d_c2[kWsT];
#pragma unroll
    for (int st = 0; st < NT; ++st) {
        const float* row = cb + (32 * st + p31) * 8;                     // padding rows hold zeros
        d_raw[st] = row[0]; d_c0[st] = row[1]; d_c1[st] = row[2]; d_c2[st] = row[3];
    }
    // F-form images of the ten encoding blocks (weight-gradient operands), from the P-form images of the forward; dealt round-robin
    {
        char* ef = lds + LD::EF + lo16;
#pragma unroll
        for (int j = 0; j < 5 * NT; ++j) {
            if (j % NWV != wave) continue;
            const int st = j / 5, eb = j - 5 * st;
            const char* src = eb < 3 ? e1x + st * LD::E_ST + 2 * eb * I::XCH : e2x + st * LD::E_ST + 2 * (eb - 3) * I::XCH;
            unsigned h[8], m[8], f[16];
            const u32x4 h0 = lds16(src), m0 = lds16(src + 1024);
            h[0] = h0[0]; h[1] = h0[1]; h[2] = h0[2]; h[3] = h0[3]; m[0] = m0[0]; m[1] = m0[1]; m[2] = m0[2]; m[3] = m0[3];
            if (eb < 4) {
                const u32x4 h1 = lds16(src + I::XCH), m1 = lds16(src + I::XCH + 1024);
                h[4] = h1[0]; h[5] = h1[1]; h[6] = h1[2]; h[7] = h1[3]; m[4] = m1[0]; m[5] = m1[1]; m[6] = m1[2]; m[7] = m1[3];
            } else {
                h[4] = h[5] = h[6] = h[7] = 0u; m[4] = m[5] = m[6] = m[7] = 0u;
            }
            to_F<4>(f, tile, h, m, p31, hi, TL);
            if (LD::EF2_GLOBAL && eb >= 3) put_F_g(wgs + LD::EF2_OFF + (st * 2 + eb - 3) * 4096, vlo16, f);
            else put_F(ef + st * LD::EF_ST + eb * 4096, f);
        }
    }
    // F-form of the heads' delta (features 0..3 = d raw alpha, d raw colour), both tiles: every wave for itself
    unsigned dF[kWsT][16];
    float dv[kWsT][16];
#pragma unroll
    for (int st = 0; st < NT; ++st) {
        unsigned dh[8], dm[8], dl[8];
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[st][r] = 0.0f;
        if (hi == 0) { dv[st][0] = d_raw[st]; dv[st][1] = d_c0[st]; dv[st][2] = d_c1[st]; dv[st][3] = d_c2[st]; }
        split_planes<16, 2>(dv[st], dh, dm, dl);
        to_F<4>(dF[st], tile, dh, dm, p31, hi, TL);
    }
    __syncthreads();                                                     // encoding F images complete; the P-form images are dead
    WS_MARK(8);
    using RW = RowWs<NB>;
    constexpr int AH = NB > 4 ? VK_WS8_AH : 2;                             // dw_layer: hidden 256 keeps ONE set of earlier sums in flight
    float* outR = out + RW::slot(0, wave, 0);                           // this output block's slots of the workgroup's row
    const unsigned lane4 = 4u * (unsigned)lane;
    char* dlt_own = lds + LD::DLT + wave * 2 * I::DCH + lo16;
    const char* dltx = lds + LD::DLT + lo16;
    char* xf_own = lds + LD::XF + wave * 4096 + lo16;
    const char* xfx = lds + LD::XF + lo16;
    const char* efx = lds + LD::EF + lo16;
    f32x16 accw, accd[kWsT];
    float dproj[kWsT][11];
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int i = 0; i < 11; ++i) dproj[st][i] = 0.0f;
    // the fetched activation block (ah, am) -> its F-form images, the layer input of everybody's weight gradients
    auto publish_x = [&]() {
#pragma unroll
        for (int st = 0; st < NT; ++st) {
            unsigned xF[16];
            to_F<4>(xF, tile, ah[st], am[st], p31, hi, TL);
            put_F(xf_own + st * LD::XF_ST, xF);
        }
    };
    // the wave's delta block (float32 registers dv[st]) -> planes -> P-form image + F-form registers dF
    auto publish_d = [&]() {
#pragma unroll
        for (int st = 0; st < NT; ++st) {
            unsigned dh[8], dm[8], dl[8];
            split_planes<16, 2>(dv[st], dh, dm, dl);
            put_image<2, I::DCH>(dlt_own + st * LD::DLT_ST, dh, dm, dl);
            to_F<4>(dF[st], tile, dh, dm, p31, hi, TL);
        }
    };
    // d-prop into the wave's own hidden block of layer ct (both tiles); the first W^T chunks (tp) were fetched at the phase start
    TPre tp, tpe;
    auto hidden_ptr = [&](int ct_base) {
        return gWT + ((long long)(ct_base + wave * JS)) * I::DCH;
    };
    auto dprop_hidden = [&](int ct_base, bool add_alpha) {
#pragma unroll
        for (int st = 0; st < NT; ++st) {
            if (add_alpha) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accd[st][r] = SM[I::W_A + 32 * wave + phi(r, hi)] * d_raw[st];
            } else zero_acc(accd[st]);
        }
        bwd_run<W3, JS, NT>(accd, tp, hidden_ptr(ct_base), vlo16, dltx, LD::DLT_ST);
    };
    // d-prop into an encoding block -> d(proj) through the cos factors (cfr: fetched at the phase start, with tpe)
    float cfr[kWsT][16];
    auto enc_ptr = [&](int ct_chunk) {
        return gWT + (long long)ct_chunk * I::DCH;
    };
    auto enc_fetch = [&](int ct_chunk, int group, int blk) {
        tpre_load<W3>(tpe, enc_ptr(ct_chunk), vlo16);
#pragma unroll
        for (int st = 0; st < NT; ++st) {
            // (this kernel keeps the [66][64 lanes] form of the tile: the vector form - cf_load, step_main_wp - costs it registers and time,
            // background 69.0 -> 71.9 us: profiles/round6d_cos_factor_layout_ab.jsonl)
            const char* cfu = reinterpret_cast<const char*>(cfs + st * kCfTile);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = 16 * blk + r;
                const int idx = group == 1 ? (R < 44 ? 6 * (R >> 2) + (R & 3) : -1) : (R < 22 ? 6 * (R >> 1) + 4 + (R & 1) : -1);
                cfr[st][r] = idx >= 0 ? *reinterpret_cast<const float*>(cfu + idx * 256 + (unsigned)lane * 4u) : 0.0f;
            }
        }
        wv::sched_fence();
    };
    auto dprop_enc = [&](int ct_chunk, int group, int blk) {
        f32x16 acce[kWsT];
#pragma unroll
        for (int st = 0; st < NT; ++st) zero_acc(acce[st]);
        bwd_run<W3, JS, NT>(acce, tpe, enc_ptr(ct_chunk), vlo16, dltx, LD::DLT_ST);
#pragma unroll
        for (int st = 0; st < NT; ++st) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = 16 * blk + r;
                if (group == 1) { if (R < 44) dproj[st][R >> 2] = fmaf(acce[st][r], cfr[st][r], dproj[st][R >> 2]); }
                else { if (R < 22) dproj[st][R >> 1] = fmaf(acce[st][r], cfr[st][r], dproj[st][R >> 1]); }
            }
        }
    };
    auto xf_img = [&](int kb, const char*& p, int& st) { p = xfx + kb * 4096; st = LD::XF_ST; };
    // -- heads: d W_a = (d raw)^T h4, d W_oc = (d colour)^T hc; rows 0..3 of one block each --
    if (own) {
        FImg xi;
        publish_x();                                                     // h4 block: color_linear's weight-gradient operand
        fetch(4);                                                        // hc
        wv::wave_lds_fence();
        fimg_load<NT>(xi, xf_own, LD::XF_ST);
        dw_mm_pair<NT>(accw, dF, xi);
        if (hi == 0) store_one(out + RW::W_A + 32 * wave + p31, accw[0], first);
        if (wave == 0) {
            f32x16 accb;
            db_pair<NT>(accb, dF);
            if (lane == 0) {
                store_one(out + RW::B_A, accb[0], first);
                store_one(out + RW::B_OC + 0, accb[1], first); store_one(out + RW::B_OC + 1, accb[2], first); store_one(out + RW::B_OC + 2, accb[3], first);
            }
        }
        zero_acc(accw);
#pragma unroll
        for (int st = 0; st < NT; ++st) {
            unsigned x0[16];
            to_F<4>(x0, tile, ah[st], am[st], p31, hi, TL);
            dw_mm_s(accw, dF[st], x0);
        }
        if (hi == 0) {
            store_one(out + RW::W_OC + 0 * H + 32 * wave + p31, accw[1], first);
            store_one(out + RW::W_OC + 1 * H + 32 * wave + p31, accw[2], first);
            store_one(out + RW::W_OC + 2 * H + 32 * wave + p31, accw[3], first);
        }
    }
    // -- delta 0 = d hc (through the ReLU; ah = hc's hi plane) --
    if (own) {
#pragma unroll
        for (int st = 0; st < NT; ++st) {
            f32x16 v;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * wave + phi(r, hi);
                v[r] = SM[I::W_OC + j] * d_c0[st] + SM[I::W_OC + H + j] * d_c1[st] + SM[I::W_OC + 2 * H + j] * d_c2[st];
            }
            mask_by(dv[st], v, ah[st]);
        }
        publish_d();
        fetch(3);                                                        // h4 again: the mask of delta 1
    }
    __syncthreads();
    WS_MARK(9);
    WS_DMARK(0);
    // color_linear: weight gradients (h4 blocks, second-group blocks + bias column), d-prop -> d h4 (+ W_a d raw), d(second group)
    constexpr int EC0 = NB >= 4 ? 0 : 2, EC1 = NB >= 4 ? 1 : 3;          // the waves that d-prop color_linear's two encoding blocks
    if (own) tpre_load<W3>(tp, hidden_ptr(I::CT_C), vlo16);
    if (wave == EC0) enc_fetch(I::CT_C + (NB + 0) * JS, 2, 0);
    if (wave == EC1) enc_fetch(I::CT_C + (NB + 1) * JS, 2, 1);
    WS_DMARK(1);
    if (own)
        dw_layer<NB + 2, NT, LD::EF2_GLOBAL ? NB : 99, AH>(dF, first,
            [&](int kb, const char*& p, int& st) {
                if (kb < NB) xf_img(kb, p, st);
                else if (LD::EF2_GLOBAL) { p = wgs + LD::EF2_OFF + (kb - NB) * 4096; st = 2 * 4096; }
                else { p = efx + (3 + kb - NB) * 4096; st = LD::EF_ST; }
            },
            [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
                WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_C, 0, kb), lane4, v, old)));
            }, vlo16);
    WS_DMARK(2);
    if (wave == EC0) dprop_enc(I::CT_C + (NB + 0) * JS, 2, 0);
    if (wave == EC1) dprop_enc(I::CT_C + (NB + 1) * JS, 2, 1);
    WS_DMARK(3);
    if (own) {
        dprop_hidden(I::CT_C, true);
        WS_DMARK(4);
#pragma unroll
        for (int st = 0; st < NT; ++st) mask_by(dv[st], accd[st], ah[st]);  // delta 1 = d h4
        fetch(2);                                                        // h3: mid2's input and the mask of delta 2
        tpre_load<W3>(tp, hidden_ptr(I::CT_M2), vlo16);
    }
    WS_DMARK(5);
    __syncthreads();
    WS_DMARK(6);
    if (own) publish_d();
    WS_DMARK(7);
    if (own) publish_x();
    WS_DMARK(8);
    __syncthreads();
    WS_DMARK(9);
    WS_MARK(10);
    // mid2
    if (own) {
        dw_layer<NB, NT, 99, AH>(dF, first, xf_img, [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
            WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_M2, 0, kb), lane4, v, old)));
        });
        WS_DMARK(10);
        db_pair<NT>(accw, dF);
        if (p31 == 0) store_rows<1>(out + RW::B_M2 + 32 * wave, (unsigned)(4 * hi), accw, first);
        WS_DMARK(11);
        dprop_hidden(I::CT_M2, false);
        WS_DMARK(12);
#pragma unroll
        for (int st = 0; st < NT; ++st) mask_by(dv[st], accd[st], ah[st]);  // delta 2 = d h3
        WS_DMARK(13);
        fetch(1);                                                        // h2
        tpre_load<W3>(tp, hidden_ptr(I::CT_CAT), vlo16);
    }
    WS_DMARK(14);
    if (wave == 1) enc_fetch(I::CT_CAT + (NB + 0) * JS, 1, 0);
    if (wave == 2) enc_fetch(I::CT_CAT + (NB + 1) * JS, 1, 1);
    if (wave == 3) enc_fetch(I::CT_CAT + (NB + 2) * JS, 1, 2);
    __syncthreads();
    if (own) { publish_d(); publish_x(); }
    __syncthreads();
    WS_MARK(11);
    WS_DMARK(15);
    // cat_layer
    if (own)
        dw_layer<NB + 3, NT, 99, AH>(dF, first,
            [&](int kb, const char*& p, int& st) { if (kb < NB) xf_img(kb, p, st); else { p = efx + (kb - NB) * 4096; st = LD::EF_ST; } },
            [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
                WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_CAT, 0, kb), lane4, v, old)));
            });
    if (wave == 1) dprop_enc(I::CT_CAT + (NB + 0) * JS, 1, 0);
    if (wave == 2) dprop_enc(I::CT_CAT + (NB + 1) * JS, 1, 1);
    if (wave == 3) dprop_enc(I::CT_CAT + (NB + 2) * JS, 1, 2);
    if (own) {
        dprop_hidden(I::CT_CAT, false);
#pragma unroll
        for (int st = 0; st < NT; ++st) mask_by(dv[st], accd[st], ah[st]);  // delta 3 = d h2
        fetch(0);                                                        // h1
        tpre_load<W3>(tp, hidden_ptr(I::CT_M1), vlo16);
    }
    __syncthreads();
    if (own) { publish_d(); publish_x(); }
    __syncthreads();
    WS_MARK(12);
    // mid1
    if (own) {
        dw_layer<NB, NT, 99, AH>(dF, first, xf_img, [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
            WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_M1, 0, kb), lane4, v, old)));
        });
        db_pair<NT>(accw, dF);
        if (p31 == 0) store_rows<1>(out + RW::B_M1 + 32 * wave, (unsigned)(4 * hi), accw, first);
        dprop_hidden(I::CT_M1, false);
#pragma unroll
        for (int st = 0; st < NT; ++st) mask_by(dv[st], accd[st], ah[st]);  // delta 4 = d h1
    }
    if (wave == 0) enc_fetch(I::CT_IN + 0 * JS, 1, 0);
    if (wave == 2) enc_fetch(I::CT_IN + 1 * JS, 1, 1);
    if (wave == 3) enc_fetch(I::CT_IN + 2 * JS, 1, 2);
    __syncthreads();
    if (own) publish_d();
    __syncthreads();
    WS_MARK(13);
    // in_layer
    if (own)
        dw_layer<3, NT, 99, AH>(dF, first, [&](int kb, const char*& p, int& st) { p = efx + kb * 4096; st = LD::EF_ST; },
                    [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
                        WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_IN, 0, kb), lane4, v, old)));
                    });
    if (wave == 0) dprop_enc(I::CT_IN + 0 * JS, 1, 0);
    if (wave == 2) dprop_enc(I::CT_IN + 1 * JS, 1, 1);
    if (wave == 3) dprop_enc(I::CT_IN + 2 * JS, 1, 2);
    // B_layer.weight: dB[d][j] = sum_points d(proj)[d] t[j].  d(proj) = sum of the four waves' parts (exchange through LDS);
    // t = the x, y, z slots of the hi = 0 lanes = columns 24..26 of first-group block 2
    __syncthreads();
    WS_MARK(14);
    {
        float* px = reinterpret_cast<float*>(lds + LD::XF);               // [wave][tile][11][64]
#pragma unroll
        for (int st = 0; st < NT; ++st)
#pragma unroll
            for (int i = 0; i < 11; ++i) px[((wave * LD::TT + st) * 11 + i) * 64 + lane] = dproj[st][i];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int st = 0; st < NT; ++st) {
                unsigned dh[8], dm[8], dl[8];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dv[st][r] = r < 11 ? (px[((0 * LD::TT + st) * 11 + r) * 64 + lane] + px[((1 * LD::TT + st) * 11 + r) * 64 + lane]) +
                                         (px[((2 * LD::TT + st) * 11 + r) * 64 + lane] + px[((3 * LD::TT + st) * 11 + r) * 64 + lane]) : 0.0f;
                split_planes<16, 2>(dv[st], dh, dm, dl);
                to_F<4>(dF[st], tile, dh, dm, p31, hi, TL);
            }
            FImg xi;
            fimg_load<NT>(xi, efx + 2 * 4096, LD::EF_ST);
            dw_mm_pair<NT>(accw, dF, xi);
            if (p31 >= 24 && p31 < 27) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = hi ? 11 + r : r;                          // row phi(r, hi) <-> direction
                    if (r < (hi ? 10 : 11)) store_one(out + RW::PE_B + 3 * d + (p31 - 24), accw[r], first);
                }
            }
        }
    }
    WS_MARK(15);
    }   // BWD
    }   // rounds
#undef WS_MARK
#undef WS_DMARK
    __syncthreads();
    if (tid_k == 0) {
        float* pl = a.part_loss + (obj * a.NW + wgo) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pl[k] = (loss_cells[k] + loss_cells[4 + k]) + (loss_cells[8 + k] + loss_cells[12 + k]);
            if (NWV == 8) pl[k] += (loss_cells[16 + k] + loss_cells[20 + k]) + (loss_cells[24 + k] + loss_cells[28 + k]);
        }
        pl[3] = 0.0f;
    }
}

}  // namespace vk
