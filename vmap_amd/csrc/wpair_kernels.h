// wpair_kernels.h - step_main_wp<NB>: the step_main_ws scheme (wsplit_kernels.h: same images, same numerics, same round of two
// 32-point tiles) with TWO waves per output block.  step_main_ws runs one wave per SIMD: nothing overlaps a wave's exposed
// latencies (its matrix pipe is 26 % busy), and at hidden 64 only two of the four waves own an output block.  Here wave
// w = (output block ob = w % NB, half kh = w / NB), 2 NB waves per workgroup (hidden 128: 8 waves = two per SIMD):
//
//   * a layer's 16-deep steps (and a d-prop chain's) are split between the two waves of a block - each still feeds both tiles
//     from one weight operand, so the operand traffic per matrix instruction is unchanged;
//   * the two partial sums meet through LDS: wave kh sends its partial of tile 1 - kh, receives the partner's partial of tile
//     kh and finishes tile kh alone (ReLU, split into planes, images) - the exchange rides on the barrier a layer has anyway;
//   * weight-gradient blocks of a block row are dealt alternately to the two waves; each needs the delta's F-form of both tiles:
//     its own tile's from its registers, the partner's rebuilt from the P-form image the partner published;
//   * the encoding blocks' F images live in the workgroup's L2 scratch (LDS holds the partial-sum exchange instead).
#pragma once
#include <type_traits>

#include "wsplit_kernels.h"

namespace vk {

template <int NB>
struct LdsWp {
    using I = ImgWs<NB>;
    static constexpr int NWV = 2 * NB, NTH = 64 * NWV;
    static constexpr bool EG = NB == 2;      // hidden 64: the encoding's P-form images live in the L2 scratch too - 78 KB of LDS, two workgroups per CU
    static constexpr int ACT = 0, ACT_ST = I::ACT_ST, ACT_BYTES = I::ACT_BYTES;   // forward: layer input images
    static constexpr int XCG = 0, XCG_W = 4096;                                    // backward: partial-sum exchange, one slot per wave
    static constexpr int PX_BYTES = NWV * 2 * 11 * 64 * 4;                         // end of the backward: d(proj) exchange
    static constexpr int R0a = ACT_BYTES > NWV * XCG_W ? ACT_BYTES : NWV * XCG_W;
    static constexpr int R0_BYTES = R0a > PX_BYTES ? R0a : PX_BYTES;
    static constexpr int EIM = R0_BYTES, E_ST = I::E_ST, E2_OFF = I::E2_OFF;       // forward: encoding images
    static constexpr int DLT = EIM, DLT_ST = I::DLT_ST;                            // backward: delta images
    static constexpr int XF = DLT + 2 * DLT_ST, XF_ST = I::XF_ST;                  // backward: F-form images of the layer input
    static constexpr int R1_BYTES = !EG && 2 * E_ST > 2 * DLT_ST + 2 * XF_ST ? 2 * E_ST : 2 * DLT_ST + 2 * XF_ST;
    static constexpr int SCRT = EIM + R1_BYTES;                                    // one transpose tile per wave (forward: the exchange slot)
    static constexpr int HP = SCRT + NWV * Img32s::TILE;                           // head partial sums [block][tile][32 points][4]
    static constexpr int HX = XCG + NWV * XCG_W;                                   // backward (behind the exchange slots): head-gradient partials of the kh = 1 waves [block][8][64]
    static constexpr int CBO = HP + NB * 2 * 32 * 4 * 4;
    static constexpr int LOSS = CBO + 64 * 8 * 4;
    static constexpr int LDS_BYTES = LOSS + kWaves * 4 * 4;
    // scratch per workgroup (global memory, L2): cos factors [tile][66][lane] | activation planes [layer][tile][plane][step][block][lane]
    // | F-form images of the encoding blocks [tile][5][4 KiB]
    static constexpr int ACTS_OFF = I::ACTS_OFF;
    static constexpr int ACTS_CH = NB * 1024;                                      // one (layer, tile, plane, step) chunk: NB blocks x 64 lanes x 16 B
    static constexpr int EFG_OFF = ACTS_OFF + 5 * 2 * 2 * 2 * ACTS_CH;
    static constexpr int EIG_OFF = EFG_OFF + 2 * 5 * 4096;                         // (EG) encoding P-form images [tile][9 steps][3 planes][1 KiB]
    static constexpr int WG_SCRATCH = EIG_OFF + (EG ? 2 * E_ST : 0);
};
static_assert(LdsWp<4>::LDS_BYTES <= 160 * 1024 && LdsWp<2>::LDS_BYTES <= 80 * 1024, "LDS budget (hidden 64: two workgroups per CU)");
static_assert(LdsWp<4>::HX + 4 * 8 * 64 * 4 <= LdsWp<4>::R0_BYTES && LdsWp<2>::HX + 2 * 8 * 64 * 4 <= LdsWp<2>::R0_BYTES, "head partials fit behind the exchange slots");

// partial sum of one tile -> the wave's exchange slot / + the partner's slot
__device__ __forceinline__ void acc_send(char* slot, const f32x16& a) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
        *reinterpret_cast<wv::f32x4*>(slot + c * 1024) = wv::f32x4{a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]};
}
__device__ __forceinline__ void acc_recv_add(f32x16& a, const char* slot) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const wv::f32x4 v = *reinterpret_cast<const wv::f32x4*>(slot + c * 1024);
        a[4 * c] += v[0]; a[4 * c + 1] += v[1]; a[4 * c + 2] += v[2]; a[4 * c + 3] += v[3];
    }
}
// weight-gradient blocks kb = kh, kh + 2, ... < N of one layer; xload(img, kb) fills the block's two F images, io(mode, kb, acc,
// old): slot_io of the block.  One block at a time (the partner wave on the SIMD covers the LDS round trip; registers are
// 256 per wave here); later rounds read the earlier sums in front of the matrix instructions.
template <int N, class XL, class IO>
__device__ __forceinline__ void dw_layer_half(const unsigned (&dF)[2][16], bool first, int kh, XL&& xload, IO&& io) {
    constexpr int NH = (N + 1) / 2;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int kb = 2 * i + kh;
        if (kb < N) {
            FImg x;
            f32x16 acc;
            float old[16];
            xload(x, kb);
            if (!first) io(1, kb, acc, old);
            wv::sched_fence();
            dw_mm_pair<2, kRowT>(acc, dF, x);
            io(first ? 0 : 2, kb, acc, old);
        }
    }
}

template <int NB, bool BWD, bool W3, bool STAMPS = false>
__global__ __launch_bounds__(128 * NB, 2) void step_main_wp(const WsArgs ga) {
    static_assert(NB == 4 || NB == 2, "two waves per output block, at most eight waves");
    using I = ImgWs<NB>;
    using LD = LdsWp<NB>;
    constexpr int H = I::H, JS = I::JS, NWV = LD::NWV, NTH = LD::NTH;
    const StepArgs& a = ga.s;
    char* lds = reinterpret_cast<char*>(wv::lds_base());
    const int tid_k = threadIdx.x;
    // xcd_affine (the launcher's choice, from eight objects on): an object's workgroups all on ONE XCD (block b runs on XCD b % 8), so the
    // object's weight images are fetched into one L2 instead of into as many as it has workgroups
    int obj, wgo;
    if (a.xcd_affine) {
        const int slot = blockIdx.x >> 3, og = slot / a.NW;
        obj = og * 8 + (blockIdx.x & 7);
        wgo = slot - og * a.NW;
        if (obj >= a.n_obj) return;
    } else {
        obj = blockIdx.x / a.NW;
        wgo = blockIdx.x - obj * a.NW;
    }
    const int wg_index = obj * a.NW + wgo;               // this workgroup's row / scratch / stamp slot (the block index without the map)
    const char* gimg = reinterpret_cast<const char*>(a.wimg) + (long long)obj * I::BYTES;
    const float* SM = reinterpret_cast<const float*>(gimg + I::SMALL_OFF);
    float* loss_cells = reinterpret_cast<float*>(lds + LD::LOSS);
    if (tid_k < kWaves * 4) loss_cells[tid_k] = 0.0f;
    float* out_k = a.part_grad + (long long)wg_index * a.PR;
    float* cb = reinterpret_cast<float*>(lds + LD::CBO);
    float* hp = reinterpret_cast<float*>(lds + LD::HP);
    float* hx = reinterpret_cast<float*>(lds + LD::HX);
    const float scale = a.pe_scale.p[obj * a.pe_scale.stride];
    const float* Bg = SM + I::PE_B;
    char* wgs_k = ga.scratch + (long long)wg_index * LD::WG_SCRATCH;
    unsigned* tmark = STAMPS && a.timing ? a.timing + ((long long)wg_index * kWaves + ((tid_k >> 6) & 3)) * kMarks : nullptr;
#define WP_MARK(i) do { if constexpr (STAMPS) { if (tmark && first && tid_k < 256 && (tid_k & 63) == 0) tmark[i] = wv::clock32(); } } while (0)

    for (int grp = wgo; grp < a.NG; grp += a.NW) {
    const bool first = grp == wgo;
    const unsigned uz = wv::opaque_uzero();
    const char* gW = gimg + uz;
    const char* gWT = gimg + I::WT_OFF + uz;
    float* out = out_k + uz;
    char* wgs = wgs_k + uz;
    float* cfs = reinterpret_cast<float*>(wgs);
    WP_MARK(0);
    const int tid = wv::opaque_iter(tid_k), lane = tid & 63, wave = wv::uniform(tid >> 6), p31 = lane & 31, hi = lane >> 5;
    const int ob = wave % NB, kh = wave / NB;                            // output block, half (= the tile this wave finishes)
    const TrLane TL = tr_lane(lane);
    char* tile = lds + LD::SCRT + wave * Img32s::TILE;
    const char* tile_partner = lds + LD::SCRT + (ob + (1 - kh) * NB) * Img32s::TILE;
    char* acts = wgs + LD::ACTS_OFF;
    const unsigned ob16 = (unsigned)(ob * 64 + lane) * 16u;              // this lane's place inside an activation-plane chunk
    const int lo16 = lane * 16;
    const unsigned vlo16 = (unsigned)lane * 16u;
    __syncthreads();                                                     // previous round done with LDS
    for (int i = tid; i < I::kPts * 8; i += NTH) cb[i] = 0.0f;
    const int ray0 = grp * a.G;
    const int nrays = min(a.G, a.R - ray0);
    const int npts = nrays * a.S;                                        // <= 64
    float zv = 0.0f;
    if (wave < 2 && hi == 0 && 32 * wave + p31 < npts) {
        const int pt = 32 * wave + p31, lray = pt / a.S, smp = pt - lray * a.S;
        zv = a.z[obj * a.z_so + (ray0 + lray) * a.z_sr + smp * a.z_ss];
    }
    RayMeta rmeta{};
    if (wave < kWaves) rmeta = load_ray_meta(a, obj, ray0 + min(4 * wave + (lane >> 4), nrays - 1));
    // ---- forward step lists: a layer = NE encoding steps (weights at chunk JSoff.., inputs from the encoding images) followed by
    //      NH hidden steps; wave kh takes the first / second half of the concatenated list ----
    constexpr bool EG = LD::EG;
    // encoding images: LDS (lane offset included) or the workgroup's scratch (wave-uniform base, lane offset = vlo16)
    char* eimg = EG ? wgs + LD::EIG_OFF : lds + LD::EIM + lo16;
    const unsigned evo = EG ? vlo16 : 0u;
    const char* e1x = eimg;
    const char* e2x = e1x + LD::E2_OFF;
    const char* actx = lds + LD::ACT + lo16;
    auto wchunk = [&](int base, int ks, int s) __attribute__((always_inline)) { return gW + ((long long)(base + ob * ks + s)) * I::XCH; };     // wave-uniform
    f32x16 acc[2];
    WPre pre;
    // NE / NH: steps of the two parts; we / wh: weight chunk of step 0 of each part; xe / xh: input chunk of step 0
#define WP_SPLIT(NE, NH)                                                                                   \
    constexpr int TOT = (NE) + (NH), H0 = (TOT + 1) / 2, A0 = (NE) < H0 ? (NE) : H0, B0 = H0 - A0, A1 = (NE) - A0, B1 = (NH) - B0
    auto layer_pre = [&](auto ne_c, auto nh_c, const char* we, const char* wh) __attribute__((always_inline)) {
        constexpr int NE = decltype(ne_c)::value, NH = decltype(nh_c)::value;
        WP_SPLIT(NE, NH);
        if (kh == 0) wpre_load<W3, A0, B0>(pre, we, wh, vlo16);
        else wpre_load<W3, A1, B1>(pre, we + A0 * I::XCH, wh + B0 * I::XCH, vlo16);
    };
    auto layer_run = [&](auto ne_c, auto nh_c, const char* we, const char* xe, const char* wh, const char* xh) __attribute__((always_inline)) {
        constexpr int NE = decltype(ne_c)::value, NH = decltype(nh_c)::value;
        WP_SPLIT(NE, NH);
        // acc[0] = this wave's tile (kh), acc[1] = the partner's: the inputs are read in that order (tile stride +- one tile)
        if (kh == 0) fwd_run<W3, A0, B0, EG>(acc, pre, we, xe, LD::E_ST, wh, xh, LD::ACT_ST, vlo16);
        else fwd_run<W3, A1, B1, EG>(acc, pre, we + A0 * I::XCH, xe + A0 * I::XCH + LD::E_ST, -LD::E_ST, wh + B0 * I::XCH, xh + B0 * I::XCH + LD::ACT_ST, -LD::ACT_ST, vlo16);
    };
    using C0 = std::integral_constant<int, 0>;
    using C3 = std::integral_constant<int, 3>;
    using C6 = std::integral_constant<int, 6>;
    using CJ = std::integral_constant<int, JS>;
    layer_pre(C6{}, C0{}, wchunk(I::CW_IN, I::KS_IN, 0), nullptr);       // in_layer's first chunks: fetched behind the encoding
    // ---- encoding (embedding.py:82-91): wave = (tile est, slot group dq): 12 slots of a lane half dealt to NWV / 2 waves ----
    {
        constexpr int NQ = NWV / 2, SL = 12 / NQ;                        // slot groups, slots per wave (6 or 3)
        const int est = wave & 1, dq = wave >> 1;
        const int pt = 32 * est + p31;
        const bool valid = pt < npts;
        const int lray = valid ? pt / a.S : 0, smp = valid ? pt - lray * a.S : 0, ray = ray0 + lray;
        float px3[3] = {0.0f, 0.0f, 0.0f};
        if (valid) load_point(a, obj, ray, smp, px3[0], px3[1], px3[2]);      // the points tensor, or (o + d z) - c of a ray batch (ABI v7)
        const float t[3] = {px3[0] / scale, px3[1] / scale, px3[2] / scale};          // embedding.py:83
        float proj[SL];
        float amax = 0.0f;
#pragma unroll
        for (int ii = 0; ii < SL; ++ii) {
            const int i = SL * dq + ii;
            const int d = hi ? min(11 + i, 20) : min(i, 10);
            proj[ii] = fmaf(t[2], Bg[3 * d + 2], fmaf(t[1], Bg[3 * d + 1], t[0] * Bg[3 * d]));      // embedding.py:84
            amax = fmaxf(amax, fabsf(proj[ii]));
        }
        const bool fast = !wv::wave_any(!(amax * (32.0f * kPi) < kSinCosFastLimit));
        char* e1img = eimg + est * LD::E_ST + evo;
        char* e2img = e1img + LD::E2_OFF;
        float* cf_out = cfs + est * kCfTile;
        unsigned p1[3][4], p2[3][SL];                                    // (SL == 6) planes of a slot pair of group 1 / of the wave's slots of group 2
#pragma unroll
        for (int ii = 0; ii < SL; ++ii) {
            const int i = SL * dq + ii;
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
                float cv[6];
#pragma unroll
                for (int f = 0; f < 6; ++f) cv[f] = own ? c[f] * (kPi * (float)(1 << f)) : 0.0f;
                cf_store(cf_out, i, lane, cv);
            }
            unsigned h1[2], m1[2], l1[2], h2[1], m2[1], l2[1];
            split_planes<4, 3>(v1, h1, m1, l1);
            split_planes<2, 3>(v2, h2, m2, l2);
            if constexpr (SL == 6) {
                // A lane's 16 bytes of a chunk hold two slots of the first group / four of the second.  Six slots per wave (hidden 64, where
                // these images live in the L2-side scratch): the slots are gathered and leave as whole 16-byte pieces (8 bytes where the
                // wave owns half of a lane's sixteen) - a wave store that leaves gaps in its lines costs the memory system twice (round 6d).
                p1[0][2 * (ii & 1)] = h1[0]; p1[0][2 * (ii & 1) + 1] = h1[1];
                p1[1][2 * (ii & 1)] = m1[0]; p1[1][2 * (ii & 1) + 1] = m1[1];
                p1[2][2 * (ii & 1)] = l1[0]; p1[2][2 * (ii & 1) + 1] = l1[1];
                if (ii & 1) {                                            // (a wave's first slot is even)
                    char* q1 = e1img + (i >> 1) * I::XCH;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4*>(q1 + pl * 1024) = u32x4{p1[pl][0], p1[pl][1], p1[pl][2], p1[pl][3]};
                }
                p2[0][ii] = h2[0]; p2[1][ii] = m2[0]; p2[2][ii] = l2[0];
            } else {
                char* q1 = e1img + (i >> 1) * I::XCH + (i & 1) * 8;
                *reinterpret_cast<u32x2*>(q1) = u32x2{h1[0], h1[1]};
                *reinterpret_cast<u32x2*>(q1 + 1024) = u32x2{m1[0], m1[1]};
                *reinterpret_cast<u32x2*>(q1 + 2048) = u32x2{l1[0], l1[1]};
                char* q2 = e2img + (i >> 2) * I::XCH + (i & 3) * 4;
                *reinterpret_cast<unsigned*>(q2) = h2[0];
                *reinterpret_cast<unsigned*>(q2 + 1024) = m2[0];
                *reinterpret_cast<unsigned*>(q2 + 2048) = l2[0];
            }
        }
        if constexpr (SL == 6) {
            // second group: slots 0..3 | 4, 5 (dq = 0) or 6, 7 | 8..11 (dq = 1) = chunk 0 whole + the first half of chunk 1, or the second
            // half of chunk 1 + chunk 2 whole
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                if (dq == 0) {
                    *reinterpret_cast<u32x4*>(e2img + pl * 1024) = u32x4{p2[pl][0], p2[pl][1], p2[pl][2], p2[pl][3]};
                    *reinterpret_cast<u32x2*>(e2img + I::XCH + pl * 1024) = u32x2{p2[pl][4], p2[pl][5]};
                } else {
                    *reinterpret_cast<u32x2*>(e2img + I::XCH + 8 + pl * 1024) = u32x2{p2[pl][0], p2[pl][1]};
                    *reinterpret_cast<u32x4*>(e2img + 2 * I::XCH + pl * 1024) = u32x4{p2[pl][2], p2[pl][3], p2[pl][4], p2[pl][5]};
                }
            }
        }
    }
    __syncthreads();
    WP_MARK(1);
    // ---- forward (model.py:59-83) ----
    char* act_own = lds + LD::ACT + kh * LD::ACT_ST + ob * 2 * I::XCH + lo16;     // the layer-input image of (tile kh, block ob)
    // after a layer's run: partial of the partner's tile -> exchange slot; (barrier); + the partner's partial of this wave's tile;
    // epilogue of tile kh: ReLU, heads' partial sums, split into planes, planes -> next layer's input image + scratch
    auto send_partial = [&]() __attribute__((always_inline)) { acc_send(tile + lo16, acc[1]); };
    auto finish_tile = [&](int layer) __attribute__((always_inline)) {
        f32x16 v = acc[0];
        acc_recv_add(v, tile_partner + lo16);
        float hf[16];
        unsigned ph[8], pm[8], pl[8];
        relu_to(hf, v);
        if (layer >= 3) {
            float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * ob + phi(r, hi);
                if (layer == 3) r0 = fmaf(SM[I::W_A + j], hf[r], r0);                               // :71 out_alpha
                else {                                                                              // :82 out_color
                    r0 = fmaf(SM[I::W_OC + j], hf[r], r0);
                    r1 = fmaf(SM[I::W_OC + H + j], hf[r], r1);
                    r2 = fmaf(SM[I::W_OC + 2 * H + j], hf[r], r2);
                }
            }
            r0 += wv::swap_half(r0);
            if (layer == 4) { r1 += wv::swap_half(r1); r2 += wv::swap_half(r2); }
            if (hi == 0) {
                float* cell = hp + ((ob * 2 + kh) * 32 + p31) * 4;
                if (layer == 3) cell[0] = r0;
                else { cell[1] = r0; cell[2] = r1; cell[3] = r2; }
            }
        }
        split_planes<16, 3>(hf, ph, pm, pl);
        if (layer < 4) put_image<3, I::XCH>(act_own, ph, pm, pl);
        if (BWD) {
            char* q = acts + (layer * 2 + kh) * 4 * LD::ACTS_CH;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                *reinterpret_cast<u32x4*>(q + s * LD::ACTS_CH + ob16) = u32x4{ph[4 * s], ph[4 * s + 1], ph[4 * s + 2], ph[4 * s + 3]};
                *reinterpret_cast<u32x4*>(q + (2 + s) * LD::ACTS_CH + ob16) = u32x4{pm[4 * s], pm[4 * s + 1], pm[4 * s + 2], pm[4 * s + 3]};
            }
        }
    };
    auto init_acc = [&](int bias_off) __attribute__((always_inline)) {                                  // bias: only in the kh = 0 partial
        if (bias_off >= 0 && kh == 0) { load_bias(acc[0], SM + bias_off + 32 * ob, hi); acc[1] = acc[0]; }
        else { zero_acc(acc[0]); zero_acc(acc[1]); }
    };
    init_acc(-1);                                                        // :59 in_layer (bias rides in the constant-1 column)
    layer_run(C6{}, C0{}, wchunk(I::CW_IN, I::KS_IN, 0), e1x, nullptr, nullptr);
    layer_pre(C0{}, CJ{}, nullptr, wchunk(I::CW_M1, I::KS_M, 0));
    send_partial();
    __syncthreads();
    finish_tile(0);
    __syncthreads();
    WP_MARK(2);
    init_acc(I::B_M1);                                                   // :60 mid1
    layer_run(C0{}, CJ{}, nullptr, nullptr, wchunk(I::CW_M1, I::KS_M, 0), actx);
    layer_pre(C6{}, CJ{}, wchunk(I::CW_CAT, I::KS_CAT, JS), wchunk(I::CW_CAT, I::KS_CAT, 0));
    send_partial();
    __syncthreads();                                                     // everybody has read h1; partials are out
    finish_tile(1);
    __syncthreads();
    WP_MARK(3);
    init_acc(-1);                                                        // :63-64 cat_layer: encoding part, then h2
    layer_run(C6{}, CJ{}, wchunk(I::CW_CAT, I::KS_CAT, JS), e1x, wchunk(I::CW_CAT, I::KS_CAT, 0), actx);
    layer_pre(C0{}, CJ{}, nullptr, wchunk(I::CW_M2, I::KS_M, 0));
    send_partial();
    __syncthreads();
    finish_tile(2);
    __syncthreads();
    WP_MARK(4);
    init_acc(I::B_M2);                                                   // :67 mid2
    layer_run(C0{}, CJ{}, nullptr, nullptr, wchunk(I::CW_M2, I::KS_M, 0), actx);
    layer_pre(C3{}, CJ{}, wchunk(I::CW_C, I::KS_C, JS), wchunk(I::CW_C, I::KS_C, 0));
    send_partial();
    __syncthreads();
    finish_tile(3);
    __syncthreads();
    WP_MARK(5);
    init_acc(-1);                                                        // :81 color_linear: second encoding group, then h4
    layer_run(C3{}, CJ{}, wchunk(I::CW_C, I::KS_C, JS), e2x, wchunk(I::CW_C, I::KS_C, 0), actx);
    send_partial();
    __syncthreads();
    finish_tile(4);
    __syncthreads();
    WP_MARK(6);
    if (wave < 2 && hi == 0) {                                           // heads of tile `wave`: sum of the blocks' partials
        const int pt = 32 * wave + p31;
        if (pt < npts) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = hp[((0 * 2 + wave) * 32 + p31) * 4 + c] + hp[((1 * 2 + wave) * 32 + p31) * 4 + c];
                if (NB == 4) v[c] += hp[((2 * 2 + wave) * 32 + p31) * 4 + c] + hp[((3 * 2 + wave) * 32 + p31) * 4 + c];
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
    if (wave < kWaves) {
        const StepArgs& al = wv::kernarg_late(ga).s;
        composite_phase<BWD>(al, cb, loss_cells, obj, ray0, nrays, wave, lane, tid, rmeta);
    }
    __syncthreads();
    WP_MARK(7);
    if (BWD) {
    // ---- backward ----
    // planes (hi, mid) of this wave's (block ob, tile kh) activations, from the scratch: two sets, used alternately
    unsigned pah[8], pam[8], pbh[8], pbm[8];
    auto fetch = [&](unsigned (&h)[8], unsigned (&m)[8], int layer) {
        const char* q = acts + (layer * 2 + kh) * 4 * LD::ACTS_CH;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 v = ldgu(q + s * LD::ACTS_CH, ob16), w = ldgu(q + (2 + s) * LD::ACTS_CH, ob16);
            h[4 * s] = v[0]; h[4 * s + 1] = v[1]; h[4 * s + 2] = v[2]; h[4 * s + 3] = v[3];
            m[4 * s] = w[0]; m[4 * s + 1] = w[1]; m[4 * s + 2] = w[2]; m[4 * s + 3] = w[3];
        }
        wv::sched_fence();
    };
    fetch(pah, pam, 3);                                                  // h4
    fetch(pbh, pbm, 4);                                                  // hc
    const float* crow = cb + (32 * kh + p31) * 8;                        // this lane's point of tile kh (padding rows hold zeros)
    const float d_raw = crow[0], d_c0 = crow[1], d_c1 = crow[2], d_c2 = crow[3];
    // F-form images of the ten encoding blocks -> the workgroup's scratch (weight-gradient operands), dealt round-robin
    char* efg = wgs + LD::EFG_OFF;
    {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            if ((j % NWV) != wave) continue;
            const int st = j / 5, eb = j - 5 * st;
            const char* src = eb < 3 ? e1x + st * LD::E_ST + 2 * eb * I::XCH : e2x + st * LD::E_ST + 2 * (eb - 3) * I::XCH;
            unsigned h[8], m[8], f[16];
            const u32x4 h0 = *reinterpret_cast<const u32x4*>(src + evo), m0 = *reinterpret_cast<const u32x4*>(src + 1024 + evo);
            h[0] = h0[0]; h[1] = h0[1]; h[2] = h0[2]; h[3] = h0[3]; m[0] = m0[0]; m[1] = m0[1]; m[2] = m0[2]; m[3] = m0[3];
            if (eb < 4) {
                const u32x4 h1 = *reinterpret_cast<const u32x4*>(src + I::XCH + evo), m1 = *reinterpret_cast<const u32x4*>(src + I::XCH + 1024 + evo);
                h[4] = h1[0]; h[5] = h1[1]; h[6] = h1[2]; h[7] = h1[3]; m[4] = m1[0]; m[5] = m1[1]; m[6] = m1[2]; m[7] = m1[3];
            } else {
                h[4] = h[5] = h[6] = h[7] = 0u; m[4] = m[5] = m[6] = m[7] = 0u;
            }
            to_F<4>(f, tile, h, m, p31, hi, TL);
            char* dst = efg + st * 5 * 4096 + eb * 4096;
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(dst + c * 1024 + vlo16) = u32x4{f[4 * c], f[4 * c + 1], f[4 * c + 2], f[4 * c + 3]};
        }
    }
    unsigned dF[2][16];                                                  // F-form of the current delta block: [0] this wave's tile (kh), [1] the partner's
    float dv[16];                                                        // the delta of (block ob, tile kh) in float32
    f32x16 accd[2];
    float dproj[2][11];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int i = 0; i < 11; ++i) dproj[st][i] = 0.0f;
    using RW = RowWs<NB>;
    float* outR = out + RW::slot(0, ob, 0);                           // this output block's slots of the workgroup's row
    const unsigned lane4 = 4u * (unsigned)lane;
    char* dlt_own = lds + LD::DLT + kh * LD::DLT_ST + ob * 2 * I::DCH + lo16;
    const char* dlt_partner = lds + LD::DLT + (1 - kh) * LD::DLT_ST + ob * 2 * I::DCH + lo16;
    const char* dltx = lds + LD::DLT + lo16;
    char* xf_own = lds + LD::XF + kh * LD::XF_ST + ob * 4096 + lo16;
    const char* xfx = lds + LD::XF + lo16;
    char* xcg_own = lds + LD::XCG + wave * LD::XCG_W + lo16;
    const char* xcg_partner = lds + LD::XCG + (ob + (1 - kh) * NB) * LD::XCG_W + lo16;
    // an activation block of tile kh -> its F-form image (the layer input of everybody's weight gradients)
    auto publish_x = [&](const unsigned (&h)[8], const unsigned (&m)[8]) {
        unsigned xF[16];
        to_F<4>(xF, tile, h, m, p31, hi, TL);
        put_F(xf_own, xF);
    };
    // this wave's delta (tile kh) -> planes -> P-form image + F-form registers
    auto publish_d = [&]() __attribute__((always_inline)) {
        unsigned dh[8], dm[8], dl[8];
        split_planes<16, 2>(dv, dh, dm, dl);
        put_image<2, I::DCH>(dlt_own, dh, dm, dl);
        to_F<4>(dF[0], tile, dh, dm, p31, hi, TL);
    };
    // after the barrier: the partner's delta (tile 1 - kh) from its P-form image -> F-form registers
    auto partner_dF = [&]() __attribute__((always_inline)) {
        unsigned dh[8], dm[8];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 v = lds16(dlt_partner + s * I::DCH), w = lds16(dlt_partner + s * I::DCH + 1024);
            dh[4 * s] = v[0]; dh[4 * s + 1] = v[1]; dh[4 * s + 2] = v[2]; dh[4 * s + 3] = v[3];
            dm[4 * s] = w[0]; dm[4 * s + 1] = w[1]; dm[4 * s + 2] = w[2]; dm[4 * s + 3] = w[3];
        }
        to_F<4>(dF[1], tile, dh, dm, p31, hi, TL);
    };
    TPre tp, tpe;
    constexpr int JH = JS / 2;                                           // d-prop steps per wave of a pair
    auto hidden_ptr = [&](int ct_base) __attribute__((always_inline)) { return gWT + ((long long)(ct_base + ob * JS + kh * JH)) * I::DCH; };
    // d-prop into block ob: this wave's half of the chain (both tiles), partial of the partner's tile -> exchange slot
    auto dprop_hidden = [&](int ct_base, bool add_alpha) __attribute__((always_inline)) {
#pragma unroll
        for (int st = 0; st < 2; ++st) zero_acc(accd[st]);
        if (add_alpha) {                                                 // + W_a d raw: tile kh's term, in the partial this wave keeps
#pragma unroll
            for (int r = 0; r < 16; ++r) accd[0][r] = SM[I::W_A + 32 * ob + phi(r, hi)] * d_raw;
        }
        // accd[0] = this wave's tile, accd[1] = the partner's (delta images read in that order)
        bwd_run<W3, JH>(accd, tp, hidden_ptr(ct_base), vlo16, dltx + kh * JH * I::DCH + kh * LD::DLT_ST, (1 - 2 * kh) * LD::DLT_ST);
        acc_send(xcg_own, accd[1]);
    };
    // after the barrier: + the partner's partial, through the ReLU of the activation whose hi plane is mh -> dv
    auto finish_delta = [&](const unsigned (&mh)[8]) {
        f32x16 v = accd[0];
        acc_recv_add(v, xcg_partner);
        mask_by(dv, v, mh);
    };
    auto enc_ptr = [&](int ct_chunk) __attribute__((always_inline)) { return gWT + (long long)ct_chunk * I::DCH; };
    // d-prop into an encoding block (both tiles, the whole chain) -> d(proj) through the cos factors
    auto dprop_enc = [&](int ct_chunk, int group, int blk) __attribute__((always_inline)) {
        f32x16 acce[2];
        zero_acc(acce[0]); zero_acc(acce[1]);
        tpre_load<W3>(tpe, enc_ptr(ct_chunk), vlo16);
        bwd_run<W3, JS>(acce, tpe, enc_ptr(ct_chunk), vlo16, dltx, LD::DLT_ST);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float cf[16];
            cf_load(cf, cfs + st * kCfTile, group, blk, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = 16 * blk + r;
                if (group == 1) { if (R < 44) dproj[st][R >> 2] = fmaf(acce[st][r], cf[r], dproj[st][R >> 2]); }
                else { if (R < 22) dproj[st][R >> 1] = fmaf(acce[st][r], cf[r], dproj[st][R >> 1]); }
            }
        }
    };
    // the waves that d-prop the encoding blocks: the last three / two (at hidden 128 all of them have kh = 1)
    constexpr int EW1 = NWV - 3, EW2 = NWV - 2, EW3 = NWV - 1;
    // F images of a block, this wave's tile first (matches dF)
    auto xload_hidden = [&](FImg& x, int kb) __attribute__((always_inline)) { fimg_load(x, xfx + kb * 4096 + kh * LD::XF_ST, (1 - 2 * kh) * LD::XF_ST); };
    const char* efo = efg + kh * 5 * 4096;                               // encoding F images, this wave's tile first
    const int efs = (1 - 2 * kh) * 5 * 4096;
    auto bias_rows = [&](float* q) __attribute__((always_inline)) {                                     // bias gradient of mid1 / mid2: dY^T . ones, rows of block ob
        f32x16 accb;
        db_pair(accb, dF);
        if (p31 == 0) store_rows<1>(q + 32 * ob, (unsigned)(4 * hi), accb, first);
    };
    __syncthreads();                                                     // forward images dead: the exchange slots / delta images may be written
    WP_MARK(8);
    // -- heads: F-form of (d raw alpha, d raw colour) of tile kh; its products with h4 / hc of (block ob, tile kh) = rows 0..3 of a
    //    block; the two tiles' partial sums meet in LDS (kh = 1 sends, kh = 0 stores) --
    f32x16 ga4, gc4, gb4;
    float* cell = hx + (ob * 8) * 64 + lane;                             // [block][8][64]: rows 0..3 head weights, 4..7 head biases
    {
        unsigned dHF[16], xF[16];
        {
            float hv[16];
            unsigned dh[8], dm[8], dl[8];
#pragma unroll
            for (int r = 0; r < 16; ++r) hv[r] = 0.0f;
            if (hi == 0) { hv[0] = d_raw; hv[1] = d_c0; hv[2] = d_c1; hv[3] = d_c2; }
            split_planes<16, 2>(hv, dh, dm, dl);
            to_F<4>(dHF, tile, dh, dm, p31, hi, TL);
        }
        to_F<4>(xF, tile, pah, pam, p31, hi, TL);                        // h4 of (ob, kh)
        put_F(xf_own, xF);                                               // = color_linear's weight-gradient operand
        zero_acc(ga4); dw_mm_s(ga4, dHF, xF);
        to_F<4>(xF, tile, pbh, pbm, p31, hi, TL);                        // hc of (ob, kh)
        zero_acc(gc4); dw_mm_s(gc4, dHF, xF);
        zero_acc(gb4);
        if (ob == 0) {                                                   // b_a, b_oc: sums over the tile's points
            const u32x4 ones = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                gb4 = wv::mfma_bf16(u32x4{dHF[8 + 4 * s], dHF[8 + 4 * s + 1], dHF[8 + 4 * s + 2], dHF[8 + 4 * s + 3]}, ones, gb4);
                gb4 = wv::mfma_bf16(u32x4{dHF[4 * s], dHF[4 * s + 1], dHF[4 * s + 2], dHF[4 * s + 3]}, ones, gb4);
            }
        }
        if (kh == 1) {
            cell[0 * 64] = ga4[0]; cell[1 * 64] = gc4[1]; cell[2 * 64] = gc4[2]; cell[3 * 64] = gc4[3];
            if (ob == 0) { cell[4 * 64] = gb4[0]; cell[5 * 64] = gb4[1]; cell[6 * 64] = gb4[2]; cell[7 * 64] = gb4[3]; }
        }
        // delta 0 = d hc (through the ReLU of hc), tile kh
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 32 * ob + phi(r, hi);
            v[r] = SM[I::W_OC + j] * d_c0 + SM[I::W_OC + H + j] * d_c1 + SM[I::W_OC + 2 * H + j] * d_c2;
        }
        mask_by(dv, v, pbh);
    }
    publish_d();
    fetch(pbh, pbm, 2);                                                  // h3: mid2's input, the mask of delta 2
    tpre_load<W3>(tp, hidden_ptr(I::CT_C), vlo16);
    __syncthreads();
    if (kh == 0 && hi == 0) {
        store_one(out + RW::W_A + 32 * ob + p31, ga4[0] + cell[0 * 64], first);
        store_one(out + RW::W_OC + 0 * H + 32 * ob + p31, gc4[1] + cell[1 * 64], first);
        store_one(out + RW::W_OC + 1 * H + 32 * ob + p31, gc4[2] + cell[2 * 64], first);
        store_one(out + RW::W_OC + 2 * H + 32 * ob + p31, gc4[3] + cell[3 * 64], first);
        if (ob == 0 && lane == 0) {
            store_one(out + RW::B_A, gb4[0] + cell[4 * 64], first);
            store_one(out + RW::B_OC + 0, gb4[1] + cell[5 * 64], first);
            store_one(out + RW::B_OC + 1, gb4[2] + cell[6 * 64], first);
            store_one(out + RW::B_OC + 2, gb4[3] + cell[7 * 64], first);
        }
    }
    WP_MARK(9);
    // color_linear: weight gradients (h4 blocks, second-group blocks + bias column), d-prop -> d h4 (+ W_a d raw), d(second group)
    partner_dF();
    dw_layer_half<NB + 2>(dF, first, kh,
        [&](FImg& x, int kb) { if (kb < NB) xload_hidden(x, kb); else fimg_load_g(x, efo + (3 + kb - NB) * 4096, efs, vlo16); },
        [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
            WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_C, 0, kb), lane4, v, old)));
        });
    if (wave == EW2) dprop_enc(I::CT_C + (NB + 0) * JS, 2, 0);
    if (wave == EW3) dprop_enc(I::CT_C + (NB + 1) * JS, 2, 1);
    dprop_hidden(I::CT_C, true);
    tpre_load<W3>(tp, hidden_ptr(I::CT_M2), vlo16);
    __syncthreads();
    finish_delta(pah);                                                   // delta 1 = d h4
    publish_d();
    publish_x(pbh, pbm);                                                 // h3
    fetch(pah, pam, 1);                                                  // h2: cat_layer's input, the mask of delta 3
    __syncthreads();
    WP_MARK(10);
    // mid2
    partner_dF();
    dw_layer_half<NB>(dF, first, kh, xload_hidden, [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
        WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_M2, 0, kb), lane4, v, old)));
    });
    if (kh == 1) bias_rows(out + RW::B_M2);
    dprop_hidden(I::CT_M2, false);
    tpre_load<W3>(tp, hidden_ptr(I::CT_CAT), vlo16);
    __syncthreads();
    finish_delta(pbh);                                                   // delta 2 = d h3
    publish_d();
    publish_x(pah, pam);                                                 // h2
    fetch(pbh, pbm, 0);                                                  // h1: mid1's input, the mask of delta 4
    __syncthreads();
    WP_MARK(11);
    // cat_layer
    partner_dF();
    dw_layer_half<NB + 3>(dF, first, kh,
        [&](FImg& x, int kb) { if (kb < NB) xload_hidden(x, kb); else fimg_load_g(x, efo + (kb - NB) * 4096, efs, vlo16); },
        [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
            WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_CAT, 0, kb), lane4, v, old)));
        });
    if (wave == EW1) dprop_enc(I::CT_CAT + (NB + 0) * JS, 1, 0);
    if (wave == EW2) dprop_enc(I::CT_CAT + (NB + 1) * JS, 1, 1);
    if (wave == EW3) dprop_enc(I::CT_CAT + (NB + 2) * JS, 1, 2);
    dprop_hidden(I::CT_CAT, false);
    tpre_load<W3>(tp, hidden_ptr(I::CT_M1), vlo16);
    __syncthreads();
    finish_delta(pah);                                                   // delta 3 = d h2
    publish_d();
    publish_x(pbh, pbm);                                                 // h1
    __syncthreads();
    WP_MARK(12);
    // mid1
    partner_dF();
    dw_layer_half<NB>(dF, first, kh, xload_hidden, [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
        WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_M1, 0, kb), lane4, v, old)));
    });
    if (kh == 1) bias_rows(out + RW::B_M1);
    dprop_hidden(I::CT_M1, false);
    __syncthreads();
    finish_delta(pbh);                                                   // delta 4 = d h1
    publish_d();
    __syncthreads();
    WP_MARK(13);
    // in_layer
    partner_dF();
    dw_layer_half<3>(dF, first, kh, [&](FImg& x, int kb) { fimg_load_g(x, efo + kb * 4096, efs, vlo16); },
                     [&](int mode, int kb, const f32x16& v, float (&old)[16]) {
                         WS_IO(mode, (slot_io<M>(outR + RW::slot(RW::S_IN, 0, kb), lane4, v, old)));
                     });
    if (wave == EW1) dprop_enc(I::CT_IN + 0 * JS, 1, 0);
    if (wave == EW2) dprop_enc(I::CT_IN + 1 * JS, 1, 1);
    if (wave == EW3) dprop_enc(I::CT_IN + 2 * JS, 1, 2);
    // B_layer.weight: dB[d][j] = sum_points d(proj)[d] t[j].  d(proj) = sum of the waves' parts (exchange through LDS);
    // t = the x, y, z slots of the hi = 0 lanes = columns 24..26 of first-group block 2
    __syncthreads();
    WP_MARK(14);
    {
        float* px = reinterpret_cast<float*>(lds + LD::XCG);             // [wave][tile][11][64]
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int i = 0; i < 11; ++i) px[((wave * 2 + st) * 11 + i) * 64 + lane] = dproj[st][i];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                unsigned dh[8], dm[8], dl[8];
                float ds[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float acc_r = 0.0f;
                    if (r < 11) {
                        // only the waves EW1..EW3 contribute (the others hold zeros): fixed order
                        acc_r = (px[((EW1 * 2 + st) * 11 + r) * 64 + lane] + px[((EW2 * 2 + st) * 11 + r) * 64 + lane]) + px[((EW3 * 2 + st) * 11 + r) * 64 + lane];
                    }
                    ds[r] = acc_r;
                }
                split_planes<16, 2>(ds, dh, dm, dl);
                to_F<4>(dF[st], tile, dh, dm, p31, hi, TL);
            }
            FImg xi;
            fimg_load_g(xi, efg + 2 * 4096, 5 * 4096, vlo16);
            f32x16 accB;
            dw_mm_pair(accB, dF, xi);
            if (p31 >= 24 && p31 < 27) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int d = hi ? 11 + r : r;                          // row phi(r, hi) <-> direction
                    if (r < (hi ? 10 : 11)) store_one(out + RW::PE_B + 3 * d + (p31 - 24), accB[r], first);
                }
            }
        }
    }
    WP_MARK(15);
    }   // BWD
    }   // rounds
#undef WP_MARK
#undef WP_SPLIT
    __syncthreads();
    if (tid_k == 0) {
        float* pl = a.part_loss + (obj * a.NW + wgo) * 4;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            pl[k] = (loss_cells[k] + loss_cells[4 + k]) + (loss_cells[8 + k] + loss_cells[12 + k]);
        pl[3] = 0.0f;
    }
}

}  // namespace vk
