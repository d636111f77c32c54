// wave_ops.h - the handful of gfx950 wave-level primitives the step kernels are written against.
//
// Device implementation (CDNA4, wave64).  Everything the kernels need from the hardware beyond plain
// per-lane arithmetic goes through this header: the fp32 matrix instruction, the half-wave exchange,
// the intra-wave LDS ordering point and the dynamic LDS base.  tests/sim/ provides a host-side model of
// exactly this interface so that the kernel source can be executed lane-by-lane on a CPU in the
// `-m "not gpu"` tier; the product never uses that model.
#pragma once
#include <hip/hip_runtime.h>

// occupancy request of a kernel: exactly n waves per SIMD (the register allocator's budget is 512 / n)
#define WV_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))

namespace wv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // four floats at any 4-byte boundary (global_load / store_dwordx4: gfx950 runs in unaligned access mode)

// D = A(32x2) * B(2x32) + C, exact fp32 (bitwise an fmaf chain over k), 64 cycles per SIMD.
//   a: lane l supplies A[i = l & 31][k = l >> 5]
//   b: lane l supplies B[k = l >> 5][j = l & 31]
//   c/d: lane l, register r holds D[i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][j = l & 31]
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- bf16 matrix pipe (split-bf16 step kernel, split_kernels.h) ----
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// D = A(32x16) * B(16x32) + C on v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate, 32 cycles per SIMD (1/16 of the
// exact-fp32 form per MAC) and - unlike the fp32 form - issued next to VALU work of the same wave (measured:
// profiles/r02a_bf16_probe.jsonl: up to ~6 VALU instructions per matrix instruction are free).
//   a: lane l supplies A[i = l & 31][k = 8 * (l >> 5) + t], t = 0..7 as four dwords of two bf16 each (low half first)
//   b: lane l supplies B[k = 8 * (l >> 5) + t][j = l & 31]
//   c/d: as mfma32
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// D = A(16x32) * B(32x16) + C on v_mfma_f32_16x16x32_bf16 (16 cycles): a: lane l supplies A[l & 15][8 (l >> 4) + t]; b: lane l
// supplies B[8 (l >> 4) + t][l & 15]; c / d: lane l, register r <-> D[4 (l >> 4) + r][l & 15]  (checked against a host product:
// tests/tools/mfma16_probe.hip)
typedef float f32x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4m mfma16_bf16(u32x4 a, u32x4 b, f32x4m c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// two floats -> one dword of two bfloat16 (round to nearest even), lo in the low half: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// ds_read_b64_tr_b16: within each 16-lane group, lane c receives element (c & 3) of the four consecutive 16-bit words lane
// 4 * j + (c >> 2) points at, for j = 0..3 (measured: profiles/r02a_bf16_probe.jsonl) - a 4 x 4 transpose across lanes.
// `p` must be 8-byte aligned.
__device__ __forceinline__ u32x2 lds_tr16(const void* p) {
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    return __builtin_bit_cast(u32x2, v);
}

// value held by the partner lane in the other 32-lane half of the wave (lane ^ 32): v_permlane32_swap + select
// (VALU only; the ds_bpermute form costs an LDS round trip of ~100 cycles that one wave per SIMD cannot hide)
__device__ __forceinline__ float swap_half(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// value held by lane J of this lane's 16-lane row (DPP row_newbcast: one VALU op, often fused into its consumer)
template <int J>
__device__ __forceinline__ float row_bcast(float x) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x150 + J, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}
// sum over this lane's 16-lane row, result on every lane: quad butterflies, then half-row and row mirrors
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp_mov<0xB1>(x);     // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);     // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);    // row_half_mirror
    x += dpp_mov<0x140>(x);    // row_mirror
    return x;
}

// sum of x over the 32 lanes of this lane's HALF of the wave (lanes with the same lane >> 5), valid on the lanes of the half's second
// 16-lane row (lane & 16): the row sums, then DPP row_bcast:15 (lane 15 of rows 0 / 2 -> every lane of rows 1 / 3; a GFX9 control)
__device__ __forceinline__ float half_sum32_hi_row(float x) {
    x = row_sum16(x);
    const float t = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x142, 0xA, 0xF, false));
    return x + t;
}

// value of x held by lane `src` (0..63) of this wave
__device__ __forceinline__ float shfl(float x, int src) { return __shfl(x, src & 63); }

// asynchronous 16-byte-per-lane copy global -> LDS (LDS-DMA, no VGPR round trip): lane l's 16 bytes at `g` land at
// lds_wave_base + 16*l.  Completion is covered by the vmcnt drain of the next workgroup barrier.
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// true on every lane iff pred holds on at least one lane of the wave (wave-uniform result)
__device__ __forceinline__ bool wave_any(bool pred) { return __any(pred) != 0; }

// Orders this wave's earlier LDS writes before its later LDS reads (data exchanged between lanes of ONE
// wave through a wave-private LDS region).  A wave's DS instructions execute in order, so no hardware
// barrier is needed - only the compiler must not move accesses across this point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds_bytes[];
__device__ __forceinline__ float* lds_base() { return reinterpret_cast<float*>(dyn_lds_bytes); }

// shader-clock timestamp (low 32 bits of s_memtime)
// The kernel's by-value argument struct, re-read from the kernarg segment at the point of use.  Fields accessed
// through the function parameter are all fetched in the prologue; a kernel that keeps 500 vector registers live has
// no scalar registers to park them in, and every later use becomes a v_readlane of a spilled SGPR.  A phase that
// needs a dozen pointers and strides reads them here instead (scalar loads, one latency per phase).
template <class T>
__device__ __forceinline__ const T& kernarg_late(const T&) {
    auto p = (const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const T*)p;
}

// max(x, 0) for finite x as ONE instruction: v_med3_f32(x, 0, FLT_MAX).  `x > 0 ? x : 0` and fmaxf() both cost two
// (a canonicalising v_max in front: the matrix-instruction result is not known to be free of signalling NaNs).  +inf maps
// to FLT_MAX instead of +inf, which only matters once the "loss explode" flag (render_rays.py:88-90) is up anyway.
__device__ __forceinline__ float relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 3.4028234663852886e38f); }

// The value, with its origin hidden from the optimiser (no instruction).  The backward's ReLU masks test opaque(h) > 0:
// given the plain h = max(acc, 0) the compiler proves (h > 0) == (acc > 0), evaluates all 80 masks during the forward
// and parks them in 160 spilled SGPRs.
__device__ __forceinline__ float opaque(float x) {
    asm("" : "+v"(x));
    return x;
}

__device__ __forceinline__ unsigned opaque_u(unsigned x) {
    asm("" : "+v"(x));
    return x;
}

// Ordering ties for hand-interleaved instruction streams (no instruction): the returned copy of x exists only after y does,
// so work that consumes the copy cannot be scheduled in front of the producer of y.  (Scheduling fences alone do not pin
// side-effect-free arithmetic: it is placed before instruction scheduling sees the fences.)
__device__ __forceinline__ float after(float x, float y) {
    asm volatile("" : "+v"(x) : "v"(y));
    return x;
}
__device__ __forceinline__ unsigned after_u(unsigned x, unsigned y) {
    asm volatile("" : "+v"(x) : "v"(y));
    return x;
}

// An integer the optimiser must assume changes at every execution (no instruction).  step_main_h32's multi-pass loop
// re-derives its lane coordinates from opaque_iter(threadIdx.x): otherwise every lane mask and LDS address of the
// 6000-instruction body is loop-invariant, gets hoisted in front of the loop and is kept in (spilled) registers.
__device__ __forceinline__ int opaque_iter(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
// a value the program knows to be the same in every lane of the wave, moved to a scalar register (addresses built from
// it use the scalar-base form of the global instructions)
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// the same trick as opaque_iter for wave-uniform addresses: a scalar zero the optimiser cannot see through.  Addresses formed
// as base + opaque_uzero() inside a loop are not loop-invariant, so they are formed where they are used instead of living in
// (spilled) scalar registers; the base keeps its provenance (global address space).
__device__ __forceinline__ unsigned opaque_uzero() {
    unsigned z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
}

// nothing may be moved across this point by the instruction scheduler (hand-placed software pipelining)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// Scheduling-region directive: the next instructions of the region are emitted as NM groups of {1 matrix instruction, NV VALU
// instructions} (LLVM sched_group_barrier; masks: VALU 0x2, MFMA 0x8).  One wave per SIMD issues in order: a matrix
// instruction that waits for the pipe blocks the VALU work behind it, so independent VALU work only hides behind matrix
// instructions when it is INTERLEAVED with them (<= 6 per bf16 matrix instruction are free, profiles/r02a_bf16_probe.jsonl).
template <int NM, int NV>
__device__ __forceinline__ void interleave_mfma_valu() {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, NV, 0);
    }
}

__device__ __forceinline__ unsigned clock32() { return (unsigned)__builtin_amdgcn_s_memtime(); }

__device__ __forceinline__ void lds_add(float* p, float v) { atomicAdd(p, v); }   // ds_add_f32

// four wave-uniform words are made to exist in scalar registers at this point (no instruction): pins where a block of
// kernel arguments is fetched
__device__ __forceinline__ void touch_scalar4(unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
    asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
}

// ---- hand-off between workgroups of ONE launch (MI355X_MICROARCH.md, inter-workgroup visibility: a CU's vector L1 is
// never refreshed by another CU's stores and the XCDs' L2s are not coherent with each other).  Recipe used here:
// write-through ("sc1") payload stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> one agent-scope counter increment;
// the consumer polls the counter with L1-bypassing loads and then reads the payload with "sc1" loads as well. ----
__device__ __forceinline__ void store_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// glds16 with the sc1 cache policy (aux bit 4 on gfx940+): the copy is served past this CU's L1
__device__ __forceinline__ void glds16_wt(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 16);
}
// every memory instruction this wave has issued is complete (the compiler may drop the wait it would otherwise emit in
// front of a relaxed atomic; the explicit form cannot be dropped)
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void signal_add(unsigned* c) { __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane polls until *c >= target; bounded (a workgroup that is not resident cannot signal: the caller raises an error
// flag instead of hanging the device).  Returns false on timeout.
__device__ __forceinline__ bool wait_ge(const unsigned* c, unsigned target, int max_polls) {
    for (int i = 0; i < max_polls; ++i) {
        if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

}  // namespace wv
