// Measurement probe (not part of the product; round 5): would step_finalize_ws read the background step's 200 gradient rows faster if the rows were
// stored chunk-major ([block][row][96 quads]: 300 KB contiguous per finalize block) instead of row-major ([row][PP]: 200 segments of 1.5 KB, 377 KB
// apart)?  Same block shape as the kernel (96 quads x 8 row groups, loads eight deep, 246 blocks), rows freshly written by another kernel each time.
// Build:  hipcc --offload-arch=gfx950 -O3 -o fin_layout_probe.out tests/tools/fin_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kQuads = 96, kGroups = 8, kNW = 200, kBlocks = 246, kPPq = kBlocks * kQuads;     // 23616 quads = 94464 floats per row (377.9 KB)

__global__ __launch_bounds__(1024) void writer(f4* rows, float v) {
    const long long n = (long long)kNW * kPPq;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n; i += (long long)gridDim.x * 1024) rows[i] = f4{v, v + 1.0f, v + 2.0f, v + 3.0f};
}

// the rows as step_main_ws leaves them: ONE workgroup writes a whole row (its own 377 KB), 200 workgroups side by side, in pieces over time
__global__ __launch_bounds__(256) void writer_row_per_block(f4* rows, float v, int spin) {
    f4* row = rows + (long long)blockIdx.x * kPPq;
    for (int i = threadIdx.x; i < kPPq; i += 256) {
        row[i] = f4{v, v + 1.0f, v + 2.0f, v + 3.0f};
        if (spin) __builtin_amdgcn_s_sleep(8);
    }
}

template <bool CHUNK_MAJOR, int DEEP>
__global__ __launch_bounds__(kGroups * kQuads) void reader(const f4* rows, f4* out) {
    __shared__ f4 red[kGroups * kQuads];
    const int ql = threadIdx.x % kQuads, rg = threadIdx.x / kQuads;
    const int per = (kNW + kGroups - 1) / kGroups, u_begin = min(kNW, rg * per), u_end = min(kNW, u_begin + per);
    const f4* pg = CHUNK_MAJOR ? rows + (long long)blockIdx.x * kNW * kQuads + ql : rows + (long long)blockIdx.x * kQuads + ql;
    const long long qs = CHUNK_MAJOR ? kQuads : kPPq;
    f4 g = {0.0f, 0.0f, 0.0f, 0.0f};
    int u0 = u_begin;
    for (; u0 + DEEP <= u_end; u0 += DEEP) {
        f4 t[DEEP];
#pragma unroll
        for (int u = 0; u < DEEP; ++u) t[u] = pg[(u0 + u) * qs];
#pragma unroll
        for (int u = 0; u < DEEP; ++u) g += t[u];
    }
    for (; u0 < u_end; ++u0) g += pg[u0 * qs];
    red[rg * kQuads + ql] = g;
    __syncthreads();
    if (rg) return;
    f4 s = red[ql];
    for (int i = 1; i < kGroups; ++i) s += red[i * kQuads + ql];
    out[blockIdx.x * kQuads + ql] = s;
}

// the hidden-32 finalize's shape: 20 objects x 12 blocks of 256 threads, one thread per quad, 10 rows of 2848 quads per object, + the
// update's own traffic (moments and parameters read and written back) - what is the floor of such a launch?
constexpr int sObj = 20, sNW = 10, sPPq = 2848, sBpo = 12;
template <bool UPDATE>
__global__ __launch_bounds__(256) void reader_s32(const f4* rows, f4* m, f4* v, f4* p) {
    const int obj = blockIdx.x / sBpo, q = (blockIdx.x % sBpo) * 256 + threadIdx.x;
    if (q >= sPPq) return;
    const f4* pg = rows + (long long)obj * sNW * sPPq + q;
    const long long s = (long long)obj * sPPq + q;
    f4 m4, v4, p4;
    if (UPDATE) { m4 = m[s]; v4 = v[s]; p4 = p[s]; }
    f4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = pg[u * (long long)sPPq];
    f4 g = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 8; ++u) g += t[u];
    g += pg[8ll * sPPq];
    g += pg[9ll * sPPq];
    if (UPDATE) { m4 = m4 * 0.9f + g * 0.1f; v4 = v4 * 0.999f + g * g * 0.001f; p4 = p4 - m4 * 1e-3f; m[s] = m4; v[s] = v4; p[s] = p4; }
    else p[s] = g;
}
__global__ void empty_kernel() {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <class K>
static float run(K k, const char* name, f4* rows, f4* out, int reps, int wmode = 0) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0.0f, best = 1e9f;
    for (int r = 0; r < reps + 3; ++r) {
        if (wmode == 0) writer<<<512, 1024>>>(rows, (float)r);
        else writer_row_per_block<<<kNW, 256>>>(rows, (float)r, wmode == 2 ? 8 : 0);
        CK(hipEventRecord(e0));
        k<<<kBlocks, kGroups * kQuads>>>(rows, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { tot += ms; best = ms < best ? ms : best; }
    }
    const double bytes = (double)kNW * kPPq * 16;
    printf("{\"layout\": \"%s\", \"mean_us\": %.2f, \"min_us\": %.2f, \"TB_per_s_at_mean\": %.2f}\n", name, 1e3 * tot / reps, 1e3 * best, bytes / (tot / reps * 1e-3) / 1e12);
    return tot / reps;
}
template <class K>
static void run_s32(K k, const char* name, f4* rows, f4* m, f4* v, f4* p, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0.0f, best = 1e9f;
    for (int r = 0; r < reps + 3; ++r) {
        writer<<<512, 1024>>>(rows, (float)r);
        CK(hipEventRecord(e0));
        k<<<sObj * sBpo, 256>>>(rows, m, v, p);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) { tot += ms; best = ms < best ? ms : best; }
    }
    printf("{\"layout\": \"%s\", \"mean_us\": %.2f, \"min_us\": %.2f}\n", name, 1e3 * tot / reps, 1e3 * best);
}
int main() {
    {
        f4 *rows, *m, *v, *p;
        CK(hipMalloc(&rows, (size_t)kNW * kPPq * 16)); CK(hipMalloc(&m, (size_t)sObj * sPPq * 16)); CK(hipMalloc(&v, (size_t)sObj * sPPq * 16)); CK(hipMalloc(&p, (size_t)sObj * sPPq * 16));
        for (int pass = 0; pass < 2; ++pass) {
            run_s32(reader_s32<false>, "hidden-32 shape: ten row reads + one store per quad", rows, m, v, p, 200);
            run_s32(reader_s32<true>, "hidden-32 shape: + moments and parameters read and written", rows, m, v, p, 200);
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float tot = 0.0f;
        for (int r = 0; r < 203; ++r) {
            writer<<<512, 1024>>>(rows, (float)r);
            CK(hipEventRecord(e0)); empty_kernel<<<1, 64>>>(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r >= 3) tot += ms;
        }
        printf("{\"layout\": \"an empty kernel between the same two events\", \"mean_us\": %.2f}\n", 1e3 * tot / 200);
        CK(hipFree(rows)); CK(hipFree(m)); CK(hipFree(v)); CK(hipFree(p));
    }
    f4 *rows, *out;
    CK(hipMalloc(&rows, (size_t)kNW * kPPq * 16)); CK(hipMalloc(&out, (size_t)kPPq * 16));
    for (int pass = 0; pass < 2; ++pass) {
        run(reader<false, 8>, "row_major_8_deep (the kernel's)", rows, out, 100);
        run(reader<true, 8>, "chunk_major_8_deep", rows, out, 100);
        run(reader<false, 16>, "row_major_16_deep", rows, out, 100);
        run(reader<true, 16>, "chunk_major_16_deep", rows, out, 100);
        run(reader<true, 25>, "chunk_major_25_deep", rows, out, 100);
        run(reader<false, 8>, "row_major_8_deep, rows written one workgroup per row", rows, out, 100, 1);
        run(reader<false, 8>, "row_major_8_deep, rows written one workgroup per row, slowly (~80 us)", rows, out, 100, 2);
    }
    return 0;
}
