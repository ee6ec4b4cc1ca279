// VALU issue-rate microbenchmark for gfx950: scalar f32 FMA vs packed f32 (v_pk_fma/mul/add) vs v_rcp, IEEE div.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_t p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const float m = 0.999f + seed * 1e-9f, c = 1e-3f;
    const float2_t m2 = {m, m}, c2 = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 8 scalar fma
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        } else if (MODE == 1) {  // 4 packed fma (= 8 fma)
            p0 = __builtin_elementwise_fma(p0, m2, c2); p1 = __builtin_elementwise_fma(p1, m2, c2);
            p2 = __builtin_elementwise_fma(p2, m2, c2); p3 = __builtin_elementwise_fma(p3, m2, c2);
        } else if (MODE == 2) {  // 4 packed mul + 4 packed add (= 8 mul + 8 add)
            p0 = p0 * m2; p1 = p1 * m2; p2 = p2 * m2; p3 = p3 * m2;
            p0 = p0 + c2; p1 = p1 + c2; p2 = p2 + c2; p3 = p3 + c2;
        } else if (MODE == 3) {  // 8 scalar mul + 8 scalar add
            a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
            a0 = a0 + c; a1 = a1 + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; a5 = a5 + c; a6 = a6 + c; a7 = a7 + c;
        } else if (MODE == 4) {  // 8 IEEE divisions
            a0 = m / a0; a1 = m / a1; a2 = m / a2; a3 = m / a3; a4 = m / a4; a5 = m / a5; a6 = m / a6; a7 = m / a7;
        } else if (MODE == 5) {  // 8 v_min + 8 v_max
            a0 = fminf(a0, a1); a1 = fmaxf(a1, a2); a2 = fminf(a2, a3); a3 = fmaxf(a3, a4);
            a4 = fminf(a4, a5); a5 = fmaxf(a5, a6); a6 = fminf(a6, a7); a7 = fmaxf(a7, a0);
            a0 = fmaxf(a0, c); a1 = fminf(a1, m); a2 = fmaxf(a2, c); a3 = fminf(a3, m);
            a4 = fmaxf(a4, c); a5 = fminf(a5, m); a6 = fmaxf(a6, c); a7 = fminf(a7, m);
        } else if (MODE == 6) {  // 8 cmp+cndmask
            a0 = a0 > a1 ? a2 : a0; a1 = a1 > a2 ? a3 : a1; a2 = a2 > a3 ? a4 : a2; a3 = a3 > a4 ? a5 : a3;
            a4 = a4 > a5 ? a6 : a4; a5 = a5 > a6 ? a7 : a5; a6 = a6 > a7 ? a0 : a6; a7 = a7 > a0 ? a1 : a7;
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
double run(const char* name, int ops_per_iter, int waves_per_simd) {
    float* d; hipMalloc(&d, 4);
    int iters = 20000;
    int blocks = 256 * waves_per_simd;  // 4 waves per block, 256 CUs
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * iters * ops_per_iter;
    double per_simd_per_s = wave_instr / (ms * 1e-3) / 1024.0;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.3f G wave-ops/s/SIMD  => %.2f cycles/op @2.4GHz\n", name, waves_per_simd, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
    hipFree(d);
    return per_simd_per_s;
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("scalar v_fma_f32 (8)", 8, w);
        run<1>("v_pk_fma_f32 (4 = 8 fma)", 4, w);
        run<2>("v_pk_mul + v_pk_add (8)", 8, w);
        run<3>("scalar mul + add (16)", 16, w);
        run<4>("IEEE div (8 divs)", 8, w);
        run<5>("v_min/v_max (16)", 16, w);
        run<6>("cmp+cndmask (8 pairs)", 8, w);
    }
    return 0;
}
