// More gfx950 VALU issue costs (integer / compare / select / special), 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    uint64_t q0 = a0 | ((uint64_t)a1 << 32), q1 = a2 | ((uint64_t)a3 << 32), q2 = a4 | ((uint64_t)a5 << 32), q3 = a6 | ((uint64_t)a7 << 32);
    float f0 = a0 * 1e-3f + 1, f1 = a1 * 1e-9f + 1, f2 = a2 * 1e-9f + 1, f3 = a3 * 1e-9f + 1, f4 = a4 * 1e-9f + 1, f5 = a5 * 1e-9f + 1, f6 = a6 * 1e-9f + 1, f7 = a7 * 1e-9f + 1;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 8 x v_mul_lo_u32
#define X(n) a##n = a##n * 0x9e3779b9u + 1u;
            REP8(X)
#undef X
        } else if (MODE == 1) {  // 4 x 64-bit multiply by constant (+add)
            q0 = q0 * 0xbf58476d1ce4e5b9ull + 1; q1 = q1 * 0xbf58476d1ce4e5b9ull + 1; q2 = q2 * 0xbf58476d1ce4e5b9ull + 1; q3 = q3 * 0xbf58476d1ce4e5b9ull + 1;
        } else if (MODE == 2) {  // 8 x xor+shift (32-bit)
#define X(n) a##n ^= a##n >> 7;
            REP8(X)
#undef X
        } else if (MODE == 3) {  // 4 x 64-bit xorshift
            q0 ^= q0 >> 30; q1 ^= q1 >> 27; q2 ^= q2 >> 31; q3 ^= q3 << 17;
        } else if (MODE == 4) {  // 4 x rotl64 + add64
            q0 = ((q0 << 23) | (q0 >> 41)) + q1; q1 = ((q1 << 45) | (q1 >> 19)) + q2; q2 = ((q2 << 23) | (q2 >> 41)) + q3; q3 = ((q3 << 45) | (q3 >> 19)) + q0;
        } else if (MODE == 5) {  // 8 x v_rcp_f32
#define X(n) f##n = __builtin_amdgcn_rcpf(f##n) + 0.5f;
            REP8(X)
#undef X
        } else if (MODE == 6) {  // 8 x sqrtf (IEEE)
#define X(n) f##n = sqrtf(f##n) + 1.0f;
            REP8(X)
#undef X
        } else if (MODE == 7) {  // 8 x (v_cmp -> ballot -> uniform branch)
#define X(n) if (__ballot(f##n > 0.5f) != 0) f##n = f##n * 0.999f + 0.001f;
            REP8(X)
#undef X
        } else if (MODE == 8) {  // 8 x v_max3
#define X(n) f##n = __builtin_fmaxf(__builtin_fmaxf(f##n * 0.99f, 0.5f), f0 * 0.5f);
            REP8(X)
#undef X
        } else if (MODE == 9) {  // 8 x plain v_cmp + v_cndmask with independent chains
#define X(n) f##n = (f##n > 1.0f) ? f##n * 0.5f : f##n + 0.25f;
            REP8(X)
#undef X
        }
    }
    uint32_t r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(q0 + q1 + q2 + q3) + (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
    if (r == 0x12345678u) out[0] = r;
}
template <int MODE>
void run(const char* name, int ops) {
    uint32_t* d; (void)hipMalloc(&d, 4);
    int iters = 20000, blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 100, 1); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double per = (double)blocks * 4 * iters * ops / (ms * 1e-3) / 1024.0;
    printf("%-40s %.3f ms  => %.2f cycles per source op @2.4GHz\n", name, ms, 2.4e9 / per);
    (void)hipFree(d);
}
int main() {
    run<0>("v_mul_lo_u32 (+add)", 8); run<1>("u64 * const + 1", 4); run<2>("xor+shift 32", 8); run<3>("xorshift 64", 4);
    run<4>("rotl64 + add64", 4); run<5>("v_rcp_f32 (+add)", 8); run<6>("sqrtf IEEE (+add)", 8); run<7>("cmp+ballot+uniform branch (+fma)", 8);
    run<8>("mul + max3", 8); run<9>("cmp + cndmask + mul/add", 8);
    return 0;
}
