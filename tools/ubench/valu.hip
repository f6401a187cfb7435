// Issue rate of the VALU instructions the aligner / extension kernels are made of (round 3): eight independent accumulators per
// lane, 8 waves per SIMD, every CU busy; cycles per wave-instruction per SIMD at the shader clock MEASURED inside each launch
// (s_memtime against the 100 MHz s_memrealtime; round 3 assumed a nominal 2.4 GHz).  Also: the same with ONE accumulator
// (a dependent chain) -- what a recurrence like the banded DP sees.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu.hip -o tools/ubench/valu && tools/ubench/valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define OPS(X) X(0, "v_add_u32 %0, %0, %1") X(1, "v_pk_add_u16 %0, %0, %1") X(2, "v_pk_max_i16 %0, %0, %1") X(3, "v_perm_b32 %0, %0, %1, %1") \
               X(4, "v_and_or_b32 %0, %0, %1, %1") X(5, "v_max_i32 %0, %0, %1") X(6, "v_max3_i32 %0, %0, %1, %1") X(7, "v_lshl_or_b32 %0, %0, 3, %1") \
               X(8, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") X(9, "v_cndmask_b32 %0, %0, %1, vcc") X(10, "v_pk_add_i16 %0, %0, %1 clamp") \
               X(11, "v_sub_u32 %0, %0, %1") X(12, "v_and_b32 %0, %0, %1") X(13, "v_or_b32 %0, %0, %1") X(14, "v_xor_b32 %0, %0, %1") X(15, "v_lshlrev_b32 %0, 1, %0") \
               X(16, "v_min_u32 %0, %0, %1") X(17, "v_max_u32 %0, %0, %1") X(18, "v_add3_u32 %0, %0, %1, %1") X(19, "v_mov_b32 %0, %1") X(20, "v_bfe_u32 %0, %0, 3, 8") \
               X(21, "v_mad_u32_u24 %0, %0, %1, %1") X(22, "v_add_u16 %0, %0, %1") X(23, "v_ashrrev_i32 %0, 1, %0") X(24, "v_cmp_lt_u32 vcc, %0, %1") \
               X(25, "v_add_f32 %0, %0, %1") X(26, "v_fma_f32 %0, %0, %1, %1") X(27, "v_sad_u8 %0, %0, %1, %1") X(28, "v_alignbit_b32 %0, %0, %1, 8") X(29, "v_lshl_add_u32 %0, %0, 2, %1") \
               X(30, "v_max_i16 %0, %0, %1") X(31, "v_pk_sub_i16 %0, %0, %1") X(32, "v_add_co_u32 %0, vcc, %0, %1") X(33, "v_max_f32 %0, %0, %1") X(34, "v_pk_max_f16 %0, %0, %1") \
               /* round 4: the select as the kernels use it -- behind the compare that produced its mask (op 9 reads a VCC nothing in the loop \
                  has written: its 23 cycles were that, see the two lines below) -- through VCC and through an SGPR pair; TWO instructions each */ \
               X(35, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc") X(36, "v_cmp_lt_u32_e64 s[40:41], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[40:41]") \
               X(37, "v_cmp_lt_i32_e64 s[40:41], %0, %1\n\tv_max_i32 %0, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[40:41]")

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void k(int iters, unsigned seed, unsigned* out, unsigned long long* clk) {
    unsigned a[8], b = seed + threadIdx.x;
    // shader clock of THIS launch: s_memtime (core clock ticks) against s_memrealtime (constant 100 MHz), first lane of workgroup 0
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned& x = a[CHAINS == 1 ? 0 : i];
#define X(N, S) if (OP == N) asm volatile(S : "+v"(x) : "v"(b) : "vcc", "s40", "s41");
                OPS(X)
#undef X
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 0x12345) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = clock64() - c0, clk[1] = wall_clock64() - w0;
}

template <int OP, int CHAINS>
int run(const char* name, unsigned* o, unsigned long long* dclk, int per_op = 1) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = 256 * 8, iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, iters, 1u, o, dclk);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long hc[2] = {0, 0};
    CK(hipMemcpy(hc, dclk, sizeof hc, hipMemcpyDeviceToHost));
    const double mhz = hc[1] ? (double)hc[0] / (double)hc[1] * 100.0 : 0.0;   // measured shader clock of the launch
    const double per_simd = (double)blocks * 4 / 1024 * iters * 32 * per_op;   // wave-instructions per SIMD
    printf("%-70s %s: %7.3f ms = %5.2f cycles per wave-instruction per SIMD at the MEASURED %.0f MHz (%5.2f at a nominal 2400)\n", name,
           CHAINS == 1 ? "1 chain " : "8 chains", ms, ms * 1e-3 * mhz * 1e6 / per_simd, mhz, ms * 1e-3 * 2.4e9 / per_simd);
    return 0;
}

int main() {
    unsigned* o;
    unsigned long long* dclk;
    CK(hipMalloc(&o, 64));
    CK(hipMalloc(&dclk, 64));
#define X(N, S) run<N, 8>(S, o, dclk, N == 37 ? 3 : N >= 35 ? 2 : 1); run<N, 1>(S, o, dclk, N == 37 ? 3 : N >= 35 ? 2 : 1);
    OPS(X)
#undef X
    return 0;
}
