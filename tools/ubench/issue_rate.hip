// Issue-rate microbenchmark for gfx950: how many cycles does a SIMD spend per instruction when 5 waves per SIMD run a mix of
// independent VALU and SALU instructions?  (Shapes the instruction budget of k_threshold_lean: DESIGN.md.)
// build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip ; run: ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *clk) {
    extern __shared__ char smem[];
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f;
    unsigned long long m0 = 1, m1 = 2, m2 = 3, m3 = 4;
    int s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 64 VALU
            asm volatile(REP16("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (MODE == 1) {   // 64 SALU (32-bit)
            asm volatile(REP16("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3\n")
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 2) {   // 64 VALU + 64 SALU interleaved
            asm volatile(REP16("v_add_f32 %0, %0, %0\n s_add_u32 %4, %4, %4\n v_add_f32 %1, %1, %1\n s_add_u32 %5, %5, %5\n v_add_f32 %2, %2, %2\n s_add_u32 %6, %6, %6\n v_add_f32 %3, %3, %3\n s_add_u32 %7, %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 3) {   // 64 SALU 64-bit mask ops
            asm volatile(REP16("s_or_b64 %0, %0, %1\n s_and_b64 %1, %1, %2\n s_or_b64 %2, %2, %3\n s_and_b64 %3, %3, %0\n")
                         : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");
        } else if (MODE == 4) {   // 64 v_cmp into SGPR pairs (VALU writing SGPR)
            asm volatile(REP16("v_cmp_lt_f32 %4, %0, %1\n v_cmp_lt_f32 %5, %1, %2\n v_cmp_lt_f32 %6, %2, %3\n v_cmp_lt_f32 %7, %3, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3));
        } else if (MODE == 5) {   // dependent VALU chain of 64
            asm volatile(REP16("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n") : "+v"(a0));
        } else if (MODE == 6) {   // v_cmp -> s_or (VALU->SALU dependency), 32 + 32
            asm volatile(REP16("v_cmp_lt_f32 %4, %0, %1\n s_or_b64 %5, %5, %4\n v_cmp_lt_f32 %6, %2, %3\n s_or_b64 %7, %7, %6\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");
        } else if (MODE == 7) {   // 64 v_pk_add_f32
            asm volatile(REP16("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n")
                         : "+v"(m0), "+v"(m1));
        }
    }
    if (a0 + a1 + a2 + a3 + (float)(s0 + s1 + s2 + s3) + (float)(m0 + m1 + m2 + m3) == 12345.f) out[0] = a0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

template <int MODE>
void run(const char *name, int per_iter, int waves_per_simd) {
    float *d;
    hipMalloc(&d, 4);
    unsigned long long *dc, hc[2];
    hipMalloc(&dc, 16);
    const int iters = 2000;
    const size_t lds = 160 * 1024 / waves_per_simd - 512;   // one 256-thread block = 1 wave per SIMD; LDS limits blocks per CU
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, iters, dc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, iters, dc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)hc[0] / ((double)hc[1] * 10.0);   // wall_clock64 ticks at 100 MHz
    const double cyc = ms * 1e-3 * ghz * 1e9;
    const double inst_per_simd = (double)iters * per_iter * waves_per_simd;
    printf("%-44s waves/SIMD %d: %.3f ms, %.2f cycles per instruction per SIMD (shader clock %.2f GHz by s_memtime / s_memrealtime)\n", name, waves_per_simd, ms, cyc / inst_per_simd, ghz);
}

int main() {
    for (int w : {1, 2, 5}) {
        run<0>("64 independent VALU (v_add_f32)", 64, w);
        run<5>("64 dependent VALU", 64, w);
        run<7>("64 v_pk_add_f32", 64, w);
        run<1>("64 SALU (s_add_u32)", 64, w);
        run<3>("64 SALU (s_or/and_b64)", 64, w);
        run<2>("64 VALU + 64 SALU interleaved (128)", 128, w);
        run<4>("64 v_cmp -> SGPR pair", 64, w);
        run<6>("32 v_cmp + 32 dependent s_or_b64 (64)", 64, w);
    }
    return 0;
}
