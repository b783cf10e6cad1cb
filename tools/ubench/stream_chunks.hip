// How fast can MI355X read 0.8 GB when every wavefront streams its own contiguous time chunk (the access pattern of the
// threshold kernel: 5074 waves x 19712 samples x 8 B, 2 KB per wave and step) -- against a plain grid-wide streaming read?
// build: hipcc --offload-arch=gfx950 -O3 -o stream_chunks stream_chunks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// one wave per chunk, `PF` steps of 4 x 512 B in flight per wave
template <int PF>
__global__ __launch_bounds__(256) void k_chunks(const float2 *in, size_t n, uint32_t C, float *out) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * 4 + wave;
    const size_t m0 = (size_t)c * C;
    if (m0 >= n) return;
    const size_t m1 = m0 + C < n ? m0 + C : n;
    float2 r[PF][4];
    float acc = 0.f;
    const size_t last = m1 - 256;
#pragma unroll
    for (int k = 0; k < PF; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) r[k][j] = in[(m0 + 256 * k < last ? m0 + 256 * k : last) + 64 * j + lane];
    for (size_t b = m0; b + 256 * PF <= m1; b += 256 * PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) s += r[k][j].x * r[k][j].x + r[k][j].y * r[k][j].y;
            acc += s;
            const size_t nb = b + 256 * (k + PF);
#pragma unroll
            for (int j = 0; j < 4; j++) r[k][j] = in[(nb < last ? nb : last) + 64 * j + lane];
        }
    }
    if (acc == 12345.f) out[0] = acc;
}
// plain streaming: consecutive waves read consecutive 1 KB (float4 per lane), grid-stride
__global__ __launch_bounds__(256) void k_plain(const float4 *in, size_t n4, float *out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = in[i];
        acc += v.x * v.x + v.y * v.y + v.z + v.w;
    }
    if (acc == 12345.f) out[0] = acc;
}
// chunked like k_chunks, but a chunk per WORKGROUP of 4 waves (each step 4 x 2 KB = 8 KB contiguous per block)
template <int PF>
__global__ __launch_bounds__(256) void k_chunks_wg(const float2 *in, size_t n, uint32_t C, float *out) {
    extern __shared__ char smem[];
    const size_t m0 = (size_t)blockIdx.x * C;
    if (m0 >= n) return;
    const size_t m1 = m0 + C < n ? m0 + C : n;
    float acc = 0.f;
    for (size_t b = m0 + threadIdx.x; b < m1; b += 256 * PF) {
        float2 r[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) r[k] = in[b + 256 * k < m1 ? b + 256 * k : m1 - 1];
#pragma unroll
        for (int k = 0; k < PF; k++) acc += r[k].x * r[k].x + r[k].y * r[k].y;
    }
    if (acc == 12345.f) out[0] = acc;
}

// the round-3/4 threshold kernel's pattern (k_threshold_wg): a chunk per 256-thread workgroup, four resident per CU; a ROUND is 1024
// samples, wave w takes its 256-sample step (four 512-byte rows) of every round, asked for D rounds ahead; one barrier per round
template <int D, bool NT = false>
__global__ __launch_bounds__(256) void k_chunks_round(const float2 *in, size_t n, uint32_t C, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t m0 = (size_t)blockIdx.x * C;
    if (m0 >= n) return;
    const size_t m1 = m0 + C < n ? m0 + C : n;
    const size_t rounds = (m1 - m0) / 1024;
    float2 r[D][4];
    float acc = 0.f;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    auto at = [&](size_t rd, int j) {
        const size_t q = rd < rounds ? rd : rounds - 1;
        const size_t i = m0 + q * 1024 + 256 * wave + 64 * j + lane;
        if constexpr (NT) {   // (the non-temporal hint, as k_threshold_wg's loads carry it since the end of round 4)
            const f32x2_t v = __builtin_nontemporal_load((const f32x2_t *)in + i);
            return make_float2(v.x, v.y);
        } else {
            return in[i];
        }
    };
#pragma unroll
    for (int k = 0; k < D; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) r[k][j] = at(k, j);
    for (size_t rd = 0; rd < rounds; rd += D) {
#pragma unroll
        for (int k = 0; k < D; k++) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) s += r[k][j].x * r[k][j].x + r[k][j].y * r[k][j].y;
            acc += s;
#pragma unroll
            for (int j = 0; j < 4; j++) r[k][j] = at(rd + k + D, j);
            __syncthreads();
        }
    }
    if (acc == 12345.f) out[0] = acc;
}

template <class F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) f();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const size_t n = 100000000;
    float2 *d;
    float *o;
    hipMalloc(&d, n * 8);
    hipMalloc(&o, 4);
    hipMemset(d, 0, n * 8);
    const double gb = n * 8 / 1e9;
    for (int lds : {80000, 32000}) {
        const uint32_t waves = 256 * (160 * 1024 / lds) * 4;
        uint32_t C = (uint32_t)((n + waves - 1) / waves);
        C = (C + 1023) / 1024 * 1024;
        const uint32_t nch = (uint32_t)((n + C - 1) / C);
        const uint32_t blocks = (nch + 3) / 4;
        hipFuncSetAttribute((const void *)k_chunks<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void *)k_chunks<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void *)k_chunks<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        float ms;
        ms = timeit([&] { hipLaunchKernelGGL(k_chunks<1>, dim3(blocks), dim3(256), lds, 0, d, n, C, o); });
        printf("wave chunks (C %u, %u waves, LDS %d) PF 1: %.3f ms  %.2f TB/s\n", C, nch, lds, ms, gb / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_chunks<2>, dim3(blocks), dim3(256), lds, 0, d, n, C, o); });
        printf("wave chunks (C %u, %u waves, LDS %d) PF 2: %.3f ms  %.2f TB/s\n", C, nch, lds, ms, gb / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_chunks<4>, dim3(blocks), dim3(256), lds, 0, d, n, C, o); });
        printf("wave chunks (C %u, %u waves, LDS %d) PF 4: %.3f ms  %.2f TB/s\n", C, nch, lds, ms, gb / ms);
        hipFuncSetAttribute((const void *)k_chunks<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        ms = timeit([&] { hipLaunchKernelGGL(k_chunks<8>, dim3(blocks), dim3(256), lds, 0, d, n, C, o); });
        printf("wave chunks (C %u, %u waves, LDS %d) PF 8: %.3f ms  %.2f TB/s\n", C, nch, lds, ms, gb / ms);
    }
    {
        const int lds = 32000;
        const uint32_t wgs = 256 * 5;
        uint32_t C = (uint32_t)((n + wgs - 1) / wgs);
        C = (C + 1023) / 1024 * 1024;
        const uint32_t blocks = (uint32_t)((n + C - 1) / C);
        hipFuncSetAttribute((const void *)k_chunks_wg<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        float ms = timeit([&] { hipLaunchKernelGGL(k_chunks_wg<4>, dim3(blocks), dim3(256), lds, 0, d, n, C, o); });
        printf("workgroup chunks (C %u, %u blocks) 4 loads in flight per lane: %.3f ms  %.2f TB/s\n", C, blocks, ms, gb / ms);
    }
    for (int per_cu : {4, 3, 8}) {
        const uint32_t wgs = 256 * per_cu;
        uint32_t C = (uint32_t)((n + wgs - 1) / wgs);
        C = (C + 1023) / 1024 * 1024;
        const uint32_t blocks = (uint32_t)((n + C - 1) / C);
        float ms = timeit([&] { hipLaunchKernelGGL(k_chunks_round<1>, dim3(blocks), dim3(256), 0, 0, d, n, C, o); });
        printf("k_threshold_wg's pattern (chunk per workgroup, %d per CU: C %u, %u blocks; a wave's step of every 1024-sample round, 1 round ahead, a barrier per round): %.3f ms  %.2f TB/s\n", per_cu, C, blocks, ms, gb / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_chunks_round<2>, dim3(blocks), dim3(256), 0, 0, d, n, C, o); });
        printf("   ... 2 rounds ahead: %.3f ms  %.2f TB/s\n", ms, gb / ms);
        ms = timeit([&] { hipLaunchKernelGGL((k_chunks_round<1, true>), dim3(blocks), dim3(256), 0, 0, d, n, C, o); });
        printf("   ... 1 round ahead, non-temporal loads: %.3f ms  %.2f TB/s\n", ms, gb / ms);
    }
    for (int g : {256 * 8, 256 * 20, 256 * 64}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_plain, dim3(g), dim3(256), 0, 0, (const float4 *)d, n / 2, o); });
        printf("plain streaming read, %d blocks: %.3f ms  %.2f TB/s\n", g, ms, gb / ms);
    }
    return 0;
}
