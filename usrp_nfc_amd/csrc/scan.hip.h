// scan.hip.h -- device-wide ordered exclusive scan for gfx950 (wave64), generic in
// the (possibly non-commutative) operator.  Reduce-then-scan in three launches:
//   k_scan_reduce   : one aggregate per tile of BLOCK*ITEMS items
//   k_scan_partials : one workgroup turns the tile aggregates into exclusive prefixes
//   k_scan_apply    : re-reads the items, hands each its exclusive prefix
// Items are produced by a Load functor (so a stage can compute its item on the
// fly from whatever it reads) and consumed by a Store functor.  A blocked
// arrangement (thread t owns ITEMS consecutive items) keeps the order, which the
// transducer compositions need.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nfc {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_WAVES = SCAN_BLOCK / 64;

// ---- operator traits -------------------------------------------------------
struct AddU32 {
    using T = uint32_t;
    static __device__ __forceinline__ T identity() { return 0u; }
    static __device__ __forceinline__ T op(T a, T b) { return a + b; }
    static __device__ __forceinline__ T shfl_up(T v, int d) { return (T)__shfl_up((int)v, d, 64); }
    static __device__ __forceinline__ T shfl(T v, int l) { return (T)__shfl((int)v, l, 64); }
};

struct AddU64 {  // also used as two packed u32 counters (no carry between halves while each < 2^32)
    using T = uint64_t;
    static __device__ __forceinline__ T identity() { return 0ull; }
    static __device__ __forceinline__ T op(T a, T b) { return a + b; }
    static __device__ __forceinline__ T shfl_up(T v, int d) {
        int lo = __shfl_up((int)(uint32_t)v, d, 64), hi = __shfl_up((int)(uint32_t)(v >> 32), d, 64);
        return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    }
    static __device__ __forceinline__ T shfl(T v, int l) {
        int lo = __shfl((int)(uint32_t)v, l, 64), hi = __shfl((int)(uint32_t)(v >> 32), l, 64);
        return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
    }
};

// State maps packed 4 bits per state: h = "f then g", h[s] = g[f[s]].
template <int NSTATES>
__device__ __forceinline__ uint64_t compose_map(uint64_t f, uint64_t g) {
    uint64_t h = 0;
#pragma unroll
    for (int s = 0; s < NSTATES; s++) {
        const uint32_t fs = (uint32_t)(f >> (4 * s)) & 15u;
        h |= ((g >> (4 * fs)) & 15ull) << (4 * s);
    }
    return h;
}
__host__ __device__ constexpr uint64_t identity_map(int nstates) {
    uint64_t m = 0;
    for (int s = 0; s < nstates; s++) m |= (uint64_t)s << (4 * s);
    return m;
}

// Both decoders at once: Miller map (16 states, 64 bits) + Manchester map (8 states, 32 bits).
struct DecMaps {
    uint64_t mil;
    uint32_t man;
};
struct ComposeDec {
    using T = DecMaps;
    static __device__ __forceinline__ T identity() { return T{identity_map(16), (uint32_t)identity_map(8)}; }
    static __device__ __forceinline__ T op(T a, T b) {
        return T{compose_map<16>(a.mil, b.mil), (uint32_t)compose_map<8>(a.man, b.man)};
    }
    static __device__ __forceinline__ T shfl_up(T v, int d) {
        T r;
        r.mil = AddU64::shfl_up(v.mil, d);
        r.man = (uint32_t)__shfl_up((int)v.man, d, 64);
        return r;
    }
    static __device__ __forceinline__ T shfl(T v, int l) {
        T r;
        r.mil = AddU64::shfl(v.mil, l);
        r.man = (uint32_t)__shfl((int)v.man, l, 64);
        return r;
    }
};

// Packet framing: 2 states (started or not) -> 8-bit map, kept in a u32.
struct ComposePkt {
    using T = uint32_t;
    static __device__ __forceinline__ T identity() { return (uint32_t)identity_map(2); }
    static __device__ __forceinline__ T op(T a, T b) { return (uint32_t)compose_map<2>(a, b); }
    static __device__ __forceinline__ T shfl_up(T v, int d) { return (T)__shfl_up((int)v, d, 64); }
    static __device__ __forceinline__ T shfl(T v, int l) { return (T)__shfl((int)v, l, 64); }
};

// ---- block-level helpers ----------------------------------------------------
// Inclusive scan of one value per thread across the block; returns the exclusive
// prefix of this thread and the block total.  lds must hold SCAN_WAVES entries.
template <class Tr>
__device__ __forceinline__ typename Tr::T block_exclusive(typename Tr::T v, typename Tr::T *lds,
                                                          typename Tr::T &block_total) {
    using T = typename Tr::T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T up = Tr::shfl_up(inc, d);
        if (lane >= d) inc = Tr::op(up, inc);
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    T wave_prefix = Tr::identity();
    T total = Tr::identity();
#pragma unroll
    for (int w = 0; w < SCAN_WAVES; w++) {
        T x = lds[w];
        if (w < wave) wave_prefix = Tr::op(wave_prefix, x);
        total = Tr::op(total, x);
    }
    __syncthreads();
    block_total = total;
    T excl = Tr::shfl_up(inc, 1);
    if (lane == 0) excl = Tr::identity();
    return Tr::op(wave_prefix, excl);
}

// ---- the three kernels --------------------------------------------------------
template <class Tr, int ITEMS, class Load>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_reduce(size_t n, Load load, typename Tr::T *partials) {
    using T = typename Tr::T;
    __shared__ T lds[SCAN_WAVES];
    const size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * ITEMS;
    T agg = Tr::identity();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const size_t idx = base + i;
        if (idx < n) agg = Tr::op(agg, load(idx));
    }
    T total;
    (void)block_exclusive<Tr>(agg, lds, total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// partials[i] <- op(seed, exclusive prefix of partials)[i]; total (with the seed) stored to *total_out.
template <class Tr>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_partials(size_t nparts, typename Tr::T *partials,
                                                             typename Tr::T seed, typename Tr::T *total_out) {
    using T = typename Tr::T;
    __shared__ T lds[SCAN_WAVES];
    T carry = seed;
    for (size_t base = 0; base < nparts; base += SCAN_BLOCK) {
        const size_t i = base + threadIdx.x;
        T v = (i < nparts) ? partials[i] : Tr::identity();
        T total;
        T excl = block_exclusive<Tr>(v, lds, total);
        if (i < nparts) partials[i] = Tr::op(carry, excl);
        carry = Tr::op(carry, total);
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <class Tr, int ITEMS, class Load, class Store>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(size_t n, Load load, Store store,
                                                          const typename Tr::T *partials) {
    using T = typename Tr::T;
    __shared__ T lds[SCAN_WAVES];
    const size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * ITEMS;
    T item[ITEMS];
    T agg = Tr::identity();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const size_t idx = base + i;
        item[i] = (idx < n) ? load(idx) : Tr::identity();
        agg = Tr::op(agg, item[i]);
    }
    T total;
    T excl = block_exclusive<Tr>(agg, lds, total);
    T run = Tr::op(partials[blockIdx.x], excl);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const size_t idx = base + i;
        if (idx < n) store(idx, run, item[i]);
        run = Tr::op(run, item[i]);
    }
}

// Host-side driver.  `partials` must hold scan_num_tiles(n, ITEMS) entries.
template <int ITEMS>
inline size_t scan_num_tiles(size_t n) {
    const size_t tile = (size_t)SCAN_BLOCK * ITEMS;
    return (n + tile - 1) / tile;
}

// Phase 1: tile aggregates -> exclusive tile prefixes (seeded) and the grand total in *total_out.
template <class Tr, int ITEMS, class Load>
inline void scan_phase1(hipStream_t st, size_t n, Load load, typename Tr::T seed, typename Tr::T *partials,
                        typename Tr::T *total_out) {
    const size_t tiles = scan_num_tiles<ITEMS>(n);
    if (tiles)
        hipLaunchKernelGGL((k_scan_reduce<Tr, ITEMS, Load>), dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, n, load, partials);
    hipLaunchKernelGGL((k_scan_partials<Tr>), dim3(1), dim3(SCAN_BLOCK), 0, st, tiles, partials, seed, total_out);
}
// Phase 2: every item gets its exclusive prefix.
template <class Tr, int ITEMS, class Load, class Store>
inline void scan_phase2(hipStream_t st, size_t n, Load load, Store store, const typename Tr::T *partials) {
    const size_t tiles = scan_num_tiles<ITEMS>(n);
    if (tiles)
        hipLaunchKernelGGL((k_scan_apply<Tr, ITEMS, Load, Store>), dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, n, load,
                           store, partials);
}
template <class Tr, int ITEMS, class Load, class Store>
inline void device_scan(hipStream_t st, size_t n, Load load, Store store, typename Tr::T seed,
                        typename Tr::T *partials, typename Tr::T *total_out) {
    scan_phase1<Tr, ITEMS, Load>(st, n, load, seed, partials, total_out);
    scan_phase2<Tr, ITEMS, Load, Store>(st, n, load, store, partials);
}

}  // namespace nfc
