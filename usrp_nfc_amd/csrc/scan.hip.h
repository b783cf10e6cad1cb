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

// ---- finite-state maps ---------------------------------------------------------
// A map sends every state to its successor; "f then g" is h[s] = g[f[s]].
// Decoder maps are stored one byte per state so that v_perm_b32 does four look-ups at once.
__host__ __device__ constexpr uint64_t identity_map(int nstates) {   // nibble form (framing machine, host tables)
    uint64_t m = 0;
    for (int s = 0; s < nstates; s++) m |= (uint64_t)s << (4 * s);
    return m;
}
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
// four look-ups into a 16-byte table g0..g3; sel holds four indices 0..15
__device__ __forceinline__ uint32_t lookup16x4(uint32_t g0, uint32_t g1, uint32_t g2, uint32_t g3, uint32_t sel) {
    const uint32_t s7 = sel & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(g1, g0, s7);
    const uint32_t hi = __builtin_amdgcn_perm(g3, g2, s7);
    const uint32_t m = ((sel >> 3) & 0x01010101u) * 0xFFu;
    return (hi & m) | (lo & ~m);
}

// Both decoders at once: Miller map (16 states, 16 bytes) + Manchester map (8 states, 8 bytes).
struct DecMaps {
    uint32_t mil[4];
    uint32_t man[2];
};
struct ComposeDec {
    using T = DecMaps;
    static __host__ __device__ __forceinline__ T identity() {
        return T{{0x03020100u, 0x07060504u, 0x0B0A0908u, 0x0F0E0D0Cu}, {0x03020100u, 0x07060504u}};
    }
    static T identity_host() { return identity(); }
    static __device__ __forceinline__ T op(const T &a, const T &b) {
        T r;
#pragma unroll
        for (int k = 0; k < 4; k++) r.mil[k] = lookup16x4(b.mil[0], b.mil[1], b.mil[2], b.mil[3], a.mil[k]);
        r.man[0] = __builtin_amdgcn_perm(b.man[1], b.man[0], a.man[0]);
        r.man[1] = __builtin_amdgcn_perm(b.man[1], b.man[0], a.man[1]);
        return r;
    }
    static __device__ __forceinline__ T shfl_up(const T &v, int d) {
        T r;
#pragma unroll
        for (int k = 0; k < 4; k++) r.mil[k] = (uint32_t)__shfl_up((int)v.mil[k], d, 64);
        r.man[0] = (uint32_t)__shfl_up((int)v.man[0], d, 64);
        r.man[1] = (uint32_t)__shfl_up((int)v.man[1], d, 64);
        return r;
    }
    // states packed as miller | manchester << 4
    static __device__ __forceinline__ uint32_t step(const T &m, uint32_t st) {
        const uint32_t a = lookup16x4(m.mil[0], m.mil[1], m.mil[2], m.mil[3], st & 15u) & 15u;
        const uint32_t b = __builtin_amdgcn_perm(m.man[1], m.man[0], (st >> 4) & 7u) & 15u;
        return a | (b << 4);
    }
};

// Packet framing: 2 states (started or not) -> 8-bit nibble map, kept in a u32.
struct ComposePkt {
    using T = uint32_t;
    static __host__ __device__ __forceinline__ T identity() { return (uint32_t)identity_map(2); }
    static T identity_host() { return identity(); }
    static __device__ __forceinline__ T op(T a, T b) { return (uint32_t)compose_map<2>(a, b); }
    static __device__ __forceinline__ T shfl_up(T v, int d) { return (T)__shfl_up((int)v, d, 64); }
    static __device__ __forceinline__ uint32_t step(T m, uint32_t st) { return (m >> (4 * st)) & 1u; }
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
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_reduce(size_t n, const uint32_t *n_dev, Load load, typename Tr::T *partials) {
    using T = typename Tr::T;
    if (n_dev) n = min(n, (size_t)*n_dev);   // the item count may live on the device (n is then the capacity)
    if ((size_t)blockIdx.x * SCAN_BLOCK * ITEMS >= n) return;
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
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_partials(size_t nparts, const uint32_t *n_dev, uint32_t tile,
                                                             typename Tr::T *partials, typename Tr::T seed,
                                                             typename Tr::T *total_out) {
    using T = typename Tr::T;
    if (n_dev) nparts = min(nparts, ((size_t)*n_dev + tile - 1) / tile);
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
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(size_t n, const uint32_t *n_dev, Load load, Store store,
                                                          const typename Tr::T *partials) {
    using T = typename Tr::T;
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * SCAN_BLOCK * ITEMS >= n) return;
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

// tiles of BLOCK*ITEMS items
template <int ITEMS>
inline size_t scan_num_tiles(size_t n) {
    const size_t tile = (size_t)SCAN_BLOCK * ITEMS;
    return (n + tile - 1) / tile;
}

// ---- state-tracking variant for finite-state maps ---------------------------------
// Pass 1 also keeps every thread's aggregate, so that pass 2 needs one block scan per thread and then
// walks its items applying each map to a STATE (one look-up) instead of composing maps.
template <class Tr, int ITEMS, class Load>
__global__ __launch_bounds__(SCAN_BLOCK) void k_fsm_reduce(size_t n, const uint32_t *n_dev, Load load, typename Tr::T *partials,
                                                          typename Tr::T *aggs) {
    using T = typename Tr::T;
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * SCAN_BLOCK * ITEMS >= n) return;
    __shared__ T lds[SCAN_WAVES];
    const size_t tid = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const size_t base = tid * ITEMS;
    T agg = Tr::identity();
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const size_t idx = base + i;
        if (idx < n) agg = Tr::op(agg, load(idx));
    }
    aggs[tid] = agg;
    T total;
    (void)block_exclusive<Tr>(agg, lds, total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}
template <class Tr, int ITEMS, class Load, class Visit>
__global__ __launch_bounds__(SCAN_BLOCK) void k_fsm_apply(size_t n, const uint32_t *n_dev, Load load, Visit visit,
                                                         const typename Tr::T *partials, const typename Tr::T *aggs,
                                                         uint32_t state0) {
    using T = typename Tr::T;
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * SCAN_BLOCK * ITEMS >= n) return;
    __shared__ T lds[SCAN_WAVES];
    const size_t tid = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const size_t base = tid * ITEMS;
    T total;
    const T excl = block_exclusive<Tr>(aggs[tid], lds, total);
    uint32_t st = Tr::step(Tr::op(partials[blockIdx.x], excl), state0);
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const size_t idx = base + i;
        if (idx < n) {
            const T item = load(idx);
            visit(idx, st, item);
            st = Tr::step(item, st);
        }
    }
}
template <int ITEMS>
inline size_t fsm_num_threads(size_t n) { return scan_num_tiles<ITEMS>(n) * SCAN_BLOCK; }

// visit(i, state before item i, item map); *total_out = composition of all maps.
// n_dev != nullptr: the true item count is *n_dev on the device and n is only the capacity the grid is sized for.
template <class Tr, int ITEMS, class Load, class Visit>
inline void device_fsm_scan(hipStream_t st, size_t n, const uint32_t *n_dev, Load load, Visit visit, uint32_t state0,
                            typename Tr::T *partials, typename Tr::T *aggs, typename Tr::T *total_out) {
    const size_t tiles = scan_num_tiles<ITEMS>(n);
    if (tiles)
        hipLaunchKernelGGL((k_fsm_reduce<Tr, ITEMS, Load>), dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, n, n_dev, load, partials,
                           aggs);
    hipLaunchKernelGGL((k_scan_partials<Tr>), dim3(1), dim3(SCAN_BLOCK), 0, st, tiles, n_dev, (uint32_t)(SCAN_BLOCK * ITEMS),
                       partials, Tr::identity_host(), total_out);
    if (tiles)
        hipLaunchKernelGGL((k_fsm_apply<Tr, ITEMS, Load, Visit>), dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, n, n_dev, load,
                           visit, partials, aggs, state0);
}

// Host-side driver.  `partials` must hold scan_num_tiles(n, ITEMS) entries.

// Phase 1: tile aggregates -> exclusive tile prefixes (seeded) and the grand total in *total_out.
template <class Tr, int ITEMS, class Load>
inline void scan_phase1(hipStream_t st, size_t n, const uint32_t *n_dev, Load load, typename Tr::T seed, typename Tr::T *partials,
                        typename Tr::T *total_out) {
    const size_t tiles = scan_num_tiles<ITEMS>(n);
    if (tiles)
        hipLaunchKernelGGL((k_scan_reduce<Tr, ITEMS, Load>), dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, n, n_dev, load, partials);
    hipLaunchKernelGGL((k_scan_partials<Tr>), dim3(1), dim3(SCAN_BLOCK), 0, st, tiles, n_dev, (uint32_t)(SCAN_BLOCK * ITEMS), partials,
                       seed, total_out);
}
// Phase 2: every item gets its exclusive prefix.
template <class Tr, int ITEMS, class Load, class Store>
inline void scan_phase2(hipStream_t st, size_t n, const uint32_t *n_dev, Load load, Store store, const typename Tr::T *partials) {
    const size_t tiles = scan_num_tiles<ITEMS>(n);
    if (tiles)
        hipLaunchKernelGGL((k_scan_apply<Tr, ITEMS, Load, Store>), dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, st, n, n_dev, load,
                           store, partials);
}
template <class Tr, int ITEMS, class Load, class Store>
inline void device_scan(hipStream_t st, size_t n, const uint32_t *n_dev, Load load, Store store, typename Tr::T seed,
                        typename Tr::T *partials, typename Tr::T *total_out) {
    scan_phase1<Tr, ITEMS, Load>(st, n, n_dev, load, seed, partials, total_out);
    scan_phase2<Tr, ITEMS, Load, Store>(st, n, n_dev, load, store, partials);
}

}  // namespace nfc
