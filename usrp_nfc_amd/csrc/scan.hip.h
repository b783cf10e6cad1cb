// scan.hip.h -- device-wide ordered exclusive scan for gfx950 (wave64), generic in
// the (possibly non-commutative) operator.  Reduce-then-scan:
//   a stage's first kernel: one aggregate per tile (k_edge_reduce, k_dec_reduce; a pass may also leave the aggregates of the
//                     NEXT scan, as k_dec_apply does for the framing scan)
//   tile prefixes   : while the tiles are few, every tile's workgroup folds its predecessors' aggregates itself
//                     (tile_prefix); beyond OWN_PREFIX_MAX_TILES one workgroup does it in a launch (k_scan_partials)
//   the stage's later kernels re-read their items and give each its exclusive prefix.
// A blocked arrangement (thread t owns consecutive items) keeps the order, which the
// transducer compositions need.
// Every launch costs a few microseconds whatever it does, so stages chain their scans (a pass leaves the tile aggregates of
// the next scan), and the single-workgroup partials pass takes an epilogue for the one-thread bookkeeping that follows a scan.
// (Measured and dropped: long batches folding in two levels -- groups of tiles, then the own group -- instead of the partials
// launch: every tile's workgroup then pays two more block scans, 0.315 vs 0.260 ms for the edge stage of a 1e9-sample batch.)
#pragma once
#include <hip/hip_runtime.h>

#include "launch_check.h"
#include <stdint.h>

namespace nfc {

#ifdef NFC_TAIL_PROF
// (a profiling build: s_memtime stamps of every workgroup's thread 0, [kernel][workgroup][stamp]; read by nfc_debug_tail_prof)
constexpr int TP_WGS = 4096;
__device__ unsigned long long g_tail_prof[4 * TP_WGS * 8];
#define TP_DECL() unsigned long long tp_t[8]; int tp_n = 0; tp_t[tp_n++] = clock64()
#define TP_MARK() tp_t[tp_n++] = clock64()
#define TP_DONE(slot)                                                                                             \
    do {                                                                                                          \
        tp_t[tp_n++] = clock64();                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < TP_WGS) {                                                            \
            unsigned long long *o_ = g_tail_prof + ((size_t)(slot) * TP_WGS + blockIdx.x) * 8;                    \
            for (int i_ = 0; i_ < 8; i_++) o_[i_] = i_ < tp_n ? tp_t[i_] : 0ull;                                  \
        }                                                                                                         \
    } while (0)
#else
#define TP_DECL() ((void)0)
#define TP_MARK() ((void)0)
#define TP_DONE(slot) ((void)0)
#endif

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_WAVES = SCAN_BLOCK / 64;

// ---- operator traits -------------------------------------------------------
struct AddU32 {
    using T = uint32_t;
    static __device__ __forceinline__ T identity() { return 0u; }
    static __device__ __forceinline__ T op(T a, T b) { return a + b; }
    static __device__ __forceinline__ T shfl_up(T v, int d) { return (T)__shfl_up((int)v, d, 64); }
};

// ---- finite-state maps ---------------------------------------------------------
// A map sends every state to its successor; "f then g" is h[s] = g[f[s]].
// Decoder maps are stored one byte per state so that v_perm_b32 does four look-ups at once.
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

// ---- wave-level scan on DPP -----------------------------------------------------
// An ordered inclusive scan across the 64 lanes with data-parallel-primitive moves (one VALU instruction per word and
// step; a __shfl_up is an address computation plus an LDS permute per word).  Lanes a move does not reach keep `old`,
// which is the operator's identity, so every step is an unconditional op(up, x).
template <int CTRL, int ROWMASK, class T>
__device__ __forceinline__ T dpp_move(const T &old, const T &v) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    constexpr int W = sizeof(T) / 4;
    uint32_t o[W], x[W], r[W];
    __builtin_memcpy(o, &old, sizeof(T));
    __builtin_memcpy(x, &v, sizeof(T));
#pragma unroll
    for (int i = 0; i < W; i++) r[i] = (uint32_t)__builtin_amdgcn_update_dpp((int)o[i], (int)x[i], CTRL, ROWMASK, 0xF, false);
    T out;
    __builtin_memcpy(&out, r, sizeof(T));
    return out;
}
// The helpers below take the operator as an OBJECT (type T, identity(), operator()(a, b)), so that an operator may carry
// values (edges.hip.h: EdgeAggOp divides by max_len); the trait forms (static identity / op) wrap into StaticOp.
template <class Tr>
struct StaticOp {
    using T = typename Tr::T;
    __device__ __forceinline__ T identity() const { return Tr::identity(); }
    __device__ __forceinline__ T operator()(const T &a, const T &b) const { return Tr::op(a, b); }
};
template <class Op>
__device__ __forceinline__ typename Op::T wave_inclusive_with(const Op &op, typename Op::T x) {
    using T = typename Op::T;
    const T id = op.identity();
    x = op(dpp_move<0x111, 0xF>(id, x), x);   // row_shr:1   (rows of 16 lanes)
    x = op(dpp_move<0x112, 0xF>(id, x), x);   // row_shr:2
    x = op(dpp_move<0x114, 0xF>(id, x), x);   // row_shr:4
    x = op(dpp_move<0x118, 0xF>(id, x), x);   // row_shr:8
    x = op(dpp_move<0x142, 0xA>(id, x), x);   // row_bcast:15 -> rows 1 and 3
    x = op(dpp_move<0x143, 0xC>(id, x), x);   // row_bcast:31 -> rows 2 and 3
    return x;
}
template <class Tr>
__device__ __forceinline__ typename Tr::T wave_inclusive(typename Tr::T x) {
    return wave_inclusive_with(StaticOp<Tr>{}, x);
}
// the inclusive value of the lane below (identity in lane 0)
template <class Op>
__device__ __forceinline__ typename Op::T wave_shift_up1_with(const Op &op, const typename Op::T &inc) {
    return dpp_move<0x138, 0xF>(op.identity(), inc);   // wave_shr:1
}
template <class Tr>
__device__ __forceinline__ typename Tr::T wave_shift_up1(const typename Tr::T &inc) {
    return wave_shift_up1_with(StaticOp<Tr>{}, inc);
}

// ---- block-level helpers ----------------------------------------------------
// Inclusive scan of one value per thread across the block; returns the exclusive
// prefix of this thread and the block total.  lds must hold WAVES entries.
template <int WAVES, class Op>
__device__ __forceinline__ typename Op::T block_exclusive_with(const Op &op, typename Op::T v, typename Op::T *lds,
                                                               typename Op::T &block_total) {
    using T = typename Op::T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T inc = wave_inclusive_with(op, v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    T wave_prefix = op.identity();
    T total = op.identity();
#pragma unroll
    for (int w = 0; w < WAVES; w++) {
        T x = lds[w];
        if (w < wave) wave_prefix = op(wave_prefix, x);
        total = op(total, x);
    }
    __syncthreads();
    block_total = total;
    return op(wave_prefix, wave_shift_up1_with(op, inc));
}
template <class Tr, int WAVES = SCAN_WAVES>
__device__ __forceinline__ typename Tr::T block_exclusive(typename Tr::T v, typename Tr::T *lds,
                                                          typename Tr::T &block_total) {
    return block_exclusive_with<WAVES>(StaticOp<Tr>{}, v, lds, block_total);
}
// A tile's exclusive prefix from the aggregates of its predecessors, computed by the tile's OWN workgroup: every
// thread folds a contiguous share of aggs[0 .. b), one ordered block reduction joins the shares.  The aggregates
// are a few tens of KB and stay in L2, so while there are only a few thousand tiles this is cheaper than a
// single-workgroup prefix launch between two passes (5-10 us each, plus the boundary); the total traffic is
// quadratic in the tile count, so long batches keep the launch (OWN_PREFIX_MAX_TILES).
constexpr uint32_t OWN_PREFIX_MAX_TILES = 4096;
template <int BLOCK, class Op>
__device__ __forceinline__ typename Op::T tile_prefix_with(const Op &op, const typename Op::T *aggs, uint32_t b, typename Op::T *lds) {
    using T = typename Op::T;
    const uint32_t per = (b + BLOCK - 1) / BLOCK;
    const uint32_t lo = min(b, (uint32_t)threadIdx.x * per), hi = min(b, lo + per);
    T acc = op.identity();
    constexpr int G = sizeof(T) <= 8 ? 8 : 4;   // loads in flight per thread: the fold must not wait for them one by one
    for (uint32_t i = lo; i < hi; i += G) {
        T v[G];
#pragma unroll
        for (int k = 0; k < G; k++) v[k] = (i + k < hi) ? aggs[i + k] : op.identity();
#pragma unroll
        for (int k = 0; k < G; k++) acc = op(acc, v[k]);
    }
    T total;
    (void)block_exclusive_with<BLOCK / 64>(op, acc, lds, total);
    return total;
}
template <class Tr, int BLOCK = 256>
__device__ __forceinline__ typename Tr::T tile_prefix(const typename Tr::T *aggs, uint32_t b, typename Tr::T *lds) {
    return tile_prefix_with<BLOCK>(StaticOp<Tr>{}, aggs, b, lds);
}

// ---- the prefix launch of long batches ------------------------------------------
// partials[i] <- op(seed, exclusive prefix of partials)[i]; the total (with the seed) goes to *total_out and
// to the epilogue, which thread 0 runs once (carried-state bookkeeping that would otherwise be a launch).
// One workgroup, eight consecutive partials per thread, so that every load of a round is in flight at once;
// the workgroup is sized for a single round (256 / 512 / 1024 threads cover 2048 / 4096 / 8192 tiles).
constexpr int PART_ITEMS = 8;
struct NoEpilogue {
    template <class T>
    __device__ __forceinline__ void operator()(const T &) const {}
};
template <class Op, int BLOCK, class Epi>
__global__ __launch_bounds__(BLOCK) void k_scan_partials(size_t nparts, const uint32_t *n_dev, uint32_t tile,
                                                        typename Op::T *partials, typename Op::T seed,
                                                        typename Op::T *total_out, Epi epi, Op op) {
    using T = typename Op::T;
    constexpr int WAVES = BLOCK / 64;
    if (n_dev) nparts = min(nparts, ((size_t)*n_dev + tile - 1) / tile);
    __shared__ T lds[WAVES + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T carry = seed;
    for (size_t base = 0; base < nparts; base += (size_t)BLOCK * PART_ITEMS) {
        const size_t i0 = base + (size_t)threadIdx.x * PART_ITEMS;
        T v[PART_ITEMS];
#pragma unroll
        for (int k = 0; k < PART_ITEMS; k++) v[k] = (i0 + k < nparts) ? partials[i0 + k] : op.identity();
        T inc = v[0];
#pragma unroll
        for (int k = 1; k < PART_ITEMS; k++) inc = op(inc, v[k]);
        inc = wave_inclusive_with(op, inc);
        const T excl = wave_shift_up1_with(op, inc);
        if (lane == 63) lds[wave] = inc;
        __syncthreads();
        if (wave == 0) {   // scan of the wave totals (lanes past them hold the identity)
            const T x = wave_inclusive_with(op, (lane < WAVES) ? lds[lane] : op.identity());
            const T ex = wave_shift_up1_with(op, x);
            if (lane < WAVES) lds[lane] = ex;
            if (lane == WAVES - 1) lds[WAVES] = x;
        }
        __syncthreads();
        T run = op(carry, op(lds[wave], excl));
        carry = op(carry, lds[WAVES]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PART_ITEMS; k++) {
            if (i0 + k < nparts) partials[i0 + k] = run;
            run = op(run, v[k]);
        }
    }
    if (threadIdx.x == 0) {
        if (total_out) *total_out = carry;
        epi(carry);
    }
}
template <class Op, class Epi = NoEpilogue>
inline void scan_partials_with(hipStream_t st, Op op, size_t tiles, const uint32_t *n_dev, uint32_t tile, typename Op::T *partials,
                               typename Op::T seed, typename Op::T *total_out, Epi epi = Epi()) {
    if (tiles <= 256 * PART_ITEMS)
        NFC_LAUNCH((k_scan_partials<Op, 256, Epi>), dim3(1), dim3(256), 0, st, tiles, n_dev, tile, partials, seed, total_out, epi, op);
    else if (tiles <= 512 * PART_ITEMS)
        NFC_LAUNCH((k_scan_partials<Op, 512, Epi>), dim3(1), dim3(512), 0, st, tiles, n_dev, tile, partials, seed, total_out, epi, op);
    else
        NFC_LAUNCH((k_scan_partials<Op, 1024, Epi>), dim3(1), dim3(1024), 0, st, tiles, n_dev, tile, partials, seed, total_out, epi, op);
}
template <class Tr, class Epi = NoEpilogue>
inline void scan_partials(hipStream_t st, size_t tiles, const uint32_t *n_dev, uint32_t tile, typename Tr::T *partials,
                          typename Tr::T seed, typename Tr::T *total_out, Epi epi = Epi()) {
    scan_partials_with(st, StaticOp<Tr>{}, tiles, n_dev, tile, partials, seed, total_out, epi);
}

}  // namespace nfc
