// decode.hip.h -- Modified-Miller / Manchester symbol decode (miller.py:153-197,
// manchester.py:30-61) and packet framing (packets.py:67-79) as parallel
// finite-state transducers over the edge list.
//
// Every edge is routed by its type exactly as background.py:30-35 does (t == 1 ->
// Miller, t == 0 -> Manchester, t == -1 dropped).  An edge is a state map looked
// up in a host-built LUT (decoder_tables.h); an ordered scan of map compositions
// gives each edge its incoming decoder state; a second scan of emission counts
// places the 0-2 symbols it produces.  Framing is the same pattern with a
// two-state machine (started / not started) over the symbol stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nfc_amd.h"
#include "scan.hip.h"

namespace nfc {

struct DecTables {
    const uint4 *mil_map;      // [(cur+1) * nd + d]  16 states, one byte each
    const uint2 *man_map;      //                      8 states, one byte each
    const uint16_t *mil_step;  // [((cur+1) * nd + d) * 16 + state] = next state | out byte << 8
    const uint16_t *man_step;  // [((cur+1) * nd + d) * 8 + state]
    int32_t nd;                // max_len + 1
    int32_t reader, tag;
};

struct DecCarry {
    int32_t mil_state, man_state;
    int32_t pkt_started[2];
    uint32_t pending[2];  // bits of the open packet kept from earlier batches, per type
};

constexpr int DEC_ITEMS = 16;                       // edges (or symbols) per thread: one 16-byte load of bytes
constexpr int DEC_TILE = SCAN_BLOCK * DEC_ITEMS;
constexpr int DEC_LDS_ROWS = 512;                   // LUT rows staged in LDS: 4 (max_len + 1) <= 512 (cur = -1 .. 2)
inline size_t dec_num_tiles(size_t n) { return (n + DEC_TILE - 1) / DEC_TILE; }

// Edges arrive as 16-bit codes (edges.hip.h: edge_code): LUT row | route << 14.  A thread's sixteen, two per word.
__device__ __forceinline__ void load_codes(const uint16_t *ecode, size_t base, size_t n, uint32_t (&c)[8]) {
    if (base + DEC_ITEMS <= n) {
        const uint4 a = *(const uint4 *)(ecode + base), b = *(const uint4 *)(ecode + base + 8);
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
        c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const size_t i = base + 2 * k;
            c[k] = (i < n ? (uint32_t)ecode[i] : 0u) | (i + 1 < n ? (uint32_t)ecode[i + 1] << 16 : 0u);   // code 0 is dropped
        }
    }
}
__device__ __forceinline__ void load_bytes16(const uint8_t *p, size_t base, size_t n, uint32_t fill, uint32_t (&w)[4]) {
    if (base + 16 <= n) {
        const uint4 a = *(const uint4 *)(p + base);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t v = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const size_t i = base + 4 * k + q;
                v |= (i < n ? (uint32_t)p[i] : fill) << (8 * q);
            }
            w[k] = v;
        }
    }
}

// ---- symbols and framing live in the EDGE domain ------------------------------------------------------
// A symbol belongs to the edge that emitted it, and a thread's sixteen out-bytes hold its symbols in stream order.  So
// PacketProcessor.append_bit (packets.py:67-79: two states per type, started or not) needs no scan over the symbol
// arrays: the framing map of a span of symbols is the latest of its symbols that is not the identity (a start-bit value
// starts, an error symbol stops, any other bit changes nothing), and it rides in the same aggregate as the symbol counts.
constexpr uint32_t PM_STOP = 0u, PM_START = 1u, PM_ID = 2u;
struct SymAgg {
    uint32_t cnt[2];   // symbols: [0] Manchester / tag, [1] Miller / reader
    uint32_t map[2];   // framing map over them: PM_*
};
struct SymAggOp {
    using T = SymAgg;
    static __host__ __device__ __forceinline__ T identity() { return T{{0u, 0u}, {PM_ID, PM_ID}}; }
    static __device__ __forceinline__ T op(const T &a, const T &b) {
        return T{{a.cnt[0] + b.cnt[0], a.cnt[1] + b.cnt[1]}, {b.map[0] == PM_ID ? a.map[0] : b.map[0], b.map[1] == PM_ID ? a.map[1] : b.map[1]}};
    }
    static __device__ __forceinline__ T shfl_up(const T &v, int d) {
        return T{{(uint32_t)__shfl_up((int)v.cnt[0], d, 64), (uint32_t)__shfl_up((int)v.cnt[1], d, 64)},
                 {(uint32_t)__shfl_up((int)v.map[0], d, 64), (uint32_t)__shfl_up((int)v.map[1], d, 64)}};
    }
};
__device__ __forceinline__ uint32_t pm_apply(uint32_t map, uint32_t started) { return map == PM_ID ? started : map; }
__device__ __forceinline__ uint32_t start_bit_of(int type) { return type == 0 ? 1u : 0u; }   // packets.py:24-28

// f(type, symbol, k) for every symbol in a thread's sixteen out-bytes, in stream order (k: the edge within the thread)
template <class F>
__device__ __forceinline__ void for_each_symbol(const uint32_t (&ow)[4], F f) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t w = (ow[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        const uint32_t q = w & 3u;
        if (q == 0u) continue;
        if (q == 3u) {
            f(0, (w >> 2) & 7u, k);
        } else {
            f(1, (w >> 2) & 7u, k);
            if (q == 2u) f(1, (w >> 5) & 7u, k);
        }
    }
}
__device__ __forceinline__ SymAgg sym_agg_of(const uint32_t (&ow)[4]) {
    SymAgg a = SymAggOp::identity();
    for_each_symbol(ow, [&](int t, uint32_t s, int) {
        a.cnt[t]++;
        a.map[t] = s > 1u ? PM_STOP : (s == start_bit_of(t) ? PM_START : a.map[t]);
    });
    return a;
}
// per symbol: bit 0 = appended to the packet, bit 1 = closes a started packet (packets.py:67-79); updates `started`
__device__ __forceinline__ uint32_t frame_symbol(int t, uint32_t s, uint32_t &started) {
    uint32_t f;
    if (s > 1u) {
        f = started ? 2u : 0u;
        started = 0u;
    } else {
        const bool sb = s == start_bit_of(t);
        f = (!started && sb) ? 0u : 1u;
        started = (started || sb) ? 1u : 0u;
    }
    return f;
}
// per type: appended bits in the low half, closes in the high half
struct PktCnt {
    uint64_t v[2];
};
struct PktCntOp {
    using T = PktCnt;
    static __host__ __device__ __forceinline__ T identity() { return T{{0ull, 0ull}}; }
    static __device__ __forceinline__ T op(const T &a, const T &b) { return T{{a.v[0] + b.v[0], a.v[1] + b.v[1]}}; }
    static __device__ __forceinline__ T shfl_up(const T &v, int d) { return T{{AddU64::shfl_up(v.v[0], d), AddU64::shfl_up(v.v[1], d)}}; }
};

// ---- pass 1: every edge is a pair of state maps (Miller, Manchester); a thread composes its sixteen ----
// Each edge is routed as background.py:30-35 does: route 2 -> Miller, 1 -> Manchester, 0 dropped.
template <bool LDS>
__global__ __launch_bounds__(SCAN_BLOCK) void k_dec_reduce(const uint16_t *ecode, size_t n, const uint32_t *n_dev, DecTables T,
                                                          DecMaps *partials, DecMaps *aggs) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ uint4 s_mil[LDS ? DEC_LDS_ROWS : 1];
    __shared__ uint2 s_man[LDS ? DEC_LDS_ROWS : 1];
    __shared__ DecMaps lds[SCAN_WAVES];
    if (LDS) {
        const int rows = 4 * T.nd;
        for (int i = threadIdx.x; i < rows; i += SCAN_BLOCK) {
            if (T.reader) s_mil[i] = T.mil_map[i];
            if (T.tag) s_man[i] = T.man_map[i];
        }
        __syncthreads();
    }
    const size_t tid = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint32_t c[8];
    load_codes(ecode, tid * DEC_ITEMS, n, c);
    DecMaps agg = ComposeDec::identity();
#pragma unroll
    for (int k = 0; k < DEC_ITEMS; k++) {
        const uint32_t code = (c[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        const uint32_t li = code & 0x3FFFu, route = code >> 14;
        if (route == 2u && T.reader) {
            const uint4 v = LDS ? s_mil[li] : T.mil_map[li];
#pragma unroll
            for (int q = 0; q < 4; q++) agg.mil[q] = lookup16x4(v.x, v.y, v.z, v.w, agg.mil[q]);
        } else if (route == 1u && T.tag) {
            const uint2 v = LDS ? s_man[li] : T.man_map[li];
            agg.man[0] = __builtin_amdgcn_perm(v.y, v.x, agg.man[0]);
            agg.man[1] = __builtin_amdgcn_perm(v.y, v.x, agg.man[1]);
        }
    }
    aggs[tid] = agg;
    DecMaps total;
    (void)block_exclusive<ComposeDec>(agg, lds, total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// ---- pass 2: a thread walks its edges from its incoming states, one LUT look-up per edge ----
// What an edge emits, one byte: bits 0-1 = 0 nothing, 1 / 2 Miller symbols, 3 one Manchester symbol;
// bits 2-4 first symbol, bits 5-7 second symbol.  The tile's symbol counts AND the framing maps over its symbols
// (SymAgg) are the aggregates of the scan that places the symbols and hands every tile its framing state.
template <bool LDS>
__global__ __launch_bounds__(SCAN_BLOCK) void k_dec_apply(const uint16_t *ecode, size_t n, const uint32_t *n_dev, DecTables T,
                                                         const DecMaps *partials, const DecMaps *aggs, uint32_t state0,
                                                         uint8_t *outw, SymAgg *sym_aggs) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ __attribute__((aligned(16))) uint16_t s_mil[LDS ? DEC_LDS_ROWS * 16 : 8];
    __shared__ __attribute__((aligned(16))) uint16_t s_man[LDS ? DEC_LDS_ROWS * 8 : 8];
    __shared__ DecMaps lds[SCAN_WAVES];
    __shared__ SymAgg lds2[SCAN_WAVES];
    if (LDS) {
        const int rows = 4 * T.nd;
        if (T.reader)
            for (int i = threadIdx.x; i < rows * 2; i += SCAN_BLOCK) ((uint4 *)s_mil)[i] = ((const uint4 *)T.mil_step)[i];
        if (T.tag)
            for (int i = threadIdx.x; i < rows; i += SCAN_BLOCK) ((uint4 *)s_man)[i] = ((const uint4 *)T.man_step)[i];
    }   // the block scan below synchronises before the tables are read
    const size_t tid = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const size_t base = tid * DEC_ITEMS;
    uint32_t c[8];
    load_codes(ecode, base, n, c);
    DecMaps total;
    const DecMaps excl = block_exclusive<ComposeDec>(aggs[tid], lds, total);
    uint32_t st = ComposeDec::step(ComposeDec::op(partials[blockIdx.x], excl), state0);
    const uint16_t *mil = LDS ? s_mil : T.mil_step;
    const uint16_t *man = LDS ? s_man : T.man_step;
    uint32_t ow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < DEC_ITEMS; k++) {
        const uint32_t code = (c[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        const uint32_t li = code & 0x3FFFu, route = code >> 14;
        uint32_t w = 0;
        if (route == 2u && T.reader) {
            const uint32_t e = mil[li * 16u + (st & 15u)];
            w = e >> 8;
            st = (st & ~15u) | (e & 15u);
        } else if (route == 1u && T.tag) {
            const uint32_t e = man[li * 8u + ((st >> 4) & 7u)];
            const uint32_t m = e >> 8;
            w = (m & 3u) ? ((m & 0xFCu) | 3u) : 0u;
            st = (st & 15u) | ((e & 15u) << 4);
        }
        ow[k >> 2] |= w << (8 * (k & 3));
    }
    if (base < n) *(uint4 *)(outw + base) = make_uint4(ow[0], ow[1], ow[2], ow[3]);   // outw has 16 bytes of slack
    SymAgg total_sa;
    (void)block_exclusive<SymAggOp>(sym_agg_of(ow), lds2, total_sa);
    if (threadIdx.x == 0) sym_aggs[blockIdx.x] = total_sa;
}

// ---- pass 3: symbols to their arrays; framing flags counted -------------------------------------------------
struct SymOut {
    uint8_t *sym[2];   // [0] Manchester / tag, [1] Miller / reader
    uint32_t cap[2];   // buffer capacities (an overflow is detected by the host from the totals)
};
__global__ __launch_bounds__(SCAN_BLOCK) void k_sym_frame(const uint8_t *outw, size_t n, const uint32_t *n_dev, const SymAgg *tile_pre,
                                                         uint32_t started0, uint32_t started1, SymOut S, PktCnt *pk_sums) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ SymAgg lds[SCAN_WAVES];
    __shared__ PktCnt lds2[SCAN_WAVES];
    const size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * DEC_ITEMS;
    uint32_t ow[4] = {0u, 0u, 0u, 0u};
    if (base < n) {   // k_dec_apply wrote whole 16-byte groups, zero past n
        const uint4 a = *(const uint4 *)(outw + base);
        ow[0] = a.x; ow[1] = a.y; ow[2] = a.z; ow[3] = a.w;
    }
    SymAgg total;
    const SymAgg pre = SymAggOp::op(tile_pre[blockIdx.x], block_exclusive<SymAggOp>(sym_agg_of(ow), lds, total));
    uint32_t off[2] = {pre.cnt[0], pre.cnt[1]};
    uint32_t started[2] = {pm_apply(pre.map[0], started0), pm_apply(pre.map[1], started1)};
    uint32_t nb[2] = {0u, 0u}, nc[2] = {0u, 0u};
    for_each_symbol(ow, [&](int t, uint32_t s, int) {
        if (off[t] + 1 < S.cap[t]) S.sym[t][off[t]] = (uint8_t)s;
        off[t]++;
        const uint32_t f = frame_symbol(t, s, started[t]);
        nb[t] += f & 1u;
        nc[t] += f >> 1;
    });
    PktCnt tot_pk;
    (void)block_exclusive<PktCntOp>(PktCnt{{(uint64_t)nb[0] | ((uint64_t)nc[0] << 32), (uint64_t)nb[1] | ((uint64_t)nc[1] << 32)}}, lds2, tot_pk);
    if (threadIdx.x == 0) pk_sums[blockIdx.x] = tot_pk;
}

// ---- pass 4: packet bits and closes to their places ------------------------------------------------------------
struct PktOut {
    const nfc_edge *edges;
    uint8_t *bits[2];       // appended bits per type, starting with the pending ones of earlier batches
    uint32_t *close_end[2]; // per close: number of bits appended before it (= end offset of the packet)
    uint64_t *close_idx[2]; // per close: sample index of the closing edge
    uint32_t cap_bits[2], cap_close[2];   // more than estimated: dropped here, seen by the host in the totals, the stage repeated
};
__global__ __launch_bounds__(SCAN_BLOCK) void k_pkt_write(const uint8_t *outw, size_t n, const uint32_t *n_dev, const SymAgg *tile_pre,
                                                         const PktCnt *pk_pre, uint32_t started0, uint32_t started1, PktOut P) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ SymAgg lds[SCAN_WAVES];
    __shared__ PktCnt lds2[SCAN_WAVES];
    const size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * DEC_ITEMS;
    uint32_t ow[4] = {0u, 0u, 0u, 0u};
    if (base < n) {
        const uint4 a = *(const uint4 *)(outw + base);
        ow[0] = a.x; ow[1] = a.y; ow[2] = a.z; ow[3] = a.w;
    }
    SymAgg total;
    const SymAgg pre = SymAggOp::op(tile_pre[blockIdx.x], block_exclusive<SymAggOp>(sym_agg_of(ow), lds, total));
    const uint32_t st0[2] = {pm_apply(pre.map[0], started0), pm_apply(pre.map[1], started1)};
    uint32_t started[2] = {st0[0], st0[1]};
    uint32_t nb[2] = {0u, 0u}, nc[2] = {0u, 0u};
    for_each_symbol(ow, [&](int t, uint32_t s, int) {
        const uint32_t f = frame_symbol(t, s, started[t]);
        nb[t] += f & 1u;
        nc[t] += f >> 1;
    });
    PktCnt tot_pk;
    const PktCnt mine{{(uint64_t)nb[0] | ((uint64_t)nc[0] << 32), (uint64_t)nb[1] | ((uint64_t)nc[1] << 32)}};
    const PktCnt run0 = PktCntOp::op(pk_pre[blockIdx.x], block_exclusive<PktCntOp>(mine, lds2, tot_pk));
    if (!(mine.v[0] | mine.v[1])) return;
    uint64_t run[2] = {run0.v[0], run0.v[1]};
    started[0] = st0[0];
    started[1] = st0[1];
    for_each_symbol(ow, [&](int t, uint32_t s, int k) {
        const uint32_t f = frame_symbol(t, s, started[t]);
        if (f & 2u) {
            const uint32_t j = (uint32_t)(run[t] >> 32);
            if (j < P.cap_close[t]) {
                P.close_end[t][j] = (uint32_t)run[t];
                P.close_idx[t][j] = P.edges[base + k].idx;
            }
            run[t] += 1ull << 32;
        } else if (f & 1u) {
            if ((uint32_t)run[t] < P.cap_bits[t]) P.bits[t][(uint32_t)run[t]] = (uint8_t)s;
            run[t] += 1ull;
        }
    });
}

// After framing: keep the open packet's bits for the next batch and publish the carry.  Reads nothing that it (or a
// sibling carry epilogue) writes, so the edge / decode stages of a batch can be repeated as a whole.  One workgroup
// per packet type.
struct PktFinish {
    const uint8_t *bits[2];
    uint8_t *pending_next[2];   // (the other half of the double buffer)
    const uint32_t *close_end[2];
    const PktCnt *totals;       // per type: (appended incl. pending) | closes << 32
    const SymAgg *sym_total;    // framing map over the batch's symbols
    DecCarry *carry;
    int32_t enabled[2], started_in[2];
    uint32_t pending_cap[2];
    uint32_t cap_bits[2], cap_close[2];
};
__global__ __launch_bounds__(256) void k_pkt_finish(PktFinish F) {
    const int t = blockIdx.x;
    if (!F.enabled[t]) return;
    const uint64_t tot = F.totals->v[t];
    const uint32_t nbits = (uint32_t)tot, ncl = (uint32_t)(tot >> 32);
    if (nbits > F.cap_bits[t] || ncl > F.cap_close[t]) return;   // the host repeats the stage with room
    const uint32_t from = ncl ? F.close_end[t][ncl - 1] : 0u;
    const uint32_t keep = nbits - from;
    for (uint32_t i = threadIdx.x; i < keep && i < F.pending_cap[t]; i += blockDim.x) F.pending_next[t][i] = F.bits[t][from + i];
    if (threadIdx.x == 0) {
        F.carry->pending[t] = keep;
        F.carry->pkt_started[t] = (int32_t)pm_apply(F.sym_total->map[t], (uint32_t)F.started_in[t]);
    }
}

// Decoder states after the batch, from the total of the map scan; also publishes the per-type symbol counts the
// host checks against the capacities.  Epilogue of the symbol scan's partials pass.
struct DecCarryEpilogue {
    const DecMaps *total;
    uint32_t state_in;
    DecCarry *carry;
    uint32_t *nsym;
    __device__ __forceinline__ void operator()(const SymAgg &sym_total) const {
        const uint32_t st = ComposeDec::step(*total, state_in);
        carry->mil_state = (int32_t)(st & 15u);
        carry->man_state = (int32_t)(st >> 4);
        nsym[1] = sym_total.cnt[1];   // Miller / reader
        nsym[0] = sym_total.cnt[0];   // Manchester / tag
    }
};

}  // namespace nfc
