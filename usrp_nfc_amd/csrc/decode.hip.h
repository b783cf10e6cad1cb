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
// bits 2-4 first symbol, bits 5-7 second symbol.  The tile's symbol counts (Miller low half, Manchester
// high half) are the aggregates of the scan that places the symbols.
template <bool LDS>
__global__ __launch_bounds__(SCAN_BLOCK) void k_dec_apply(const uint16_t *ecode, size_t n, const uint32_t *n_dev, DecTables T,
                                                         const DecMaps *partials, const DecMaps *aggs, uint32_t state0,
                                                         uint8_t *outw, uint64_t *sym_sums) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ __attribute__((aligned(16))) uint16_t s_mil[LDS ? DEC_LDS_ROWS * 16 : 8];
    __shared__ __attribute__((aligned(16))) uint16_t s_man[LDS ? DEC_LDS_ROWS * 8 : 8];
    __shared__ DecMaps lds[SCAN_WAVES];
    __shared__ uint64_t lds2[SCAN_WAVES];
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
    uint32_t n_mil = 0, n_man = 0;
#pragma unroll
    for (int k = 0; k < DEC_ITEMS; k++) {
        const uint32_t code = (c[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        const uint32_t li = code & 0x3FFFu, route = code >> 14;
        uint32_t w = 0;
        if (route == 2u && T.reader) {
            const uint32_t e = mil[li * 16u + (st & 15u)];
            w = e >> 8;
            st = (st & ~15u) | (e & 15u);
            n_mil += w & 3u;
        } else if (route == 1u && T.tag) {
            const uint32_t e = man[li * 8u + ((st >> 4) & 7u)];
            const uint32_t m = e >> 8;
            w = (m & 3u) ? ((m & 0xFCu) | 3u) : 0u;
            st = (st & 15u) | ((e & 15u) << 4);
            n_man += (m & 3u) ? 1u : 0u;
        }
        ow[k >> 2] |= w << (8 * (k & 3));
    }
    if (base < n) *(uint4 *)(outw + base) = make_uint4(ow[0], ow[1], ow[2], ow[3]);   // outw has 16 bytes of slack
    const uint64_t sum = block_sum<AddU64>((uint64_t)n_mil | ((uint64_t)n_man << 32), lds2);
    if (threadIdx.x == 0) sym_sums[blockIdx.x] = sum;
}

// ---- symbols: placed by a scan of the emission counts ---------------------------------------
struct SymOut {
    uint8_t *sym[2];   // [0] Manchester / tag, [1] Miller / reader
    uint32_t *src[2];  // index of the producing edge -- written for error symbols only (the ones that can close a packet)
    uint32_t cap[2];   // buffer capacities (an overflow is detected by the host from the totals)
};
__global__ __launch_bounds__(SCAN_BLOCK) void k_sym_store(const uint8_t *outw, size_t n, const uint32_t *n_dev,
                                                         const uint64_t *tile_base, SymOut S) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ uint64_t lds[SCAN_WAVES];
    const size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * DEC_ITEMS;
    uint32_t ow[4] = {0u, 0u, 0u, 0u};
    if (base < n) {   // k_dec_apply wrote whole 16-byte groups, zero past n
        const uint4 a = *(const uint4 *)(outw + base);
        ow[0] = a.x; ow[1] = a.y; ow[2] = a.z; ow[3] = a.w;
    }
    uint64_t mine = 0;
#pragma unroll
    for (int k = 0; k < DEC_ITEMS; k++) {
        const uint32_t q = (ow[k >> 2] >> (8 * (k & 3))) & 3u;
        mine += (q == 3u) ? (1ull << 32) : (uint64_t)q;
    }
    uint64_t total;
    uint64_t run = tile_base[blockIdx.x] + block_exclusive<AddU64>(mine, lds, total);
    if (!mine) return;
#pragma unroll
    for (int k = 0; k < DEC_ITEMS; k++) {
        const uint32_t w = (ow[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        const uint32_t q = w & 3u;
        if (q == 0u) continue;
        const int type = (q == 3u) ? 0 : 1;
        const uint32_t off = type == 1 ? (uint32_t)run : (uint32_t)(run >> 32);
        if (off + 1 < S.cap[type]) {
            const uint32_t s0 = (w >> 2) & 7u;
            S.sym[type][off] = (uint8_t)s0;
            if (s0 > 1u) S.src[type][off] = (uint32_t)(base + k);   // only a symbol that can close a packet needs its edge
            if (q == 2u) {
                const uint32_t s1 = (w >> 5) & 7u;
                S.sym[type][off + 1] = (uint8_t)s1;
                if (s1 > 1u) S.src[type][off + 1] = (uint32_t)(base + k);
            }
        }
        run += (q == 3u) ? (1ull << 32) : (uint64_t)q;
    }
}

// ---- framing: PacketProcessor.append_bit (packets.py:67-79) ---------------------
// state 0 = not started, 1 = started; nibble map
__device__ __forceinline__ uint32_t pkt_map(uint32_t s, int start_bit) {
    if (s > 1u) return 0x00u;                      // error symbol: started -> not started, not started stays
    if ((int)s == start_bit) return 0x11u;         // start bit: not started -> started (dropped); started stays
    return 0x10u;                                  // other bit: identity
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_pkt_reduce(const uint8_t *sym, size_t n, const uint32_t *n_dev, int start_bit,
                                                          uint32_t *partials, uint32_t *aggs) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ uint32_t lds[SCAN_WAVES];
    const size_t tid = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint32_t w[4];
    load_bytes16(sym, tid * DEC_ITEMS, n, (uint32_t)(1 - start_bit), w);   // padding: a non-start bit is the identity
    // the three maps (all -> 0, all -> 1, identity) compose to the latest one that is not the identity
    uint32_t agg = ComposePkt::identity();
#pragma unroll
    for (int k = 0; k < DEC_ITEMS; k++) {
        const uint32_t m = pkt_map((w[k >> 2] >> (8 * (k & 3))) & 0xFFu, start_bit);
        agg = (m == 0x10u) ? agg : m;
    }
    aggs[tid] = agg;
    uint32_t total;
    (void)block_exclusive<ComposePkt>(agg, lds, total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}
// per symbol: bit 0 = appended to the packet, bit 1 = closes a started packet; the tile sums of
// (appended, closes << 32) are the aggregates of the placing scan
__global__ __launch_bounds__(SCAN_BLOCK) void k_pkt_apply(const uint8_t *sym, size_t n, const uint32_t *n_dev, int start_bit,
                                                         const uint32_t *partials, const uint32_t *aggs, uint32_t state0,
                                                         uint8_t *pflags, uint64_t *sums) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ uint32_t lds[SCAN_WAVES];
    __shared__ uint64_t lds2[SCAN_WAVES];
    const size_t tid = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const size_t base = tid * DEC_ITEMS;
    uint32_t w[4];
    load_bytes16(sym, base, n, 0xFFu, w);
    uint32_t total;
    const uint32_t excl = block_exclusive<ComposePkt>(aggs[tid], lds, total);
    uint32_t started = ComposePkt::step(ComposePkt::op(partials[blockIdx.x], excl), state0);
    uint32_t fw[4] = {0u, 0u, 0u, 0u};
    uint32_t n_bits = 0, n_close = 0;
#pragma unroll
    for (int k = 0; k < DEC_ITEMS; k++) {
        const uint32_t s = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        if (base + k < n) {
            uint32_t f;
            if (s > 1u) {
                f = started ? 2u : 0u;
                started = 0u;
            } else {
                f = (!started && (int)s == start_bit) ? 0u : 1u;
                started = (started || (int)s == start_bit) ? 1u : 0u;
            }
            n_bits += f & 1u;
            n_close += f >> 1;
            fw[k >> 2] |= f << (8 * (k & 3));
        }
    }
    if (base < n) *(uint4 *)(pflags + base) = make_uint4(fw[0], fw[1], fw[2], fw[3]);
    const uint64_t sum = block_sum<AddU64>((uint64_t)n_bits | ((uint64_t)n_close << 32), lds2);
    if (threadIdx.x == 0) sums[blockIdx.x] = sum;
}
struct PktOut {
    const uint32_t *src;
    const nfc_edge *edges;
    uint8_t *bits;       // appended bits, starting with the pending ones of earlier batches
    uint32_t *close_end; // per close: number of bits appended before it (= end offset of the packet)
    uint64_t *close_idx; // per close: sample index of the closing edge
};
__global__ __launch_bounds__(SCAN_BLOCK) void k_pkt_store(const uint8_t *sym, const uint8_t *pflags, size_t n, const uint32_t *n_dev,
                                                         const uint64_t *tile_base, PktOut P) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ uint64_t lds[SCAN_WAVES];
    const size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * DEC_ITEMS;
    uint32_t w[4], fw[4] = {0u, 0u, 0u, 0u};
    load_bytes16(sym, base, n, 0u, w);
    if (base < n) {
        const uint4 a = *(const uint4 *)(pflags + base);
        fw[0] = a.x; fw[1] = a.y; fw[2] = a.z; fw[3] = a.w;
    }
    uint64_t mine = 0;
#pragma unroll
    for (int k = 0; k < DEC_ITEMS; k++) {
        const uint32_t f = (fw[k >> 2] >> (8 * (k & 3))) & 3u;
        mine += (uint64_t)(f & 1u) | ((uint64_t)(f >> 1) << 32);
    }
    uint64_t total;
    uint64_t run = tile_base[blockIdx.x] + block_exclusive<AddU64>(mine, lds, total);
    if (!mine) return;
#pragma unroll
    for (int k = 0; k < DEC_ITEMS; k++) {
        const uint32_t f = (fw[k >> 2] >> (8 * (k & 3))) & 3u;
        if (f & 2u) {
            const uint32_t j = (uint32_t)(run >> 32);
            P.close_end[j] = (uint32_t)run;
            P.close_idx[j] = P.edges[P.src[base + k]].idx;
            run += 1ull << 32;
        } else if (f & 1u) {
            P.bits[(uint32_t)run] = (uint8_t)((w[k >> 2] >> (8 * (k & 3))) & 0xFFu);
            run += 1ull;
        }
    }
}

// After framing: keep the open packet's bits for the next batch and publish the carry.  Reads nothing that it
// (or a sibling carry kernel) writes, so the edge / decode stages of a batch can be repeated as a whole.
struct PktFinish {
    const uint8_t *bits;
    uint8_t *pending_next;   // [cap]  (the other half of the double buffer)
    const uint32_t *close_end;
    const uint64_t *totals;  // (appended incl. pending) | closes << 32
    const uint32_t *map_total;
    DecCarry *carry;
    int type;
    int32_t started_in;
    uint32_t pending_cap;
};
__global__ __launch_bounds__(256) void k_pkt_finish(PktFinish F) {
    const uint64_t tot = *F.totals;
    const uint32_t nbits = (uint32_t)tot, ncl = (uint32_t)(tot >> 32);
    const uint32_t from = ncl ? F.close_end[ncl - 1] : 0u;
    const uint32_t keep = nbits - from;
    for (uint32_t i = threadIdx.x; i < keep && i < F.pending_cap; i += blockDim.x) F.pending_next[i] = F.bits[from + i];
    if (threadIdx.x == 0) {
        F.carry->pending[F.type] = keep;
        F.carry->pkt_started[F.type] = (int32_t)((*F.map_total >> (4 * F.started_in)) & 1u);
    }
}

// Decoder states after the batch, from the total of the map scan; also splits the packed symbol totals into the
// two per-type counts the framing scans read from the device.  Epilogue of the symbol-count partials pass.
struct DecCarryEpilogue {
    const DecMaps *total;
    uint32_t state_in;
    DecCarry *carry;
    uint32_t *nsym;
    __device__ __forceinline__ void operator()(uint64_t sym_total) const {
        const uint32_t st = ComposeDec::step(*total, state_in);
        carry->mil_state = (int32_t)(st & 15u);
        carry->man_state = (int32_t)(st >> 4);
        nsym[1] = (uint32_t)sym_total;           // Miller / reader
        nsym[0] = (uint32_t)(sym_total >> 32);   // Manchester / tag
    }
};

}  // namespace nfc
