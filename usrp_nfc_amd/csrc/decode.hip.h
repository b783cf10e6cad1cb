// decode.hip.h -- Modified-Miller / Manchester symbol decode (miller.py:153-197,
// manchester.py:30-61) and packet framing (packets.py:67-79) as parallel
// finite-state transducers over the edge list.
//
// Every edge is routed by its type exactly as background.py:30-35 does (t == 1 ->
// Miller, t == 0 -> Manchester, t == -1 dropped).  An edge is a state map looked
// up in a host-built LUT (decoder_tables.h); an ordered scan of map compositions
// gives each edge its incoming decoder state.  What an edge emits (0-2 symbols)
// stays with the edge; ONE more scan, of an aggregate that carries symbol counts,
// the framing map and bit / close counts under both entry states, places the
// symbols, the packet bits and the packet ends of both packet types.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nfc_amd.h"
#include "scan.hip.h"

namespace nfc {

struct DecTables {
    const uint4 *mil_map;      // [(cur+1) * nd + d]  16 states, one byte each
    const uint2 *man_map;      //                      8 states, one byte each (+ one row more at 4 nd: the identity)
    const uint16_t *mil_step;  // [((cur+1) * nd + d) * 16 + state] = next state | out byte << 8
    const uint16_t *man_step;  // [((cur+1) * nd + d) * 8 + state]
    int32_t nd;                // max_len + 1
    int32_t reader, tag;
    // The Miller decoder as its QUOTIENT machine (decoder_tables.h: miller_quotient): states that no sequence of edges can tell
    // apart -- the 16 states fall into 9 classes, the ones reachable from the initial state into 6 -- are one class, so a state map
    // is 8 bytes and composing two of them is two v_perm_b32 (a 16-byte map takes 28 instructions).  The speculative decode works
    // on classes; the three-launch form keeps the 16 states.  Both publish the CANONICAL state of the class they end in.
    const uint2 *qmil_map;     // [(cur+1) * nd + d] 8 classes, one byte each (unused class ids follow class 0; + the identity row at 4 nd)
    const uint16_t *qmil_step; // [((cur+1) * nd + d) * 8 + class] = next class | out byte << 8
    uint32_t q_rep[2];         // class -> its canonical state
    uint32_t canon[4];         // state -> the canonical state of its class (itself where the quotient machine does not know it)
    int32_t q_ok;              // the reachable classes fit 8
};
__device__ __forceinline__ uint32_t byte_of(const uint32_t *w, uint32_t i) { return (w[i >> 2] >> (8u * (i & 3u))) & 0xFFu; }

struct DecCarry {
    int32_t mil_state, man_state;
    int32_t pkt_started[2];
    uint32_t pending[2];  // bits of the open packet kept from earlier batches, per type
};

constexpr int DEC_ITEMS = 16;                       // edges per group: one 16-byte load of out-bytes, 32 symbol slots in a word
#ifndef NFC_DEC_GROUPS
#define NFC_DEC_GROUPS 2
#endif
constexpr int DEC_GROUPS = NFC_DEC_GROUPS;          // groups per thread in the multi-launch stage: a tile's block scans (the
                                                    // larger part of k_dec_reduce's instructions) are paid per 32 edges, not 16
constexpr int DEC_PER_THREAD = DEC_ITEMS * DEC_GROUPS;
constexpr int DEC_TILE = SCAN_BLOCK * DEC_PER_THREAD;
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
// arrays, and no scan of its own: ONE aggregate per span of edges carries, per packet type,
//   cnt  symbols emitted (places the symbols),
//   map  the framing map over them: the latest symbol that is not the identity decides (a start-bit value starts, an
//        error symbol stops, any other bit changes nothing),
//   nb / nc  bits appended / packets closed if the span is entered in state "started", and
//   db / dc  whether entering it "not started" gives one fewer: the two hypotheses merge at the span's first symbol
//        that is not the identity -- a start bit (dropped instead of appended) or an error symbol (nothing to close).
constexpr uint32_t PM_STOP = 0u, PM_START = 1u, PM_ID = 2u;   // map, FrameAgg.fl bits 0-1
constexpr uint32_t FA_DB = 4u, FA_DC = 8u;                    // FrameAgg.fl bits 2, 3
struct FrameAgg {
    uint32_t cnt[2];   // [0] Manchester / tag, [1] Miller / reader
    uint32_t nb[2], nc[2];
    uint32_t fl[2];
};
struct FrameAggOp {
    using T = FrameAgg;
    static __host__ __device__ __forceinline__ T identity() { return T{{0u, 0u}, {0u, 0u}, {0u, 0u}, {PM_ID, PM_ID}}; }
    static __device__ __forceinline__ T op(const T &a, const T &b) {
        T r;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t ma = a.fl[t] & 3u, mb = b.fl[t] & 3u;
            const bool into_stopped = ma == PM_STOP;   // b is entered "not started" whatever a was entered in
            r.cnt[t] = a.cnt[t] + b.cnt[t];
            r.nb[t] = a.nb[t] + b.nb[t] - (into_stopped ? (b.fl[t] >> 2) & 1u : 0u);
            r.nc[t] = a.nc[t] + b.nc[t] - (into_stopped ? (b.fl[t] >> 3) & 1u : 0u);
            r.fl[t] = (mb == PM_ID ? ma : mb) | ((ma == PM_ID ? b.fl[t] : a.fl[t]) & (FA_DB | FA_DC));
        }
        return r;
    }
    static __device__ __forceinline__ T shfl_up(const T &v, int d) {
        T r;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            r.cnt[t] = (uint32_t)__shfl_up((int)v.cnt[t], d, 64);
            r.nb[t] = (uint32_t)__shfl_up((int)v.nb[t], d, 64);
            r.nc[t] = (uint32_t)__shfl_up((int)v.nc[t], d, 64);
            r.fl[t] = (uint32_t)__shfl_up((int)v.fl[t], d, 64);
        }
        return r;
    }
};
// Inside a tile (<= 2 * DEC_TILE = 16384 symbols of a type) the counts fit 16 bits: the block scans run on this packed
// form, half the words to shuffle.  a = cnt | nb << 16, b = nc | fl << 16.
struct FramePk {
    uint32_t a[2], b[2];
};
struct FramePkOp {
    using T = FramePk;
    static __device__ __forceinline__ T pack(const FrameAgg &f) {
        return T{{f.cnt[0] | (f.nb[0] << 16), f.cnt[1] | (f.nb[1] << 16)}, {f.nc[0] | (f.fl[0] << 16), f.nc[1] | (f.fl[1] << 16)}};
    }
    static __device__ __forceinline__ FrameAgg unpack(const T &p) {
        return FrameAgg{{p.a[0] & 0xFFFFu, p.a[1] & 0xFFFFu}, {p.a[0] >> 16, p.a[1] >> 16}, {p.b[0] & 0xFFFFu, p.b[1] & 0xFFFFu}, {p.b[0] >> 16, p.b[1] >> 16}};
    }
    static __device__ __forceinline__ T identity() { return T{{0u, 0u}, {PM_ID << 16, PM_ID << 16}}; }
    static __device__ __forceinline__ T op(const T &x, const T &y) {
        T r;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint32_t ma = (x.b[t] >> 16) & 3u, mb = (y.b[t] >> 16) & 3u;
            const uint32_t stopped = ma == PM_STOP ? 1u : 0u;
            r.a[t] = x.a[t] + y.a[t] - ((stopped & (y.b[t] >> 18)) << 16);
            const uint32_t nc = (x.b[t] & 0xFFFFu) + (y.b[t] & 0xFFFFu) - (stopped & (y.b[t] >> 19));
            const uint32_t fl = (mb == PM_ID ? ma : mb) | (((ma == PM_ID ? y.b[t] : x.b[t]) >> 16) & (FA_DB | FA_DC));
            r.b[t] = nc | (fl << 16);
        }
        return r;
    }
};
__device__ __forceinline__ uint32_t pm_apply(uint32_t fl, uint32_t started) { return (fl & 3u) == PM_ID ? started : (fl & 3u); }
// bits appended / packets closed over a span entered in state `started`
__device__ __forceinline__ uint32_t fa_bits(const FrameAgg &a, int t, uint32_t started) { return a.nb[t] - (started ? 0u : (a.fl[t] >> 2) & 1u); }
__device__ __forceinline__ uint32_t fa_closes(const FrameAgg &a, int t, uint32_t started) { return a.nc[t] - (started ? 0u : (a.fl[t] >> 3) & 1u); }
__device__ __forceinline__ uint32_t start_bit_of(int type) { return type == 0 ? 1u : 0u; }   // packets.py:24-28

// PacketProcessor.append_bit per symbol (packets.py:67-79): an error symbol (> 1) closes a started packet and leaves the
// state "not started"; a bit is appended unless it is the start-bit value arriving while not started (it starts the
// packet and is dropped); the start-bit value leaves the state "started".
// A thread never walks its symbols to apply this: a thread's 16 out-bytes are 32 symbol slots (slot 2k + j: symbol j of
// edge k); per packet type three 32-bit masks say which slots hold a symbol, an error symbol, a start-bit value, and
// the started / not-started latch over them is the carry chain of ONE addition (a start generates, an error kills,
// everything else propagates).
__device__ __forceinline__ uint32_t transpose_pairs(uint32_t m0, uint32_t m1, uint32_t m2, uint32_t m3) {
    // m_i: bits 0-1 of byte j = the slot pair of edge 4 i + j.  Returns the pairs in stream order (pair 4 i + j at bits 8 i + 2 j).
    uint32_t x = m0 | (m1 << 2) | (m2 << 4) | (m3 << 6);   // byte j: [m0.j, m1.j, m2.j, m3.j] -- a 4 x 4 matrix of pairs, transposed:
    uint32_t t = (x ^ (x >> 6)) & 0x00CC00CCu;
    x ^= t ^ (t << 6);
    t = (x ^ (x >> 12)) & 0x0000F0F0u;
    x ^= t ^ (t << 12);
    return x;
}
struct SlotMasks {
    uint32_t V[2], ST[2], SA[2];   // per type: slots that hold a symbol / an error symbol / a start-bit value
};
__device__ __forceinline__ SlotMasks slot_masks(const uint32_t (&ow)[4]) {
    constexpr uint32_t M = 0x01010101u;
    uint32_t V1[4], ST1[4], SA1[4], V0[4], ST0[4], SA0[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t w = ow[i];
        const uint32_t b0 = w & M, b1 = (w >> 1) & M;
        const uint32_t v0 = b0 & b1;            // q == 3: one Manchester symbol
        const uint32_t v1e = b0 ^ b1;           // q == 1 or 2: a first Miller symbol
        const uint32_t v1o = b1 & ~b0;          // q == 2: a second one
        const uint32_t c2 = (w >> 2) & M, hi0 = ((w >> 3) | (w >> 4)) & M;   // first symbol: bit 0; > 1
        const uint32_t c5 = (w >> 5) & M, hi1 = ((w >> 6) | (w >> 7)) & M;   // second symbol
        const uint32_t z0 = M ^ (c2 | hi0), z1 = M ^ (c5 | hi1);             // == 0 (Miller's start-bit value, packets.py:24-28)
        const uint32_t o0 = c2 & ~hi0;                                       // == 1 (Manchester's)
        V1[i] = v1e | (v1o << 1);
        ST1[i] = (hi0 & v1e) | ((hi1 & v1o) << 1);
        SA1[i] = (z0 & v1e) | ((z1 & v1o) << 1);
        V0[i] = v0;
        ST0[i] = hi0 & v0;
        SA0[i] = o0 & v0;
    }
    return SlotMasks{{transpose_pairs(V0[0], V0[1], V0[2], V0[3]), transpose_pairs(V1[0], V1[1], V1[2], V1[3])},
                     {transpose_pairs(ST0[0], ST0[1], ST0[2], ST0[3]), transpose_pairs(ST1[0], ST1[1], ST1[2], ST1[3])},
                     {transpose_pairs(SA0[0], SA0[1], SA0[2], SA0[3]), transpose_pairs(SA1[0], SA1[1], SA1[2], SA1[3])}};
}
// started / not started before every slot, entered in state `started`: the carry into the slot of (SA | keep) + SA + started,
// keep = slots that are neither a start nor an error (a start generates a carry, an error kills it, the rest propagate)
__device__ __forceinline__ uint32_t started_before(const SlotMasks &m, int t, uint32_t started) {
    const uint32_t x = m.SA[t] | ~(m.ST[t] | m.SA[t]), y = m.SA[t];
    return (x + y + started) ^ x ^ y;
}
__device__ __forceinline__ FrameAgg frame_agg_of(const uint32_t (&ow)[4]) {
    const SlotMasks m = slot_masks(ow);
    FrameAgg a;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const uint32_t nonid = m.ST[t] | m.SA[t];
        const uint32_t before = started_before(m, t, 1u);
        const uint32_t appended = m.V[t] & ~m.ST[t] & (before | ~m.SA[t]);
        const uint32_t closes = m.ST[t] & before;
        const uint32_t first = nonid & (0u - nonid);
        a.cnt[t] = (uint32_t)__popc(m.V[t]);
        a.nb[t] = (uint32_t)__popc(appended);
        a.nc[t] = (uint32_t)__popc(closes);
        a.fl[t] = nonid == 0u ? PM_ID
                              : ((m.ST[t] > m.SA[t] ? PM_STOP : PM_START) | ((first & m.SA[t]) ? FA_DB : 0u) | ((first & m.ST[t]) ? FA_DC : 0u));
    }
    return a;
}
// ---- packet bits leave the multi-launch stage PACKED, 32 to a word (round 4) ---------------------------------------------
// A thread's appended bits are its symbols' low bits at the slots of `appended`, in slot order: compress (Hacker's Delight 7-4,
// the parallel-suffix form: five rounds of shifts and xors, no loop over the symbols and nothing that diverges) squeezes them to
// the bottom of a word, and the words of a tile are or-ed together in LDS and stored whole.  Before: a loop per symbol with a
// byte store each -- two thirds of k_frame_write's time.  nfc_read_packet_bits unpacks (one byte per bit, as ever).
__device__ __forceinline__ uint32_t compress32(uint32_t x, uint32_t m) {
    x &= m;
    uint32_t mk = ~m << 1;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        uint32_t mp = mk ^ (mk << 1);
        mp ^= mp << 2;
        mp ^= mp << 4;
        mp ^= mp << 8;
        mp ^= mp << 16;
        const uint32_t mv = mp & m;
        m = (m ^ mv) | (mv >> (1 << i));
        const uint32_t t = x & mv;
        x = (x ^ t) | (t >> (1 << i));
        mk &= ~mp;
    }
    return x;
}
// bit 0 of the symbol in every slot (slot 2k + j: symbol j of edge k), in slot order
__device__ __forceinline__ uint32_t symbol_low_bits(const uint32_t (&ow)[4]) {
    constexpr uint32_t M = 0x01010101u;
    uint32_t b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) b[i] = ((ow[i] >> 2) & M) | (((ow[i] >> 5) & M) << 1);
    return transpose_pairs(b[0], b[1], b[2], b[3]);
}

// per type: appended bits in the low half, closes in the high half
struct PktCnt {
    uint64_t v[2];
};

// The packed bit arrays are or-ed into (k_frame_write): the stage's first launch clears their words, all its workgroups together.
struct ZeroJob {
    uint32_t *p[2];
    uint32_t n[2];   // words
};
__device__ __forceinline__ void zero_words(const ZeroJob &Z) {
#pragma unroll
    for (int t = 0; t < 2; t++)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < Z.n[t]; i += gridDim.x * blockDim.x) Z.p[t][i] = 0u;
}

// ---- pass 1: every edge is a pair of state maps (Miller, Manchester); a thread composes its sixteen ----
// Each edge is routed as background.py:30-35 does: route 2 -> Miller, 1 -> Manchester, 0 dropped.
template <bool LDS>
__global__ __launch_bounds__(SCAN_BLOCK) void k_dec_reduce(const uint16_t *ecode, size_t n, const uint32_t *n_dev, DecTables T,
                                                          DecMaps *partials, DecMaps *aggs, ZeroJob Z) {
    zero_words(Z);
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    TP_DECL();
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
    uint32_t c[DEC_GROUPS][8];
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) load_codes(ecode, tid * DEC_PER_THREAD + DEC_ITEMS * g, n, c[g]);
    TP_MARK();   // 1: tables to LDS, codes asked for
    DecMaps agg = ComposeDec::identity();
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) {
#pragma unroll
        for (int k = 0; k < DEC_ITEMS; k++) {
            const uint32_t code = (c[g][k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
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
    }
    TP_MARK();   // 2: the compositions
    aggs[tid] = agg;
    DecMaps total;
    (void)block_exclusive<ComposeDec>(agg, lds, total);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
    TP_DONE(1);   // 3: block scan
}

// ---- pass 2: a thread walks its edges from its incoming states, one LUT look-up per edge ----
// What an edge emits, one byte: bits 0-1 = 0 nothing, 1 / 2 Miller symbols, 3 one Manchester symbol;
// bits 2-4 first symbol, bits 5-7 second symbol.  The tile's symbol counts AND the framing maps over its symbols
// (FrameAgg) are the aggregates of the ONE scan that places symbols, packet bits and packet ends.
template <bool LDS>
__global__ __launch_bounds__(SCAN_BLOCK) void k_dec_apply(const uint16_t *ecode, size_t n, const uint32_t *n_dev, DecTables T,
                                                         const DecMaps *partials, const DecMaps *aggs, uint32_t state0,
                                                         uint8_t *outw, FrameAgg *frame_aggs, FramePk *thread_aggs, bool own_prefix, DecMaps *total_out) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if (own_prefix && n == 0 && blockIdx.x == 0 && threadIdx.x == 0) *total_out = ComposeDec::identity();
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    TP_DECL();
    __shared__ __attribute__((aligned(16))) uint16_t s_mil[LDS ? DEC_LDS_ROWS * 16 : 8];
    __shared__ __attribute__((aligned(16))) uint16_t s_man[LDS ? DEC_LDS_ROWS * 8 : 8];
    __shared__ DecMaps lds[SCAN_WAVES];
    __shared__ FramePk lds2[SCAN_WAVES];
    // own_prefix: `partials` still holds the tiles' maps (scan.hip.h: tile_prefix; first, while few registers are live)
    const DecMaps pre = own_prefix ? tile_prefix<ComposeDec, SCAN_BLOCK>(partials, blockIdx.x, lds) : partials[blockIdx.x];
    TP_MARK();   // 1: tile prefix
    if (LDS) {
        const int rows = 4 * T.nd;
        if (T.reader)
            for (int i = threadIdx.x; i < rows * 2; i += SCAN_BLOCK) ((uint4 *)s_mil)[i] = ((const uint4 *)T.mil_step)[i];
        if (T.tag)
            for (int i = threadIdx.x; i < rows; i += SCAN_BLOCK) ((uint4 *)s_man)[i] = ((const uint4 *)T.man_step)[i];
    }   // the block scan below synchronises before the tables are read
    const size_t tid = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const size_t base = tid * DEC_PER_THREAD;
    uint32_t c[DEC_GROUPS][8];
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) load_codes(ecode, base + DEC_ITEMS * g, n, c[g]);
    DecMaps total;
    const DecMaps excl = block_exclusive<ComposeDec>(aggs[tid], lds, total);
    // (own_prefix: the last tile publishes the total)
    if (own_prefix && threadIdx.x == 0 && ((size_t)blockIdx.x + 1) * DEC_TILE >= n) *total_out = ComposeDec::op(pre, total);
    uint32_t st = ComposeDec::step(ComposeDec::op(pre, excl), state0);
    TP_MARK();   // 2: tables, codes, block scan
    const uint16_t *mil = LDS ? s_mil : T.mil_step;
    const uint16_t *man = LDS ? s_man : T.man_step;
    FramePk mine = FramePkOp::identity();
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) {
        uint32_t ow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < DEC_ITEMS; k++) {
            const uint32_t code = (c[g][k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
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
        const size_t gb = base + (size_t)DEC_ITEMS * g;
        if (gb < n) *(uint4 *)(outw + gb) = make_uint4(ow[0], ow[1], ow[2], ow[3]);   // outw has 16 bytes of slack
        mine = FramePkOp::op(mine, FramePkOp::pack(frame_agg_of(ow)));
    }
    TP_MARK();   // 3: the walk
    FramePk total_fa;
    *(uint4 *)(thread_aggs + tid) = make_uint4(mine.a[0], mine.a[1], mine.b[0], mine.b[1]);   // (k_frame_write scans them again)
    (void)block_exclusive<FramePkOp>(mine, lds2, total_fa);
    if (threadIdx.x == 0) frame_aggs[blockIdx.x] = FramePkOp::unpack(total_fa);
    TP_DONE(2);   // 4: block scan
}

// ---- passes 1 + 2 in ONE launch: every tile derives its incoming decoder states by itself (round 4) ----------------------------
// Both decoders re-synchronise at a frame gap: an edge whose duration is out of range resets them (miller.py:165-176,
// manchester.py:40-43), so the composed map of a stretch of edges that spans a gap is CONSTANT -- whatever state the stretch is
// entered in, it leaves in the same one.  A tile therefore composes a RUN-IN of `runin` edges right before its own (its
// predecessor's last ones) and takes the run-in's map at the carried state of the batch: where that map is constant, the tile's
// incoming state is known without any scan over the tiles before it, and k_dec_reduce, the tile-prefix fold over its maps and the
// re-load of codes and maps between two launches all go away.  Where it is not (a frame longer than the run-in across the tile's
// first edge; a decoder that has seen no edge for longer than that) the assumption may be wrong, so nothing is taken on trust:
// the tile leaves its own composed map, the state it assumed, and -- per decoder -- whether any of its outputs depends on that
// assumption (a thread whose map from the run-in's start to its first edge is constant does not); dec_verify (one extra workgroup
// of k_frame_write) scans the tiles' maps, compares every assumption that matters with the true state, publishes the decoder
// states after the batch from the true scan, and leaves a verdict in the state block.  The host repeats the decode stage with
// the three-launch form (k_dec_reduce / k_dec_apply: exact whatever the edges look like) when a tile was wrong.
// The Miller decoder runs as its quotient machine here (DecTables: classes, not states): maps of 8 bytes for both decoders.
// What a tile of the speculative decode leaves for k_concat (round 4: the bits and packet ends of a tile are worked out where its
// symbols are, in k_dec_spec -- no second pass over the out-bytes, no second block scan): per packet type the tile's appended bits
// and packet ends AS IF THE TILE WERE ENTERED IN STATE "started" (tile-local offsets), and which of its bits is not appended when
// it is entered "not started" -- the two cases differ in the tile's first start-bit value or in its first packet end, FrameAgg's
// FA_DB / FA_DC.  k_concat knows the state every tile is really entered in (the scan of the tiles' FrameAgg) and moves the bits to
// their place in the stream.
constexpr int FW_WORDS = DEC_TILE * 2 / 32 + 4;   // bit words of a tile: two symbols per edge at most (+ the tile's phase in the stream, k_frame_write)
constexpr uint32_t ST_CLOSES = 1024;              // packet ends staged per tile and type (125 on the bench captures); a tile with more fails the check
struct TileStage {
    uint32_t *bits[2];        // [tile][FW_WORDS], NULL: packet type not decoded
    uint32_t *close_bit[2];   // [tile][ST_CLOSES]: bits of the tile appended before the packet end
    uint64_t *close_idx[2];   // [tile][ST_CLOSES]: sample index of the closing edge
    uint32_t *drop_bit[2];    // [tile]: the tile-local bit that is dropped when the tile is entered "not started" (FA_DB set)
    FrameAgg *own;            // [tile]: the tile's own aggregate (the scan may overwrite the other copy with prefixes)
    const uint32_t *epos;     // per edge: batch-local sample position ...
    uint64_t g0;
    const uint64_t *idx64;    // ... or the caller's own indices (nfc_push_edges)
};
// nbits bits from bit src_bit of src to bit dst_bit of dst (zeroed words, shared with other copies: or-ed in), by the whole workgroup
__device__ __forceinline__ void bitcopy_or(uint32_t *dst, uint32_t dst_bit, const uint32_t *src, uint32_t src_bit, uint32_t nbits, uint32_t cap_bits) {
    if (!nbits) return;
    const uint32_t w0 = dst_bit >> 5, nw = ((dst_bit & 31u) + nbits + 31u) >> 5, capw = (cap_bits + 31u) >> 5;
    for (uint32_t j = threadIdx.x; j < nw; j += blockDim.x) {
        if (w0 + j >= capw) break;   // (an estimate too small: the host sees it in the totals and repeats the stage with room)
        const uint32_t lo = max(dst_bit, (w0 + j) << 5), hi = min(dst_bit + nbits, (w0 + j + 1u) << 5);   // this word's bits of the range
        const uint32_t sb = src_bit + (lo - dst_bit), k = sb & 31u;
        const uint32_t a = src[sb >> 5], b = k ? src[(sb >> 5) + 1u] : 0u;
        uint32_t v = (a >> k) | (k ? b << (32u - k) : 0u);
        const uint32_t cnt = hi - lo;
        if (cnt < 32u) v &= (1u << cnt) - 1u;
        v <<= (lo & 31u);
        if (v) atomicOr(dst + w0 + j, v);
    }
}

struct QMaps {
    uint32_t mil[2];   // Miller: class -> class, one byte each
    uint32_t man[2];   // Manchester: state -> state
};
struct ComposeQ {
    using T = QMaps;
    static __host__ __device__ __forceinline__ T identity() { return T{{0x03020100u, 0x07060504u}, {0x03020100u, 0x07060504u}}; }
    static __device__ __forceinline__ T op(const T &a, const T &b) {   // a, then b
        return T{{__builtin_amdgcn_perm(b.mil[1], b.mil[0], a.mil[0]), __builtin_amdgcn_perm(b.mil[1], b.mil[0], a.mil[1])},
                 {__builtin_amdgcn_perm(b.man[1], b.man[0], a.man[0]), __builtin_amdgcn_perm(b.man[1], b.man[0], a.man[1])}};
    }
    // states packed as miller class | manchester state << 4
    static __device__ __forceinline__ uint32_t step(const T &m, uint32_t st) {
        const uint32_t a = __builtin_amdgcn_perm(m.mil[1], m.mil[0], st & 7u) & 15u;
        const uint32_t b = __builtin_amdgcn_perm(m.man[1], m.man[0], (st >> 4) & 7u) & 15u;
        return a | (b << 4);
    }
};
struct DecSpec {
    QMaps map;        // composed map of the tile's own edges
    uint32_t s_in;    // the incoming state the tile assumed (miller class | manchester state << 4)
    uint32_t needs;   // bit 0 / 1: Miller / Manchester outputs of the tile depend on that assumption
    uint32_t pad[2];
};
static_assert(sizeof(DecSpec) == 32, "two 16-byte loads");
__device__ __forceinline__ bool map8_constant(const uint32_t (&m)[2]) { return m[0] == m[1] && m[0] == (m[0] & 0xFFu) * 0x01010101u; }
constexpr int DEC_RUNIN_MAX = 8;   // run-in edges per thread at most (runin = 2, 4 or 8 x SCAN_BLOCK: 512, 1024 or 2048 edges)
// (k_dec_spec and k_frame_write squeeze a thread's appended bits -- 32 slots per group -- into one 64-bit accumulator, and the run-in
// must lie inside the tile before: both break silently with more groups per thread)
static_assert(DEC_GROUPS >= 1 && DEC_GROUPS <= 2, "NFC_DEC_GROUPS: a thread's packed bits are accumulated in 64 bits (32 slots per group)");
static_assert(DEC_RUNIN_MAX * SCAN_BLOCK <= DEC_TILE, "the longest run-in fits inside the tile before");
template <bool LDS>
__global__ __launch_bounds__(SCAN_BLOCK) void k_dec_spec(const uint16_t *ecode, size_t n, const uint32_t *n_dev, DecTables T, uint32_t state0, int runin_per_thread,
                                                        uint8_t *outw, FrameAgg *frame_aggs, DecSpec *spec, ZeroJob Z, TileStage S) {
    zero_words(Z);
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    TP_DECL();
    __shared__ uint2 s_milmap[LDS ? DEC_LDS_ROWS + 1 : 1];
    __shared__ uint2 s_manmap[LDS ? DEC_LDS_ROWS + 1 : 1];
    __shared__ __attribute__((aligned(16))) uint16_t s_mil[LDS ? DEC_LDS_ROWS * 8 : 8];
    __shared__ __attribute__((aligned(16))) uint16_t s_man[LDS ? DEC_LDS_ROWS * 8 : 8];
    __shared__ QMaps lds[SCAN_WAVES];
    __shared__ FramePk lds2[SCAN_WAVES];
    __shared__ uint32_t s_needs;
    __shared__ uint32_t s_bits[2][FW_WORDS];   // the tile's appended bits, entered "started"
    __shared__ uint32_t s_drop[2];             // the bit that is dropped when it is entered "not started"
    for (int i = threadIdx.x; i < 2 * FW_WORDS; i += SCAN_BLOCK) (&s_bits[0][0])[i] = 0u;
    if (threadIdx.x < 2) s_drop[threadIdx.x] = 0xFFFFFFFFu;
    const uint32_t ident = 4u * (uint32_t)T.nd;   // the identity row of both map tables
    if (LDS) {
        const int rows = 4 * T.nd;
        for (int i = threadIdx.x; i <= rows; i += SCAN_BLOCK) {
            if (T.reader) s_milmap[i] = T.qmil_map[i];
            if (T.tag) s_manmap[i] = T.man_map[i];
        }
        for (int i = threadIdx.x; i < rows; i += SCAN_BLOCK) {
            if (T.reader) ((uint4 *)s_mil)[i] = ((const uint4 *)T.qmil_step)[i];
            if (T.tag) ((uint4 *)s_man)[i] = ((const uint4 *)T.man_step)[i];
        }
    }
    if (threadIdx.x == 0) s_needs = 0u;
    const size_t tid = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const size_t base = tid * DEC_PER_THREAD;
    uint32_t c[DEC_GROUPS][8];
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) load_codes(ecode, base + DEC_ITEMS * g, n, c[g]);
    // the run-in: this thread's share of the `runin` edges before the tile (tile 0 has none: its incoming state is the carried one)
    uint32_t rc[DEC_RUNIN_MAX / 2];
    const bool have_runin = blockIdx.x > 0;
    {
        const size_t r0 = (size_t)blockIdx.x * DEC_TILE - (size_t)runin_per_thread * SCAN_BLOCK + (size_t)threadIdx.x * runin_per_thread;
#pragma unroll
        for (int k = 0; k < DEC_RUNIN_MAX / 2; k++) rc[k] = (have_runin && 2 * k < runin_per_thread) ? *(const uint32_t *)(ecode + r0 + 2 * k) : 0u;   // (code 0 is dropped)
    }
    __syncthreads();   // the tables are staged
    TP_MARK();   // 1: tables, codes
    // An edge that is not routed to a decoder takes that table's identity row: a select on the index, no branch.  Four edges are
    // composed as a tree (two pairs, then the pair of pairs) before they meet the accumulated map: a chain of 8 dependent
    // look-ups per 32 edges instead of 32.
    auto mil_row = [&](uint32_t code) __attribute__((always_inline)) -> uint2 {
        const uint32_t idx = (code >> 14) == 2u ? (code & 0x3FFFu) : ident;
        return LDS ? s_milmap[idx] : T.qmil_map[idx];
    };
    auto man_row = [&](uint32_t code) __attribute__((always_inline)) -> uint2 {
        const uint32_t idx = (code >> 14) == 1u ? (code & 0x3FFFu) : ident;
        return LDS ? s_manmap[idx] : T.man_map[idx];
    };
    auto then = [](const uint2 &a, const uint2 &b) __attribute__((always_inline)) -> uint2 {   // a, then b
        return make_uint2(__builtin_amdgcn_perm(b.y, b.x, a.x), __builtin_amdgcn_perm(b.y, b.x, a.y));
    };
    auto routes_of = [](uint32_t pair) __attribute__((always_inline)) -> uint32_t {   // bit 0 / 1: a Miller / Manchester edge among the two codes of a word
        const uint32_t r0 = (pair >> 14) & 3u, r1 = pair >> 30;
        return ((r0 == 2u || r1 == 2u) ? 1u : 0u) | ((r0 == 1u || r1 == 1u) ? 2u : 0u);
    };
    auto compose4 = [&](uint2 &am, uint2 &an, uint32_t w0, uint32_t w1) __attribute__((always_inline)) {   // four edges (two words of codes)
        const uint32_t c0 = w0 & 0xFFFFu, c1 = w0 >> 16, c2 = w1 & 0xFFFFu, c3 = w1 >> 16;
        if (T.reader) am = then(am, then(then(mil_row(c0), mil_row(c1)), then(mil_row(c2), mil_row(c3))));
        if (T.tag) an = then(an, then(then(man_row(c0), man_row(c1)), then(man_row(c2), man_row(c3))));
    };
    const QMaps idm = ComposeQ::identity();
    QMaps runin = idm;
    if (have_runin) {   // (uniform)
        uint2 am = make_uint2(idm.mil[0], idm.mil[1]), an = make_uint2(idm.man[0], idm.man[1]);
#pragma unroll
        for (int k = 0; k < DEC_RUNIN_MAX / 2; k += 2)
            if (2 * k < runin_per_thread) compose4(am, an, rc[k], rc[k + 1]);   // (words past the run-in hold code 0: not routed)
        const QMaps ra{{am.x, am.y}, {an.x, an.y}};
        (void)block_exclusive<ComposeQ>(ra, lds, runin);
    }
    uint32_t seen = 0u;
    uint2 am = make_uint2(idm.mil[0], idm.mil[1]), an = make_uint2(idm.man[0], idm.man[1]);
    // (the maps of the thread's first 8, 16, 24 ... edges are kept: the walk below runs a chain per eight edges)
#ifndef NFC_DEC_CHAIN
#define NFC_DEC_CHAIN 8
#endif
    // edges per chain of the walk.  Measured (round 5, -DNFC_DEC_CHAIN): eight chains of 4 -- half the dependent look-ups, twice the
    // start states to derive -- 24.5 -> 30.2 us for the launch, two chains of 16 the same 24.6: the kernel issues instructions, it does not wait
    constexpr int CHAIN = NFC_DEC_CHAIN;
    static_assert(CHAIN == 16 || CHAIN == 8 || CHAIN == 4, "NFC_DEC_CHAIN");
    constexpr int NCH = DEC_PER_THREAD / CHAIN;        // chains per thread
    uint2 pm[NCH], pn[NCH];
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) {
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            if (k % (CHAIN / 2) == 0) {   // (k counts words of two codes)
                pm[(16 * g + 2 * k) / CHAIN] = am;
                pn[(16 * g + 2 * k) / CHAIN] = an;
            }
            compose4(am, an, c[g][k], c[g][k + 1]);
            seen |= routes_of(c[g][k]) | routes_of(c[g][k + 1]);
        }
    }
    if (!T.reader) seen &= ~1u;
    if (!T.tag) seen &= ~2u;
    const QMaps agg{{am.x, am.y}, {an.x, an.y}};
    TP_MARK();   // 2: the compositions
    QMaps total;
    const QMaps excl = block_exclusive<ComposeQ>(agg, lds, total);
    // the map from the run-in's first edge to this thread's first: constant for a decoder = its state here is known whatever came before
    const QMaps upto = ComposeQ::op(runin, excl);
    const uint32_t dep = ((seen & 1u) && !map8_constant(upto.mil) ? 1u : 0u) | ((seen & 2u) && !map8_constant(upto.man) ? 2u : 0u);
    if (blockIdx.x > 0 && dep) atomicOr(&s_needs, dep);   // (rare; tile 0 starts from the carried state itself)
    // The walk: one table look-up per edge from the state before it -- a chain of dependent LDS reads.  A thread's 32 edges are
    // FOUR chains of eight, each started from the state the maps composed above give it, and walked side by side.
    uint32_t st[NCH];
#pragma unroll
    for (int j = 0; j < NCH; j++) st[j] = ComposeQ::step(ComposeQ::op(upto, QMaps{{pm[j].x, pm[j].y}, {pn[j].x, pn[j].y}}), state0);
    TP_MARK();   // 3: block scans, incoming states
    const uint16_t *mil = LDS ? s_mil : T.qmil_step;
    const uint16_t *man = LDS ? s_man : T.man_step;
    uint32_t ow[DEC_GROUPS][4];
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) ow[g][0] = ow[g][1] = ow[g][2] = ow[g][3] = 0u;
#pragma unroll
    for (int e = 0; e < CHAIN; e++) {
#pragma unroll
        for (int j = 0; j < NCH; j++) {
            const int g = (j * CHAIN) / DEC_ITEMS, k = (j * CHAIN) % DEC_ITEMS + e;   // chain j: the thread's edges j CHAIN .. j CHAIN + CHAIN - 1
            const uint32_t code = (c[g][k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
            const uint32_t li = code & 0x3FFFu, route = code >> 14;
            uint32_t w = 0;
            if (route == 2u && T.reader) {
                const uint32_t en = mil[li * 8u + (st[j] & 7u)];
                w = en >> 8;
                st[j] = (st[j] & ~15u) | (en & 15u);
            } else if (route == 1u && T.tag) {
                const uint32_t en = man[li * 8u + ((st[j] >> 4) & 7u)];
                const uint32_t m = en >> 8;
                w = (m & 3u) ? ((m & 0xFCu) | 3u) : 0u;
                st[j] = (st[j] & 15u) | ((en & 15u) << 4);
            }
            ow[g][k >> 2] |= w << (8 * (k & 3));
        }
    }
    FramePk mine = FramePkOp::identity();
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) {
        const size_t gb = base + (size_t)DEC_ITEMS * g;
        if (gb < n) *(uint4 *)(outw + gb) = make_uint4(ow[g][0], ow[g][1], ow[g][2], ow[g][3]);   // outw has 16 bytes of slack
        mine = FramePkOp::op(mine, FramePkOp::pack(frame_agg_of(ow[g])));
    }
    TP_MARK();   // 4: the walk
    FramePk total_fa;
    const FramePk in_tile = block_exclusive<FramePkOp>(mine, lds2, total_fa);   // (its barriers: every thread's atomicOr has landed)
    // The tile's packet bits and packet ends, entered "started": a thread's groups in order, each from everything before it in the
    // tile -- the appended bits squeezed together (compress32), up to 64 per thread and type, or-ed into the tile's words in LDS.
    {
        FrameAgg before = FramePkOp::unpack(in_tile);
        uint64_t acc[2] = {0ull, 0ull};
        uint32_t cnt[2] = {0u, 0u};
        const uint32_t bo0[2] = {before.nb[0], before.nb[1]};   // (fa_bits entered "started")
#pragma unroll
        for (int g = 0; g < DEC_GROUPS; g++) {
            if (ow[g][0] | ow[g][1] | ow[g][2] | ow[g][3]) {
                const SlotMasks m = slot_masks(ow[g]);
                const uint32_t low = symbol_low_bits(ow[g]);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    if (!m.V[t] || !S.bits[t]) continue;
                    const uint32_t started = pm_apply(before.fl[t], 1u);
                    const uint32_t bef = started_before(m, t, started);
                    const uint32_t appended = m.V[t] & ~m.ST[t] & (bef | ~m.SA[t]);
                    uint32_t cl = m.ST[t] & bef;
                    const uint32_t bo = bo0[t] + cnt[t];
                    const uint32_t nonid = m.ST[t] | m.SA[t];
                    if ((before.fl[t] & 3u) == PM_ID && nonid) {   // the tile's first symbol that is not the identity lies here
                        const uint32_t first = nonid & (0u - nonid);
                        if (first & m.SA[t]) s_drop[t] = bo + (uint32_t)__popc(appended & (first - 1u));   // (appended now, dropped if "not started")
                    }
                    acc[t] |= (uint64_t)compress32(low, appended) << cnt[t];
                    cnt[t] += (uint32_t)__popc(appended);
                    if (cl) {   // packet ends: one per frame
                        const uint32_t co = before.nc[t], closes = cl;
                        while (cl) {
                            const uint32_t lowb = cl & (0u - cl);
                            const int k = (__ffs((int)cl) - 1) >> 1;
                            cl ^= lowb;
                            const uint32_t j = co + (uint32_t)__popc(closes & (lowb - 1u));
                            if (j < ST_CLOSES) {
                                const size_t e = base + (size_t)DEC_ITEMS * g + k;
                                S.close_bit[t][(size_t)blockIdx.x * ST_CLOSES + j] = bo + (uint32_t)__popc(appended & (lowb - 1u));
                                S.close_idx[t][(size_t)blockIdx.x * ST_CLOSES + j] = S.idx64 ? S.idx64[e] : S.g0 + (uint64_t)S.epos[e];
                            } else {
                                atomicOr(&s_needs, 4u);   // (more packet ends than a tile stages: the three-launch form takes the batch)
                            }
                        }
                    }
                }
                if (g + 1 < DEC_GROUPS) before = FrameAggOp::op(before, frame_agg_of(ow[g]));
            }
        }
#pragma unroll
        for (int t = 0; t < 2; t++) {
            if (!cnt[t]) continue;
            const uint32_t w = bo0[t] >> 5, sh = bo0[t] & 31u;
            const uint64_t lo = acc[t] << sh;
            const uint32_t hi = sh ? (uint32_t)(acc[t] >> (64u - sh)) : 0u;
            if (w + 2 < (uint32_t)FW_WORDS) {
                if ((uint32_t)lo) atomicOr(&s_bits[t][w], (uint32_t)lo);
                if ((uint32_t)(lo >> 32)) atomicOr(&s_bits[t][w + 1], (uint32_t)(lo >> 32));
                if (hi) atomicOr(&s_bits[t][w + 2], hi);
            }
        }
    }
    __syncthreads();
    const FrameAgg tile_fa = FramePkOp::unpack(total_fa);
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (!S.bits[t]) continue;
        const uint32_t nw = min((uint32_t)FW_WORDS, (tile_fa.nb[t] + 31u) / 32u + 1u);   // (one word more: k_concat reads pairs)
        for (uint32_t j = threadIdx.x; j < nw; j += SCAN_BLOCK) S.bits[t][(size_t)blockIdx.x * FW_WORDS + j] = s_bits[t][j];
    }
    if (threadIdx.x == 0) {
        frame_aggs[blockIdx.x] = tile_fa;
        S.own[blockIdx.x] = tile_fa;
        if (S.bits[0]) S.drop_bit[0][blockIdx.x] = s_drop[0];
        if (S.bits[1]) S.drop_bit[1][blockIdx.x] = s_drop[1];
        const uint32_t s_in = ComposeQ::step(runin, state0);
        ((uint4 *)(spec + blockIdx.x))[0] = make_uint4(total.mil[0], total.mil[1], total.man[0], total.man[1]);
        ((uint4 *)(spec + blockIdx.x))[1] = make_uint4(s_in, s_needs, 0u, 0u);
    }
    TP_DONE(1);   // 5: block scan
}

// The check of k_dec_spec's assumptions: ONE workgroup scans the tiles' maps from the carried state, compares the state every tile
// assumed with the true one wherever the tile said its outputs depend on it, and publishes the decoder states after the batch.
struct DecVerify {
    const DecSpec *spec;   // NULL: the three-launch form ran (nothing to check)
    uint32_t state0;       // miller CLASS | manchester state << 4
    uint32_t q_rep[2];     // Miller class -> canonical state
    DecCarry *carry;
    uint32_t *verdict;     // 0: every assumption that mattered was right
};
constexpr int DV_ITEMS = 4;
__device__ __forceinline__ void dec_verify(const DecVerify &V, size_t ntiles, QMaps *lds) {
    QMaps before = ComposeQ::identity();   // the map of all tiles before this round's
    int bad = 0;
    for (size_t b0 = 0; b0 < ntiles; b0 += (size_t)SCAN_BLOCK * DV_ITEMS) {
        const size_t i0 = b0 + (size_t)threadIdx.x * DV_ITEMS;
        QMaps mp[DV_ITEMS];
        uint32_t s_in[DV_ITEMS], needs[DV_ITEMS];
#pragma unroll
        for (int k = 0; k < DV_ITEMS; k++) {
            if (i0 + k < ntiles) {
                const uint4 a = ((const uint4 *)(V.spec + i0 + k))[0], b = ((const uint4 *)(V.spec + i0 + k))[1];
                mp[k] = QMaps{{a.x, a.y}, {a.z, a.w}};
                s_in[k] = b.x;
                needs[k] = b.y;
            } else {
                mp[k] = ComposeQ::identity();
                s_in[k] = 0u;
                needs[k] = 0u;
            }
        }
        QMaps agg = mp[0];
#pragma unroll
        for (int k = 1; k < DV_ITEMS; k++) agg = ComposeQ::op(agg, mp[k]);
        QMaps total;
        QMaps run = ComposeQ::op(before, block_exclusive<ComposeQ>(agg, lds, total));
#pragma unroll
        for (int k = 0; k < DV_ITEMS; k++) {
            const uint32_t diff = ComposeQ::step(run, V.state0) ^ s_in[k];
            if (((needs[k] & 1u) && (diff & 15u)) || ((needs[k] & 2u) && (diff >> 4)) || (needs[k] & 4u)) bad = 1;   // (4: more packet ends than a tile stages)
            run = ComposeQ::op(run, mp[k]);
        }
        before = ComposeQ::op(before, total);
    }
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
        const uint32_t st = ComposeQ::step(before, V.state0);
        V.carry->mil_state = (int32_t)byte_of(V.q_rep, st & 7u);
        V.carry->man_state = (int32_t)(st >> 4);
        *V.verdict = bad ? 1u : 0u;
    }
}

// ---- pass 3: symbols, packet bits and packet ends to their places ------------------------------------------------
struct FrameOut {
    uint8_t *sym[2];        // [0] Manchester / tag, [1] Miller / reader
    uint32_t cap_sym[2];    // buffer capacities (an overflow is seen by the host in the totals, the stage repeated)
    const uint32_t *epos;   // per edge: batch-local sample position (edges.hip.h); its index is g0 + position ...
    uint64_t g0;
    const uint64_t *idx64;  // ... unless the caller brought the edges with indices of its own (nfc_push_edges)
    uint8_t *bits[2];       // appended bits per type, starting with the pending ones of earlier batches
    uint32_t *close_end[2]; // per close: number of bits appended before it (= end offset of the packet)
    uint64_t *close_idx[2]; // per close: sample index of the closing edge
    uint32_t cap_bits[2], cap_close[2];
    const uint8_t *pending[2];   // the open packets' bits of earlier batches (workgroup 0 puts them in front)
    uint32_t pend[2], started_in[2];
    // (the multi-launch stage: bits[t] holds WORDS, bit i of the stream at word i / 32, bit i % 32; cap_bits counts bits as ever.
    // The one-launch stage of short batches keeps a byte per bit.  The symbol arrays are only written when somebody reads
    // them: k_symbols_write.)
};
__device__ __forceinline__ void copy_pending(const FrameOut &P, int tid, int nthreads) {
#pragma unroll
    for (int t = 0; t < 2; t++)
        for (uint32_t i = tid; i < P.pend[t] && i < P.cap_bits[t]; i += nthreads) P.bits[t][i] = P.pending[t][i];
}
// a thread's symbols, given the aggregate of everything before them: one round per symbol it holds
__device__ __forceinline__ void frame_write(const FrameOut &P, const FrameAgg &pre, const uint32_t (&ow)[4], size_t base) {
    const SlotMasks m = slot_masks(ow);
    const uint64_t lo = (uint64_t)ow[0] | ((uint64_t)ow[1] << 32), hi = (uint64_t)ow[2] | ((uint64_t)ow[3] << 32);
#pragma unroll
    for (int t = 0; t < 2; t++) {
        uint32_t v = m.V[t];
        if (!v) continue;
        const uint32_t started = pm_apply(pre.fl[t], P.started_in[t]);
        const uint32_t before = started_before(m, t, started);
        const uint32_t appended = v & ~m.ST[t] & (before | ~m.SA[t]);
        const uint32_t closes = m.ST[t] & before;
        uint32_t off = pre.cnt[t];
        const uint32_t bo = P.pend[t] + fa_bits(pre, t, P.started_in[t]), co = fa_closes(pre, t, P.started_in[t]);
        if (__all(off + 33u < P.cap_sym[t] && bo + 32u < P.cap_bits[t])) {
            // every lane's symbols and bits fit their buffers (the usual case): the same walk without the per-symbol capacity
            // checks, the appended bits' places counted along instead of recounted, the (rare) packet ends in a loop of their own
            uint8_t *const sp = P.sym[t], *const bp = P.bits[t];
            uint32_t bi = bo;
            while (v) {
                const uint32_t low = v & (0u - v);
                const int slot = __ffs((int)v) - 1, k = slot >> 1;
                v ^= low;
                const uint32_t byte = (uint32_t)((k < 8 ? lo : hi) >> (8 * (k & 7))) & 0xFFu;
                const uint32_t s = (byte >> ((slot & 1) ? 5 : 2)) & 7u;
                sp[off++] = (uint8_t)s;
                if (appended & low) bp[bi++] = (uint8_t)s;
            }
            uint32_t cl = closes;
            while (cl) {
                const uint32_t low = cl & (0u - cl);
                const int k = (__ffs((int)cl) - 1) >> 1;
                cl ^= low;
                const uint32_t j = co + (uint32_t)__popc(closes & (low - 1u));
                if (j < P.cap_close[t]) {
                    P.close_end[t][j] = bo + (uint32_t)__popc(appended & (low - 1u));
                    P.close_idx[t][j] = P.idx64 ? P.idx64[base + k] : P.g0 + (uint64_t)P.epos[base + k];
                }
            }
            continue;
        }
        while (v) {
            const uint32_t low = v & (0u - v);
            const int slot = __ffs((int)v) - 1, k = slot >> 1;
            v ^= low;
            const uint32_t byte = (uint32_t)((k < 8 ? lo : hi) >> (8 * (k & 7))) & 0xFFu;
            const uint32_t s = (byte >> ((slot & 1) ? 5 : 2)) & 7u;
            if (off + 1 < P.cap_sym[t]) P.sym[t][off] = (uint8_t)s;
            off++;
            if (appended & low) {
                const uint32_t i = bo + (uint32_t)__popc(appended & (low - 1u));
                if (i < P.cap_bits[t]) P.bits[t][i] = (uint8_t)s;
            } else if (closes & low) {
                const uint32_t j = co + (uint32_t)__popc(closes & (low - 1u));
                if (j < P.cap_close[t]) {
                    P.close_end[t][j] = bo + (uint32_t)__popc(appended & (low - 1u));
                    P.close_idx[t][j] = P.idx64 ? P.idx64[base + k] : P.g0 + (uint64_t)P.epos[base + k];
                }
            }
        }
    }
}
// Decoder states after the batch, from the total of the map scan; also publishes the per-type symbol counts the
// host checks against the capacities and the per-type bit / close totals.  Runs once after the framing scan: in the last
// tile's workgroup of k_frame_write, or as the epilogue of the scan's prefix launch (long batches).
struct DecCarryEpilogue {
    const DecMaps *total;
    uint32_t state_in;
    DecCarry *carry;
    uint32_t *nsym;
    PktCnt *pk_total;
    uint32_t pend[2], started_in[2];
    uint32_t canon[4];
    __device__ __forceinline__ void operator()(const FrameAgg &ft) const {
        if (total) {   // (NULL: the speculative decode ran -- dec_verify publishes the decoder states from its own scan)
            const uint32_t st = ComposeDec::step(*total, state_in);
            carry->mil_state = (int32_t)byte_of(canon, st & 15u);   // (the canonical state of its class: DecTables)
            carry->man_state = (int32_t)(st >> 4);
        }
        nsym[1] = ft.cnt[1];   // Miller / reader
        nsym[0] = ft.cnt[0];   // Manchester / tag
#pragma unroll
        for (int t = 0; t < 2; t++)
            pk_total->v[t] = (uint64_t)(pend[t] + fa_bits(ft, t, started_in[t])) | ((uint64_t)fa_closes(ft, t, started_in[t]) << 32);
    }
};

// (V.spec set: the grid has ONE workgroup more, in front -- the check of the speculative decode, dec_verify, beside the tiles)
__device__ __forceinline__ void pack_pending(const FrameOut &P, int tid, int nthreads) {   // the open packets' bits of earlier batches go in front
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (!P.bits[t]) continue;
        uint32_t *gb = (uint32_t *)P.bits[t];
        for (uint32_t i = tid; i < P.pend[t] && i < P.cap_bits[t]; i += nthreads)
            if (P.pending[t][i] & 1u) atomicOr(gb + (i >> 5), 1u << (i & 31));
    }
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_frame_write(const uint8_t *outw, size_t n, const uint32_t *n_dev, const FrameAgg *tile_pre,
                                                           const FramePk *thread_aggs, FrameOut P, bool own_prefix, FrameAgg *total_out,
                                                           DecCarryEpilogue epi, DecVerify V) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if (V.spec) {
        if (blockIdx.x == 0) {
            __shared__ QMaps lds_v[SCAN_WAVES];
            dec_verify(V, (n + DEC_TILE - 1) / DEC_TILE, lds_v);
            return;
        }
    }
    const uint32_t bid = blockIdx.x - (V.spec ? 1u : 0u);   // the tile
    if (bid == 0) pack_pending(P, threadIdx.x, SCAN_BLOCK);
    if (own_prefix && n == 0 && bid == 0 && threadIdx.x == 0) {
        *total_out = FrameAggOp::identity();
        epi(FrameAggOp::identity());
    }
    if ((size_t)bid * DEC_TILE >= n) return;
    TP_DECL();
    __shared__ FramePk lds[SCAN_WAVES];
    __shared__ FrameAgg lds_pre[SCAN_WAVES];
    __shared__ uint32_t s_bits[2][FW_WORDS];
    for (int i = threadIdx.x; i < 2 * FW_WORDS; i += SCAN_BLOCK) (&s_bits[0][0])[i] = 0u;   // (the scans' barriers come before the first or)
    // own_prefix: tile_pre still holds the tiles' aggregates (scan.hip.h: tile_prefix; first, while few registers are live)
    const FrameAgg pre = own_prefix ? tile_prefix<FrameAggOp, SCAN_BLOCK>(tile_pre, bid, lds_pre) : tile_pre[bid];
    const size_t base = ((size_t)bid * SCAN_BLOCK + threadIdx.x) * DEC_PER_THREAD;
    TP_MARK();   // 1: tile prefix
    uint32_t ow[DEC_GROUPS][4];
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) {
        ow[g][0] = ow[g][1] = ow[g][2] = ow[g][3] = 0u;
        const size_t gb = base + (size_t)DEC_ITEMS * g;
        if (gb < n) {   // the decode pass wrote whole 16-byte groups, zero past n
            const uint4 a = *(const uint4 *)(outw + gb);
            ow[g][0] = a.x; ow[g][1] = a.y; ow[g][2] = a.z; ow[g][3] = a.w;
        }
    }
    const uint4 m4 = *(const uint4 *)(thread_aggs + (size_t)bid * SCAN_BLOCK + threadIdx.x);
    FramePk total;
    const FramePk in_tile = block_exclusive<FramePkOp>(FramePk{{m4.x, m4.y}, {m4.z, m4.w}}, lds, total);
    const FrameAgg all = FrameAggOp::op(pre, FramePkOp::unpack(total));
    // (own_prefix: the last tile publishes the total and the carries)
    if (own_prefix && threadIdx.x == 0 && ((size_t)bid + 1) * DEC_TILE >= n) {
        *total_out = all;
        epi(all);
    }
    TP_MARK();   // 2: loads + block scan
    // a thread's groups in order, each from everything before it: the appended bits squeezed together (compress32), up to 64 of
    // them per thread and packet type, then or-ed into the tile's words in LDS at the phase the tile has in the stream
    FrameAgg before = FrameAggOp::op(pre, FramePkOp::unpack(in_tile));
    uint64_t acc[2] = {0ull, 0ull};
    uint32_t cnt[2] = {0u, 0u}, bo0[2], tile_bit0[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        bo0[t] = P.pend[t] + fa_bits(before, t, P.started_in[t]);        // the thread's first bit in the stream
        tile_bit0[t] = P.pend[t] + fa_bits(pre, t, P.started_in[t]);     // the tile's
    }
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) {
        if (ow[g][0] | ow[g][1] | ow[g][2] | ow[g][3]) {
            const SlotMasks m = slot_masks(ow[g]);
            const uint32_t low = symbol_low_bits(ow[g]);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (!m.V[t] || !P.bits[t]) continue;
                const uint32_t started = pm_apply(before.fl[t], P.started_in[t]);
                const uint32_t bef = started_before(m, t, started);
                const uint32_t appended = m.V[t] & ~m.ST[t] & (bef | ~m.SA[t]);
                uint32_t cl = m.ST[t] & bef;
                const uint32_t bo = bo0[t] + cnt[t];
                acc[t] |= (uint64_t)compress32(low, appended) << cnt[t];
                cnt[t] += (uint32_t)__popc(appended);
                if (cl) {   // packet ends: rare (one per frame)
                    const uint32_t co = fa_closes(before, t, P.started_in[t]);
                    const uint32_t closes = cl;
                    while (cl) {
                        const uint32_t lowb = cl & (0u - cl);
                        const int k = (__ffs((int)cl) - 1) >> 1;
                        cl ^= lowb;
                        const uint32_t j = co + (uint32_t)__popc(closes & (lowb - 1u));
                        if (j < P.cap_close[t]) {
                            P.close_end[t][j] = bo + (uint32_t)__popc(appended & (lowb - 1u));
                            const size_t e = base + (size_t)DEC_ITEMS * g + k;
                            P.close_idx[t][j] = P.idx64 ? P.idx64[e] : P.g0 + (uint64_t)P.epos[e];
                        }
                    }
                }
            }
            if (g + 1 < DEC_GROUPS) before = FrameAggOp::op(before, frame_agg_of(ow[g]));
        }
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (!cnt[t]) continue;
        const uint32_t lb = (tile_bit0[t] & 31u) + (bo0[t] - tile_bit0[t]);   // the thread's first bit in the tile's words
        const uint32_t w = lb >> 5, sh = lb & 31u;
        const uint64_t lo = acc[t] << sh;
        const uint32_t hi = sh ? (uint32_t)(acc[t] >> (64u - sh)) : 0u;
        if (w + 2 < (uint32_t)FW_WORDS) {
            if ((uint32_t)lo) atomicOr(&s_bits[t][w], (uint32_t)lo);
            if ((uint32_t)(lo >> 32)) atomicOr(&s_bits[t][w + 1], (uint32_t)(lo >> 32));
            if (hi) atomicOr(&s_bits[t][w + 2], hi);
        }
    }
    TP_MARK();   // 3: compress + or
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (!P.bits[t]) continue;
        const uint32_t nbits = P.pend[t] + fa_bits(all, t, P.started_in[t]) - tile_bit0[t];   // the tile's appended bits
        if (!nbits) continue;
        uint32_t *gb = (uint32_t *)P.bits[t];
        const uint32_t w0 = tile_bit0[t] >> 5, nw = ((tile_bit0[t] & 31u) + nbits + 31u) >> 5, capw = (P.cap_bits[t] + 31u) >> 5;
        for (uint32_t j = threadIdx.x; j < nw; j += SCAN_BLOCK) {
            if (w0 + j >= capw) break;   // (an estimate too small: the host sees it in the totals and repeats the stage with room)
            const uint32_t v = s_bits[t][j];
            if (j == 0 || j == nw - 1) atomicOr(gb + w0 + j, v);   // (words shared with the neighbouring tiles, or with the pending bits)
            else gb[w0 + j] = v;
        }
    }
    TP_DONE(3);   // 4: the stores
}

// The speculative decode's second launch: every tile's staged bits and packet ends (TileStage, entered "started") to their place
// in the stream, from the state the tile is really entered in -- the scan of the tiles' FrameAgg says it.  Entered "not started",
// the tile's first start-bit value is not appended (FA_DB: one bit fewer, everything behind it one place down) or its first
// packet end does not happen (FA_DC).  One workgroup more in front: dec_verify.
__global__ __launch_bounds__(SCAN_BLOCK) void k_concat(size_t n, const uint32_t *n_dev, const FrameAgg *tile_pre, TileStage S, FrameOut P, bool own_prefix,
                                                      FrameAgg *total_out, DecCarryEpilogue epi, DecVerify V) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if (blockIdx.x == 0) {
        __shared__ QMaps lds_v[SCAN_WAVES];
        dec_verify(V, (n + DEC_TILE - 1) / DEC_TILE, lds_v);
        return;
    }
    const uint32_t bid = blockIdx.x - 1u;   // the tile
    if (bid == 0) pack_pending(P, threadIdx.x, SCAN_BLOCK);
    if (own_prefix && n == 0 && bid == 0 && threadIdx.x == 0) {
        *total_out = FrameAggOp::identity();
        epi(FrameAggOp::identity());
    }
    if ((size_t)bid * DEC_TILE >= n) return;
    TP_DECL();
    __shared__ FrameAgg lds_pre[SCAN_WAVES];
    const FrameAgg pre = own_prefix ? tile_prefix<FrameAggOp, SCAN_BLOCK>(tile_pre, bid, lds_pre) : tile_pre[bid];
    const FrameAgg own = S.own[bid];
    TP_MARK();   // 1: tile prefix
    if (own_prefix && threadIdx.x == 0 && ((size_t)bid + 1) * DEC_TILE >= n) {   // the last tile publishes the total and the carries
        const FrameAgg all = FrameAggOp::op(pre, own);
        *total_out = all;
        epi(all);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (!P.bits[t] || !S.bits[t]) continue;
        const uint32_t started = pm_apply(pre.fl[t], P.started_in[t]);
        const uint32_t bit0 = P.pend[t] + fa_bits(pre, t, P.started_in[t]);   // the tile's first bit in the stream
        const uint32_t nb = own.nb[t];                                        // staged (entered "started")
        const bool drop = !started && (own.fl[t] & FA_DB);
        const uint32_t q = drop ? S.drop_bit[t][bid] : 0xFFFFFFFFu;
        const uint32_t *src = S.bits[t] + (size_t)bid * FW_WORDS;
        uint32_t *dst = (uint32_t *)P.bits[t];
        if (!drop) {
            bitcopy_or(dst, bit0, src, 0u, nb, P.cap_bits[t]);
        } else if (q < nb) {
            bitcopy_or(dst, bit0, src, 0u, q, P.cap_bits[t]);
            bitcopy_or(dst, bit0 + q, src, q + 1u, nb - 1u - q, P.cap_bits[t]);
        }
        const bool dropc = !started && (own.fl[t] & FA_DC);
        const uint32_t ncl = min(own.nc[t], ST_CLOSES), co = fa_closes(pre, t, P.started_in[t]);
        for (uint32_t i = threadIdx.x; i < ncl; i += SCAN_BLOCK) {
            if (dropc && i == 0) continue;
            const uint32_t j = co + i - (dropc ? 1u : 0u);
            if (j >= P.cap_close[t]) continue;
            const uint32_t e = S.close_bit[t][(size_t)bid * ST_CLOSES + i];
            P.close_end[t][j] = bit0 + e - ((drop && e > q) ? 1u : 0u);
            P.close_idx[t][j] = S.close_idx[t][(size_t)bid * ST_CLOSES + i];
        }
    }
    TP_DONE(3);   // 2: bits and packet ends to their places
    // (Measured and dropped, round 4: what k_pkt_finish does -- the open packets' bits, the carry, the mirror -- by the first
    // workgroup of this launch behind dec_verify, from the tiles' staging and a scan of their aggregates of its own.  Exact, but
    // that workgroup is a chain of 20 us (two scans over the tiles, dependent loads, the mirror) beside tiles that live 8:
    // the launch took 22-30 us against 8 + 4 for k_pkt_finish and the boundary before it.)
}

// The symbol arrays (what the decoders emitted, error codes included: the reference hands them to PacketProcessor.append_bit and
// keeps nothing) are written when somebody asks for them -- nfc_read_symbols -- from what the decode pass left per edge.
__global__ __launch_bounds__(SCAN_BLOCK) void k_symbols_write(const uint8_t *outw, size_t n, const uint32_t *n_dev, const FrameAgg *tile_pre,
                                                             FrameOut P, bool own_prefix) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ FramePk lds[SCAN_WAVES];
    __shared__ FrameAgg lds_pre[SCAN_WAVES];
    const FrameAgg pre = own_prefix ? tile_prefix<FrameAggOp, SCAN_BLOCK>(tile_pre, blockIdx.x, lds_pre) : tile_pre[blockIdx.x];
    const size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * DEC_PER_THREAD;
    uint32_t owa[DEC_GROUPS][4];
    FramePk mine = FramePkOp::identity();   // (the thread's aggregate again, from its out-bytes)
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) {
        owa[g][0] = owa[g][1] = owa[g][2] = owa[g][3] = 0u;
        const size_t gb = base + (size_t)DEC_ITEMS * g;
        if (gb < n) {
            const uint4 a = *(const uint4 *)(outw + gb);
            owa[g][0] = a.x; owa[g][1] = a.y; owa[g][2] = a.z; owa[g][3] = a.w;
        }
        mine = FramePkOp::op(mine, FramePkOp::pack(frame_agg_of(owa[g])));
    }
    FramePk total;
    const FramePk in_tile = block_exclusive<FramePkOp>(mine, lds, total);
    FrameAgg before = FrameAggOp::op(pre, FramePkOp::unpack(in_tile));
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) {
        const size_t gb = base + (size_t)DEC_ITEMS * g;
        if (gb >= n) break;
        const uint32_t ow[4] = {owa[g][0], owa[g][1], owa[g][2], owa[g][3]};
        if (!(ow[0] | ow[1] | ow[2] | ow[3])) continue;
        const SlotMasks m = slot_masks(ow);
        const uint64_t lo = (uint64_t)ow[0] | ((uint64_t)ow[1] << 32), hi = (uint64_t)ow[2] | ((uint64_t)ow[3] << 32);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            uint32_t v = m.V[t], off = before.cnt[t];
            while (v) {
                const int slot = __ffs((int)v) - 1, k = slot >> 1;
                v &= v - 1u;
                const uint32_t byte = (uint32_t)((k < 8 ? lo : hi) >> (8 * (k & 7))) & 0xFFu;
                if (off + 1 < P.cap_sym[t]) P.sym[t][off] = (uint8_t)((byte >> ((slot & 1) ? 5 : 2)) & 7u);
                off++;
            }
        }
        before = FrameAggOp::op(before, frame_agg_of(ow));
    }
}

// After framing: keep the open packets' bits for the next batch and publish the carry.  Reads nothing that it (or a
// sibling carry epilogue) writes, so the edge / decode stages of a batch can be repeated as a whole.  One workgroup,
// the batch's last: it also copies the stream state block into the host's pinned mirror (mapped memory), which saves
// the copy-engine launch that would otherwise follow.
struct PktFinish {
    const uint8_t *bits[2];
    uint8_t *pending_next[2];   // (the other half of the double buffer)
    const uint32_t *close_end[2];
    const PktCnt *totals;       // per type: (appended incl. pending) | closes << 32
    const FrameAgg *frame_total;   // framing map over the batch's symbols
    DecCarry *carry;
    int32_t enabled[2], started_in[2];
    uint32_t pending_cap[2];
    uint32_t cap_bits[2], cap_close[2];
    const uint32_t *mirror_src;   // the device state block, or NULL
    uint32_t *mirror_dst;         // the host mirror as the device sees it
    uint32_t mirror_words;
    uint32_t stamp_word, stamp;   // the batch's number, written into word stamp_word of the mirror: this launch wrote it
    int32_t packed;               // bits[t] holds words (the multi-launch stage), not a byte per bit
};
__global__ __launch_bounds__(256) void k_pkt_finish(PktFinish F) {
    for (int t = 0; t < 2; t++) {
        if (!F.enabled[t]) continue;
        const uint64_t tot = F.totals->v[t];
        const uint32_t nbits = (uint32_t)tot, ncl = (uint32_t)(tot >> 32);
        if (nbits > F.cap_bits[t] || ncl > F.cap_close[t]) continue;   // the host repeats the stage with room
        const uint32_t from = ncl ? F.close_end[t][ncl - 1] : 0u;
        const uint32_t keep = nbits - from;
        const uint32_t *bw = (const uint32_t *)F.bits[t];
        for (uint32_t i = threadIdx.x; i < keep && i < F.pending_cap[t]; i += blockDim.x)
            F.pending_next[t][i] = F.packed ? (uint8_t)((bw[(from + i) >> 5] >> ((from + i) & 31u)) & 1u) : F.bits[t][from + i];
        if (threadIdx.x == 0) {
            F.carry->pending[t] = keep;
            F.carry->pkt_started[t] = (int32_t)pm_apply(F.frame_total->fl[t], (uint32_t)F.started_in[t]);
        }
    }
    if (F.mirror_src) {
        __threadfence_block();   // (workgroup scope: a device-scope release writes the L2 back, and only this workgroup reads the words)
        __syncthreads();   // thread 0's carry words are part of the block
        // (the stamp LAST, behind a system-scope fence: a host that watches the stamp instead of waiting for the stream -- host_threshold.h:
        // wait_for_stamp -- finds every other word of the block in place when it sees it)
        for (uint32_t i = threadIdx.x; i < F.mirror_words; i += blockDim.x)
            if (i != F.stamp_word) F.mirror_dst[i] = F.mirror_src[i];
        __threadfence_system();   // (every thread's own words are in place system-wide ...)
        __syncthreads();          // (... before the barrier lets thread 0 send the stamp)
        if (threadIdx.x == 0) {
            ((volatile uint32_t *)F.mirror_dst)[F.stamp_word] = F.stamp;
        }
    }
}

}  // namespace nfc
