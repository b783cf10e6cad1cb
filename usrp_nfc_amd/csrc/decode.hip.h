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
    const uint64_t *mil_map;  // [(cur+1) * nd + d]
    const uint32_t *man_map;
    const uint8_t *mil_out;   // [((cur+1) * nd + d) * 16 + state]
    const uint8_t *man_out;   // [((cur+1) * nd + d) * 8 + state]
    int32_t nd;               // max_len + 1
    int32_t reader, tag;
};

struct DecCarry {
    int32_t mil_state, man_state;
    int32_t pkt_started[2];
    uint32_t pending[2];  // bits of the open packet kept from earlier batches, per type
};

__device__ __forceinline__ int edge_lut_index(const nfc_edge &e, int nd) {
    const int d = e.d < nd ? e.d : nd - 1;
    return (e.v + 1) * nd + d;
}

// ---- pass 1: incoming decoder state of every edge ----------------------------
struct LoadEdgeMaps {
    const nfc_edge *edges;
    DecTables T;
    __device__ __forceinline__ DecMaps operator()(size_t i) const {
        const nfc_edge e = edges[i];
        DecMaps m{identity_map(16), (uint32_t)identity_map(8)};
        const int li = edge_lut_index(e, T.nd);
        if (e.t == 1 && T.reader) m.mil = T.mil_map[li];
        else if (e.t == 0 && T.tag) m.man = T.man_map[li];
        return m;
    }
};
struct StoreEdgeStates {
    uint8_t *states;  // mil | man << 4
    int32_t mil0, man0;
    __device__ __forceinline__ void operator()(size_t i, DecMaps excl, DecMaps) const {
        const uint32_t ms = (uint32_t)(excl.mil >> (4 * mil0)) & 15u;
        const uint32_t ts = (excl.man >> (4 * man0)) & 15u;
        states[i] = (uint8_t)(ms | (ts << 4));
    }
};

// ---- pass 2: symbols ------------------------------------------------------------
// count word: Miller symbols in the low half, Manchester in the high half
__device__ __forceinline__ uint8_t edge_out_word(const nfc_edge &e, uint8_t st, const DecTables &T, int &type) {
    const int li = edge_lut_index(e, T.nd);
    if (e.t == 1 && T.reader) { type = 1; return T.mil_out[(size_t)li * 16 + (st & 15)]; }
    if (e.t == 0 && T.tag) { type = 0; return T.man_out[(size_t)li * 8 + ((st >> 4) & 7)]; }
    type = -1;
    return 0;
}
struct LoadSymCounts {
    const nfc_edge *edges;
    const uint8_t *states;
    DecTables T;
    __device__ __forceinline__ uint64_t operator()(size_t i) const {
        int type;
        const uint8_t w = edge_out_word(edges[i], states[i], T, type);
        const uint64_t n = w & 3u;
        return type == 1 ? n : (type == 0 ? (n << 32) : 0ull);
    }
};
struct StoreSymbols {
    const nfc_edge *edges;
    const uint8_t *states;
    DecTables T;
    uint8_t *sym[2];   // [0] Manchester / tag, [1] Miller / reader
    uint32_t *src[2];  // index of the producing edge
    __device__ __forceinline__ void operator()(size_t i, uint64_t excl, uint64_t) const {
        int type;
        const uint8_t w = edge_out_word(edges[i], states[i], T, type);
        const int n = w & 3;
        if (type < 0 || n == 0) return;
        const uint32_t off = type == 1 ? (uint32_t)excl : (uint32_t)(excl >> 32);
        sym[type][off] = (w >> 2) & 7u;
        src[type][off] = (uint32_t)i;
        if (n > 1) {
            sym[type][off + 1] = (w >> 5) & 7u;
            src[type][off + 1] = (uint32_t)i;
        }
    }
};

// ---- framing: PacketProcessor.append_bit (packets.py:67-79) ---------------------
// state 0 = not started, 1 = started
__device__ __forceinline__ uint32_t pkt_map(uint8_t s, int start_bit) {
    if (s > 1) return 0x00u;                       // error symbol: started -> not started, not started stays
    if ((int)s == start_bit) return 0x11u;         // start bit: not started -> started (dropped); started stays
    return 0x10u;                                  // other bit: identity
}
struct LoadPktMaps {
    const uint8_t *sym;
    int start_bit;
    __device__ __forceinline__ uint32_t operator()(size_t i) const { return pkt_map(sym[i], start_bit); }
};
struct StorePktStarted {
    uint8_t *started;
    int32_t started0;
    __device__ __forceinline__ void operator()(size_t i, uint32_t excl, uint32_t) const {
        started[i] = (uint8_t)((excl >> (4 * started0)) & 1u);
    }
};
// count word: appended bits in the low half, closes in the high half
struct LoadPktCounts {
    const uint8_t *sym;
    const uint8_t *started;
    int start_bit;
    __device__ __forceinline__ uint64_t operator()(size_t i) const {
        const uint8_t s = sym[i];
        const bool st = started[i];
        if (s > 1) return st ? (1ull << 32) : 0ull;
        return (!st && (int)s == start_bit) ? 0ull : 1ull;
    }
};
struct StorePkt {
    const uint8_t *sym;
    const uint8_t *started;
    const uint32_t *src;
    const nfc_edge *edges;
    int start_bit;
    uint8_t *bits;       // appended bits, starting with the pending ones of earlier batches
    uint32_t *close_end; // per close: number of bits appended before it (= end offset of the packet)
    uint64_t *close_idx; // per close: sample index of the closing edge
    __device__ __forceinline__ void operator()(size_t i, uint64_t excl, uint64_t) const {
        const uint8_t s = sym[i];
        const bool st = started[i];
        if (s > 1) {
            if (st) {
                const uint32_t k = (uint32_t)(excl >> 32);
                close_end[k] = (uint32_t)excl;
                close_idx[k] = edges[src[i]].idx;
            }
        } else if (st || (int)s != start_bit) {
            bits[(uint32_t)excl] = s;
        }
    }
};

// After framing: keep the open packet's bits for the next batch and update the carry.
struct PktFinish {
    uint8_t *bits;
    uint8_t *pending;        // [cap]
    const uint32_t *close_end;
    const uint64_t *totals;  // (appended incl. pending) | closes << 32
    const uint32_t *map_total;
    DecCarry *carry;
    int type;
    uint32_t pending_cap;
};
__global__ __launch_bounds__(256) void k_pkt_finish(PktFinish F) {
    const uint64_t tot = *F.totals;
    const uint32_t nbits = (uint32_t)tot, ncl = (uint32_t)(tot >> 32);
    const uint32_t from = ncl ? F.close_end[ncl - 1] : 0u;
    const uint32_t keep = nbits - from;
    for (uint32_t i = threadIdx.x; i < keep && i < F.pending_cap; i += blockDim.x) F.pending[i] = F.bits[from + i];
    if (threadIdx.x == 0) {
        F.carry->pending[F.type] = keep;
        F.carry->pkt_started[F.type] = (int32_t)((*F.map_total >> (4 * F.carry->pkt_started[F.type])) & 1u);
    }
}

// Decoder states after the batch, from the total of the map scan.
__global__ void k_dec_carry(const DecMaps *total, DecCarry *carry) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    carry->mil_state = (int32_t)((total->mil >> (4 * carry->mil_state)) & 15u);
    carry->man_state = (int32_t)((total->man >> (4 * carry->man_state)) & 15u);
}

}  // namespace nfc
