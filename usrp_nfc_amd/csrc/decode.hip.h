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
    const uint4 *mil_map;     // [(cur+1) * nd + d]  16 states, one byte each
    const uint2 *man_map;     //                      8 states, one byte each
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

// ---- pass 1: every edge as a pair of state maps; pass 2 visits it with its incoming states ----
// Edges arrive as 16-bit codes (edges.hip.h: edge_code): LUT row | route << 14.
struct LoadEdgeMaps {
    const uint16_t *ecode;
    DecTables T;
    __device__ __forceinline__ DecMaps operator()(size_t i) const {
        const uint32_t c = ecode[i];
        const uint32_t li = c & 0x3FFFu, route = c >> 14;
        DecMaps m = ComposeDec::identity();
        if (route == 2u && T.reader) {
            const uint4 v = T.mil_map[li];
            m.mil[0] = v.x; m.mil[1] = v.y; m.mil[2] = v.z; m.mil[3] = v.w;
        } else if (route == 1u && T.tag) {
            const uint2 v = T.man_map[li];
            m.man[0] = v.x; m.man[1] = v.y;
        }
        return m;
    }
};
// What an edge emits, one byte: bits 0-1 = 0 nothing, 1 / 2 Miller symbols, 3 one Manchester symbol;
// bits 2-4 first symbol, bits 5-7 second symbol.
struct VisitEdgeOut {
    const uint16_t *ecode;
    DecTables T;
    uint8_t *outw;
    __device__ __forceinline__ void operator()(size_t i, uint32_t st, const DecMaps &) const {
        const uint32_t c = ecode[i];
        const uint32_t li = c & 0x3FFFu, route = c >> 14;
        uint8_t w = 0;
        if (route == 2u && T.reader) w = T.mil_out[(size_t)li * 16 + (st & 15u)];
        else if (route == 1u && T.tag) {
            const uint8_t m = T.man_out[(size_t)li * 8 + ((st >> 4) & 7u)];
            w = (m & 3u) ? (uint8_t)((m & 0xFCu) | 3u) : (uint8_t)0;
        }
        outw[i] = w;
    }
};

// ---- symbols ---------------------------------------------------------------------
// count word: Miller symbols in the low half, Manchester in the high half
struct LoadSymCounts {
    const uint8_t *outw;
    __device__ __forceinline__ uint64_t operator()(size_t i) const {
        const uint32_t k = outw[i] & 3u;
        return k == 3u ? (1ull << 32) : (uint64_t)k;
    }
};
struct StoreSymbols {
    const uint8_t *outw;
    uint8_t *sym[2];   // [0] Manchester / tag, [1] Miller / reader
    uint32_t *src[2];  // index of the producing edge
    uint32_t cap[2];   // buffer capacities (an overflow is detected by the host from the totals)
    __device__ __forceinline__ void operator()(size_t i, uint64_t excl, uint64_t) const {
        const uint8_t w = outw[i];
        const uint32_t k = w & 3u;
        if (k == 0) return;
        const int type = (k == 3u) ? 0 : 1;
        const uint32_t off = type == 1 ? (uint32_t)excl : (uint32_t)(excl >> 32);
        if (off + 1 >= cap[type]) return;
        sym[type][off] = (w >> 2) & 7u;
        src[type][off] = (uint32_t)i;
        if (k == 2u) {
            sym[type][off + 1] = (w >> 5) & 7u;
            src[type][off + 1] = (uint32_t)i;
        }
    }
};

// ---- framing: PacketProcessor.append_bit (packets.py:67-79) ---------------------
// state 0 = not started, 1 = started; nibble map
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
// per symbol: bit 0 = appended to the packet, bit 1 = closes a started packet
struct VisitPktFlags {
    const uint8_t *sym;
    int start_bit;
    uint8_t *pflags;
    __device__ __forceinline__ void operator()(size_t i, uint32_t started, uint32_t) const {
        const uint8_t s = sym[i];
        uint8_t f;
        if (s > 1) f = started ? 2 : 0;
        else f = (!started && (int)s == start_bit) ? 0 : 1;
        pflags[i] = f;
    }
};
// count word: appended bits in the low half, closes in the high half
struct LoadPktCounts {
    const uint8_t *pflags;
    __device__ __forceinline__ uint64_t operator()(size_t i) const {
        const uint32_t f = pflags[i];
        return (uint64_t)(f & 1u) | ((uint64_t)(f >> 1) << 32);
    }
};
struct StorePkt {
    const uint8_t *sym;
    const uint8_t *pflags;
    const uint32_t *src;
    const nfc_edge *edges;
    uint8_t *bits;       // appended bits, starting with the pending ones of earlier batches
    uint32_t *close_end; // per close: number of bits appended before it (= end offset of the packet)
    uint64_t *close_idx; // per close: sample index of the closing edge
    __device__ __forceinline__ void operator()(size_t i, uint64_t excl, uint64_t) const {
        const uint32_t f = pflags[i];
        if (f & 2u) {
            const uint32_t k = (uint32_t)(excl >> 32);
            close_end[k] = (uint32_t)excl;
            close_idx[k] = edges[src[i]].idx;
        } else if (f & 1u) {
            bits[(uint32_t)excl] = sym[i];
        }
    }
};

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

// Decoder states after the batch, from the total of the map scan.
// Also splits the packed symbol totals into the two per-type counts the framing scans read from the device.
__global__ void k_dec_carry(const DecMaps *total, uint32_t state_in, DecCarry *carry, const uint64_t *sym_total,
                            uint32_t *nsym) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t st = ComposeDec::step(*total, state_in);
    carry->mil_state = (int32_t)(st & 15u);
    carry->man_state = (int32_t)(st >> 4);
    const uint64_t t = *sym_total;
    nsym[1] = (uint32_t)t;           // Miller / reader
    nsym[0] = (uint32_t)(t >> 32);   // Manchester / tag
}

}  // namespace nfc
