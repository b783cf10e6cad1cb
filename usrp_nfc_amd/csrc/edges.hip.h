// edges.hip.h -- run-length / edge timing (transition_sink.py:84-99) from the classification
// bit planes, one thread per 64-sample word.
//
// The reference walks samples keeping (_last_bit, _dur, _current_state).  Those three values at any
// sample are a closed form of the last two positions where val changed (and of the carried values
// when fewer than two changes precede it):
//   _last_bit        = val at the last change
//   _dur             = ((samples since that change - 1) mod max_len) + 1     (time-outs restart it)
//   _current_state   = 2 / 1 inside a LOW / HIGH run unless its latest sample fired a time-out;
//                      inside a val-0 run: 0 once the run is longer than max_len, else whatever
//                      the previous run left (which, being a LOW/HIGH run, depends on its length only)
// So an ordered scan hands every word the last two change positions before it; the word's thread then
// replays the reference loop over its 64 samples, jumping from event to event (val change or
// time-out), first to count its entries, then -- after a prefix sum -- to write them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nfc_amd.h"
#include "scan.hip.h"

namespace nfc {

// carried transition_sink variables (device resident, host mirrored)
struct EdgeCarry {
    int32_t state, last_bit, dur;
    int32_t pad;
};

constexpr int32_t POS_NONE = INT32_MIN;

struct Last2 {   // the two most recent positions (batch-local sample indices) where val changed
    int32_t s1, s2;
};
struct Last2Op {
    using T = Last2;
    static __host__ __device__ __forceinline__ T identity() { return T{POS_NONE, POS_NONE}; }
    static T identity_host() { return identity(); }
    static __device__ __forceinline__ T op(T a, T b) {
        if (b.s1 == POS_NONE) return a;
        if (b.s2 == POS_NONE) return T{b.s1, a.s1};
        return b;
    }
    static __device__ __forceinline__ T shfl_up(T v, int d) {
        return T{__shfl_up(v.s1, d, 64), __shfl_up(v.s2, d, 64)};
    }
};

struct EdgeArgs {
    const uint64_t *neg, *pos;  // classification bit planes (LOW / HIGH), 64 samples per word
    uint32_t n, skip;           // samples in the batch; samples before skip belong to the fill phase
    int32_t mx;
    int32_t dur_in, last_bit_in, state_in;   // carried _dur / _last_bit / _current_state
    int32_t nd;                 // max_len + 1 (decoder LUT row length)
    uint64_t g0;                // global index of batch sample 0

    __device__ __forceinline__ int val_at(int32_t p) const {
        if ((neg[p >> 6] >> (p & 63)) & 1ull) return -1;
        return (int)((pos[p >> 6] >> (p & 63)) & 1ull);
    }
    // samples of word w whose val differs from the previous sample's (run starts), within [skip, n)
    __device__ __forceinline__ uint64_t change_mask(size_t w, uint64_t &ng, uint64_t &ps) const {
        ng = neg[w];
        ps = pos[w];
        uint64_t pn, pp;
        if (w == 0) {
            pn = last_bit_in == -1 ? 1ull : 0ull;
            pp = last_bit_in == 1 ? 1ull : 0ull;
        } else {
            pn = neg[w - 1] >> 63;
            pp = pos[w - 1] >> 63;
        }
        uint64_t m = (ng ^ ((ng << 1) | pn)) | (ps ^ ((ps << 1) | pp));
        const long long first = (long long)w * 64;
        const long long lo = (long long)skip - first, hi = (long long)n - first;
        if (lo > 0) m &= (lo >= 64) ? 0ull : (~0ull << lo);
        if (hi < 64) m &= (hi <= 0) ? 0ull : (~0ull >> (64 - hi));
        return m;
    }
    __device__ __forceinline__ int run_state(int v, int len) const {   // after `len` samples of a LOW / HIGH run
        return (len > 1 && (len - 1) % mx == 0) ? 0 : (v == -1 ? 2 : 1);
    }
    // (_last_bit, _dur, _current_state) after sample p - 1, given the last two change positions before p
    __device__ __forceinline__ void state_before(int32_t p, Last2 c, int &lb, int &dur, int &st) const {
        if (c.s1 == POS_NONE) {                       // still in the run carried into the batch
            lb = last_bit_in;
            if (p <= (int32_t)skip) { dur = dur_in; st = state_in; return; }
            const int len = p - ((int32_t)skip - dur_in);
            dur = ((len - 1) % mx) + 1;
            if (lb == 0) st = (len >= mx + 1) ? 0 : state_in;
            else st = run_state(lb, len);
            return;
        }
        lb = val_at(c.s1);
        const int len = p - c.s1;
        dur = ((len - 1) % mx) + 1;
        if (lb != 0) { st = run_state(lb, len); return; }
        if (len >= mx + 1) { st = 0; return; }
        // a short val-0 run keeps what the previous (LOW / HIGH, possibly carried) run left
        if (c.s2 != POS_NONE) { st = run_state(val_at(c.s2), c.s1 - c.s2); return; }
        if (c.s1 == (int32_t)skip) { st = state_in; return; }      // the carried run had no sample in this batch
        const int len0 = c.s1 - ((int32_t)skip - dur_in);
        if (last_bit_in == 0) st = (len0 >= mx + 1) ? 0 : state_in;
        else st = run_state(last_bit_in, len0);
    }
};

// Replay transition_sink.py:84-99 over word w from event to event.  emit(sample, v, d, t) per entry.
template <class Emit>
__device__ __forceinline__ uint32_t replay_word(const EdgeArgs &A, size_t w, Last2 ctx, Emit emit) {
    const int32_t w0 = (int32_t)(w * 64);
    const int32_t lo = max(w0, (int32_t)A.skip), hi = min(w0 + 64, (int32_t)A.n);
    if (lo >= hi) return 0;
    uint64_t ng, ps;
    uint64_t m = A.change_mask(w, ng, ps);
    int lb, dur, st;
    A.state_before(lo, ctx, lb, dur, st);
    const int mx = A.mx;
    uint32_t cnt = 0;
    int32_t p = lo;
    while (p < hi) {
        const int32_t nc = m ? w0 + (__ffsll((long long)m) - 1) : hi;
        if (nc == p) {   // val changes at p (transition_sink.py:86-92)
            const int b = p - w0;
            const int v = ((ng >> b) & 1ull) ? -1 : (int)((ps >> b) & 1ull);
            const int prev_st = st;
            if (v == -1) st = 2;
            else if (v == 1) st = 1;
            emit(p, (st == 2) ? lb + 1 : lb, (prev_st == 0) ? mx : dur, st - 1);
            cnt++;
            dur = 1;
            lb = v;
            m &= m - 1;
            p++;
            continue;
        }
        int nrem = nc - p;   // samples that continue the run
        while (nrem > 0) {
            const int t = mx + 1 - dur;   // samples until _dur exceeds max_len (transition_sink.py:95-99)
            if (t <= nrem) {
                const int cs = (lb == -1) ? 2 : ((lb == 1) ? 1 : st);
                emit(p + t - 1, (cs == 2) ? lb + 1 : lb, mx, cs - 1);
                cnt++;
                dur = 1;
                st = 0;
                p += t;
                nrem -= t;
            } else {
                dur += nrem;
                if (lb == -1) st = 2;
                else if (lb == 1) st = 1;
                p += nrem;
                nrem = 0;
            }
        }
    }
    return cnt;
}

// ---- scan 1: last two change positions before every word; its apply also counts the word's entries ----
struct LoadLast2 {
    EdgeArgs A;
    __device__ __forceinline__ Last2 operator()(size_t w) const {
        uint64_t ng, ps;
        const uint64_t m = A.change_mask(w, ng, ps);
        if (!m) return Last2{POS_NONE, POS_NONE};
        const int b1 = 63 - __clzll((long long)m);
        const uint64_t m2 = m & ~(1ull << b1);
        return Last2{(int32_t)(w * 64) + b1, m2 ? (int32_t)(w * 64) + (63 - __clzll((long long)m2)) : POS_NONE};
    }
};
struct StoreCtxAndCount {
    EdgeArgs A;
    Last2 *ctx;
    uint32_t *cnt;
    __device__ __forceinline__ void operator()(size_t w, Last2 excl, Last2) const {
        ctx[w] = excl;
        cnt[w] = replay_word(A, w, excl, [](int32_t, int, int, int) {});
    }
};

// ---- scan 2: where each word's entries go; its apply writes them -------------------------
struct LoadWordCount {
    const uint32_t *cnt;
    __device__ __forceinline__ uint32_t operator()(size_t w) const { return cnt[w]; }
};
// per edge, for the decoders: LUT row (v + 1) * nd + d in the low 14 bits, route in the top two
// (0 dropped, 1 Manchester / tag->reader, 2 Miller / reader->tag; background.py:30-35)
__device__ __forceinline__ uint16_t edge_code(int v, int d, int t, int nd) {
    const int dd = d < nd ? d : nd - 1;
    return (uint16_t)(((v + 1) * nd + dd) | ((t + 1) << 14));
}
struct StoreWordEdges {
    EdgeArgs A;
    const Last2 *ctx;
    nfc_edge *edges;
    uint16_t *ecode;
    uint32_t cap;
    __device__ __forceinline__ void operator()(size_t w, uint32_t excl, uint32_t count) const {
        if (!count) return;
        uint32_t k = excl;
        const EdgeArgs &a = A;
        nfc_edge *e = edges;
        uint16_t *ec = ecode;
        const uint32_t cp = cap;
        replay_word(A, w, ctx[w], [&](int32_t p, int v, int d, int t) {
            if (k < cp) {
                nfc_edge o;
                o.idx = a.g0 + (uint64_t)p;
                o.d = d;
                o.v = (int8_t)v;
                o.t = (int8_t)t;
                o.pad = 0;
                e[k] = o;
                ec[k] = edge_code(v, d, t, a.nd);
            }
            k++;
        });
    }
};

// The writer proper: a workgroup owns 512 consecutive words (two per thread).  It places its entries with a
// block scan, replays its words into LDS, and copies the staged entries out with full-width coalesced
// stores (a thread's own entries are only ~4 x 16 B apart from its neighbour's -- written directly they
// touch a cache line per lane).  Tiles with more entries than fit the stage write directly.
constexpr int EW_ITEMS = 2;
constexpr int EW_CAP = 3072;   // staged entries per workgroup: 48 KB of nfc_edge + 6 KB of codes
__global__ __launch_bounds__(SCAN_BLOCK) void k_write_edges(EdgeArgs A, size_t nwords, const Last2 *ctx, const uint32_t *wcnt,
                                                           const uint32_t *tile_base, nfc_edge *edges, uint16_t *ecode,
                                                           uint32_t cap) {
    __shared__ __attribute__((aligned(16))) nfc_edge s_edge[EW_CAP];
    __shared__ uint16_t s_code[EW_CAP];
    __shared__ uint32_t s_scan[SCAN_WAVES];
    const size_t w0 = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * EW_ITEMS;
    uint32_t c[EW_ITEMS], mine = 0;
#pragma unroll
    for (int i = 0; i < EW_ITEMS; i++) {
        c[i] = (w0 + i < nwords) ? wcnt[w0 + i] : 0u;
        mine += c[i];
    }
    uint32_t total;
    uint32_t off = block_exclusive<AddU32>(mine, s_scan, total);
    const uint32_t gbase = tile_base[blockIdx.x];
    const bool staged = total <= (uint32_t)EW_CAP;
#pragma unroll
    for (int i = 0; i < EW_ITEMS; i++) {
        if (!c[i]) continue;
        uint32_t k = off;
        const int nd = A.nd;
        const uint64_t g0 = A.g0;
        if (staged) {
            replay_word(A, w0 + i, ctx[w0 + i], [&](int32_t p, int v, int d, int t) {
                nfc_edge o;
                o.idx = g0 + (uint64_t)p;
                o.d = d;
                o.v = (int8_t)v;
                o.t = (int8_t)t;
                o.pad = 0;
                s_edge[k] = o;
                s_code[k] = edge_code(v, d, t, nd);
                k++;
            });
        } else {
            replay_word(A, w0 + i, ctx[w0 + i], [&](int32_t p, int v, int d, int t) {
                const uint32_t g = gbase + k;
                if (g < cap) {
                    nfc_edge o;
                    o.idx = g0 + (uint64_t)p;
                    o.d = d;
                    o.v = (int8_t)v;
                    o.t = (int8_t)t;
                    o.pad = 0;
                    edges[g] = o;
                    ecode[g] = edge_code(v, d, t, nd);
                }
                k++;
            });
        }
        off += c[i];
    }
    if (staged) {
        __syncthreads();
        const uint4 *src = (const uint4 *)s_edge;
        uint4 *dst = (uint4 *)(edges + gbase);
        for (uint32_t i = threadIdx.x; i < total; i += SCAN_BLOCK)
            if (gbase + i < cap) {
                dst[i] = src[i];
                ecode[gbase + i] = s_code[i];
            }
    }
}

// carried (_last_bit, _dur, _current_state) after the batch, from the scan-1 total
__global__ void k_edge_carry(EdgeArgs A, const Last2 *total, EdgeCarry *carry) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (A.skip >= A.n) return;  // nothing but fill samples: unchanged
    int lb, dur, st;
    A.state_before((int32_t)A.n, *total, lb, dur, st);
    carry->last_bit = lb;
    carry->dur = dur;
    carry->state = st;
}

}  // namespace nfc
