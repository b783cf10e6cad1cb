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
// So an ordered scan hands every word the last two change positions before it.  The word's thread then
// marks the samples that produce an entry -- val changes, and time-outs every max_len samples after a
// run's first sample -- in a 64-bit event mask; after a prefix sum over the popcounts one thread per
// ENTRY rebuilds the three values before its sample from the closed form, takes the reference's step
// for that one sample (transition_sink.py:84-99) and writes the entry: stores are dense and in order.
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
    uint32_t mx_magic;          // floor(2^32 / max_len) (0xFFFFFFFF for max_len 1): modulo without a divide
    uint64_t g0;                // global index of batch sample 0
    uint64_t per_mask;          // bits 0, max_len, 2 max_len, ... < 64

    __device__ __forceinline__ int mod_mx(int x) const {   // x mod max_len for 0 <= x < 2^31
        const uint32_t q = __umulhi((uint32_t)x, mx_magic);   // floor(x / mx) or one less
        const uint32_t r = (uint32_t)x - q * (uint32_t)mx;
        return (int)(r >= (uint32_t)mx ? r - (uint32_t)mx : r);
    }

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
        return (len > 1 && mod_mx(len - 1) == 0) ? 0 : (v == -1 ? 2 : 1);
    }
    // (_last_bit, _dur, _current_state) after sample p - 1, given the last two change positions before p
    __device__ __forceinline__ void state_before(int32_t p, Last2 c, int &lb, int &dur, int &st) const {
        state_before(p, c, lb, dur, st, [this](int32_t q) { return val_at(q); });
    }
    // the same with the caller's way of reading val at a position (a tile's planes staged in LDS)
    template <class Val>
    __device__ __forceinline__ void state_before(int32_t p, Last2 c, int &lb, int &dur, int &st, Val val) const {
        if (c.s1 == POS_NONE) {                       // still in the run carried into the batch
            lb = last_bit_in;
            if (p <= (int32_t)skip) { dur = dur_in; st = state_in; return; }
            const int len = p - ((int32_t)skip - dur_in);
            dur = mod_mx(len - 1) + 1;
            if (lb == 0) st = (len >= mx + 1) ? 0 : state_in;
            else st = run_state(lb, len);
            return;
        }
        lb = val(c.s1);
        const int len = p - c.s1;
        dur = mod_mx(len - 1) + 1;
        if (lb != 0) { st = run_state(lb, len); return; }
        if (len >= mx + 1) { st = 0; return; }
        // a short val-0 run keeps what the previous (LOW / HIGH, possibly carried) run left
        if (c.s2 != POS_NONE) { st = run_state(val(c.s2), c.s1 - c.s2); return; }
        if (c.s1 == (int32_t)skip) { st = state_in; return; }      // the carried run had no sample in this batch
        const int len0 = c.s1 - ((int32_t)skip - dur_in);
        if (last_bit_in == 0) st = (len0 >= mx + 1) ? 0 : state_in;
        else st = run_state(last_bit_in, len0);
    }
};

__device__ __forceinline__ uint64_t low_mask(int k) { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); }   // bits [0, k)

// Samples of word w that produce an entry: the val changes m, and the time-outs -- sample s + k max_len
// (k >= 1) of a run whose first sample is s, up to the run's end (transition_sink.py:95-99: _dur is 1 at s
// and exceeds max_len max_len samples later, where it restarts at 1).
__device__ __forceinline__ uint64_t event_mask(const EdgeArgs &A, size_t w, Last2 ctx, uint64_t m) {
    const int32_t w0 = (int32_t)(w * 64);
    const int32_t lo = max(w0, (int32_t)A.skip), hi = min(w0 + 64, (int32_t)A.n);
    if (lo >= hi) return 0ull;
    // the run carried into the word started at s (before the batch: where its carried _dur puts it)
    const int32_t s = (ctx.s1 != POS_NONE) ? ctx.s1 : (int32_t)A.skip - A.dur_in;
    const int32_t pmin = max(lo, s + 1);
    const int r = A.mod_mx(pmin - s);
    const int32_t p0 = r ? pmin + (A.mx - r) : pmin;   // its first time-out at or after lo
    uint64_t t = 0ull;
    if (p0 - w0 < 64) t = (A.per_mask << (p0 - w0)) & low_mask(m ? __ffsll((long long)m) - 1 : 64);
    if (A.mx < 64) {   // runs that start inside the word can time out inside it
        uint64_t rest = m;
        while (rest) {
            const int b = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            t |= (A.per_mask << b) & ~(1ull << b) & low_mask(rest ? __ffsll((long long)rest) - 1 : 64);
        }
    }
    return (m | t) & low_mask(hi - w0);
}

// The entry of the event at bit b of a word (ng / ps: its planes, m: its changes, ctx: the two changes before it).
template <class Val>
__device__ __forceinline__ void event_entry(const EdgeArgs &A, int32_t w0, int b, uint64_t ng, uint64_t ps, uint64_t m,
                                            Last2 ctx, int &v, int &d, int &t, Val val) {
    const uint64_t mb = m & low_mask(b);
    Last2 c = ctx;
    if (mb) {
        const int b1 = 63 - __clzll((long long)mb);
        const uint64_t m2 = mb & ~(1ull << b1);
        c.s2 = m2 ? w0 + (63 - __clzll((long long)m2)) : ctx.s1;
        c.s1 = w0 + b1;
    }
    int lb, dur, st;
    A.state_before(w0 + b, c, lb, dur, st, val);
    if ((m >> b) & 1ull) {   // val changes here (transition_sink.py:86-92)
        const int val = ((ng >> b) & 1ull) ? -1 : (int)((ps >> b) & 1ull);
        const int prev_st = st;
        if (val == -1) st = 2;
        else if (val == 1) st = 1;
        v = (st == 2) ? lb + 1 : lb;
        d = (prev_st == 0) ? A.mx : dur;
        t = st - 1;
    } else {                 // _dur exceeds max_len (transition_sink.py:95-99)
        const int cs = (lb == -1) ? 2 : ((lb == 1) ? 1 : st);
        v = (cs == 2) ? lb + 1 : lb;
        d = A.mx;
        t = cs - 1;
    }
}

__device__ __forceinline__ void event_entry(const EdgeArgs &A, int32_t w0, int b, uint64_t ng, uint64_t ps, uint64_t m,
                                            Last2 ctx, int &v, int &d, int &t) {
    event_entry(A, w0, b, ng, ps, m, ctx, v, d, t, [&A](int32_t q) { return A.val_at(q); });
}

// ---- scan 1: last two change positions before every word; its apply also marks the word's events and
// sums their counts per tile (the aggregates of scan 2: where each word's entries go) ----
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
struct StoreCtxAndEvents {
    EdgeArgs A;
    Last2 *ctx;
    uint64_t *evm;
    __device__ __forceinline__ uint32_t operator()(size_t w, Last2 excl, Last2) const {   // returns the word's entry count
        uint64_t ng, ps;
        const uint64_t m = A.change_mask(w, ng, ps);
        const uint64_t e = event_mask(A, w, excl, m);
        ctx[w] = excl;
        evm[w] = e;
        return (uint32_t)__popcll(e);
    }
};

// per edge, for the decoders: LUT row (v + 1) * nd + d in the low 14 bits, route in the top two
// (0 dropped, 1 Manchester / tag->reader, 2 Miller / reader->tag; background.py:30-35)
__device__ __forceinline__ uint16_t edge_code(int v, int d, int t, int nd) {
    const int dd = d < nd ? d : nd - 1;
    return (uint16_t)(((v + 1) * nd + dd) | ((t + 1) << 14));
}

// carried (_last_bit, _dur, _current_state) after the batch, from the scan-1 total; runs once after the entry-count
// scan: as the epilogue of its prefix launch, or in the workgroup of the last tile of k_write_edges
struct EdgeCarryEpilogue {
    EdgeArgs A;
    const Last2 *total;
    EdgeCarry *carry;
    __device__ __forceinline__ void operator()(uint32_t) const {
        if (A.skip >= A.n) return;  // nothing but fill samples: unchanged
        int lb, dur, st;
        A.state_before((int32_t)A.n, *total, lb, dur, st);
        carry->last_bit = lb;
        carry->dur = dur;
        carry->state = st;
    }
};

// The writer: a workgroup owns 512 consecutive words (two per thread).  It places its entries with a block
// scan, lists them as (word, bit) in LDS, and then works one thread per entry, so that consecutive lanes
// store consecutive 16-byte entries.
constexpr int EW_ITEMS = 2;
constexpr int EW_WORDS = SCAN_BLOCK * EW_ITEMS;
constexpr int EW_CAP = 4096;   // entries listed per round (a tile holds 2150 on the bench workloads, 32768 at most)
__global__ __launch_bounds__(SCAN_BLOCK) void k_write_edges(EdgeArgs A, size_t nwords, const Last2 *ctx, const uint64_t *evm,
                                                           const uint32_t *tile_base, nfc_edge *edges, uint16_t *ecode,
                                                           uint32_t cap, bool own_prefix, uint32_t *total_out, const Last2 *last2_total,
                                                           EdgeCarry *carry_out) {
    __shared__ uint64_t s_ng[EW_WORDS], s_ps[EW_WORDS], s_m[EW_WORDS];
    __shared__ Last2 s_ctx[EW_WORDS];
    __shared__ uint16_t s_ev[EW_CAP];
    __shared__ uint32_t s_scan[SCAN_WAVES];
    const int wl0 = (int)threadIdx.x * EW_ITEMS;
    const size_t wt = (size_t)blockIdx.x * EW_WORDS;   // first word of the tile
    uint64_t ev[EW_ITEMS];
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < EW_ITEMS; i++) {
        const size_t w = wt + wl0 + i;
        ev[i] = 0ull;
        if (w < nwords) {
            uint64_t ng, ps;
            s_m[wl0 + i] = A.change_mask(w, ng, ps);
            s_ng[wl0 + i] = ng;
            s_ps[wl0 + i] = ps;
            s_ctx[wl0 + i] = ctx[w];
            ev[i] = evm[w];
        }
        mine += (uint32_t)__popcll(ev[i]);
    }
    uint32_t total;
    const uint32_t off = block_exclusive<AddU32>(mine, s_scan, total);
    // own_prefix: tile_base still holds the tiles' entry counts; the last tile publishes the total and the carry
    const uint32_t gbase = own_prefix ? tile_prefix<AddU32, SCAN_BLOCK>(tile_base, blockIdx.x, s_scan) : tile_base[blockIdx.x];
    if (own_prefix && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        *total_out = gbase + total;
        EdgeCarryEpilogue{A, last2_total, carry_out}(gbase + total);
    }
    for (uint32_t rbase = 0; rbase < total; rbase += EW_CAP) {
        uint32_t k = off - rbase;   // wraps below the round: the unsigned compare drops those
#pragma unroll
        for (int i = 0; i < EW_ITEMS; i++) {
            uint64_t e = ev[i];
            while (e) {
                if (k < (uint32_t)EW_CAP) s_ev[k] = (uint16_t)(((wl0 + i) << 6) | (__ffsll((long long)e) - 1));
                e &= e - 1;
                k++;
            }
        }
        __syncthreads();
        const uint32_t cnt = min((uint32_t)EW_CAP, total - rbase);
        for (uint32_t j = threadIdx.x; j < cnt; j += SCAN_BLOCK) {
            const uint32_t code = s_ev[j];
            const int wl = (int)(code >> 6), b = (int)(code & 63u);
            const int32_t w0 = (int32_t)((wt + wl) * 64);
            int v, d, t;
            // val at an earlier change position: from the staged planes when it lies in this tile (nearly always)
            auto val = [&](int32_t q) {
                const long long ql = (long long)(q >> 6) - (long long)wt;
                if (ql >= 0) {
                    const int sh = q & 63;
                    if ((s_ng[ql] >> sh) & 1ull) return -1;
                    return (int)((s_ps[ql] >> sh) & 1ull);
                }
                return A.val_at(q);
            };
            event_entry(A, w0, b, s_ng[wl], s_ps[wl], s_m[wl], s_ctx[wl], v, d, t, val);
            const uint32_t g = gbase + rbase + j;
            if (g < cap) {
                nfc_edge o;
                o.idx = A.g0 + (uint64_t)(w0 + b);
                o.d = d;
                o.v = (int8_t)v;
                o.t = (int8_t)t;
                o.pad = 0;
                edges[g] = o;
                ecode[g] = edge_code(v, d, t, A.nd);
            }
        }
        __syncthreads();
    }
}

}  // namespace nfc
