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
//
// The number of entries before a position is a closed form too: the val changes before it, plus
// floor((length - 1) / max_len) time-outs per finished run, plus those of the run in progress.  So ONE
// scan (EdgeAgg: first change, last two changes, entries of the runs that begin and end inside the span)
// gives a tile both what the old two scans did -- the change positions before it and where its entries go.
// Two launches: a reduce over the planes (16 bytes per tile), and the writer, which re-reads its tile's
// planes, folds its predecessors' aggregates itself and needs no per-word arrays from global memory.
//
// An entry is stored as its sample position (u32, batch-local) and its 16-bit code (LUT row | route):
// 6 bytes instead of the 16 of nfc_edge.  (v, d, t) decode from the code, the index is g0 + position;
// nfc_read_edges builds the records on the host from these.
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

struct MinI32 {
    using T = int32_t;
    static __device__ __forceinline__ T identity() { return INT32_MAX; }
    static __device__ __forceinline__ T op(T a, T b) { return a < b ? a : b; }
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
    // How the words are cut into tiles: a SUPER is sw consecutive words (EW_WORDS * EW_SUPER as k_edge_reduce leaves the aggregates),
    // cut into tps tiles of EW_WORDS words, the last of which may be short.  Tile b is tile b % tps of super b / tps.
    uint32_t sw, tps;

    __device__ __forceinline__ int mod_mx(int x) const {   // x mod max_len for 0 <= x < 2^31
        const uint32_t q = __umulhi((uint32_t)x, mx_magic);   // floor(x / mx) or one less
        const uint32_t r = (uint32_t)x - q * (uint32_t)mx;
        return (int)(r >= (uint32_t)mx ? r - (uint32_t)mx : r);
    }

    __device__ __forceinline__ uint32_t div_mx(int x) const {   // floor(x / max_len) for 0 <= x < 2^31
        const uint32_t q = __umulhi((uint32_t)x, mx_magic);   // the quotient or one less
        const uint32_t r = (uint32_t)x - q * (uint32_t)mx;
        return q + (r >= (uint32_t)mx ? 1u : 0u);
    }
    // time-outs of a run whose first sample is s, before position e: samples s + k max_len (k >= 1) below e
    __device__ __forceinline__ uint32_t timeouts_between(int32_t s, int32_t e) const {
        const int x = e - s - 1;
        return x > 0 ? div_mx(x) : 0u;
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
        return change_mask_of(w, ng, ps, pn, pp);
    }
    // the same from planes already loaded: pn / pp = the LOW / HIGH bit of the sample before the word
    __device__ __forceinline__ uint64_t change_mask_of(size_t w, uint64_t ng, uint64_t ps, uint64_t pn, uint64_t pp) const {
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

// ---- the stage's scan ------------------------------------------------------------------------------------------
// A span of words: where its first val change is, its last two, and the entries it is sure of whatever surrounds it --
// its val changes and the time-outs of the runs that begin AND end in it.  The time-outs of the run that enters the span
// and of the one that leaves it depend on the neighbours: the operator adds them when two spans meet (the run between
// a's last change and b's first), entries_before() adds the two at the ends.
struct EdgeAgg {
    int32_t first;   // POS_NONE: no change in the span
    Last2 l;
    uint32_t sum;
};
struct EdgeAggOp {
    using T = EdgeAgg;
    int32_t mx;
    uint32_t mx_magic;
    __host__ __device__ __forceinline__ T identity() const { return T{POS_NONE, Last2{POS_NONE, POS_NONE}, 0u}; }
    __device__ __forceinline__ T operator()(const T &a, const T &b) const {   // (selects, no branches: it runs inside scans)
        const bool ae = a.first == POS_NONE, be = b.first == POS_NONE;
        const int x = (int)((uint32_t)b.first - (uint32_t)a.l.s1 - 1u);   // samples between a's last change and b's first
        const uint32_t q = __umulhi((uint32_t)x, mx_magic);
        const uint32_t r = (uint32_t)x - q * (uint32_t)mx;
        const uint32_t tmo = (ae || be || x <= 0) ? 0u : q + (r >= (uint32_t)mx ? 1u : 0u);
        T o;
        o.first = ae ? b.first : a.first;
        o.l.s1 = be ? a.l.s1 : b.l.s1;
        o.l.s2 = be ? a.l.s2 : (b.l.s2 != POS_NONE ? b.l.s2 : a.l.s1);
        o.sum = a.sum + b.sum + tmo;
        return o;
    }
};
// one word's aggregate from its change mask (w0: its first sample)
// (a run inside the word can only time out between the word's first and last change, and only if they lie more than max_len apart)
__device__ __forceinline__ bool word_may_time_out(const EdgeArgs &A, uint64_t m) {
    return m && (63 - __clzll((long long)m)) - (__ffsll((long long)m) - 1) > A.mx;
}
// inner: some lane of the wave holds a word in which a run may time out (word_may_time_out) -- the count is skipped wave-wide
// otherwise: inside frames the changes lie closer than max_len, in the gaps between them a word has none
__device__ __forceinline__ EdgeAgg word_agg(const EdgeArgs &A, int32_t w0, uint64_t m, bool inner = true) {
    if (!m) return EdgeAgg{POS_NONE, Last2{POS_NONE, POS_NONE}, 0u};
    const int b0 = __ffsll((long long)m) - 1, b1 = 63 - __clzll((long long)m);
    const uint64_t m2 = m & ~(1ull << b1);
    uint32_t sum = (uint32_t)__popcll(m);
    if (!inner) {
        // (uniform: no word of this wave has two changes more than max_len apart)
    } else if (A.mx >= 32) {
        if (A.mx < 63) {
            // a run inside the word can time out at most once (2 max_len > 63): count the stretches of max_len or more
            // samples without a change between the word's first and last change.  x marks the samples that begin
            // max_len change-free samples (and-ing shifted copies, doubling the proven length); a stretch of r such
            // samples (max_len <= r < 2 max_len) leaves r - max_len + 1 marks in a row.
            uint64_t x = ~m & ((1ull << b1) - 1ull) & ~((2ull << b0) - 1ull);
            for (int have = 1; have < A.mx;) {
                const int sh = min(have, A.mx - have);
                x &= x >> sh;
                have += sh;
            }
            sum += (uint32_t)__popcll(x & ~(x >> 1));
        }
    } else if (b1 - b0 > A.mx) {   // short max_len: walk the runs
        uint64_t rest = m & (m - 1);
        int prev = b0;
        while (rest) {
            const int b = __ffsll((long long)rest) - 1;
            rest &= rest - 1;
            sum += A.timeouts_between(prev, b);
            prev = b;
        }
    }
    return EdgeAgg{w0 + b0, Last2{w0 + b1, m2 ? w0 + (63 - __clzll((long long)m2)) : POS_NONE}, sum};
}
// Entries before position T (the first sample of a word), from the aggregate of all the words before it: what the
// aggregate is sure of, the time-outs of the run carried into the batch (up to the first change), and those of the run
// in progress at T.  (A carried run "starts" where its carried _dur puts it: skip - dur_in.)
__device__ __forceinline__ uint32_t entries_before(const EdgeArgs &A, const EdgeAgg &pre, int32_t T) {
    const int32_t s0 = (int32_t)A.skip - A.dur_in;
    uint32_t c = pre.sum;
    if (pre.first != POS_NONE) c += A.timeouts_between(s0, pre.first);
    return c + A.timeouts_between(pre.l.s1 != POS_NONE ? pre.l.s1 : s0, T);
}

// A thread's consecutive words of the planes and their change masks; 16-byte loads when the words are all there.
template <int ITEMS>
__device__ __forceinline__ int load_words(const EdgeArgs &A, size_t w, size_t nwords, uint64_t (&ng)[ITEMS], uint64_t (&ps)[ITEMS],
                                          uint64_t (&m)[ITEMS]) {   // returns val of the sample before the first word
    static_assert(ITEMS % 2 == 0, "pairs of words");
    if (w + ITEMS <= nwords) {
#pragma unroll
        for (int i = 0; i < ITEMS; i += 2) {
            const uint4 a = *(const uint4 *)(A.neg + w + i), b = *(const uint4 *)(A.pos + w + i);
            ng[i] = (uint64_t)a.x | ((uint64_t)a.y << 32);
            ng[i + 1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
            ps[i] = (uint64_t)b.x | ((uint64_t)b.y << 32);
            ps[i + 1] = (uint64_t)b.z | ((uint64_t)b.w << 32);
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            ng[i] = (w + i < nwords) ? A.neg[w + i] : 0ull;
            ps[i] = (w + i < nwords) ? A.pos[w + i] : 0ull;
        }
    }
    uint64_t pn, pp;
    if (w == 0) {
        pn = A.last_bit_in == -1 ? 1ull : 0ull;
        pp = A.last_bit_in == 1 ? 1ull : 0ull;
    } else if (w <= nwords) {
        pn = A.neg[w - 1] >> 63;
        pp = A.pos[w - 1] >> 63;
    } else {
        pn = pp = 0ull;
    }
    const int val_before = pn ? -1 : (int)pp;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        m[i] = (w + i < nwords) ? A.change_mask_of(w + i, ng[i], ps[i], pn, pp) : 0ull;
        pn = ng[i] >> 63;
        pp = ps[i] >> 63;
    }
    return val_before;
}

#ifndef NFC_EW_ITEMS
#define NFC_EW_ITEMS 2
#endif
constexpr int EW_ITEMS = NFC_EW_ITEMS;               // words per thread (2; 4 with twice the staged entries measured 52 vs 47 us)
constexpr int EW_WORDS = SCAN_BLOCK * EW_ITEMS;      // words per tile, in both launches of the stage
#ifndef NFC_EW_CAP
#define NFC_EW_CAP 3072
#endif
inline size_t edge_num_tiles(size_t nwords) { return (nwords + EW_WORDS - 1) / EW_WORDS; }

// ---- launch 1: one aggregate per tile, and one per group of EW_SUPER tiles ----
// The reduce pass as launched: a workgroup takes EW_SUPER consecutive tiles -- one per WAVE (a thread eight consecutive words, so
// a wave's 512 words are exactly a writer tile) -- and leaves every tile's aggregate AND the aggregate of the four together.
// Short batches only use the tiles' (the writer's workgroups fold their predecessors themselves); a long batch scans the
// SUPER-aggregates in its single-workgroup prefix launch -- a quarter of the items: 7 630 instead of 30 517 at 1e9 samples, one round
// of that workgroup instead of four -- and the writer adds at most three sibling tiles to its group's prefix.
constexpr int EW_SUPER = SCAN_WAVES;               // tiles per super-aggregate
constexpr int ES_ITEMS = EW_WORDS / 64;            // words per thread here: a wave covers one tile
static_assert(ES_ITEMS % 2 == 0 && ES_ITEMS * 64 == EW_WORDS, "a wave of the reduce pass covers one writer tile");
inline size_t edge_num_supers(size_t nwords) { return (edge_num_tiles(nwords) + EW_SUPER - 1) / EW_SUPER; }
// (round 4: WPT waves per tile -- 512 threads a workgroup at WPT = 2, four words per lane; 4 and 1 measured the same within 1 us: the pass is the first read of the 25 MB of planes the threshold kernel just wrote, not its arithmetic.  With one wave per tile
// a lane folded eight words one after the other behind one round of loads, at three waves per SIMD: 12 us for 25 MB.)
#ifndef NFC_ER_WPT
#define NFC_ER_WPT 2
#endif
constexpr int ER_WPT = NFC_ER_WPT;                    // waves per tile
constexpr int ER_ITEMS = ES_ITEMS / ER_WPT;           // words per lane
constexpr int ER_BLOCK = 64 * ER_WPT * EW_SUPER;      // threads per workgroup: EW_SUPER tiles
static_assert(ER_ITEMS >= 2 && ER_ITEMS % 2 == 0 && ER_ITEMS * ER_WPT == ES_ITEMS, "pairs of words per lane");
__device__ __forceinline__ void edge_reduce_super(const EdgeArgs &A, size_t nwords, uint32_t super, EdgeAgg *partials, EdgeAgg *supers) {
    __shared__ EdgeAgg lds[EW_SUPER * ER_WPT];
    const EdgeAggOp op{A.mx, A.mx_magic};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // wave: ER_WPT consecutive ones per tile
    const size_t tile = (size_t)super * EW_SUPER + wave / ER_WPT;
    const size_t w = tile * EW_WORDS + ((size_t)(wave % ER_WPT) * 64 + lane) * ER_ITEMS;
    uint64_t ng[ER_ITEMS], ps[ER_ITEMS], m[ER_ITEMS];
    load_words<ER_ITEMS>(A, w, nwords, ng, ps, m);
    EdgeAgg agg = op.identity();
    bool may = false;
#pragma unroll
    for (int i = 0; i < ER_ITEMS; i++) may = may || word_may_time_out(A, m[i]);
    const bool inner = __any(may);
#pragma unroll
    for (int i = 0; i < ER_ITEMS; i++) agg = op(agg, word_agg(A, (int32_t)((w + i) * 64), m[i], inner));
    // The wave's aggregate without scanning the whole aggregate (round 5; the operator's division by max_len is then paid once per lane
    // instead of once per lane and scan step): the last two changes by a scan of selects, the first change by a minimum, and the entries
    // by a sum -- a lane adds the time-outs of the run that ENDS at its first change, which it knows from the changes before it in the wave
    // (none before it: the run began outside the wave, and whoever joins the waves adds its time-outs -- the operator, below).
    const Last2 l_inc = wave_inclusive<Last2Op>(agg.l);
    const Last2 l_exc = wave_shift_up1<Last2Op>(l_inc);
    const uint32_t cross = (agg.first != POS_NONE && l_exc.s1 != POS_NONE) ? A.timeouts_between(l_exc.s1, agg.first) : 0u;
    const uint32_t sum_inc = wave_inclusive<AddU32>(agg.sum + cross);
    const int32_t first_inc = wave_inclusive<MinI32>(agg.first != POS_NONE ? agg.first : INT32_MAX);
    if (lane == 63) lds[wave] = EdgeAgg{first_inc != INT32_MAX ? first_inc : POS_NONE, l_inc, sum_inc};
    __syncthreads();
    if (threadIdx.x < EW_SUPER) {   // a lane per tile folds its waves; lane 0 the tiles
        EdgeAgg t = lds[threadIdx.x * ER_WPT];
#pragma unroll
        for (int k = 1; k < ER_WPT; k++) t = op(t, lds[threadIdx.x * ER_WPT + k]);
        const size_t tl = (size_t)super * EW_SUPER + threadIdx.x;
        if (tl * EW_WORDS < nwords) partials[tl] = t;
        lds[threadIdx.x * ER_WPT] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        EdgeAgg t = lds[0];
#pragma unroll
        for (int k = 1; k < EW_SUPER; k++) t = op(t, lds[k * ER_WPT]);
        supers[super] = t;
    }
}
__global__ __launch_bounds__(ER_BLOCK) void k_edge_reduce(EdgeArgs A, size_t nwords, EdgeAgg *partials, EdgeAgg *supers) {
    edge_reduce_super(A, nwords, blockIdx.x, partials, supers);
}

// per edge, for the decoders: LUT row (v + 1) * nd + d in the low 14 bits, route in the top two
// (0 dropped, 1 Manchester / tag->reader, 2 Miller / reader->tag; background.py:30-35).  d <= max_len = nd - 1 on
// every path of event_entry, so the code holds the whole entry.
__device__ __forceinline__ uint16_t edge_code(int v, int d, int t, int nd) {
    const int dd = d < nd ? d : nd - 1;
    return (uint16_t)(((v + 1) * nd + dd) | ((t + 1) << 14));
}
// ... and back (host: nfc_read_edges)
inline void edge_decode(uint16_t code, int nd, int &v, int &d, int &t) {
    const int li = code & 0x3FFF;
    v = li / nd - 1;
    d = li % nd;
    t = (int)(code >> 14) - 1;
}

// carried (_last_bit, _dur, _current_state) after the batch, from the last two change positions of the whole batch
__device__ __forceinline__ void edge_carry_out(const EdgeArgs &A, Last2 total, EdgeCarry *carry) {
    if (A.skip >= A.n) return;  // nothing but fill samples: unchanged
    int lb, dur, st;
    A.state_before((int32_t)A.n, total, lb, dur, st);
    carry->last_bit = lb;
    carry->dur = dur;
    carry->state = st;
}
// long batches: the epilogue of the prefix launch over the tile aggregates publishes the totals and the carry
struct EdgeTotalEpilogue {
    EdgeArgs A;
    uint32_t *edges_total;
    Last2 *last2_total;
    EdgeCarry *carry;
    __device__ __forceinline__ void operator()(const EdgeAgg &tot) const {
        *edges_total = entries_before(A, tot, (int32_t)A.n);
        *last2_total = tot.l;
        edge_carry_out(A, tot.l, carry);
    }
};

// ---- tiles of a super (EdgeArgs.sw / tps) ----
struct TileSpan {
    size_t w0;        // first word
    uint32_t words;   // words of the tile (0: beyond the batch)
    uint32_t g, q;    // super, tile within it
};
__device__ __forceinline__ TileSpan tile_span(const EdgeArgs &A, size_t nwords, uint32_t b) {
    TileSpan t;
    t.g = b / A.tps;
    t.q = b % A.tps;
    t.w0 = (size_t)t.g * A.sw + (size_t)t.q * EW_WORDS;
    const size_t in_super = (size_t)t.q * EW_WORDS < A.sw ? (size_t)A.sw - (size_t)t.q * EW_WORDS : 0;
    const size_t in_batch = t.w0 < nwords ? nwords - t.w0 : 0;
    t.words = (uint32_t)min(min(in_super, in_batch), (size_t)EW_WORDS);
    return t;
}
inline size_t edge_num_tiles_of(size_t nwords, uint32_t sw, uint32_t tps) { return ((nwords + sw - 1) / sw) * tps; }
// ---- launch 2: the writer.  A workgroup owns EW_WORDS consecutive words; a thread walks its EW_ITEMS words' entries in
// stream order, which is the reference's own loop restricted to the samples that emit (transition_sink.py:84-99): between
// entries everything it carries has a closed form, so the walk keeps
//   lb      _last_bit: val of the run in progress
//   q       the sample where _dur last restarted: the run's first sample or its latest time-out (_dur = p - q)
//   timed   the run has timed out at least once
//   left    _current_state the previous run left (a val-0 run that has not timed out keeps it)
// and needs no division per entry.  Where a thread starts comes from ONE block scan of EdgeAgg on top of the tile's
// prefix: the two changes before its first word AND, by entries_before(), the offset of its first entry.  Entries go to
// LDS at their offsets and leave the workgroup as whole rows of positions and codes.
constexpr int EW_CAP = NFC_EW_CAP;   // entries staged per round (a tile of 512 words holds 2150 on the bench workloads, 32768 at most).
// 3072 entries of 2 + 2 bytes: 12 KB per workgroup -- six of them fit in the 96 KB of LDS a CU has left while the threshold
// kernel of the next batch runs on it (batches submitted ahead), which is when this kernel's occupancy matters most.
static_assert(EW_WORDS * 64 <= 65536, "tile-local positions are staged in 16 bits");
// (all of it integers: a per-lane bool that lives across the loop's back edge is kept as a lane MASK in scalar registers and costs
// three scalar instructions per update -- the walk's trip was 54 vector + 30 scalar instructions; round 4)
struct EdgeWalk {
    int lb;          // _last_bit
    int32_t q;       // where _dur last restarted
    int zst;         // _current_state inside a val-0 run: 0 once the run has timed out, else what the previous run left
    int32_t tq;      // the sample of the latest time-out if that was the latest entry (POS_NONE otherwise)
    int32_t cskip;   // A.skip while the run carried into the batch is still in progress (POS_NONE otherwise)
};
// (bid of ntiles: the writer may share a launch with another stage's workgroups -- host_context.h: k_certify_and_write)
__device__ __forceinline__ void write_edges_tile(const EdgeArgs &A, size_t nwords, const EdgeAgg *partials, const EdgeAgg *supers, uint32_t *epos,
                                                 uint16_t *ecode, uint32_t cap, bool own_prefix, uint32_t *total_out,
                                                 Last2 *last2_total, EdgeCarry *carry_out, const uint32_t bid, const uint32_t ntiles) {
    __shared__ uint32_t s_ent[EW_CAP + 1];   // tile-local sample position (a tile is EW_WORDS * 64 <= 65536 samples) | code << 16; slot EW_CAP: entries of another round
    TP_DECL();
    const TileSpan ts = tile_span(A, nwords, bid);
    const size_t wt = ts.w0;                    // first word of the tile
    const size_t wend = wt + ts.words;          // ... and where its words end (the tiles of a super's tail are short, or empty)
    const size_t w_first = wt + (size_t)threadIdx.x * EW_ITEMS;
    // own_prefix: partials still holds the tiles' aggregates and the workgroup folds its predecessors' itself
    // (first, while few registers are live); the last tile then publishes the totals and the carry
    // (the tile's words are requested first: their latency passes under the prefix)
    const EdgeAggOp op{A.mx, A.mx_magic};
    uint64_t ng[EW_ITEMS], ps[EW_ITEMS], m[EW_ITEMS];
    const int val_before = load_words<EW_ITEMS>(A, w_first, wend, ng, ps, m);
    EdgeAgg pre;
    const uint32_t g = ts.g, q = ts.q;
    if (own_prefix) {
        // The tile's prefix: the super-aggregates of the groups before its own (one per super, left by the reduce pass) and the sibling tiles before it in its group -- folded by ONE wave (round 4; all four waves folded the
        // tiles' own aggregates before and joined in a block scan: this kernel is bound by the number of vector instructions it
        // issues, and three of the four waves' shares of them are gone; the other waves' words are in flight meanwhile)
        __shared__ EdgeAgg s_pre;
        if (threadIdx.x < 64) {
            const uint32_t per = (g + 63u) / 64u;
            const uint32_t lo = min(g, (uint32_t)threadIdx.x * per), hi = min(g, lo + per);
            EdgeAgg acc = op.identity();
            for (uint32_t i = lo; i < hi; i += 4) {   // four loads in flight per lane
                EdgeAgg v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = (i + k < hi) ? supers[i + k] : op.identity();
#pragma unroll
                for (int k = 0; k < 4; k++) acc = op(acc, v[k]);
            }
            EdgeAgg sib[EW_SUPER - 1];
#pragma unroll
            for (int k = 0; k < EW_SUPER - 1; k++) sib[k] = (uint32_t)k < q ? partials[(size_t)g * A.tps + k] : op.identity();
            EdgeAgg tot = wave_inclusive_with(op, acc);   // (lane 63: all groups before this tile's)
#pragma unroll
            for (int k = 0; k < EW_SUPER - 1; k++) tot = op(tot, sib[k]);
            if (threadIdx.x == 63) s_pre = tot;
        }
        __syncthreads();
        pre = s_pre;
    } else {
        // (the prefix launch scanned the super-aggregates: the group's prefix, then the sibling tiles before this one)
        EdgeAgg sib[EW_SUPER - 1];
#pragma unroll
        for (int k = 0; k < EW_SUPER - 1; k++) sib[k] = (uint32_t)k < q ? partials[(size_t)g * A.tps + k] : op.identity();
        pre = supers[g];
#pragma unroll
        for (int k = 0; k < EW_SUPER - 1; k++) pre = op(pre, sib[k]);
    }
    const int32_t tile_p0 = (int32_t)min(wt * 64, (size_t)A.n);   // (an empty tile beyond the batch: nothing of it may be counted)
    const uint32_t gbase = entries_before(A, pre, tile_p0);
    TP_MARK();   // 1: words + tile prefix
    EdgeAgg agg = op.identity();
    bool may = false;
#pragma unroll
    for (int i = 0; i < EW_ITEMS; i++) may = may || word_may_time_out(A, m[i]);
    const bool inner = __any(may);
#pragma unroll
    for (int i = 0; i < EW_ITEMS; i++) agg = op(agg, word_agg(A, (int32_t)((w_first + i) * 64), m[i], inner));
    // Where a thread's entries go, in TWO cheap block scans instead of one of the whole aggregate (round 5: the aggregate's operator is
    // 25 vector instructions, two of them quarter-rate multiplies for the division by max_len, and a block scan applies it 16 times per
    // thread -- 420 of the writer's 1 390 vector instructions per wave).  Entries before a position are its val changes, plus per change
    // the time-outs of the run that ENDS there, plus those of the run in progress (entries_before); the time-outs of the run that ends at
    // a thread's first change need the change before it, which is all the first scan carries (Last2: selects only); the second scan adds
    // counts.  Same numbers as entries_before(op(pre, exclusive aggregate), T): the gap between two spans is counted where it ends.
    const int32_t s0 = (int32_t)A.skip - A.dur_in;   // where the run carried into the batch "starts"
    __shared__ Last2 s_l2[SCAN_WAVES];
    __shared__ uint32_t s_cnt[SCAN_WAVES];
    Last2 tile_l;
    const Last2 before_l = Last2Op::op(pre.l, block_exclusive<Last2Op>(agg.l, s_l2, tile_l));
    const Last2 all_l = Last2Op::op(pre.l, tile_l);
    const uint32_t cross = (agg.first != POS_NONE) ? A.timeouts_between(before_l.s1 != POS_NONE ? before_l.s1 : s0, agg.first) : 0u;
    uint32_t tile_cnt;
    const uint32_t excl_cnt = block_exclusive<AddU32>(agg.sum + cross, s_cnt, tile_cnt);
    const int32_t tile_end = (int32_t)min(wend * 64, (size_t)A.n), T = (int32_t)min(w_first * 64, (size_t)tile_end);
    // (entries before the tile that do not depend on where the tile begins: pre.sum and the carried run's time-outs up to the batch's first change)
    const uint32_t run_k = A.timeouts_between(before_l.s1 != POS_NONE ? before_l.s1 : s0, T);   // time-outs so far of the run in progress at T
    const uint32_t in_progress0 = A.timeouts_between(pre.l.s1 != POS_NONE ? pre.l.s1 : s0, tile_p0);   // ... at the tile's first sample (gbase holds them)
    const uint32_t off = excl_cnt + run_k - in_progress0;
    const uint32_t total = tile_cnt + A.timeouts_between(all_l.s1 != POS_NONE ? all_l.s1 : s0, tile_end) - in_progress0;
    if (own_prefix && bid == ntiles - 1 && threadIdx.x == 0) {
        *total_out = gbase + total;
        *last2_total = all_l;
        edge_carry_out(A, all_l, carry_out);
    }
    // where the walk stands at the thread's first sample
    EdgeWalk W0;
    {
        const Last2 c = before_l;
        const bool carried = c.s1 == POS_NONE;
        const int32_t s = carried ? s0 : c.s1;
        const int32_t k = (int32_t)run_k;   // time-outs of the run so far
        W0.q = s + k * A.mx;
        const bool timed = k > 0;
        int left = A.state_in;
        if (carried) {
            W0.lb = A.last_bit_in;
        } else {
            // (no change between c.s1 and T: val at T - 1 is the run's; T - 1 is the last sample of the word before the
            // thread's, or -- past the end of the batch -- nothing the walk will use)
            W0.lb = val_before;
            if (W0.lb == 0 && !timed) {   // a val-0 run that has not timed out keeps what the previous run left
                int lb, dur;
                A.state_before(T, c, lb, dur, left);
            }
        }
        W0.zst = timed ? 0 : left;
        W0.tq = timed ? W0.q : POS_NONE;
        W0.cskip = carried ? (int32_t)A.skip : POS_NONE;
    }
    TP_MARK();   // 2: block scan + where the walk stands
    for (uint32_t rbase = 0; rbase < total; rbase += EW_CAP) {   // (a second round walks again: only tiles denser than EW_CAP)
        EdgeWalk W = W0;
        uint32_t k4 = (off - rbase) * 4u;   // (byte offset of the next entry's slot) wraps below the round: the unsigned clamp sends those to the spare slot
#pragma unroll
        for (int i = 0; i < EW_ITEMS; i++) {
            const int32_t w0 = (int32_t)((w_first + i) * 64);
            const int32_t end = min(w0 + 64, tile_end);   // (words beyond the tile's: nothing to find, and the run's next time-out is not before tile_end)
            const uint32_t ng_lo = (uint32_t)ng[i], ng_hi = (uint32_t)(ng[i] >> 32), ps_lo = (uint32_t)ps[i], ps_hi = (uint32_t)(ps[i] >> 32);
            uint64_t mm = m[i];
            while (true) {
                const int b = mm ? __ffsll((long long)mm) - 1 : 64;
                const int32_t c = min(w0 + b, end);   // the next change, or the end of the word
                const int32_t nt = W.q + A.mx;        // the run's next time-out
                const bool is_to = nt < c;            // _dur exceeds max_len first (transition_sink.py:95-99)
                if (!is_to && !mm) break;
                const int32_t p = is_to ? nt : c;
                // _current_state before sample p
                const int run_st = (W.lb == -1) ? 2 : 1;                                  // inside a LOW / HIGH run ...
                int prev_st = (W.lb != 0) ? ((W.tq == p - 1) ? 0 : run_st) : W.zst;       // ... unless p - 1 timed out; inside a val-0 run
                prev_st = (p == W.cskip) ? A.state_in : prev_st;                          // the first stable sample: the carried value itself
                // a time-out keeps val and takes the run's state; a change (transition_sink.py:86-92) takes the new val's
                // (val at bit b: the half of the word first -- 64-bit shifts by a register run at a quarter of the rate)
                const uint32_t nh = b < 32 ? ng_lo : ng_hi, ph = b < 32 ? ps_lo : ps_hi;
                const int vb = __builtin_amdgcn_ubfe(nh, (uint32_t)b & 31u, 1u) ? -1 : (int)__builtin_amdgcn_ubfe(ph, (uint32_t)b & 31u, 1u);
                const int val = is_to ? W.lb : vb;
                const int st = is_to ? ((W.lb != 0) ? run_st : W.zst) : ((val == -1) ? 2 : ((val == 1) ? 1 : prev_st));
                {
                    const int v = st == 2 ? W.lb + 1 : W.lb, d = (is_to || prev_st == 0) ? A.mx : p - W.q, t = st - 1;
                    const int dd = d < A.nd ? d : A.nd - 1;   // (edge_code, with its product of two small numbers at full rate)
                    uint32_t row;   // (v + 1) * nd + dd: v_mad_u32_u24, one full-rate instruction (the compiler picks the 32-bit product, a quarter of the rate)
                    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"((uint32_t)(v + 1)), "s"((uint32_t)A.nd), "v"((uint32_t)dd));
                    const uint32_t code = row | ((uint32_t)(t + 1) << 14);
                    *(uint32_t *)((char *)s_ent + min(k4, (uint32_t)EW_CAP * 4u)) = (uint32_t)(p - tile_p0) | (code << 16);
                    k4 += 4u;
                }
                W.zst = is_to ? 0 : prev_st;
                W.tq = is_to ? p : POS_NONE;
                W.cskip = is_to ? W.cskip : POS_NONE;
                W.lb = val;
                W.q = p;
                mm &= mm - (is_to ? 0ull : 1ull);
            }
        }
        TP_MARK();   // 3: the walk
        __syncthreads();
        TP_MARK();   // 4: waiting for the slowest wave
        const uint32_t cnt = min((uint32_t)EW_CAP, total - rbase);
        for (uint32_t j = threadIdx.x; j < cnt; j += SCAN_BLOCK) {
            const uint32_t g = gbase + rbase + j;
            if (g < cap) {
                const uint32_t e = s_ent[j];
                epos[g] = (uint32_t)tile_p0 + (e & 0xFFFFu);
                ecode[g] = (uint16_t)(e >> 16);
            }
        }
        __syncthreads();
    }
    TP_DONE(0);   // 5: the stores
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_write_edges(EdgeArgs A, size_t nwords, const EdgeAgg *partials, const EdgeAgg *supers, uint32_t *epos,
                                                           uint16_t *ecode, uint32_t cap, bool own_prefix, uint32_t *total_out,
                                                           Last2 *last2_total, EdgeCarry *carry_out) {
    write_edges_tile(A, nwords, partials, supers, epos, ecode, cap, own_prefix, total_out, last2_total, carry_out, blockIdx.x, gridDim.x);
}

}  // namespace nfc
