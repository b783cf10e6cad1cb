// tail.hip.h -- everything behind the threshold stage in ONE persistent launch (round 6): a tile end to end per workgroup.
//
// Until round 5 the tail of a batch was five launches -- k_edge_reduce, k_certify_and_write, k_dec_spec, k_concat, k_pkt_finish --
// that read the classification planes twice, wrote 6 bytes per entry and read the codes back, folded tile prefixes by brute force
// and carried the decoders across tiles on a speculated run-in that a later launch had to check.  Here a workgroup takes a TILE of
// TL_WORDS plane words (a ticket: tiles are handed out in stream order), and does, without leaving the CU:
//   E1  its words' change masks and aggregates (edges.hip.h: word_agg), the tile's EdgeAgg;
//   --  publishes that aggregate, gets the aggregate of everything before the tile by DECOUPLED LOOK-BACK over the tiles' status
//       words (below), publishes its inclusive prefix;
//   E2  the walk (transition_sink.py:84-99 restricted to the samples that emit): entries into LDS, 6 bytes each to HBM;
//   D   Modified-Miller / Manchester decode of those entries FROM LDS (miller.py:153-197, manchester.py:30-61 as table walks):
//       composes its edges' state maps, publishes the tile's map, looks back for its incoming decoder states -- a frame gap makes
//       a tile's map constant, so the look-back ends at the nearest tile that saw one: no speculation, nothing to verify --, walks;
//   F   framing (packets.py:67-79): the tile's FrameAgg, a third look-back for bit / packet-end offsets and the started state,
//       packet bits through LDS to their place in the stream, packet ends.
// The last tile publishes the batch's totals and carries.  The certification of the threshold stage (nothing here reads what it
// leaves) rides along as extra tickets, as it did in k_certify_and_write.
//
// Look-back without fences.  A device-scope release on this machine writes an XCD's L2 back (measured in round 5: 811 workgroups
// each asking for one made a 7 us launch take 34).  So nothing here is published by "store, fence, flag": every status word is ONE
// 64-bit relaxed agent-scope atomic store that carries its own validity -- payload in the low half, tag = launch epoch | status in
// the high half -- and is polled with relaxed agent-scope loads.  Words of one slot may be seen in any order; a reader takes a slot
// when every word of it carries this launch's epoch.  Status 1: the tile's own aggregate; status 2: its inclusive prefix is in the
// prefix array (whose words validate themselves the same way).  No buffer is cleared between launches: the epoch does that.
// Forward progress: a tile only ever waits for tiles with smaller tickets, which are resident or done.  A poll that does not
// succeed within TL_SPIN_LIMIT tries gives up, flags the batch (the host reports a device error) and goes on with garbage that
// every store bounds-checks: the launch cannot hang.
//
// A tile whose entries do not fit the LDS staging (TL_CAP) flags the batch; the host repeats the tail with shorter tiles (TL_WORDS
// halves down to 32: 2 048 samples cannot hold more than TL_CAP entries... they can hold 2 048).
//
// MEASURED AND NOT ADOPTED (round 6; DESIGN.md section 6c has the numbers).  Bit-exact on every test and on the 1e8-sample
// workloads, and 1.8 times as long as the five launches it replaces (152 us against 83.5 on configs[1], same-call A/B): three
// look-backs per tile each cost round trips to memory that the L2s cannot serve (status words must bypass them to be seen across
// XCDs), and a tile's phases are a chain of barriers that five resident workgroups per CU do not hide.  The product build does not
// contain this file; the test build does (NFC_TAIL=1 selects it) and tests/test_gpu_parity.py keeps it exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode.hip.h"
#include "edges.hip.h"
#include "scan.hip.h"
#include "threshold.hip.h"

namespace nfc {

constexpr int TL_BLOCK = 256;
constexpr int TL_ITEMS = 2;                       // plane words per thread
constexpr int TL_WORDS_MAX = TL_BLOCK * TL_ITEMS; // words per tile at most (32 768 samples)
constexpr int TL_PER_MAX = 15;                    // entries per thread in the decode (odd strides keep the LDS banks apart; 16 out-bytes hold them)
constexpr int TL_CAP = TL_BLOCK * TL_PER_MAX;     // entries staged per tile: 3 840 (a tile of the bench workloads holds 2 200)
constexpr int TL_BITW = TL_CAP * 2 / 32 + 4;      // bit words of a tile per packet type (two symbols per entry at most, + the tile's phase)
constexpr uint32_t TL_SPIN_LIMIT = 400000;        // polls before a look-back gives up (each is a round trip to memory: ~0.5 s)
constexpr uint32_t TLV_DENSE = 2u, TLV_TIMEOUT = 4u;   // verdict bits (TOT_SPEC; 1 is the speculative decode's)

// ---- status words ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_store(uint64_t *p, uint32_t payload, uint32_t tag) {
    __hip_atomic_store(p, (uint64_t)payload | ((uint64_t)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t st_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool st_valid(uint64_t w, uint32_t epoch) { return (uint32_t)(w >> 34) == epoch; }
__device__ __forceinline__ uint32_t st_status(uint64_t w) { return (uint32_t)(w >> 32) & 3u; }

// per tile: E_A 2 words, E_P 4, M 2, F_A 3, F_P 8
constexpr int TLS_EA = 2, TLS_EP = 4, TLS_M = 2, TLS_FA = 3, TLS_FP = 8;
inline size_t tail_table_bytes(int nd) { return (size_t)(4 * nd) * 32 + (size_t)(4 * nd + 1) * 16; }
inline size_t tail_status_words(size_t ntiles) { return ntiles * (size_t)(TLS_EA + TLS_EP + TLS_M + TLS_FA + TLS_FP) + 16; }

struct TailArgs {
    EdgeArgs E;
    size_t nwords;
    uint32_t tw;          // words per tile (<= TL_WORDS_MAX)
    uint32_t ntiles;
    uint32_t *epos;
    uint16_t *ecode;
    uint32_t cap;         // entries the buffers hold
    uint8_t *outw;        // one out-byte per entry (what the decoders emitted: k_symbols_write reads it)
    int32_t decode;       // 0: the edge stage only (the three-launch decode follows)
    DecTables T;
    uint32_t state0;      // Miller class | Manchester state << 4
    FrameOut P;
    DecCarryEpilogue epi;
    uint32_t q_rep[2];
    uint32_t *edges_total;
    Last2 *last2_total;
    EdgeCarry *carry_out;
    FrameAgg *frame_total;
    uint32_t *verdict;
    uint32_t *peak_out;   // entries of the densest tile (the host sizes the next batch's tiles by it)
    uint64_t *st;         // status words (tail_status_words)
    uint32_t *ticket;
    uint32_t ticket_base, epoch;
    ZeroJob Znext;        // the NEXT batch's packed bit arrays (its tiles or into them: nothing else may clear them in time)
    CertLaunch C;         // blocks == 0: no certification rides along
};

// the F chain carries two flags beside the framing aggregate: some tile so far was too dense / gave up waiting
struct FrameLb {
    FrameAgg f;
    uint32_t flags;
};
struct FrameLbOp {
    using T = FrameLb;
    __device__ __forceinline__ T identity() const { return T{FrameAggOp::identity(), 0u}; }
    __device__ __forceinline__ T operator()(const T &a, const T &b) const { return T{FrameAggOp::op(a.f, b.f), a.flags | b.flags}; }
};

__device__ __forceinline__ uint32_t nib_pack(uint32_t lo, uint32_t hi) {   // eight bytes (values < 16) to eight nibbles
    auto half = [](uint32_t m) { return (m & 0xFu) | ((m >> 4) & 0xF0u) | ((m >> 8) & 0xF00u) | ((m >> 12) & 0xF000u); };
    return half(lo) | (half(hi) << 16);
}
__device__ __forceinline__ void nib_unpack(uint32_t n, uint32_t &lo, uint32_t &hi) {
    auto half = [](uint32_t h) { return (h & 0xFu) | ((h & 0xF0u) << 4) | ((h & 0xF00u) << 8) | ((h & 0xF000u) << 12); };
    lo = half(n & 0xFFFFu);
    hi = half(n >> 16);
}

// ---- the look-back ------------------------------------------------------------------------------------------------------------
// All 256 threads poll the 256 tiles before `hi` (lane order = stream order, wave 3 nearest); a wave folds from its nearest prefix
// on, the waves' folds meet in LDS.  Pol: T, op, base (the prefix before tile 0), poll(tile, val, isP) -> ready, load_prefix(tile).
#ifdef NFC_TAIL_PROF
__device__ uint32_t g_lb_tries;   // (polls that were repeated)
#endif
struct LbShared {
    uint32_t flag[4];   // per wave: bit 0 every lane it needs is ready, bit 1 it holds a prefix
    uint32_t give_up;
};
template <class T>
__device__ __forceinline__ T wave_bcast_lane63(const T &v) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    constexpr int W = sizeof(T) / 4;
    uint32_t x[W];
    __builtin_memcpy(x, &v, sizeof(T));
#pragma unroll
    for (int i = 0; i < W; i++) x[i] = (uint32_t)__builtin_amdgcn_readlane((int)x[i], 63);
    T o;
    __builtin_memcpy(&o, x, sizeof(T));
    return o;
}
// ONE wave polls, 64 tiles at a time (lane order = stream order, lane 63 nearest), and folds from the nearest prefix on; the other
// waves wait at the barrier that hands the result over.  (A first form polled 256 tiles per step with every thread: a million
// uncached loads per launch aimed at the same few hundred lines made every round trip four times as long.)
template <class Pol>
__device__ __forceinline__ typename Pol::T tail_lookback(const Pol &pol, int j, typename Pol::T *s_res, LbShared *, uint32_t *timed_out /* LDS */) {
    using T = typename Pol::T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        T acc = pol.op.identity();
        int hi = j;
        uint32_t tries = *timed_out ? TL_SPIN_LIMIT : 0u;   // (a workgroup that gave up once does not wait again: the batch is flagged)
        for (;;) {
            const int t = hi - 64 + lane;
            T val = pol.op.identity();
            bool ready = true, isP = false;
            if (t == -1) {
                val = pol.base;
                isP = true;
            } else if (t >= 0) {
                ready = pol.poll(t, val, isP);
            }
            const uint64_t rmask = __ballot(ready), pmask = __ballot(ready && isP);
            const int lp = pmask ? 63 - __clzll((long long)pmask) : -1;
            const uint64_t need = lp >= 0 ? (~0ull << lp) : ~0ull;
            if ((rmask & need) == need) {
                if (!Pol::in_slot && lane == lp && t >= 0) val = pol.load_prefix(t, timed_out);   // (status 2: the value is in the prefix array)
                T x = (lane < lp || !ready) ? pol.op.identity() : val;
                x = wave_bcast_lane63(wave_inclusive_with(pol.op, x));
                acc = pol.op(x, acc);
                if (lp >= 0) break;
                hi -= 64;
                continue;
            }
#ifdef NFC_TAIL_PROF
            if (lane == 0) atomicAdd(&g_lb_tries, 1u);
#endif
            if (++tries > TL_SPIN_LIMIT) {
                if (lane == 0) *timed_out = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (lane == 0) *s_res = acc;
    }
    __syncthreads();
    return *s_res;
}

struct LbEdge {
    using T = EdgeAgg;
    static constexpr bool in_slot = false;
    EdgeAggOp op;
    T base;
    const uint64_t *ea, *ep;
    uint32_t epoch, tw;
    __device__ __forceinline__ bool poll(int t, T &val, bool &isP) const {
        const uint64_t w0 = st_load(ea + (size_t)t * TLS_EA), w1 = st_load(ea + (size_t)t * TLS_EA + 1);
        if (!st_valid(w0, epoch) || !st_valid(w1, epoch)) return false;
        isP = st_status(w0) == 2u || st_status(w1) == 2u;
        const int32_t p0 = (int32_t)((uint32_t)t * tw * 64u);
        const uint32_t a = (uint32_t)w0, b = (uint32_t)w1;
        auto pos = [&](uint32_t r) { return r == 0xFFFFu ? POS_NONE : p0 + (int32_t)r; };
        val = T{pos(a & 0xFFFFu), Last2{pos(a >> 16), pos(b & 0xFFFFu)}, b >> 16};
        return true;
    }
    __device__ __forceinline__ T load_prefix(int t, uint32_t *timed_out) const {
        const uint64_t *p = ep + (size_t)t * TLS_EP;
        for (uint32_t k = 0;; k++) {
            const uint64_t w0 = st_load(p), w1 = st_load(p + 1), w2 = st_load(p + 2), w3 = st_load(p + 3);
            if (st_valid(w0, epoch) && st_valid(w1, epoch) && st_valid(w2, epoch) && st_valid(w3, epoch))
                return T{(int32_t)(uint32_t)w0, Last2{(int32_t)(uint32_t)w1, (int32_t)(uint32_t)w2}, (uint32_t)w3};
            if (k > TL_SPIN_LIMIT) {
                *timed_out = 1u;
                return op.identity();
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
};
struct LbMaps {
    using T = QMaps;
    static constexpr bool in_slot = true;
    StaticOp<ComposeQ> op;
    T base;
    const uint64_t *m;
    uint32_t epoch;
    __device__ __forceinline__ bool poll(int t, T &val, bool &isP) const {
        const uint64_t w0 = st_load(m + (size_t)t * TLS_M), w1 = st_load(m + (size_t)t * TLS_M + 1);
        if (!st_valid(w0, epoch) || !st_valid(w1, epoch)) return false;
        nib_unpack((uint32_t)w0, val.mil[0], val.mil[1]);
        nib_unpack((uint32_t)w1, val.man[0], val.man[1]);
        isP = map8_constant(val.mil) && map8_constant(val.man);   // a constant map IS a prefix: whatever came before does not matter
        return true;
    }
    __device__ __forceinline__ T load_prefix(int, uint32_t *) const { return op.identity(); }   // (never: the value is in the slot)
};
struct LbFrame {
    using T = FrameLb;
    static constexpr bool in_slot = false;
    FrameLbOp op;
    T base;
    const uint64_t *fa, *fp;
    uint32_t epoch;
    __device__ __forceinline__ bool poll(int t, T &val, bool &isP) const {
        const uint64_t *p = fa + (size_t)t * TLS_FA;
        const uint64_t w0 = st_load(p), w1 = st_load(p + 1), w2 = st_load(p + 2);
        if (!st_valid(w0, epoch) || !st_valid(w1, epoch) || !st_valid(w2, epoch)) return false;
        isP = st_status(w0) == 2u || st_status(w1) == 2u || st_status(w2) == 2u;
        const uint32_t a = (uint32_t)w0, b = (uint32_t)w1, c = (uint32_t)w2;
        val.f.cnt[0] = a & 0x3FFFu;
        val.f.cnt[1] = (a >> 14) & 0x3FFFu;
        val.f.fl[0] = a >> 28;
        val.f.nb[0] = b & 0x3FFFu;
        val.f.nb[1] = (b >> 14) & 0x3FFFu;
        val.f.fl[1] = b >> 28;
        val.f.nc[0] = c & 0x3FFFu;
        val.f.nc[1] = (c >> 14) & 0x3FFFu;
        val.flags = c >> 28;
        return true;
    }
    __device__ __forceinline__ T load_prefix(int t, uint32_t *timed_out) const {
        const uint64_t *p = fp + (size_t)t * TLS_FP;
        for (uint32_t k = 0;; k++) {
            uint64_t w[7];
            bool v = true;
#pragma unroll
            for (int i = 0; i < 7; i++) {
                w[i] = st_load(p + i);
                v = v && st_valid(w[i], epoch);
            }
            if (v) {
                T r;
                r.f.cnt[0] = (uint32_t)w[0];
                r.f.cnt[1] = (uint32_t)w[1];
                r.f.nb[0] = (uint32_t)w[2];
                r.f.nb[1] = (uint32_t)w[3];
                r.f.nc[0] = (uint32_t)w[4];
                r.f.nc[1] = (uint32_t)w[5];
                r.f.fl[0] = (uint32_t)w[6] & 0xFu;
                r.f.fl[1] = ((uint32_t)w[6] >> 4) & 0xFu;
                r.flags = (uint32_t)w[6] >> 8;
                return r;
            }
            if (k > TL_SPIN_LIMIT) {
                *timed_out = 1u;
                return op.identity();
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
};

#ifdef NFC_TAIL_PROF
#define TLP_DECL() unsigned long long tlp_acc[16] = {0}, tlp_t0 = clock64(), tlp_prev = tlp_t0
#define TLP(k) do { const unsigned long long now_ = clock64(); tlp_acc[k] += now_ - tlp_prev; tlp_prev = now_; } while (0)
#define TLP_ADD(k, v) tlp_acc[k] += (v)
#define TLP_DONE()                                                                                        \
    do {                                                                                                  \
        tlp_acc[14] = clock64() - tlp_t0;                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < TP_WGS)                                                      \
            for (int i_ = 0; i_ < 16; i_++) g_tail_prof[((size_t)(i_ >> 3) * TP_WGS + blockIdx.x) * 8 + (i_ & 7)] = tlp_acc[i_]; \
    } while (0)
#else
#define TLP_DECL() ((void)0)
#define TLP(k) ((void)0)
#define TLP_ADD(k, v) ((void)0)
#define TLP_DONE() ((void)0)
#endif

// ---- the kernel ---------------------------------------------------------------------------------------------------------------
#ifndef NFC_TAIL_WAVES
#define NFC_TAIL_WAVES 4   // waves per SIMD the register allocation aims at
#endif
template <bool LDS>
__global__ __launch_bounds__(TL_BLOCK, NFC_TAIL_WAVES) void k_tail(TailArgs A) {
    __shared__ uint32_t s_ent[TL_CAP + 1];   // tile-local sample position | code << 16; slot TL_CAP: entries beyond the staging
    __shared__ __attribute__((aligned(16))) uint8_t s_outb[TL_CAP + 16];
    __shared__ uint32_t s_bits[2][TL_BITW];
    // the decoders' tables, staged once per workgroup (dynamic LDS: tail_table_bytes -- 48 bytes per LUT row, 9.8 KB at max_len 50)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_tab[];
    const int trows = 4 * A.T.nd;
    uint16_t *const s_mil = (uint16_t *)s_tab, *const s_man = s_mil + (LDS ? trows * 8 : 0);
    uint2 *const s_milmap = (uint2 *)(s_man + (LDS ? trows * 8 : 0)), *const s_manmap = s_milmap + (LDS ? trows + 1 : 0);
    __shared__ Last2 s_l2[SCAN_WAVES];
    __shared__ uint32_t s_cnt[SCAN_WAVES];
    __shared__ QMaps s_qm[SCAN_WAVES];
    __shared__ FramePk s_fpk[SCAN_WAVES];
    __shared__ EdgeAgg s_efold[4];
    __shared__ QMaps s_mfold[4];
    __shared__ FrameLb s_ffold[4];
    __shared__ LbShared s_lb;
    __shared__ int32_t s_first;
    __shared__ uint32_t s_crossx, s_job, s_timed_out;
    const int tid = threadIdx.x;
    const EdgeArgs &E = A.E;
    const DecTables &T = A.T;
    uint64_t *const st_ea = A.st, *const st_ep = st_ea + (size_t)A.ntiles * TLS_EA, *const st_m = st_ep + (size_t)A.ntiles * TLS_EP,
                    *const st_fa = st_m + (size_t)A.ntiles * TLS_M, *const st_fp = st_fa + (size_t)A.ntiles * TLS_FA;
    const uint32_t tagA = (A.epoch << 2) | 1u, tagP = (A.epoch << 2) | 2u;

    zero_words(A.Znext);
    const uint32_t ident = 4u * (uint32_t)T.nd;   // the identity row of both map tables
    if (LDS && A.decode) {
        const int rows = 4 * T.nd;
        for (int i = tid; i <= rows; i += TL_BLOCK) {
            if (T.reader) s_milmap[i] = T.qmil_map[i];
            if (T.tag) s_manmap[i] = T.man_map[i];
        }
        for (int i = tid; i < rows; i += TL_BLOCK) {
            if (T.reader) ((uint4 *)s_mil)[i] = ((const uint4 *)T.qmil_step)[i];
            if (T.tag) ((uint4 *)s_man)[i] = ((const uint4 *)T.man_step)[i];
        }
    }
    if (tid == 0) s_timed_out = 0u;
    TLP_DECL();
    const uint32_t ncert = A.C.blocks, c0 = ncert ? 1u : 0u, njobs = A.ntiles + ncert;
    const EdgeAggOp eop{E.mx, E.mx_magic};
    const int32_t s0 = (int32_t)E.skip - E.dur_in;   // where the run carried into the batch "starts"

    uint32_t nxt = 0u;     // (thread 0: the next ticket, drawn while the tile before waited in its last look-back)
    bool have_nxt = false;
    for (;;) {
        __syncthreads();   // (the tile before: its LDS is free; the tables are staged)
        if (tid == 0) s_job = (have_nxt ? nxt : atomicAdd(A.ticket, 1u)) - A.ticket_base;
        have_nxt = false;
        __syncthreads();
        const uint32_t job = s_job;
        TLP(15);
        if (job >= njobs) break;
        if (ncert && (job == 0u || job >= c0 + A.ntiles)) {   // the certification's workgroups: the one that resolves the end-of-batch state first
            certify_block(A.C.A, A.C.cert, nullptr, A.C.ring_next, A.C.carry, A.C.sum, job == 0u ? ncert - 1u : job - c0 - A.ntiles, ncert);
            continue;
        }
        const int j = (int)(job - c0);
        const bool last = (uint32_t)j == A.ntiles - 1u;
        // ================= E1: the tile's words, their aggregates ==========================================================
        const size_t wt = (size_t)j * A.tw, wend = min(wt + A.tw, A.nwords);
        const size_t w_first = wt + (size_t)tid * TL_ITEMS;
        uint64_t ng[TL_ITEMS], ps[TL_ITEMS], m[TL_ITEMS];
        const int val_before = load_words<TL_ITEMS>(E, min(w_first, wend), wend, ng, ps, m);
        for (int i = tid; i < 2 * TL_BITW; i += TL_BLOCK) (&s_bits[0][0])[i] = 0u;
        if (tid == 0) {
            s_first = POS_NONE;
            s_crossx = 0u;
        }
        EdgeAgg agg = eop.identity();
        {
            bool may = false;
#pragma unroll
            for (int i = 0; i < TL_ITEMS; i++) may = may || word_may_time_out(E, m[i]);
            const bool inner = __any(may);
#pragma unroll
            for (int i = 0; i < TL_ITEMS; i++) agg = eop(agg, word_agg(E, (int32_t)((w_first + i) * 64), m[i], inner));
        }
        // entries before a thread's words, in the tile: its val changes, per change the time-outs of the run that ends there (the
        // change before it is what the first scan carries), in two cheap scans (edges.hip.h: write_edges_tile)
        Last2 tile_l;
        const Last2 excl_l = block_exclusive<Last2Op>(agg.l, s_l2, tile_l);
        const bool has_first = agg.first != POS_NONE, first_of_tile = has_first && excl_l.s1 == POS_NONE;
        const uint32_t cross_in = (has_first && !first_of_tile) ? E.timeouts_between(excl_l.s1, agg.first) : 0u;
        if (first_of_tile) s_first = agg.first;
        uint32_t tile_cnt_in;
        const uint32_t excl_cnt_in = block_exclusive<AddU32>(agg.sum + cross_in, s_cnt, tile_cnt_in);   // (its barriers publish s_first)
        const int32_t tile_first = s_first;
        const int32_t tile_p0 = (int32_t)min(wt * 64, (size_t)E.n);
        const int32_t tile_base = (int32_t)(wt * 64);
        if (tid == 0) {
            auto rel = [&](int32_t p) { return p == POS_NONE ? 0xFFFFu : (uint32_t)(p - tile_base); };
            st_store(st_ea + (size_t)j * TLS_EA, rel(tile_first) | (rel(tile_l.s1) << 16), tagA);
            st_store(st_ea + (size_t)j * TLS_EA + 1, rel(tile_l.s2) | (tile_cnt_in << 16), tagA);
        }
        TLP(0);
        TLP_ADD(10, 1);
        const LbEdge lbe{eop, eop.identity(), st_ea, st_ep, A.epoch, A.tw};
        const EdgeAgg pre = tail_lookback(lbe, j, s_efold, &s_lb, &s_timed_out);
        TLP(1);
        if (tid == 0) {
            const EdgeAgg inc = eop(pre, EdgeAgg{tile_first, tile_l, tile_cnt_in});
            uint64_t *p = st_ep + (size_t)j * TLS_EP;
            st_store(p, (uint32_t)inc.first, tagP);
            st_store(p + 1, (uint32_t)inc.l.s1, tagP);
            st_store(p + 2, (uint32_t)inc.l.s2, tagP);
            st_store(p + 3, inc.sum, tagP);
            auto rel = [&](int32_t p) { return p == POS_NONE ? 0xFFFFu : (uint32_t)(p - tile_base); };
            st_store(st_ea + (size_t)j * TLS_EA, rel(tile_first) | (rel(tile_l.s1) << 16), tagP);
            st_store(st_ea + (size_t)j * TLS_EA + 1, rel(tile_l.s2) | (tile_cnt_in << 16), tagP);
        }
        // ================= E2: where every thread's entries go, the walk ===================================================
        const uint32_t gbase = entries_before(E, pre, tile_p0);
        const Last2 before_l = Last2Op::op(pre.l, excl_l), all_l = Last2Op::op(pre.l, tile_l);
        if (first_of_tile) s_crossx = E.timeouts_between(pre.l.s1 != POS_NONE ? pre.l.s1 : s0, agg.first);
        __syncthreads();
        const uint32_t crossx = s_crossx;
        const uint32_t excl_cnt = excl_cnt_in + (excl_l.s1 != POS_NONE ? crossx : 0u), tile_cnt = tile_cnt_in + crossx;
        const int32_t tile_end = (int32_t)min(wend * 64, (size_t)E.n), Tq = (int32_t)min(w_first * 64, (size_t)tile_end);
        const uint32_t run_k = E.timeouts_between(before_l.s1 != POS_NONE ? before_l.s1 : s0, Tq);   // time-outs so far of the run in progress at Tq
        const uint32_t in_progress0 = E.timeouts_between(pre.l.s1 != POS_NONE ? pre.l.s1 : s0, tile_p0);
        const uint32_t off = excl_cnt + run_k - in_progress0;
        const uint32_t total = tile_cnt + E.timeouts_between(all_l.s1 != POS_NONE ? all_l.s1 : s0, tile_end) - in_progress0;
        if (last && tid == 0) {
            *A.edges_total = gbase + total;
            *A.last2_total = all_l;
            edge_carry_out(E, all_l, A.carry_out);
        }
        const bool dense = total > (uint32_t)TL_CAP;
        if (tid == 0 && total > (uint32_t)TL_CAP / 4u)   // (the launch's densest tile: later epochs outrank earlier ones, nothing clears the word)
            (void)__hip_atomic_fetch_max((unsigned long long *)(A.ticket + 2), ((unsigned long long)A.epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!dense) {
            EdgeWalk W;
            {
                const Last2 c = before_l;
                const bool carried = c.s1 == POS_NONE;
                const int32_t s = carried ? s0 : c.s1;
                const int32_t k = (int32_t)run_k;
                W.q = s + k * E.mx;
                const bool timed = k > 0;
                int left = E.state_in;
                if (carried) {
                    W.lb = E.last_bit_in;
                } else {
                    W.lb = val_before;
                    if (W.lb == 0 && !timed) {   // a val-0 run that has not timed out keeps what the previous run left
                        int lb, dur;
                        E.state_before(Tq, c, lb, dur, left);
                    }
                }
                W.zst = timed ? 0 : left;
                W.tq = timed ? W.q : POS_NONE;
                W.cskip = carried ? (int32_t)E.skip : POS_NONE;
            }
            uint32_t k4 = off * 4u;
#pragma unroll
            for (int i = 0; i < TL_ITEMS; i++) {
                const int32_t w0 = (int32_t)((w_first + i) * 64);
                const int32_t end = min(w0 + 64, tile_end);
                const uint32_t ng_lo = (uint32_t)ng[i], ng_hi = (uint32_t)(ng[i] >> 32), ps_lo = (uint32_t)ps[i], ps_hi = (uint32_t)(ps[i] >> 32);
                uint64_t mm = m[i];
                while (true) {
                    const int b = mm ? __ffsll((long long)mm) - 1 : 64;
                    const int32_t c = min(w0 + b, end);   // the next change, or the end of the word
                    const int32_t nt = W.q + E.mx;        // the run's next time-out
                    const bool is_to = nt < c;            // _dur exceeds max_len first (transition_sink.py:95-99)
                    if (!is_to && !mm) break;
                    const int32_t p = is_to ? nt : c;
                    const int run_st = (W.lb == -1) ? 2 : 1;
                    int prev_st = (W.lb != 0) ? ((W.tq == p - 1) ? 0 : run_st) : W.zst;
                    prev_st = (p == W.cskip) ? E.state_in : prev_st;
                    const uint32_t nh = b < 32 ? ng_lo : ng_hi, ph = b < 32 ? ps_lo : ps_hi;
                    const int vb = __builtin_amdgcn_ubfe(nh, (uint32_t)b & 31u, 1u) ? -1 : (int)__builtin_amdgcn_ubfe(ph, (uint32_t)b & 31u, 1u);
                    const int val = is_to ? W.lb : vb;
                    const int st = is_to ? ((W.lb != 0) ? run_st : W.zst) : ((val == -1) ? 2 : ((val == 1) ? 1 : prev_st));
                    {
                        const int v = st == 2 ? W.lb + 1 : W.lb, d = (is_to || prev_st == 0) ? E.mx : p - W.q, t = st - 1;
                        const int dd = d < E.nd ? d : E.nd - 1;
                        uint32_t row;
                        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"((uint32_t)(v + 1)), "s"((uint32_t)E.nd), "v"((uint32_t)dd));
                        const uint32_t code = row | ((uint32_t)(t + 1) << 14);
                        *(uint32_t *)((char *)s_ent + min(k4, (uint32_t)TL_CAP * 4u)) = (uint32_t)(p - tile_p0) | (code << 16);
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
        }
        __syncthreads();
        TLP(2);
        const uint32_t cnt_e = dense ? 0u : total;
        for (uint32_t jj = tid; jj < cnt_e; jj += TL_BLOCK) {
            const uint32_t g = gbase + jj;
            if (g < A.cap) {
                const uint32_t e = s_ent[jj];
                A.epos[g] = (uint32_t)tile_p0 + (e & 0xFFFFu);
                A.ecode[g] = (uint16_t)(e >> 16);
            }
        }
        // (decode == 0: the three-launch decode follows -- the tile goes through the chains below with nothing to decode, so that the
        // batch's verdict still reaches the last tile)
        const uint32_t cnt = A.decode ? cnt_e : 0u;
        TLP(3);
        // ================= D: the decoders, from the entries in LDS ========================================================
        const uint32_t per = ((cnt + TL_BLOCK - 1) / TL_BLOCK) | 1u;   // entries per thread, odd
        const uint32_t i0 = (uint32_t)tid * per;
        // (a thread's entries are read from LDS where they are needed, twice: fifteen registers fewer across the look-back)
        auto code_at = [&](int k) __attribute__((always_inline)) -> uint32_t { return (i0 + k < cnt) ? s_ent[i0 + k] >> 16 : 0u; };   // (code 0 is dropped)
        const QMaps idm = ComposeQ::identity();
        uint2 am = make_uint2(idm.mil[0], idm.mil[1]), an = make_uint2(idm.man[0], idm.man[1]);
        {
            auto then = [](const uint2 &a, const uint2 &b) __attribute__((always_inline)) -> uint2 {   // a, then b
                return make_uint2(__builtin_amdgcn_perm(b.y, b.x, a.x), __builtin_amdgcn_perm(b.y, b.x, a.y));
            };
#pragma unroll
            for (int k = 0; k < TL_PER_MAX; k++) {
                if ((uint32_t)k < per) {   // (uniform)
                    const uint32_t code = code_at(k);
                    if (T.reader) {
                        const uint32_t idx = (code >> 14) == 2u ? (code & 0x3FFFu) : ident;
                        am = then(am, LDS ? s_milmap[idx] : T.qmil_map[idx]);
                    }
                    if (T.tag) {
                        const uint32_t idx = (code >> 14) == 1u ? (code & 0x3FFFu) : ident;
                        an = then(an, LDS ? s_manmap[idx] : T.man_map[idx]);
                    }
                }
            }
        }
        QMaps tile_map;
        const QMaps excl_map = block_exclusive<ComposeQ>(QMaps{{am.x, am.y}, {an.x, an.y}}, s_qm, tile_map);
        if (!T.reader || !A.decode) tile_map.mil[0] = tile_map.mil[1] = 0u;   // (a decoder that is off: a constant map ends every look-back)
        if (!T.tag || !A.decode) tile_map.man[0] = tile_map.man[1] = 0u;
        if (tid == 0 && !dense) {
            st_store(st_m + (size_t)j * TLS_M, nib_pack(tile_map.mil[0], tile_map.mil[1]), tagA);
            st_store(st_m + (size_t)j * TLS_M + 1, nib_pack(tile_map.man[0], tile_map.man[1]), tagA);
        }
        TLP(4);
        QMaps base_map;   // before tile 0: the carried states, as constant maps
        base_map.mil[0] = base_map.mil[1] = (T.reader ? (A.state0 & 7u) : 0u) * 0x01010101u;
        base_map.man[0] = base_map.man[1] = (T.tag ? ((A.state0 >> 4) & 7u) : 0u) * 0x01010101u;
        const LbMaps lbm{StaticOp<ComposeQ>{}, base_map, st_m, A.epoch};
        const QMaps in_map = tail_lookback(lbm, j, s_mfold, &s_lb, &s_timed_out);   // constant: the states the tile is entered in
        TLP(5);
        const uint32_t st_in = (in_map.mil[0] & 7u) | ((in_map.man[0] & 7u) << 4);
        {
            const uint32_t st_out = ComposeQ::step(tile_map, st_in);
            if (tid == 0) {   // the states after the tile, as constant maps: whoever looks back stops here
                st_store(st_m + (size_t)j * TLS_M, (st_out & 7u) * 0x11111111u, tagP);
                st_store(st_m + (size_t)j * TLS_M + 1, ((st_out >> 4) & 7u) * 0x11111111u, tagP);
                if (last) {
                    A.epi.carry->mil_state = (int32_t)byte_of(A.q_rep, st_out & 7u);
                    A.epi.carry->man_state = (int32_t)(st_out >> 4);
                }
            }
        }
        uint32_t ow[4] = {0u, 0u, 0u, 0u};
        {
            uint32_t st = ComposeQ::step(excl_map, st_in);
            const uint16_t *mil = LDS ? s_mil : T.qmil_step;
            const uint16_t *man = LDS ? s_man : T.man_step;
#pragma unroll
            for (int k = 0; k < TL_PER_MAX; k++) {
                if ((uint32_t)k < per) {
                    const uint32_t code = code_at(k);
                    const uint32_t li = code & 0x3FFFu, route = code >> 14;
                    uint32_t w = 0;
                    if (route == 2u && T.reader) {
                        const uint32_t en = mil[li * 8u + (st & 7u)];
                        w = en >> 8;
                        st = (st & ~15u) | (en & 15u);
                    } else if (route == 1u && T.tag) {
                        const uint32_t en = man[li * 8u + ((st >> 4) & 7u)];
                        const uint32_t mo = en >> 8;
                        w = (mo & 3u) ? ((mo & 0xFCu) | 3u) : 0u;
                        st = (st & 15u) | ((en & 15u) << 4);
                    }
                    ow[k >> 2] |= w << (8 * (k & 3));
                    if (i0 + k < cnt) s_outb[i0 + k] = (uint8_t)w;
                }
            }
        }
        TLP(6);
        // ================= F: framing ======================================================================================
        FramePk tile_pk;
        const FramePk mine = FramePkOp::pack(frame_agg_of(ow));
        const FramePk in_tile = block_exclusive<FramePkOp>(mine, s_fpk, tile_pk);   // (its barriers: s_outb is complete)
        for (uint32_t jj = tid; A.decode && jj < cnt + (last ? 16u : 0u); jj += TL_BLOCK) {   // (the batch's last group is read whole: zeros behind the last entry)
            const uint32_t g = gbase + jj;
            if (g < A.cap + 16u) A.outw[g] = jj < cnt ? s_outb[jj] : (uint8_t)0;
        }
        const FrameAgg tile_fa = FramePkOp::unpack(tile_pk);
        const uint32_t my_flags = (dense ? 1u : 0u) | (s_timed_out ? 2u : 0u);   // (behind the block scan's barriers)
        if (tid == 0) {
            uint64_t *p = st_fa + (size_t)j * TLS_FA;
            st_store(p, tile_fa.cnt[0] | (tile_fa.cnt[1] << 14) | (tile_fa.fl[0] << 28), tagA);
            st_store(p + 1, tile_fa.nb[0] | (tile_fa.nb[1] << 14) | (tile_fa.fl[1] << 28), tagA);
            st_store(p + 2, tile_fa.nc[0] | (tile_fa.nc[1] << 14) | (my_flags << 28), tagA);
        }
        TLP(7);
        if (tid == 0) nxt = atomicAdd(A.ticket, 1u);   // (its round trip passes under the look-back)
        have_nxt = true;
        const LbFrame lbf{FrameLbOp{}, FrameLb{FrameAggOp::identity(), 0u}, st_fa, st_fp, A.epoch};
        const FrameLb pre_f = tail_lookback(lbf, j, s_ffold, &s_lb, &s_timed_out);
        TLP(8);
        const FrameAgg pre_fa = pre_f.f;
        const FrameAgg all = FrameAggOp::op(pre_fa, tile_fa);
        const uint32_t all_flags = pre_f.flags | my_flags | (s_timed_out ? 2u : 0u);
        if (tid == 0) {
            uint64_t *p = st_fp + (size_t)j * TLS_FP;
            st_store(p, all.cnt[0], tagP);
            st_store(p + 1, all.cnt[1], tagP);
            st_store(p + 2, all.nb[0], tagP);
            st_store(p + 3, all.nb[1], tagP);
            st_store(p + 4, all.nc[0], tagP);
            st_store(p + 5, all.nc[1], tagP);
            st_store(p + 6, all.fl[0] | (all.fl[1] << 4) | (all_flags << 8), tagP);
            uint64_t *q = st_fa + (size_t)j * TLS_FA;
            st_store(q, tile_fa.cnt[0] | (tile_fa.cnt[1] << 14) | (tile_fa.fl[0] << 28), tagP);
            st_store(q + 1, tile_fa.nb[0] | (tile_fa.nb[1] << 14) | (tile_fa.fl[1] << 28), tagP);
            st_store(q + 2, tile_fa.nc[0] | (tile_fa.nc[1] << 14) | (my_flags << 28), tagP);
            if (last) {
                *A.frame_total = all;
                A.epi(all);
                *A.verdict = ((all_flags & 1u) ? TLV_DENSE : 0u) | ((all_flags & 2u) ? TLV_TIMEOUT : 0u);
                const unsigned long long pk = __hip_atomic_load((unsigned long long *)(A.ticket + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *A.peak_out = (uint32_t)(pk >> 32) == A.epoch ? (uint32_t)pk : 0u;   // (a hint: an update still on its way is missed)
            }
        }
        if (j == 0 && A.decode) pack_pending(A.P, tid, TL_BLOCK);
        // a thread's appended bits squeezed together (compress32), or-ed into the tile's words in LDS at the phase the tile has
        // in the stream; its packet ends to their places (decode.hip.h: k_frame_write -- here from the out-bytes in registers)
        {
            const FrameOut &P = A.P;
            const FrameAgg before = FrameAggOp::op(pre_fa, FramePkOp::unpack(in_tile));
            if (ow[0] | ow[1] | ow[2] | ow[3]) {
                const SlotMasks sm = slot_masks(ow);
                const uint32_t low = symbol_low_bits(ow);
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    if (!sm.V[t] || !P.bits[t]) continue;
                    const uint32_t bo = P.pend[t] + fa_bits(before, t, P.started_in[t]);          // the thread's first bit in the stream
                    const uint32_t tile_bit0 = P.pend[t] + fa_bits(pre_fa, t, P.started_in[t]);   // the tile's
                    const uint32_t started = pm_apply(before.fl[t], P.started_in[t]);
                    const uint32_t bef = started_before(sm, t, started);
                    const uint32_t appended = sm.V[t] & ~sm.ST[t] & (bef | ~sm.SA[t]);
                    uint32_t cl = sm.ST[t] & bef;
                    const uint32_t acc = compress32(low, appended);
                    if (cl) {   // packet ends: one per frame
                        const uint32_t co = fa_closes(before, t, P.started_in[t]), closes = cl;
                        while (cl) {
                            const uint32_t lowb = cl & (0u - cl);
                            const int k = (__ffs((int)cl) - 1) >> 1;
                            cl ^= lowb;
                            const uint32_t jc = co + (uint32_t)__popc(closes & (lowb - 1u));
                            if (jc < P.cap_close[t]) {
                                P.close_end[t][jc] = bo + (uint32_t)__popc(appended & (lowb - 1u));
                                P.close_idx[t][jc] = P.g0 + (uint64_t)((uint32_t)tile_p0 + (s_ent[i0 + k] & 0xFFFFu));
                            }
                        }
                    }
                    if (appended) {
                        const uint32_t lb = (tile_bit0 & 31u) + (bo - tile_bit0);   // the thread's first bit in the tile's words
                        const uint32_t w = lb >> 5, sh = lb & 31u;
                        const uint64_t v = (uint64_t)acc << sh;
                        if (w + 1 < (uint32_t)TL_BITW) {
                            if ((uint32_t)v) atomicOr(&s_bits[t][w], (uint32_t)v);
                            if ((uint32_t)(v >> 32)) atomicOr(&s_bits[t][w + 1], (uint32_t)(v >> 32));
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (!P.bits[t]) continue;
                const uint32_t tile_bit0 = P.pend[t] + fa_bits(pre_fa, t, P.started_in[t]);
                const uint32_t nbits = P.pend[t] + fa_bits(all, t, P.started_in[t]) - tile_bit0;   // the tile's appended bits
                if (!nbits) continue;
                uint32_t *gb = (uint32_t *)P.bits[t];
                const uint32_t w0 = tile_bit0 >> 5, nw = ((tile_bit0 & 31u) + nbits + 31u) >> 5, capw = (P.cap_bits[t] + 31u) >> 5;
                for (uint32_t jj = tid; jj < nw && jj < (uint32_t)TL_BITW; jj += TL_BLOCK) {
                    if (w0 + jj >= capw) break;   // (an estimate too small: the host sees it in the totals and repeats the stage with room)
                    const uint32_t v = s_bits[t][jj];
                    if (jj == 0 || jj == nw - 1) {
                        if (v) atomicOr(gb + w0 + jj, v);   // (words shared with the neighbouring tiles, or with the pending bits)
                    } else {
                        gb[w0 + jj] = v;
                    }
                }
            }
        }
        TLP(9);
    }
    TLP_DONE();
}

// The symbol arrays are written when somebody reads them (decode.hip.h: k_symbols_write), from the out-bytes and the framing
// aggregates of tiles of DEC_TILE entries: after k_tail those aggregates do not exist yet -- this pass makes them.
__global__ __launch_bounds__(SCAN_BLOCK) void k_sym_reduce(const uint8_t *outw, size_t n, const uint32_t *n_dev, FrameAgg *tile_aggs) {
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * DEC_TILE >= n) return;
    __shared__ FramePk lds[SCAN_WAVES];
    const size_t base = ((size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x) * DEC_PER_THREAD;
    FramePk mine = FramePkOp::identity();
#pragma unroll
    for (int g = 0; g < DEC_GROUPS; g++) {
        uint32_t ow[4] = {0u, 0u, 0u, 0u};
        const size_t gb = base + (size_t)DEC_ITEMS * g;
        if (gb < n) {
            const uint4 a = *(const uint4 *)(outw + gb);
            ow[0] = a.x; ow[1] = a.y; ow[2] = a.z; ow[3] = a.w;
        }
        mine = FramePkOp::op(mine, FramePkOp::pack(frame_agg_of(ow)));
    }
    FramePk total;
    (void)block_exclusive<FramePkOp>(mine, lds, total);
    if (threadIdx.x == 0) tile_aggs[blockIdx.x] = FramePkOp::unpack(total);
}

}  // namespace nfc
