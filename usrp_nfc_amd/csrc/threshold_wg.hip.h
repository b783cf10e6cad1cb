// threshold_wg.hip.h -- pass 0 of the threshold stage (transition_sink.py:55-82) with a time chunk per WORKGROUP.
//
// k_threshold_lean (threshold_lean.hip.h) walks a chunk with ONE wave, so the number of waves on the machine equals the number
// of chunks, and every chunk pays a window of speculation before it and a summary after it: occupancy could only be bought
// with more of those.  Here the four waves of a workgroup share one chunk and one LDS ring:
//
//   * a ROUND is four steps of 64 NR samples; wave w takes step w of every round.  A SUPERSTEP of a few rounds is classified
//     against the thresholds of the sum tracked at its start, so its steps are independent, and their ring slots are disjoint
//     while a round fits the window.  No allowance is guessed: when the superstep closes, the drift the window sum turned out
//     to have (the larger of the sums of the positive and of the negative accepted x - prev, plus what the speculated incoming
//     ring may be off by) is compared with how close any sample came to a threshold -- farther than the drift, and the
//     classifications are the reference's (by induction over the samples); the next superstep is made longer or shorter by the
//     head-room seen;
//   * what a step needs of the samples before it is one thing only: where the last LOW sample lies, if within max_len + 1
//     ("HIGH is ignored", transition_sink.py:71) -- with max_len <= 254 that is inside the step before, whose LOW masks its
//     wave publishes in LDS before the round's first barrier (phase A: envelopes, pre-tests, LOW masks; phase B: ring update);
//   * the round closes with one exchange of (B, D, failure) per wave and a second barrier; every wave then moves the tracked
//     sum and opens the next round with the same arithmetic, so nothing is broadcast;
//   * a LOW run longer than max_len (a loss of signal) classifies like any other LOW sample; what it changes is whether a HIGH
//     sample within max_len + 1 of its END is ignored (its last sample may have ended on a time-out, transition_sink.py:95-99).
//     Only the general form ever asks: it gives up when a run in reach MAY be that long -- an aligned block of A.blk LOW samples
//     in its step or the two before it (a run longer than max_len covers one wherever step seams fall) --, the run a chunk
//     STARTS in is measured against the carried length, and a chunk that ends in reach of such a run gives up too;
//   * rounds that are not four whole steps of stable samples (the stream's first stable sample, a batch's ragged end) take the
//     masked general form in every wave, on synchronously loaded samples;
//   * speculation, the saved incoming ring and the summary are spread over the 256 threads.
//
// Everything else is as in k_threshold_lean.  In pass 0 nothing is repaired in place -- a failed check makes the whole workgroup
// give up its chunk, which the host re-runs from the exact state: with the EX instantiations of this kernel (round 6: a round that
// fails its check is taken back and evaluated exactly by the four waves, see the template's comment) where few enough chunks
// failed, with k_threshold otherwise.  Same summaries (ChunkInfo, RunMeta, ring_in, ring_out, touched), so certification and
// re-runs do not know which kernel ran.
#pragma once
#include "edges.hip.h"
#include "threshold_lean.hip.h"

namespace nfc {

constexpr int WG_WAVES = 4;
// A step is NR rows of 64 samples (lane l holds samples l, 64 + l, ...), a round four steps: NR is chosen so that a round fits the
// window (its steps' ring slots must be disjoint) -- 4 or 8 rows: 1024 or 2048 samples per round.  Everything a round
// costs once (barriers, bookkeeping, the form dispatch) is spread over that many samples.
constexpr int wg_round_samples(int nr) { return 64 * nr * WG_WAVES; }
constexpr int WG_NR_MAX = 8;
// The plane words of a regular round (a step's 2 NR dwords per plane and wave) are STAGED in LDS behind WgShared, in the planes' own
// order (per plane: round, wave, dword).  Measured (round 4): with the plane stores taken out the kernel runs 0.158 instead of 0.183 ms,
// and the same with every store aimed at one 64 KB region -- what costs is not the store instructions but 25 MB of WRITES reaching
// the HBM in a trickle between the reads of 800 MB (the 16 MB of ring summaries, written in bulk at a chunk's ends, cost 3 us).  So:
//   * where a CU's LDS holds the planes of its workgroups' whole chunks (2 bits per sample: 24 KB for the 98 304 samples of a chunk
//     at 1e8 samples per batch, four workgroups per CU) they leave when the chunk is done, all of them at once;
//   * otherwise (long windows, batches submitted ahead -- the other stages' workgroups need the LDS --, longer chunks) they leave FR
//     rounds at a time as one wave's 16-byte stores of whole lines, from a ring of 2 FR rounds.
#ifndef NFC_WG_FR
#define NFC_WG_FR 8
#endif
#ifndef NFC_WG_FLAGS_SLEEP
#define NFC_WG_FLAGS_SLEEP 0   // s_sleep between two looks at the counters (FLG; 0: none)
#endif
constexpr int wg_flush_rounds(int nr) { return nr == 4 ? NFC_WG_FR : NFC_WG_FR / 2; }   // 1 KB per plane and flush
constexpr size_t wg_stage_bytes(int nr, int rounds) { return (size_t)rounds * (size_t)(2 * WG_WAVES * 2 * nr * 4); }   // both planes
// LDS behind the ring: LOW masks of the rounds' steps, the close exchange, scratch for workgroup reductions
struct WgShared {
    uint32_t msk[3][WG_WAVES][4 * WG_NR_MAX + 4];   // per round (modulo three) and wave: the LOW masks of its step (dwords 0 .. 2 NR - 1), dword 4 NR: any LOW sample
    uint32_t scr[2][WG_WAVES][8];      // workgroup reductions (alternating halves: one barrier per reduction)
    int32_t fin[WG_WAVES][4];          // chunk end: last LOW index, last non-LOW index, latest step with LOW samples
    float4 acc[WG_WAVES][64];          // close of a superstep: every lane's (sum |x - prev|, sum (x - prev)) over what it accepted, its smallest distances to the thresholds
    uint32_t flag[WG_WAVES];           // ... and every wave's failure code
    // What only the GENERAL form of a step produces, per wave (lane 0 keeps it up to date, the chunk's summary reads it): this wave's
    // last LOW sample, its last sample that is not LOW, the smallest / largest raw bits it accepted.  In LDS and not in registers
    // because the general form runs rarely and the rounds run always: values that one cold path writes are carried through every
    // round as copies -- the compiler moved eight registers to and fro per round for these four (17 v_mov per round in all, of
    // ~120 vector instructions: round 5, read off the ISA).
    // ... and (slots 4, 5) how close a sample of such a step came to the LOW / to the HIGH threshold in the superstep in progress
    // (f32 bits; lane 0 takes them into what it hands in when the superstep closes, and resets them).
    uint32_t cold[WG_WAVES][8];
    float bc[8];                       // ... and what wave 0 makes of them: the next thresholds, the sum, the allowance, the verdict
    uint32_t pha[WG_WAVES];            // FLG: rounds whose phase A (LOW masks) each wave has published
};
constexpr size_t WG_SHARED_BYTES = (sizeof(WgShared) + 15) & ~(size_t)15;

// The workgroup's barrier without the fence __syncthreads() brings: that fence waits for EVERY vector memory operation of the
// wave (s_waitcnt vmcnt(0)) -- the samples asked for ahead included.  LDS traffic of this wave is complete (lgkmcnt) before it
// arrives; the asm statement is a compiler barrier for memory accesses as well.
// smallest value of a wave (finite, non-negative inputs), on DPP moves like wave_sum_f32
__device__ __forceinline__ float wg_wave_min_f32(float v) {
    constexpr int BIG = 0x7F7FFFFF;   // lanes without a source take this
    v = fminf(v, __int_as_float(dpp_i32<0x111, 0xF>(BIG, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x112, 0xF>(BIG, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x114, 0xF>(BIG, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x118, 0xF>(BIG, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x142, 0xA>(BIG, __float_as_int(v))));
    v = fminf(v, __int_as_float(dpp_i32<0x143, 0xC>(BIG, __float_as_int(v))));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The 64 NR samples at base (uniform) + voff (this lane's byte offset) into accumulator registers named literally (row j of an
// IQ step in a[2 j : 2 j + 1], of the one-dword kinds in a[j]; see threshold_lean.hip.h for why) -- with the address split into
// a scalar base and a 32-bit lane offset, so that walking the chunk costs scalar additions only.
// (The scalar base may have been written by a VECTOR instruction just before -- v_readlane of a spilled register,
// v_readfirstlane --, and a vector memory instruction that reads such a register within the next five issue slots gets the OLD
// value: the hardware does not interlock and the compiler pads only between instructions it emitted itself.  Measured: a
// memory access fault on whichever build happened to reload the base right in front of the statement.  Hence the "s_nop 4" at
// the head, and ONE statement for all loads of a step: nothing of the compiler's comes between the pad and the last load.)
// The samples are read ONCE: they carry the non-temporal hint, so that 800 MB of them do not push the planes, the ring summaries
// and the tables out of the L2 and the memory-side cache on their way through (measured, same-call A/B on a slow box: launch
// 0.175 -> 0.161 ms, and the step 0.292 -> 0.274 -- the stages behind the kernel find what it wrote; sc1 / sc0 sc1: no change).
#ifndef NFC_WG_LDPOL
#define NFC_WG_LDPOL " nt"
#endif
#define WG_LDS_(OP, R, K) OP " " R ", %0, %1 offset:%" #K NFC_WG_LDPOL "\n\t"
#define WG_LD4(OP, R0, R1, R2, R3, ST, ...)                                                                                  \
    asm volatile("s_nop 4\n\t" WG_LDS_(OP, R0, 2) WG_LDS_(OP, R1, 3) WG_LDS_(OP, R2, 4) OP " " R3 ", %0, %1 offset:%5" NFC_WG_LDPOL         \
                 :                                                                                                           \
                 : "v"(voff), "s"(base), "n"(0), "n"(ST), "n"(2 * (ST)), "n"(3 * (ST))                                       \
                 : "memory", __VA_ARGS__)
#define WG_LD8(OP, R0, R1, R2, R3, R4, R5, R6, R7, ST, ...)                                                                  \
    asm volatile("s_nop 4\n\t" WG_LDS_(OP, R0, 2) WG_LDS_(OP, R1, 3) WG_LDS_(OP, R2, 4) WG_LDS_(OP, R3, 5) WG_LDS_(OP, R4, 6)  \
                 WG_LDS_(OP, R5, 7) WG_LDS_(OP, R6, 8) OP " " R7 ", %0, %1 offset:%9" NFC_WG_LDPOL                           \
                 :                                                                                                           \
                 : "v"(voff), "s"(base), "n"(0), "n"(ST), "n"(2 * (ST)), "n"(3 * (ST)), "n"(4 * (ST)), "n"(5 * (ST)),        \
                   "n"(6 * (ST)), "n"(7 * (ST))                                                                              \
                 : "memory", __VA_ARGS__)
#define WG_A03 "a0", "a1", "a2", "a3"
#define WG_A47 "a4", "a5", "a6", "a7"
#define WG_A8B "a8", "a9", "a10", "a11"
#define WG_ACF "a12", "a13", "a14", "a15"
template <int KIND, int NR>
__device__ __forceinline__ void wg_load_step(uint32_t voff, const char *base) {
    static_assert(NR == 4 || NR == 8, "rows per step");
    if constexpr (KIND == IN_IQ_F32) {
        if constexpr (NR == 4) WG_LD4("global_load_dwordx2", "a[0:1]", "a[2:3]", "a[4:5]", "a[6:7]", 512, WG_A03, WG_A47);
        if constexpr (NR == 8)
            WG_LD8("global_load_dwordx2", "a[0:1]", "a[2:3]", "a[4:5]", "a[6:7]", "a[8:9]", "a[10:11]", "a[12:13]", "a[14:15]", 512, WG_A03, WG_A47, WG_A8B, WG_ACF);
    } else if constexpr (KIND == IN_I16_SQ) {
        if constexpr (NR == 4) WG_LD4("global_load_sshort", "a0", "a1", "a2", "a3", 128, WG_A03);
        if constexpr (NR == 8) WG_LD8("global_load_sshort", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", 128, WG_A03, WG_A47);
    } else {
        if constexpr (NR == 4) WG_LD4("global_load_dword", "a0", "a1", "a2", "a3", 256, WG_A03);
        if constexpr (NR == 8) WG_LD8("global_load_dword", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", 256, WG_A03, WG_A47);
    }
}
// Waits for EVERY vector memory operation of the wave (the request went out a round ago), then hands the step's samples over as
// envelopes.  (The reads that follow the wait are statements of their own: asm volatile statements keep their order.)
#define WG_RD4(WAIT, R0, R1, R2, R3, O)                                                                                                       \
    asm volatile(WAIT "v_accvgpr_read_b32 %0, " R0 "\n\tv_accvgpr_read_b32 %1, " R1 "\n\tv_accvgpr_read_b32 %2, " R2 "\n\tv_accvgpr_read_b32 %3, " R3 \
                 : "=v"(w[O]), "=v"(w[O + 1]), "=v"(w[O + 2]), "=v"(w[O + 3])                                                                  \
                 :                                                                                                                           \
                 : "memory")
#define WG_RD2(WAIT, R0, R1, O)                                                               \
    asm volatile(WAIT "v_accvgpr_read_b32 %0, " R0 "\n\tv_accvgpr_read_b32 %1, " R1           \
                 : "=v"(w[O]), "=v"(w[O + 1])                                                  \
                 :                                                                            \
                 : "memory")
template <int KIND, int NR>
__device__ __forceinline__ void wg_take(float (&x)[NR], float i16_scale) {
    if constexpr (KIND == IN_IQ_F32) {
        float w[2 * NR];
        WG_RD4("s_waitcnt vmcnt(0)\n\t", "a0", "a1", "a2", "a3", 0);
        WG_RD4("", "a4", "a5", "a6", "a7", 4);
        if constexpr (NR >= 6) WG_RD4("", "a8", "a9", "a10", "a11", 8);
        if constexpr (NR >= 8) WG_RD4("", "a12", "a13", "a14", "a15", 12);
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const float a = w[2 * j] * w[2 * j], b = w[2 * j + 1] * w[2 * j + 1];   // gnuradio complex_to_mag_squared: two products, one sum
            x[j] = a + b;
        }
    } else {
        float w[NR];
        WG_RD4("s_waitcnt vmcnt(0)\n\t", "a0", "a1", "a2", "a3", 0);
        if constexpr (NR == 8) WG_RD4("", "a4", "a5", "a6", "a7", 4);
#pragma unroll
        for (int j = 0; j < NR; j++) {
            if constexpr (KIND == IN_I16_SQ) {
                const float sv = i16_to_float(__float_as_int(w[j]), i16_scale);   // (global_load_sshort sign-extends into the register)
                x[j] = sv * sv;
            } else if constexpr (KIND == IN_ENV_F32) {
                x[j] = w[j];          // the envelope itself (what transition_sink.work receives, transition_sink.py:13-18)
            } else {
                x[j] = w[j] * w[j];   // IN_REAL_F32_SQ
            }
        }
    }
}
// the dwords of NR masks into lanes LANE0 .. LANE0 + 2 NR - 1 of pk
#define PLANE_PUT4(pk, m, I0, LANE0)                                                                                                       \
    asm volatile("s_nop 2\n\tv_writelane_b32 %0, %1, %5\n\tv_writelane_b32 %0, %2, %5+1\n\tv_writelane_b32 %0, %3, %5+2\n\tv_writelane_b32 %0, %4, %5+3" \
                 : "+v"(pk)                                                                                                                \
                 : "s"((uint32_t)(m)[I0]), "s"((uint32_t)((m)[I0] >> 32)), "s"((uint32_t)(m)[(I0) + 1]), "s"((uint32_t)((m)[(I0) + 1] >> 32)), "n"(LANE0))
template <int NR>
__device__ __forceinline__ void wg_put_masks(int &pk, const unsigned long long (&m)[NR], std::integral_constant<int, 0>) {
    PLANE_PUT8(pk, m, 0);
    if constexpr (NR == 8) {
        const unsigned long long m2[4] = {m[4], m[5], m[6], m[7]};
        PLANE_PUT8(pk, m2, 8);
    }
}
template <int NR>
__device__ __forceinline__ void wg_put_masks(int &pk, const unsigned long long (&m)[NR], std::integral_constant<int, 1>) {   // ... from lane 2 NR on
    PLANE_PUT8(pk, m, 2 * NR);
    if constexpr (NR == 8) {
        const unsigned long long m2[4] = {m[4], m[5], m[6], m[7]};
        PLANE_PUT8(pk, m2, 2 * NR + 8);
    }
}

// One row of 64 samples (lane l = sample m) classified ONCE against the window sums handed in -- one iteration of row_exact
// (threshold.hip.h) without its loop: the in-place form below iterates over the whole round instead, every row's sums taken from the
// accept masks guessed for the samples before it.  Updates the LOW bookkeeping (w_nl, w_kl) past the row.
__device__ __forceinline__ void wg_row_once(const ThrArgs &A, int lane, int m, float x, double ss_before, int &w_nl, int &w_kl,
                                            unsigned long long &accm, unsigned long long &lowm, unsigned long long &posm) {
    const int mx = A.mx;
    const int rb = m - lane;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    bool lw = false, hg = false;
    classify_one(A, (double)x, ss_before, lw, hg);
    const unsigned long long low_now = __ballot(lw);
    const unsigned long long bn = ~low_now & lane_lt;
    const int nl = bn ? rb + last_set(bn) : w_nl;   // the last sample before this one that is not LOW
    int key = KEY_NONE;
    if (lw) {
        const int p = m - nl;   // 1-based position in the LOW run
        const bool bad = (p > mx) && ((p - 1) % mx == 0);   // (its run ends on a time-out here: transition_sink.py:95-99)
        key = 2 * m + (bad ? 0 : 1);
    }
    const unsigned long long bl = low_now & lane_lt;
    const int kq = __shfl(key, bl ? last_set(bl) : 0, 64);   // the key of the last LOW sample below this lane
    const int kl = bl ? max(kq, w_kl) : w_kl;
    const bool st2 = (kl & 1) && (m - (kl >> 1)) <= mx + 1;   // HIGH is ignored within max_len + 1 of a LOW sample whose key is good
    int v = 0;
    if (lw) v = -1;
    else if (hg && !st2) v = 1;
    accm = __ballot(v == 0);
    lowm = low_now;
    posm = __ballot(v == 1);
    const unsigned long long nonlowf = ~low_now;
    w_nl = nonlowf ? rb + last_set(nonlowf) : w_nl;
    if (low_now) w_kl = max(w_kl, __builtin_amdgcn_readlane(key, last_set(low_now)));
}

// EX (re-runs from the exact state, mode 1): a round that fails its check is not the chunk's end -- every wave puts the ring slots
// of its step back, the four waves evaluate the round the way k_threshold evaluates a step (row_exact of threshold.hip.h: every sample
// against the thresholds of the exact fp64 window sum before it, the accept masks iterated to their fixed point -- here over the whole
// round at once, see the block behind the close), leave the masks and plane words where the round's waves would have, and the chunk
// goes on in the tracked form from the exact sum.  One-round supersteps; rounds of whole steps only (a ragged end, the stream's first
// stable sample: the chunk gives up as before and k_threshold takes it).
// FLG (round 6, VERDICT r5 item 3: built and measured): the first barrier of a round of whole steps is replaced by
// per-wave counters -- a wave publishes its step's LOW masks, bumps its counter and goes on as soon as the waves BEFORE it in the round
// have published this round and the waves BEHIND it the round before (so that every step whose ring slots its own may share -- four and
// more steps back -- is complete, and the mask buffer it writes next is read by nobody).  The close of a superstep keeps its barrier.
template <int KIND, int NR, bool EX = false, bool FLG = false>
__global__ __launch_bounds__(256) void k_threshold_wg(ThrArgs A) {
    static_assert(!EX || NR == 4, "the samples of an exact round travel through sh->acc: sixteen rows");
    static_assert(!(EX && FLG), "the form that re-runs keeps the barriers");
    constexpr uint32_t STEPN = 64u * NR;
    constexpr int WG_ROUND = wg_round_samples(NR);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = rfl(tid >> 6);
    const uint32_t c = A.list ? rfl(A.list[blockIdx.x]) : blockIdx.x;   // (a re-run from the exact state, mode 1: the chunks of a list)
    float *ring = (float *)smem;
    lean_lds_f *const rl = (lean_lds_f *)ring;
    WgShared *const sh = (WgShared *)(smem + (size_t)A.Lpad * 4);
    const int L = A.L;
    const int mx = A.mx;
    uint32_t m_chunk_v, chunk_len_v;
    chunk_span(A, c, m_chunk_v, chunk_len_v);
    const uint32_t m_chunk = rfl(m_chunk_v), chunk_len = rfl(chunk_len_v);
    const uint32_t n1 = min(A.n, m_chunk + chunk_len);
    const uint32_t m_start = max(m_chunk, A.skip);
    constexpr int RB = LeanRaw<KIND>::BYTES;
    const uint32_t wbase0 = m_chunk + STEPN * (uint32_t)wave;   // this wave's step of round 0
    const char *const in_first = (const char *)A.in + (size_t)wbase0 * RB;   // this wave's step of the chunk's first round (uniform)
    // (Measured and dropped, round 5: the scalar base fixed and the LANE offset walking the chunk -- one vector addition and one
    // minimum per round instead of seven scalar instructions -- and the per-step selects of xlo_acc / xhi_acc made per lane, on the
    // compare the ballot needs anyway: 0.1463 -> 0.1482 ms per launch.  Vector issue slots are what this kernel is short of.)
    const uint32_t voff = (uint32_t)lane * (uint32_t)RB;
    // The chunk's first round is asked for NOW when it is a regular one (four whole steps of stable samples): its samples arrive
    // while the state the chunk starts from is put together below -- the first take found them 2-3 us away otherwise, once per chunk,
    // with every workgroup of the launch asking at the same time.  (Into accumulator registers nothing below touches.)
    bool primed = false;
    if (m_chunk >= m_start && m_chunk + (uint32_t)wg_round_samples(NR) <= n1) {
        wg_load_step<KIND, NR>(voff, in_first);
        primed = true;
    }
    const Carry cr = *A.carry;
    uint64_t *const neg_p = A.neg, *const pos_p = A.pos;
    // lanes 0 .. 2 NR - 1 hold the neg plane's dwords of a step, 2 NR .. 4 NR - 1 the pos plane's
    const uintptr_t plane_of_lane = (lane >= 2 * NR) ? (uintptr_t)pos_p : (uintptr_t)neg_p;
    const int plane_dword = (lane >= 2 * NR) ? lane - 2 * NR : lane;
    const float i16s = A.i16_scale;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    int scr_par = 0;
    // sum / max / min of K values over the workgroup (every thread gets the result; uniform)
    auto wg_gather = [&](const uint32_t (&v)[8], int k, uint32_t (&out)[WG_WAVES][8]) __attribute__((always_inline)) {
        if (lane == 0)
            for (int i = 0; i < k; i++) sh->scr[scr_par][wave][i] = v[i];
        wg_barrier();
        for (int w = 0; w < WG_WAVES; w++)
            for (int i = 0; i < k; i++) out[w][i] = rfl(sh->scr[scr_par][w][i]);
        scr_par ^= 1;
    };

    // (steps before the chunk's first one have published nothing: their masks read as "no LOW sample"; the barriers of the
    // prologue order this before the first round)
    for (int i = tid; i < (int)(sizeof(sh->msk) / 4); i += 256) ((uint32_t *)sh->msk)[i] = 0u;
    if (tid < WG_WAVES) sh->pha[tid] = 0u;
    // FLG: rounds done (every wave counts them alike) and, per lane, which wave's counter it looks at and how far that one must have got
    // relative to this -- in VECTOR registers, laundered through asm: the kernel has no scalar register to spare (measured: with these
    // in scalars the round loop gained 77 spill moves and the launch 18 %)
    uint32_t rno = 0u;
    uint32_t pha_addr = (uint32_t)(uintptr_t)(lean_lds_u32 *)&sh->pha[lane & 3];
    uint32_t pha_bias = ((lane & 3) < wave) ? 1u : 0u;                 // the waves before this one in the round: this round; behind it: the round before
    uint32_t pha_self = ((lane & 3) == wave) ? 0xFFFFFFFFu : 0u;       // (its own counter: anything)
    uint32_t pha_mine = (uint32_t)(uintptr_t)(lean_lds_u32 *)&sh->pha[wave];
    if constexpr (FLG) asm volatile("" : "+v"(rno), "+v"(pha_addr), "+v"(pha_bias), "+v"(pha_self), "+v"(pha_mine));
    if (lane == 0) {
        sh->cold[wave][0] = (uint32_t)LL_NONE;
        sh->cold[wave][1] = (uint32_t)LL_NONE;
        sh->cold[wave][2] = 0xFFFFFFFFu;
        sh->cold[wave][3] = 0u;
        sh->cold[wave][4] = sh->cold[wave][5] = 0x7F61B1E6u;   // 3.0e38f
    }
    // does a step's LOW mask hold an aligned block of A.blk LOW samples?  (A LOW run longer than max_len covers one, wherever
    // step seams fall: the last max_len + 1 samples of such a run do.)
    auto blk_hit = [&](const unsigned long long (&m)[NR]) __attribute__((always_inline)) -> bool {
        unsigned long long hit = 0;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            unsigned long long t = m[j];
#pragma unroll
            for (int f = 0; f < 6; f++) t &= t >> A.fold_sh[f];
            hit |= t & A.selmask;
        }
        return hit != 0ull;
    };
    unsigned long long clk0 = 0, clk1 = 0, clk2 = 0;
#ifdef NFC_WG_PROF
    unsigned long long pf_take = 0, pf_b1 = 0, pf_b2 = 0, pf_rounds = 0, pf_t = 0;   // (a profiling build: where a wave waits)
#define WG_PF_BEGIN() pf_t = clock64()
#define WG_PF_END(acc) acc += clock64() - pf_t
#else
#define WG_PF_BEGIN() ((void)0)
#define WG_PF_END(acc) ((void)0)
#endif
    if (A.dbg_clk) clk0 = clock64();

    // ---------------- the state the chunk starts from (chunk_incoming / chunk_save_in of threshold.hip.h, 256 threads wide) ----------------
    uint32_t emin = 255u, emax = 0u, vtop0 = 0u;
    int nl_in, kl_in;
    float ssf, eps = 0.f;
    {
        double ss0;
        if (c == 0) {
            for (int s = tid; s < L; s += 256) ring[s] = A.ring_carry[s];
            ss0 = cr.ss;
            nl_in = carried_nl(A);
            kl_in = carried_kl(A);
            wg_barrier();
        } else if (A.mode == 1) {
            // the exact state, resolved from the predecessors' summaries by look-back (chunk_incoming of threshold.hip.h)
            double part = 0;
            for (int s = tid; s < L; s += 256) {
                const float v = resolve_slot(A, (int)c, s);
                ring[s] = v;
                part += (double)v;
            }
            const double ps = wave_sum_f64(part);
            uint32_t v[8], g[WG_WAVES][8];
            v[0] = (uint32_t)__double2loint(ps);
            v[1] = (uint32_t)__double2hiint(ps);
            wg_gather(v, 2, g);   // (its barrier also completes the ring)
            double tot = 0;
            for (int w = 0; w < WG_WAVES; w++) tot += __hiloint2double((int)g[w][1], (int)g[w][0]);
            ss0 = tot + cr.delta;
            resolve_low_state(A, (int)c, nl_in, kl_in);
        } else {
            // speculate: the L samples before the chunk, rejected-looking ones replaced by a level estimate
            eps = A.eps;
            const uint32_t w0 = m_chunk - (uint32_t)L;
            const uint32_t slot0 = (A.g0modL + w0) % (uint32_t)L;
            float mxv = 0.f;
            for (int i0 = 0; i0 < L; i0 += 2048) {   // eight independent loads in flight per thread
                typename RawOf<KIND>::T rw[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = i0 + 256 * k + tid;
                    rw[k] = (i < L) ? load_raw<KIND>(A.in, (size_t)w0 + i) : raw_zero<KIND>();
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = i0 + 256 * k + tid;
                    if (i < L) {
                        const float x = env_of<KIND>(rw[k], A.i16_scale);
                        uint32_t slot = slot0 + (uint32_t)i;
                        slot = (slot >= (uint32_t)L) ? slot - (uint32_t)L : slot;
                        ring[slot] = x;
                        mxv = fmaxf(mxv, x);
                    }
                }
            }
            uint32_t v[8], g[WG_WAVES][8];
            v[0] = __float_as_uint(wave_max_f32(mxv));   // (envelopes are >= 0: raw bits order like values)
            wg_gather(v, 1, g);                          // (its barrier also completes the ring)
            mxv = __uint_as_float(max(max(g[0][0], g[1][0]), max(g[2][0], g[3][0])));
            const float half = 0.5f * mxv;
            float sa = 0.f, na = 0.f;
            for (int s = tid; s < L; s += 256) {
                const float x = ring[s];
                if (x >= half) { sa += x; na += 1.f; }
            }
            v[0] = __float_as_uint(wave_sum_f32(sa));
            v[1] = __float_as_uint(wave_sum_f32(na));
            wg_gather(v, 2, g);
            sa = ((__uint_as_float(g[0][0]) + __uint_as_float(g[1][0])) + __uint_as_float(g[2][0])) + __uint_as_float(g[3][0]);
            na = ((__uint_as_float(g[0][1]) + __uint_as_float(g[1][1])) + __uint_as_float(g[2][1])) + __uint_as_float(g[3][1]);
            const float ca = (na > 0.f) ? sa / na : mxv;
            float sb = 0.f, nb = 0.f;
            for (int s = tid; s < L; s += 256) {
                const float x = ring[s];
                if (x >= half && x <= ca) { sb += x; nb += 1.f; }
            }
            v[0] = __float_as_uint(wave_sum_f32(sb));
            v[1] = __float_as_uint(wave_sum_f32(nb));
            wg_gather(v, 2, g);
            sb = ((__uint_as_float(g[0][0]) + __uint_as_float(g[1][0])) + __uint_as_float(g[2][0])) + __uint_as_float(g[3][0]);
            nb = ((__uint_as_float(g[0][1]) + __uint_as_float(g[1][1])) + __uint_as_float(g[2][1])) + __uint_as_float(g[3][1]);
            const float c0 = (nb > 0.f) ? sb / nb : ca;
            const float tlo = (float)A.lo * c0, thi = (float)A.hi * c0;
            // ring slot s last saw sample m = w0 + ((s - slot0) mod L)
            int ll = LL_NONE, nl = LL_NONE;
            double part = 0;
            for (int s = tid; s < L; s += 256) {
                const int rel = (s >= (int)slot0) ? s - (int)slot0 : s - (int)slot0 + L;
                const int m = (int)w0 + rel;
                float x = ring[s];
                if (x < tlo) ll = max(ll, m);
                else nl = max(nl, m);
                if (!(x >= tlo && x <= thi)) {
                    x = c0;
                    ring[s] = c0;
                }
                part += (double)x;
            }
            const double ps = wave_sum_f64(part);
            v[0] = (uint32_t)wave_max_i32(ll);
            v[1] = (uint32_t)wave_max_i32(nl);
            v[2] = (uint32_t)__double2loint(ps);
            v[3] = (uint32_t)__double2hiint(ps);
            wg_gather(v, 4, g);
            ll = max(max((int)g[0][0], (int)g[1][0]), max((int)g[2][0], (int)g[3][0]));
            nl_in = max(max((int)g[0][1], (int)g[1][1]), max((int)g[2][1], (int)g[3][1]));
            kl_in = (ll == LL_NONE) ? KEY_NONE : 2 * ll + 1;
            double tot = 0;
            for (int w = 0; w < WG_WAVES; w++) tot += __hiloint2double((int)g[w][3], (int)g[w][2]);
            ss0 = tot + cr.delta;
        }
        ssf = rfl((float)ss0);
        // keep the ring the evaluation starts from (for k_certify), mark every slot untouched, fold the exponent guard
        float *rin = A.ring_in + (size_t)c * L;
        for (int s = tid; s < L; s += 256) {
            const float v = ring[s];
            // (non-temporal, like the ring a chunk ends on below: 16 MB of summaries per batch that only the certification reads, a launch
            // later -- kept out of the L2's way they no longer compete with the planes and the samples, and the launch's end has less to
            // write back: 0.1458 -> 0.1405 ms per launch, same-call A/B, four rounds; the planes themselves the same way: no gain, and
            // the edge stage then misses them)
            __builtin_nontemporal_store(v, rin + s);
            vtop0 = max(vtop0, __float_as_uint(v));
            ring[s] = __uint_as_float(__float_as_uint(v) | 0x80000000u);
            if (v != 0.f) {
                const uint32_t e = max(f32_expfield(v), 1u);
                emin = min(emin, e);
                emax = max(emax, e);
            }
        }
        uint32_t v[8], g[WG_WAVES][8];
        v[0] = wave_max_u32(vtop0);
        v[1] = wave_min_u32(emin);
        v[2] = wave_max_u32(emax);
        wg_gather(v, 3, g);   // (its barrier: every slot carries its sign bit before the first round reads the ring)
        vtop0 = max(max(g[0][0], g[1][0]), max(g[2][0], g[3][0]));
        emin = min(min(g[0][1], g[1][1]), min(g[2][1], g[3][1]));
        emax = max(max(g[0][2], g[1][2]), max(g[2][2], g[3][2]));
    }
    nl_in = rfl(nl_in);
    kl_in = rfl(kl_in);

    if (A.dbg_clk) clk1 = clock64();
    bool good_run = A.fast_ok != 0;
    // Raw envelopes (IN_ENV_F32) come from the caller as they are: a mag^2 envelope is never negative, but nothing says so.  This
    // kernel keeps "no sample of this chunk accepted into the slot" in a ring value's SIGN bit and orders samples by their raw
    // bits, so a negative sample, an infinity or a NaN -- raw bits 0x7F800000 and up -- makes the chunk give up (k_threshold takes
    // it, as it took every chunk of this kind before): in the state it starts from (here), in a round's samples (below).
    if constexpr (KIND == IN_ENV_F32) good_run = good_run && vtop0 < 0x7F800000u;
    uint32_t why = good_run ? 0u : 1u;   // 1 parameters / sums out of range, 2 a sample inside a band, 3 LOW run, 4 allowance, 5 first stable sample
    float min_ss = 3.0e38f;
    uint32_t vmin = 0xFFFFFFFFu, vmax = vtop0;   // (wave 0's: what the superstep closes add; the general steps' share is in sh->cold)
    const float etaD = 1.0f - 9.5367431640625e-07f;  // 1 - 2^-20
    const float slU = 1.0f + 3.814697265625e-06f;
    const float loLf = (float)A.lo_L, hiLf = (float)A.hi_L;
    uint32_t slot_step = (A.g0modL + wbase0) % (uint32_t)L;
    const uint32_t slot_adv = (uint32_t)WG_ROUND % (uint32_t)L;
    float b_acc = 0.f, dl_acc = 0.f;
    float tlo = 0.f, thi = 0.f;   // the thresholds of the superstep in progress, from the sum tracked at its start
    // how close a sample of the superstep came to a threshold -- looked at when it closes, against the drift the window sum
    // turned out to have.  Per-sample distances where a step holds classified samples, the step's extremes otherwise.
    float dlo = 3.0e38f, dhi = 3.0e38f;
    uint32_t xlo_acc = 0x7F7FFFFFu, xhi_acc = 0u;   // (raw bits: envelopes are >= 0)
    int lz_base = LL_NONE;                  // base of this wave's latest step with LOW samples
    int rounds_since_sync = 0;
    // EX: what a round needs to be taken back and evaluated again -- this wave's samples, the raw ring values of its slots, the
    // bookkeeping a step in the general form may have moved; wave 0: the exact state behind a round it evaluated itself
    float xkeep[NR], snap[NR];
    uint32_t snap_slot = 0u, st_off_round = 0u, c_snap[4] = {0u, 0u, 0u, 0u};
    int lz_snap = LL_NONE;
    double ex_ss = 0;
    bool ex_valid = false;
#ifdef NFC_EX_PRINTF
    int ex_count = 0, ex_codes = 0, ex_first = -1, ex_last = -1, n_rounds = 0, ex_trips = 0;
    const unsigned long long ex_t0 = clock64();
    unsigned long long ex_t1 = 0, ex_tx = 0, ex_ta = 0;   // the rounds' start; cycles spent evaluating rounds in place
#endif
#pragma unroll
    for (int j = 0; j < NR; j++) xkeep[j] = snap[j] = 0.f;

    auto open_round = [&]() __attribute__((always_inline)) -> bool {
        tlo = rfl(ssf * loLf);
        thi = rfl(ssf * hiLf);
        if (!(ssf > 1e-30f && ssf < 1e30f && tlo > 1e-30f && thi < 1e30f)) return false;
        min_ss = fminf(min_ss, ssf * (etaD - RND_SUM));
        return true;
    };
    auto fetch_env = [&](uint32_t b, float (&x)[NR]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const uint32_t m = b + 64u * j + lane;
            x[j] = (m < A.n) ? envelope_at<KIND>(A.in, (size_t)m, A.i16_scale) : 0.f;
        }
    };

    // ---------------- the general step on wave masks (phase B) ----------------
    // masked: lanes outside [m_start, n1) are not samples.  before: the last LOW sample before the step (LL_NONE: none in reach);
    // ext_hit: one of the two steps before this one holds an aligned block of LOW samples; own_off / pred_off / ppred_off: where
    // the LOW masks of this step and of those two lie in LDS (byte offsets into sh->msk).
    auto general_step = [&](float (&x)[NR], const uint32_t base, const bool masked, int before, const bool ext_hit, const uint32_t own_off,
                            const uint32_t pred_off, const uint32_t ppred_off, int &pk) __attribute__((always_inline)) -> uint32_t {
        float prev[NR];
        uint32_t slot[NR];
        uint32_t gmin = 0xFFFFFFFFu, gmax = 0u;   // raw bits of what this step accepts
        float gdlo = 3.0e38f, gdhi = 3.0e38f;     // how close its samples come to the thresholds
#pragma unroll
        for (int j = 0; j < NR; j++) {
            uint32_t s = slot_step + 64u * j + lane;
            s = (s >= (uint32_t)L) ? s - (uint32_t)L : s;
            slot[j] = s;
            prev[j] = fabsf(ring[s]);
        }
        unsigned long long unt[NR], am[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) {
            unt[j] = 0ull;
            am[j] = ~0ull;
        }
        if (masked) {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const uint32_t m = base + 64u * j + lane;
                const bool inact = (m < m_start) || (m >= n1);
                const bool untouched = (__float_as_uint(ring[slot[j]]) >> 31) != 0u;
                unt[j] = __ballot(inact && untouched);
                am[j] = __ballot(!inact);
                if (inact) x[j] = prev[j];
            }
        }
        unsigned long long lowm[NR], posm[NR], anylow = 0;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const bool act = (am[j] >> lane) & 1ull;
            lowm[j] = __ballot(x[j] < tlo) & am[j];
            posm[j] = __ballot(x[j] > thi) & am[j];
            gdlo = fminf(gdlo, act ? fabsf(x[j] - tlo) : 3.0e38f);
            gdhi = fminf(gdhi, act ? fabsf(x[j] - thi) : 3.0e38f);
            anylow |= lowm[j];
        }
        if (anylow && masked && base < m_start) return 5u;   // (a LOW run across the first stable sample: leave it to the exact kernel)
        // A HIGH sample within max_len + 1 of a LOW sample is ignored UNLESS that LOW sample ended its run on a time-out
        // (transition_sink.py:95-99: the sample at which dur exceeds max_len resets the state): sample s + k max_len of a run that
        // starts at s, k >= 1.  Only a run longer than max_len has one: while no LOW run in reach can be that long -- no aligned
        // block of LOW samples in this step or the two before it, and the chunk does not start inside a run -- every key is good.
        // Otherwise the run's start is looked up, per HIGH sample that depends on it, in the LOW masks of those three steps
        // (before the chunk: the last sample that was not LOW according to the state the chunk starts from).
        const int carry_in = (int)m_chunk - 1 - nl_in;
        bool unknown_key = false;
        const bool exact_keys = blk_hit(lowm) || ext_hit || (carry_in > 0 && (int)base - 2 * (int)STEPN < (int)m_chunk);
        // The LOW masks of the three steps as uniform words (row k covers samples R0 + 64 k ...), and per row the last sample before
        // it that is not LOW: nl_in before the chunk's first sample, LL_NONE when nobody knows.
        const int R0 = (int)base - 2 * (int)STEPN;
        unsigned long long Mx[3 * NR];
        int Px[3 * NR];
        if (exact_keys) {
            int run = LL_NONE;
            bool known = false;   // a row inside the chunk has been seen: `run` is exact from here on
#pragma unroll
            for (int k = 0; k < 3 * NR; k++) {
                const uint32_t so = k < NR ? ppred_off : (k < 2 * NR ? pred_off : own_off);
                const uint32_t *wp = (const uint32_t *)((const char *)&sh->msk[0][0][0] + so) + 2 * (k % NR);
                const bool inside = R0 + 64 * k >= (int)m_chunk;
                if (inside && !known) {
                    run = (R0 <= (int)m_chunk) ? nl_in : LL_NONE;   // (the window reaches back to the chunk's first sample, or nobody knows)
                    known = true;
                }
                Mx[k] = inside ? ((unsigned long long)rfl(wp[0]) | ((unsigned long long)rfl(wp[1]) << 32)) : ~0ull;   // (before the chunk: nothing to find)
                Px[k] = run;
                const unsigned long long nonlow = ~Mx[k];
                run = nonlow ? R0 + 64 * k + last_set(nonlow) : run;
            }
        }
        // (m is uniform: the lanes that depend on the same LOW sample are served together, see below)
        auto key_state = [&](const int m) __attribute__((always_inline)) -> int {   // 0 good, 1 ended on a time-out, 2 cannot tell
            if (m < (int)m_chunk) return 0;   // (a run that ended before the chunk: the incoming key says it, and it is certified)
            const int q = (m - R0) >> 6, bit = (m - R0) & 63;
            unsigned long long row = ~0ull;
            int pq = LL_NONE;
#pragma unroll
            for (int k = 0; k < 3 * NR; k++) {
                row = (q == k) ? Mx[k] : row;
                pq = (q == k) ? Px[k] : pq;
            }
            const unsigned long long nonlow = ~row & ((1ull << bit) - 1ull);   // below m in its row
            const int s1 = nonlow ? m - bit + last_set(nonlow) : pq;           // the last sample before m that is not LOW
            if (s1 == LL_NONE) return 2;   // (three steps of LOW samples, or a chunk that starts from an unknown run)
            const int koff = m - (s1 + 1);
            return (koff > 0 && (koff % mx) == 0) ? 1 : 0;
        };
#pragma unroll
        for (int j = 0; j < NR; j++) {
            // HIGH is ignored within max_len + 1 samples after a LOW sample (whose key is good)
            const int rb = (int)(base + 64u * j);
            const unsigned long long below = lowm[j] & lane_lt;
            const int lastlow = below ? rb + last_set(below) : before;
            bool ign = (rb + lane - lastlow) <= mx + 1;
            if (exact_keys) {
                // the lanes whose outcome depends on a key, served one LOW sample at a time (uniform control flow)
                const bool want = (x[j] > thi) && ign;
                int ks = 0;
                unsigned long long pend = __ballot(want);
                while (pend) {
                    const int m = __builtin_amdgcn_readlane(lastlow, __ffsll((long long)pend) - 1);
                    const int st = key_state(m);
                    const bool mine = want && lastlow == m;
                    ks = mine ? st : ks;
                    pend &= ~__ballot(mine);
                }
                unknown_key = unknown_key || ks == 2;
                ign = ign && ks == 0;
            }
            const bool ps = (x[j] > thi) && !ign;
            const bool a = !(x[j] < tlo) && !ps;
            const float t = x[j] - prev[j];
            if (a) {
                b_acc += fabsf(t);
                dl_acc += t;
                ring[slot[j]] = x[j];
            }
            const uint32_t xb = __float_as_uint(x[j]);
            gmin = min(gmin, (a && xb != 0u) ? xb : 0xFFFFFFFFu);
            gmax = max(gmax, a ? xb : 0u);
            posm[j] = __ballot(ps);
            posm[j] &= am[j];
            before = lowm[j] ? rb + last_set(lowm[j]) : before;
        }
        if (__ballot(unknown_key)) return 3u;   // (the ring has been written: nothing of this chunk stands anyway)
        int step_nl = LL_NONE, step_ll = LL_NONE;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int rb = (int)(base + 64u * j);
            const unsigned long long nonlow = ~lowm[j] & am[j];
            if ((unt[j] >> lane) & 1ull) ring[slot[j]] = __uint_as_float(__float_as_uint(x[j]) | 0x80000000u);
            step_ll = lowm[j] ? rb + last_set(lowm[j]) : step_ll;
            step_nl = nonlow ? rb + last_set(nonlow) : step_nl;
        }
        {   // (sh->cold: this wave's own slots)
            const uint32_t wmin = wave_min_u32(gmin), wmax = wave_max_u32(gmax);
            const float wdlo = wg_wave_min_f32(gdlo), wdhi = wg_wave_min_f32(gdhi);
            if (lane == 0) {
                sh->cold[wave][4] = __float_as_uint(fminf(__uint_as_float(sh->cold[wave][4]), wdlo));
                sh->cold[wave][5] = __float_as_uint(fminf(__uint_as_float(sh->cold[wave][5]), wdhi));
                if (step_ll != LL_NONE) sh->cold[wave][0] = (uint32_t)step_ll;
                if (step_nl != LL_NONE) sh->cold[wave][1] = (uint32_t)step_nl;
                sh->cold[wave][2] = min(sh->cold[wave][2], wmin);
                sh->cold[wave][3] = max(sh->cold[wave][3], wmax);
            }
        }
        lz_base = rfl(anylow ? (int)base : lz_base);
#pragma unroll
        for (int k = 0; k < NR; k++) {
            PLANE_PUT(pk, lowm[k], 2 * k);
            PLANE_PUT(pk, (lowm[k] >> 32), 2 * k + 1);
            PLANE_PUT(pk, posm[k], 2 * NR + 2 * k);
            PLANE_PUT(pk, (posm[k] >> 32), 2 * NR + 2 * k + 1);
        }
        return 0u;
    };

    // ---------------- the rounds ----------------
    // A SUPERSTEP is `sup` regular rounds classified against one set of thresholds (A.ksteps), or ONE round that is not four whole
    // steps of stable samples (the stream's first stable sample, a batch's ragged end): its
    // rounds are separated by the first barrier only, the second one and the exchange behind it close the superstep.
    const int sup = EX ? 1 : max(1, A.ksteps);   // the longest superstep
    int cur_sup = 1;                    // rounds of the next one (it adapts: see the close)
    bool need_open = true;
    uint32_t rbase = m_chunk;   // base of the round
    // which of the three mask buffers this round publishes in, and the round before it did (byte offsets of this wave's row)
    uint32_t mo = 0u, mo_prev = 2u * (uint32_t)sizeof(sh->msk[0]);
    uint32_t fail = good_run ? 0u : 1u;
    // the plane words of a regular round leave one round later, beside the next request for samples (a store between a request
    // and its use would be waited for with it)
    constexpr int FR = wg_flush_rounds(NR);
    constexpr uint32_t PST_ROUND = (uint32_t)(WG_WAVES * 2 * NR);     // dwords of a round in one plane's staging
    // the staging holds A.wg_stage_rounds rounds per plane: the whole chunk (bulk: nothing leaves before the chunk is done), or 2 FR
    const int st_cap = A.wg_stage_rounds;
    const bool bulk = (uint32_t)st_cap * (uint32_t)WG_ROUND >= chunk_len + (uint32_t)WG_ROUND;
    const uint32_t PST_PLANE = (uint32_t)st_cap * PST_ROUND;          // dwords of one plane's staging
    lean_lds_u32 *const pst = (lean_lds_u32 *)(uint32_t *)(smem + (size_t)A.Lpad * 4 + WG_SHARED_BYTES);   // (LDS addresses: 32-bit arithmetic)
    const int st_wrap = bulk ? st_cap : 2 * FR;
    const uint32_t st_wrap_off = (uint32_t)st_wrap * PST_ROUND;
    uint32_t st_off = 0u;                 // the slot the next round stages in, as a dword offset into a plane's staging
    int st_a = 0, st_cnt = 0;             // the oldest staged slot (0 or FR), rounds staged
    uint32_t st_base = 0u;                // the sample base of the oldest staged round
    int st_flusher = 0;                   // the wave that sends the next block off (they take turns)
    // where this lane's dword of a step goes in the staging ring (its plane, its wave's row)
    const uint32_t st_lane = ((lane >= 2 * NR) ? PST_PLANE : 0u) + (uint32_t)wave * (uint32_t)(2 * NR) + (uint32_t)plane_dword;
    const char *in_wave = in_first;   // this wave's step of the round in progress
    uint32_t hot_last = 0u;   // base of this wave's step in the last regular round done
    int hot_done = 0;         // regular rounds done
    {
        // what the step before the chunk's first one "published": the last LOW sample before the chunk, if it is in reach
        if (tid <= 4 * NR) {
            uint32_t wv = 0u;
            const int p = kl_in >> 1, q = p - ((int)m_chunk - (int)STEPN);
            const bool live = (kl_in & 1) && q >= 0 && q < (int)STEPN;
            if (live && tid == (q >> 5)) wv = 1u << (q & 31);
            if (tid == 4 * NR) wv = live ? 1u : 0u;
            sh->msk[2][WG_WAVES - 1][tid] = wv;
        }
        // (the first round's first barrier orders this before any read)
    }
    // FR staged rounds from slot a (0 or FR: contiguous in the ring) to both planes: 16 bytes per lane
    auto flush_block = [&](const int a, const uint32_t base) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const uintptr_t g = (uintptr_t)(pl ? pos_p : neg_p) + 8 * (uintptr_t)(base >> 6);
            const lean_lds_u32 *src = pst + (pl ? PST_PLANE : 0u) + (uint32_t)a * PST_ROUND;
#pragma unroll
            for (uint32_t i0 = 0; i0 < (uint32_t)FR * PST_ROUND; i0 += 256u) {
                const uint32_t i = i0 + 4u * (uint32_t)lane;
                if (i < (uint32_t)FR * PST_ROUND) {
                    const lean_u32x4 v = *(const lean_lds_u128 *)(src + i);
                    *(lean_g_u128 *)(g + 4 * (uintptr_t)i) = v;
                }
            }
        }
    };
    // everything still staged (before a round that is not regular, at the chunk's end): wave 0, once every wave has staged its
    // words -- the caller's condition is uniform
    auto flush_planes = [&]() __attribute__((always_inline)) {
        if (st_cnt) {
            wg_barrier();
            if (bulk) {
                // (the chunk's planes at once, every thread 16 bytes at a time: slots 0 .. st_cnt - 1, contiguous per plane)
                const uint32_t ndw = (uint32_t)st_cnt * PST_ROUND;
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const uintptr_t g = (uintptr_t)(pl ? pos_p : neg_p) + 8 * (uintptr_t)(st_base >> 6);
                    const lean_lds_u32 *src = pst + (pl ? PST_PLANE : 0u);
                    for (uint32_t i = 4u * (uint32_t)tid; i < ndw; i += 1024u) {
                        const lean_u32x4 v = *(const lean_lds_u128 *)(src + i);
                        *(lean_g_u128 *)(g + 4 * (uintptr_t)i) = v;
                    }
                }
            } else if (wave == 0) {
                int slot = st_a;
                uint32_t base = st_base;
                for (int r = 0; r < st_cnt; r++) {
                    for (uint32_t e = (uint32_t)lane; e < 2u * PST_ROUND; e += 64u) {   // (plane, wave, dword) of the round
                        const uint32_t pl = e / PST_ROUND, wd = e % PST_ROUND;
                        const uintptr_t g = (uintptr_t)(pl ? pos_p : neg_p) + 8 * (uintptr_t)(base >> 6) + 4 * (uintptr_t)wd;
                        *(lean_g_u32 *)g = pst[pl * PST_PLANE + (uint32_t)slot * PST_ROUND + wd];
                    }
                    slot = (slot + 1 == st_wrap) ? 0 : slot + 1;
                    base += (uint32_t)WG_ROUND;
                }
            }
            st_cnt = 0;
            st_a = 0;
            st_off = 0u;   // (the next regular round stages behind a round's barrier: wave 0 is through by then)
        }
    };
#ifdef NFC_EX_PRINTF
    ex_t1 = clock64();
#endif
    while (good_run && rbase < n1) {
        const bool regular = rbase >= m_start && rbase + (uint32_t)WG_ROUND <= n1;
        int nr = 1;
        uint32_t whole = 0u;   // regular rounds from here on
        if (regular) {
            whole = (n1 - rbase) / (uint32_t)WG_ROUND;
            nr = (int)min((uint32_t)cur_sup, whole);
            if (!primed) {
                // this wave's step of the first regular round is asked for
                wg_load_step<KIND, NR>(voff, in_wave);
                primed = true;
                need_open = true;
            }
        } else {
            flush_planes();
            primed = false;
        }
        if (need_open) {   // (otherwise wave 0 opened the superstep when it closed the one before)
            if (!open_round() && !fail) fail = 1u;
        }
        need_open = !regular;

        // One round.  REG (compile time): four whole steps of stable samples, asked for a round ahead -- the loop below runs the
        // regular rounds of a superstep through an instantiation that holds nothing of the other kind (round 5: with `regular` a run-
        // time flag the compiler kept one merged body and steered it with flag registers -- 80 scalar instructions per round, most
        // of them moves and tests of those flags); the rounds that are not (the stream's first stable sample, a ragged end) take the
        // masked general form in every wave, on synchronously loaded samples.
        auto one_round = [&](auto reg_tag, const int k) __attribute__((always_inline)) {
            constexpr bool REG = decltype(reg_tag)::value;
            const uint32_t base = rbase + STEPN * (uint32_t)wave;
            float x[NR];
            int pk = 0;
            unsigned long long lowany, highany;
            // ---- phase A: this step's envelopes, what can classify at all, its LOW masks for the step after it ----
            if constexpr (REG) {
                WG_PF_BEGIN();
                wg_take<KIND, NR>(x, i16s);
                WG_PF_END(pf_take);
                // the plane words of the round before leave, and the registers take the next round (if the chunk has one: past
                // its end may be past the caller's buffer)
                // (the plane words of the rounds before: staged in LDS, see flush_block)
                // (FLG: a wave may be a round behind -- what it staged two rounds ago is complete)
                if (!bulk && st_cnt > FR + (FLG ? 1 : 0)) {   // (uniform) the FR oldest staged rounds are complete in every wave: one wave sends them off
                    if (wave == st_flusher) flush_block(st_a, st_base);
                    st_a = (st_a == FR) ? 0 : FR;
                    st_base += (uint32_t)(FR * WG_ROUND);
                    st_cnt -= FR;
                    st_flusher = (st_flusher + 1) & (WG_WAVES - 1);
                }
                // (asked for unconditionally; in the chunk's last regular round this round's samples are asked for again and the
                // values are never used)
                wg_load_step<KIND, NR>(voff, (uint32_t)(k + 1) < whole ? in_wave + (size_t)WG_ROUND * RB : in_wave);
                uint32_t xlo = __float_as_uint(x[0]), xhi = __float_as_uint(x[0]);   // (envelopes are >= 0: their raw bits order like their values)
#pragma unroll
                for (int j = 1; j < NR; j++) {
                    xlo = min(xlo, __float_as_uint(x[j]));
                    xhi = max(xhi, __float_as_uint(x[j]));
                }
                if constexpr (KIND == IN_ENV_F32) {   // (raw envelopes: see good_run above)
                    if (__ballot(xhi >= 0x7F800000u) && !fail) fail = 1u;
                }
                const float xmin = __uint_as_float(xlo), xmax = __uint_as_float(xhi);
                lowany = __ballot(xmin < tlo);
                highany = __ballot(xmax > thi);
                // (a step without LOW samples is as far from the LOW threshold as its smallest sample, one without HIGH samples as
                // far from the HIGH threshold as its largest: the steps that have them measure every sample, below)
                xlo_acc = lowany ? xlo_acc : min(xlo_acc, xlo);
                xhi_acc = highany ? xhi_acc : max(xhi_acc, xhi);
                if (lowany) {
                    unsigned long long lw[NR];
#pragma unroll
                    for (int j = 0; j < NR; j++) lw[j] = __ballot(x[j] < tlo);
                    wg_put_masks<NR>(pk, lw, std::integral_constant<int, 0>{});
                    PLANE_PUT(pk, 1u, 4 * NR);
                }
            } else {
                fetch_env(base, x);
                unsigned long long lw[NR];
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const uint32_t m = base + 64u * j + lane;
                    lw[j] = __ballot(x[j] < tlo && !((m < m_start) || (m >= n1)));
                }
                wg_put_masks<NR>(pk, lw, std::integral_constant<int, 0>{});
                PLANE_PUT(pk, 1u, 4 * NR);
                lowany = highany = ~0ull;
            }
            if (lane <= 4 * NR) *(uint32_t *)((char *)&sh->msk[0][wave][0] + mo + 4u * (uint32_t)lane) = (uint32_t)pk;
            WG_PF_BEGIN();
            if constexpr (FLG) {
                // (LDS takes a wave's instructions in order: whoever sees the counter sees the masks)
                asm volatile("" ::: "memory");
                if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(pha_mine), "v"(rno + 1u) : "memory");
                if constexpr (REG) {
                    for (;;) {
                        uint32_t pv;
                        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pv) : "v"(pha_addr) : "memory");
                        if (__ballot(((pv | pha_self) >= rno + pha_bias) ? 0 : 1) == 0ull) break;
#if NFC_WG_FLAGS_SLEEP
                        __builtin_amdgcn_s_sleep(NFC_WG_FLAGS_SLEEP);
#endif
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    wg_barrier();
                }
            } else {
                wg_barrier();
            }
            WG_PF_END(pf_b1);
            if constexpr (EX && REG) {
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    uint32_t s = slot_step + 64u * j + lane;
                    s = (s >= (uint32_t)L) ? s - (uint32_t)L : s;
                    xkeep[j] = x[j];
                    snap[j] = ring[s];
                }
                snap_slot = slot_step;
                st_off_round = st_off;
                lz_snap = lz_base;
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 4; i++) c_snap[i] = sh->cold[wave][i];
                }
            }

            // ---- phase B: the ring, the drift accumulators, the HIGH plane ----
            // which form: nothing classifies / only LOW / only HIGH with no LOW sample in reach / the general step.
            bool general = !REG;
            int before = LL_NONE;   // the last LOW sample before this step, if it can matter
            bool ext_hit = false;   // the two steps before this one may hold a LOW run longer than max_len
            // where the LOW masks of the step before this one and of the one before that lie: the waves before this one in the
            // round, or the last waves of the round before
            const uint32_t wrow = (uint32_t)sizeof(sh->msk[0][0]);
            if (highany) {
                const uint32_t pred_off = (wave == 0) ? mo_prev + (uint32_t)(WG_WAVES - 1) * wrow : mo + (uint32_t)(wave - 1) * wrow;
                if (!REG && base == m_chunk) {   // a chunk's first step in the general form: what the speculation (chunk 0: the carried state) says
                    before = (kl_in & 1) ? (kl_in >> 1) : LL_NONE;
                } else {
                    const uint32_t *pm = (const uint32_t *)((const char *)&sh->msk[0][0][0] + pred_off);
                    general = general || lowany != 0ull;
                    if (rfl(pm[4 * NR]) || lowany) {
                        const uint32_t ppred_off = (wave >= 2) ? mo + (uint32_t)(wave - 2) * wrow : mo_prev + (uint32_t)(wave + 2) * wrow;
                        unsigned long long pmk[NR], ppk[NR];
                        const uint32_t *pp = (const uint32_t *)((const char *)&sh->msk[0][0][0] + ppred_off);
                        const int pb = (int)base - (int)STEPN;
#pragma unroll
                        for (int j = 0; j < NR; j++) {
                            pmk[j] = (unsigned long long)rfl(pm[2 * j]) | ((unsigned long long)rfl(pm[2 * j + 1]) << 32);
                            ppk[j] = (unsigned long long)rfl(pp[2 * j]) | ((unsigned long long)rfl(pp[2 * j + 1]) << 32);
                            before = pmk[j] ? pb + 64 * j + last_set(pmk[j]) : before;
                        }
                        if (before != LL_NONE && ((int)base - before) <= mx + 1) general = true;
                        else before = LL_NONE;
                        if (general) ext_hit = blk_hit(pmk) || blk_hit(ppk);
                    }
                }
            }
            if (__builtin_expect(general, 0)) {
                if (base < n1 && !fail) {
                    const uint32_t own_off = mo + (uint32_t)wave * wrow;
                    const uint32_t pred_off = (wave == 0) ? mo_prev + (uint32_t)(WG_WAVES - 1) * wrow : mo + (uint32_t)(wave - 1) * wrow;
                    const uint32_t ppred_off = (wave >= 2) ? mo + (uint32_t)(wave - 2) * wrow : mo_prev + (uint32_t)(wave + 2) * wrow;
                    fail = general_step(x, base, !REG, before, ext_hit, own_off, pred_off, ppred_off, pk);
                    if constexpr (!REG) {
                        const uint32_t w = (base >> 6) + (uint32_t)(plane_dword >> 1);
                        // this lane's dword of its wave's plane words of the step
                        const uintptr_t pl_addr = plane_of_lane + 4 * (2 * (uintptr_t)(base >> 6) + (uintptr_t)plane_dword);
                        if (!fail && lane < 4 * NR && (size_t)w * 64 < A.n) *(lean_g_u32 *)pl_addr = (uint32_t)pk;
                    }
                }
            } else {
                // straight-line forms: ring addresses (a step wraps the ring once in L / 256 steps)
                lean_lds_f *pa[NR];
                {
                    const uint32_t s0 = slot_step + (uint32_t)lane;
                    if (slot_step + STEPN <= (uint32_t)L) {
#pragma unroll
                        for (int j = 0; j < NR; j++) pa[j] = rl + s0 + 64u * j;
                    } else {
#pragma unroll
                        for (int j = 0; j < NR; j++) {
                            const uint32_t s = s0 + 64u * j;
                            pa[j] = rl + min(s, s - (uint32_t)L);
                        }
                    }
                }
                float praw[NR];
#pragma unroll
                for (int j = 0; j < NR; j++) praw[j] = *pa[j];
                if (highany) {
                    // HIGH samples only, no LOW sample in reach: all of them are rejected (transition_sink.py:71-74)
                    unsigned long long hw[NR];
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        const float dx = x[j] - thi;
                        const bool hi = dx > 0.f;
                        hw[j] = __ballot(hi);
                        const float val = hi ? praw[j] : x[j];
                        const float ts = fabsf(val) - fabsf(praw[j]);
                        b_acc += fabsf(ts);
                        dl_acc += ts;
                        dhi = fminf(dhi, fabsf(dx));
                        *pa[j] = val;
                    }
                    wg_put_masks<NR>(pk, hw, std::integral_constant<int, 1>{});
                } else if (lowany) {
                    // LOW samples only: rejected ones keep their slot (value and sign bit)
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        const float dx = x[j] - tlo;
                        const bool lo = dx < 0.f;
                        const float val = lo ? praw[j] : x[j];
                        const float ts = fabsf(val) - fabsf(praw[j]);   // 0 for a rejected sample
                        b_acc += fabsf(ts);
                        dl_acc += ts;
                        dlo = fminf(dlo, fabsf(dx));
                        *pa[j] = val;
                    }
                    lz_base = (int)base;
                } else {
                    // nothing classifies: every sample is accepted
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        const float t = x[j] - fabsf(praw[j]);
                        b_acc += fabsf(t);
                        dl_acc += t;
                        *pa[j] = x[j];
                    }
                }
            }
            if constexpr (REG) {
                if (lane < 4 * NR) pst[st_lane + st_off] = (uint32_t)pk;
                st_off += PST_ROUND;
                st_off = (st_off == st_wrap_off) ? 0u : st_off;
                st_cnt++;
            }
            in_wave += (size_t)WG_ROUND * RB;
            slot_step += slot_adv;
            slot_step = (slot_step >= (uint32_t)L) ? slot_step - (uint32_t)L : slot_step;
#ifdef NFC_WG_PROF
            pf_rounds++;
#endif
            rbase += (uint32_t)WG_ROUND;
            if constexpr (FLG) asm volatile("v_add_u32 %0, 1, %0" : "+v"(rno));
            mo_prev = mo;
            mo = (mo == 2u * (uint32_t)sizeof(sh->msk[0])) ? 0u : mo + (uint32_t)sizeof(sh->msk[0]);
        };
        using RegT = std::integral_constant<bool, true>;
        using IrrT = std::integral_constant<bool, false>;
        if (regular) {
            if (st_cnt == 0) st_base = rbase;   // (the staging is empty: the superstep's first round is its oldest)
            for (int k = 0; k < nr; k++) one_round(RegT{}, k);
            // (what the chunk's summary asks of the regular rounds: how many, and this wave's step of the last of them)
            hot_done += nr;
            hot_last = rbase - (uint32_t)WG_ROUND + STEPN * (uint32_t)wave;
        } else {
            one_round(IrrT{}, 0);
        }
        rounds_since_sync += nr;

        // ---- the superstep closes: did every sample keep clear of the thresholds by more than the window sum drifted?  Every lane
        // hands in its sums and distances and every wave its verdict; WAVE 0 adds them up, moves the tracked sum on, chooses the
        // next superstep's length and opens it (a round of whole steps is taken for granted: anything else opens again above) --
        // the others only read the thresholds it leaves behind.
        {
            // every lane hands in (sum |x - prev|, sum (x - prev), its smallest distance to the LOW threshold, to the HIGH threshold)
            float dl2 = fminf(dlo, __uint_as_float(xlo_acc) - tlo), dh2 = fminf(dhi, thi - __uint_as_float(xhi_acc));
            if (lane == 0) {   // (what the steps in the general form measured: sh->cold)
                dl2 = fminf(dl2, __uint_as_float(sh->cold[wave][4]));
                dh2 = fminf(dh2, __uint_as_float(sh->cold[wave][5]));
                sh->cold[wave][4] = sh->cold[wave][5] = 0x7F61B1E6u;
            }
            sh->acc[wave][lane] = make_float4(b_acc, dl_acc, dl2, dh2);
            b_acc = 0.f;
            dl_acc = 0.f;
            dlo = 3.0e38f;
            dhi = 3.0e38f;
            xlo_acc = 0x7F7FFFFFu;
            xhi_acc = 0u;
            if (lane == 0) sh->flag[wave] = fail;
            const bool resync = rounds_since_sync >= 64;   // bound the rounding the f32 sum accumulates: re-derive it from the ring
            if (resync) rounds_since_sync = 0;
            WG_PF_BEGIN();
            wg_barrier();
            WG_PF_END(pf_b2);
            if (wave == 0) {
                const float4 a0 = sh->acc[0][lane], a1 = sh->acc[1][lane], a2 = sh->acc[2][lane], a3 = sh->acc[3][lane];
                const float Bs = wave_sum_f32((a0.x + a1.x) + (a2.x + a3.x)) * 1.001f;
                const float Dt = wave_sum_f32((a0.y + a1.y) + (a2.y + a3.y));
                const float dlmin = wg_wave_min_f32(fminf(fminf(a0.z, a1.z), fminf(a2.z, a3.z)));
                const float dhmin = wg_wave_min_f32(fminf(fminf(a0.w, a1.w), fminf(a2.w, a3.w)));
                const uint4 fl = *(const uint4 *)&sh->flag[0];
                uint32_t f = rfl(fl.x | fl.y | fl.z | fl.w);
                // Every partial sum of the accepted (x - prev), in stream order, lies in [-N, P]: N / P the sums of the negative /
                // the positive ones -- (Bs -/+ Dt) / 2.  So every window sum of the superstep lay within M of the tracked one
                // (eps: what the speculated incoming ring may be off by; RND: the rounding of the f32 tracking), every true
                // threshold within that share of the one used -- and the classifications are the reference's if no sample came
                // closer to it (by induction over the samples: while those so far are right, so is the drift bound).
                const float B = 0.5f * (Bs + fabsf(Dt)) * 1.0001f;
                const float M = (B + (eps + RND_SUM) * ssf) * slU + ssf * 7.62939453125e-06f;   // (+ 2^-17 of the sum: the f32 thresholds)
                const float need_lo = M * loLf, need_hi = M * hiLf;
                if (!f && !(dlmin > need_lo && dhmin > need_hi && M < 0.25f * ssf)) f = (dlmin > need_lo && dhmin > need_hi) ? 4u : 2u;
                if constexpr (EX) {
                    // a round of whole steps that drifted too far, came too close or met a LOW run it could not measure: evaluated again below
#ifdef NFC_EX_PRINTF
                    n_rounds++;
                    if (regular && (f == 2u || f == 3u || f == 4u)) {
                        ex_count++;
                        ex_codes |= 1 << f;
                        if (ex_first < 0) ex_first = n_rounds - 1;
                        ex_last = n_rounds - 1;
                    }
#endif
                    if (regular && (f == 2u || f == 3u || f == 4u)) f = 0x40u;
                    if (lane == 0) sh->bc[4] = Dt;   // (what the round added to the sum as it was tried: the first guess below leans on it)
                }
                int next_sup = cur_sup;
                if (!f) {
                    // every window sum of the superstep lay below ssf + M, and what it accepted between the thresholds
                    vmax = max(vmax, __float_as_uint((ssf + M) * slU));
                    vmin = min(vmin, __float_as_uint(tlo));
                    // the next superstep: longer while the samples keep well clear of what the drift needs, shorter when they come close
                    // (twice as long drifts twice as far: doubled when the distances seen would still be twice what that needs,
                    // halved when they are within half again of what this one needed)
                    const float M2 = (2.f * B + (eps + RND_SUM) * ssf) * slU + ssf * 7.62939453125e-06f;
                    const float head = fminf(dlmin / fmaxf(need_lo, 1e-30f), dhmin / fmaxf(need_hi, 1e-30f));
                    const float head2 = fminf(dlmin / fmaxf(M2 * loLf, 1e-30f), dhmin / fmaxf(M2 * hiLf, 1e-30f));
                    if (regular) next_sup = head2 > 2.f ? min(2 * cur_sup, sup) : (head < 1.5f ? max(1, cur_sup / 2) : cur_sup);
                    ssf = rfl(ssf + Dt);
                    if (resync) {   // (the ring is quiescent: the other waves wait for the verdict)
                        double part = 0;
#pragma unroll 8
                        for (int s2 = lane; s2 < L; s2 += 64) part += (double)fabsf(ring[s2]);
                        ssf = (float)rfl(wave_sum_f64(part) + cr.delta);
                    }
                    if (!open_round()) f = 1u;
                }
                if (lane == 0) *(float4 *)&sh->bc[0] = make_float4(tlo, thi, ssf, __uint_as_float(f | ((uint32_t)next_sup << 8)));
            }
            wg_barrier();
            float4 t4 = *(const float4 *)&sh->bc[0];
            uint32_t fw = rfl(__float_as_uint(t4.w));
            uint32_t f = fw & 0xFFu;
            bool was_exact = false;
            if constexpr (EX) {
                if (f == 0x40u) {
                    // The round is taken back and evaluated the way k_threshold evaluates a step (row_exact of threshold.hip.h: every sample
                    // against the thresholds of the exact fp64 window sum before it, the accept masks iterated to their fixed point) -- by
                    // all four waves at once, and over the whole round at once.  A step's rows need the window sum and the LOW
                    // bookkeeping at its first sample, which the steps before it decide: every wave GUESSES its step's masks (from the
                    // thresholds the round was tried with), hands in what its step would add to the sum under the guess and its LOW masks,
                    // evaluates its rows from what the waves before it handed in, and takes the outcome as its next guess.  When no wave's
                    // masks moved, every step was evaluated from the state its predecessors really leave (by induction from step 0, which
                    // starts from the exact sum): at most five trips, one when the guess was right.
                    was_exact = true;
#ifdef NFC_EX_PRINTF
                    const unsigned long long ex_tb = clock64();
#endif
                    const int rb0 = (int)rbase - WG_ROUND;   // the round's first sample (rbase has moved on)
                    const int sb = rb0 + (int)STEPN * wave;  // this wave's
                    constexpr uint32_t MB = (uint32_t)sizeof(sh->msk[0]);
                    const uint32_t cur = mo_prev, p1 = (cur == 0u) ? 2u * MB : cur - MB, p2 = mo;   // the mask buffers of this round, the one before, the one before that
                    uint32_t slot[NR];
                    float pv[NR];
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        uint32_t s = snap_slot + 64u * j + lane;
                        s = (s >= (uint32_t)L) ? s - (uint32_t)L : s;
                        slot[j] = s;
                        ring[s] = snap[j];
                        pv[j] = fabsf(snap[j]);
                    }
                    lz_base = lz_snap;
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < 4; i++) sh->cold[wave][i] = c_snap[i];
                    }
                    fail = 0u;
                    rounds_since_sync = 0;
                    // the exact sum at the round's first sample: the one the round before left if it was evaluated here, else the ring's
                    double ss0 = ex_ss;
                    if (!ex_valid) {
                        wg_barrier();   // (every slot is back)
                        double part = 0;
                        for (int s2 = tid; s2 < L; s2 += 256) part += (double)fabsf(ring[s2]);
                        const double ps = wave_sum_f64(part);
                        uint32_t v[8], g[WG_WAVES][8];
                        v[0] = (uint32_t)__double2loint(ps);
                        v[1] = (uint32_t)__double2hiint(ps);
                        wg_gather(v, 2, g);
                        double tot = 0;
                        for (int w = 0; w < WG_WAVES; w++) tot += __hiloint2double((int)g[w][1], (int)g[w][0]);
                        ss0 = tot + cr.delta;
                    }
                    double *const exd = (double *)&sh->acc[0][0];        // (sh->acc: read and done with when the verdict was given)
                    uint32_t *const exf = (uint32_t *)&sh->acc[1][0];
                    unsigned long long accG[NR], lowG[NR], posG[NR];
                    // (the first guess: thresholds of a sum that moves evenly through the round by what the round added to it as it was
                    // tried -- a window that is being overwritten with a new level drifts like that, and a guess that is nearly right
                    // saves trips: measured on the level-step capture, 3.7 trips per round with the round's own thresholds)
                    const float dt_round = rfl(sh->bc[4]);
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        const float sg = ssf + dt_round * (((float)(wave * NR + j) + 0.5f) / (float)(4 * NR));
                        lowG[j] = __ballot(xkeep[j] < sg * loLf);
                        accG[j] = ~lowG[j] & ~__ballot(xkeep[j] > sg * hiLf);
                        posG[j] = 0ull;
                    }
                    uint32_t fx = 0u;
                    double S = ss0;
                    float bsum = 0.f;
#pragma unroll
                    for (int j = 0; j < NR; j++) bsum += fabsf(xkeep[j] - pv[j]);
                    bsum = wave_sum_f32(bsum);
                    int trips = 0;
                    (void)trips;
                    for (int trip = 0;; trip++) {
                        trips = trip + 1;
                        // what this step adds to the sum under the guess; its LOW masks where the rounds' steps publish theirs
                        // (per row: the sum before every sample of it under the guess, relative to the row's first; the row's total)
                        double bel[NR], tot[NR];
                        double d = 0;
#pragma unroll
                        for (int j = 0; j < NR; j++) {
                            const double incs = wave_scan_sum_f64(((accG[j] >> lane) & 1ull) ? ((double)xkeep[j] - (double)pv[j]) : 0.0);
                            bel[j] = wave_below_f64(incs);
                            tot[j] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(incs), 63), __builtin_amdgcn_readlane(__double2loint(incs), 63));
                            d += tot[j];
                        }
                        if (lane == 0) {
                            exd[wave] = d;
                            uint32_t *mp = (uint32_t *)((char *)&sh->msk[0][wave][0] + cur);
                            unsigned long long any = 0ull;
#pragma unroll
                            for (int j = 0; j < NR; j++) {
                                mp[2 * j] = (uint32_t)lowG[j];
                                mp[2 * j + 1] = (uint32_t)(lowG[j] >> 32);
                                any |= lowG[j];
                            }
                            mp[4 * NR] = any ? 1u : 0u;
                        }
                        wg_barrier();
                        S = ss0;
                        for (int w = 0; w < wave; w++) S += exd[w];
                        S = rfl(S);
                        // The LOW bookkeeping at the step's first sample, off the LOW masks of the two rounds' worth of steps before it
                        // (lane i: word i of their 8 NR words; the steps of rounds before this one stood their check or were evaluated
                        // here, the steps before it in this round are guesses like its own).  Before the chunk's first sample: what the
                        // chunk started from.
                        int nl, kl = KEY_NONE;
                        {
                            const int W0 = sb - 2 * WG_ROUND;
                            const bool reaches = W0 <= (int)m_chunk;   // the words reach back to the chunk's first sample
                            if (m_start > m_chunk && W0 < (int)m_start) fx = 5u;   // (masked lanes among them: the stream's first chunk)
                            const int Wb = W0 + 64 * lane;
                            const bool valid = lane < 8 * NR && Wb >= (int)m_chunk;
                            unsigned long long lw = 0ull;
                            if (lane < 8 * NR) {
                                const int rel = Wb - (rb0 - 2 * WG_ROUND);   // from the first sample of the round two before this one
                                const int q = rel / WG_ROUND, in_r = (rel - q * WG_ROUND) >> 6;
                                const uint32_t bo = (q == 0) ? p2 : (q == 1 ? p1 : cur);
                                const int wv = in_r / NR, j = in_r & (NR - 1);
                                const uint32_t *wp = (const uint32_t *)((const char *)&sh->msk[0][wv][0] + bo) + 2 * j;
                                lw = (unsigned long long)wp[0] | ((unsigned long long)wp[1] << 32);
                            }
                            const unsigned long long lowv = valid ? lw : 0ull, nonlow = valid ? ~lw : 0ull;
                            nl = wave_max_i32(nonlow ? Wb + last_set(nonlow) : LL_NONE);
                            const int ll = wave_max_i32(lowv ? Wb + last_set(lowv) : LL_NONE);
                            if (nl == LL_NONE) {
                                if (reaches) nl = nl_in;
                                else fx = 3u;   // (two rounds of LOW samples: where the run began is out of sight)
                            }
                            if (ll == LL_NONE) {
                                kl = reaches ? kl_in : KEY_NONE;   // (a LOW sample further back is out of every HIGH sample's reach)
                            } else {
                                // did its run end on a time-out?  The last sample before it that is not LOW says it.
                                unsigned long long below = nonlow;
                                if (Wb > ll) below = 0ull;
                                else if (Wb + 63 >= ll) below &= (1ull << (ll - Wb)) - 1ull;
                                int s1 = wave_max_i32(below ? Wb + last_set(below) : LL_NONE);
                                bool known = true;
                                if (s1 == LL_NONE) {
                                    if (reaches) s1 = nl_in;
                                    else known = false;
                                }
                                const int koff = ll - (s1 + 1);
                                const bool bad = known && koff > 0 && (koff % mx) == 0;
                                kl = 2 * ll + (bad ? 0 : 1);
                                if (!known && sb - ll <= mx + 1) fx = 3u;
                            }
                            nl = rfl(nl);
                            kl = rfl(kl);
                        }
                        // the step's rows from that state: every sample classified once against the sum the guess implies before it
                        // (no row iterates on its own any more: the trips do, over the whole round -- a fixed point of all the masks is the
                        // sequential evaluation, by induction over the samples)
                        bool moved = false;
                        double ss = S;
#pragma unroll
                        for (int j = 0; j < NR; j++) {
                            unsigned long long am, lm, pm;
                            wg_row_once(A, lane, sb + 64 * j + lane, xkeep[j], ss + bel[j], nl, kl, am, lm, pm);
                            moved = moved || am != accG[j] || lm != lowG[j];
                            accG[j] = am;
                            lowG[j] = lm;
                            posG[j] = pm;
                            ss = rfl(ss + tot[j]);
                        }
                        if (lane == 0) exf[wave] = (moved ? 1u : 0u) | (fx << 8);
                        wg_barrier();
                        const uint4 xf = *(const uint4 *)&exf[0];
                        const uint32_t all = rfl(xf.x | xf.y | xf.z | xf.w);
                        if (all >> 8) fx = (all >> 8) > 5u ? 3u : (all >> 8);   // (a LOW run out of some wave's sight)
                        if (fx || !(all & 1u)) break;
                        if (trip >= 24) {   // (a cascade -- every sample's outcome moving the next one's: a trip settles one more of them; k_threshold's)
                            fx = 2u;
                            break;
                        }
                    }
                    if (!fx) {
                        // the round as evaluated: the ring, the plane words, the bookkeeping of a step in the general form
                        vmax = max(vmax, __float_as_uint((float)S * slU + bsum * 1.001f));   // (no sum inside the step can exceed this)
                        int step_ll = LL_NONE, step_nl = LL_NONE;
                        int pk = 0;
#pragma unroll
                        for (int j = 0; j < NR; j++) {
                            if ((accG[j] >> lane) & 1ull) {
                                ring[slot[j]] = xkeep[j];
                                const uint32_t xb = __float_as_uint(xkeep[j]);
                                vmax = max(vmax, xb);
                                vmin = min(vmin, xb ? xb : 0xFFFFFFFFu);
                            }
                            const int rb = sb + 64 * j;
                            step_ll = lowG[j] ? rb + last_set(lowG[j]) : step_ll;
                            step_nl = ~lowG[j] ? rb + last_set(~lowG[j]) : step_nl;
                            PLANE_PUT(pk, lowG[j], 2 * j);
                            PLANE_PUT(pk, (lowG[j] >> 32), 2 * j + 1);
                            PLANE_PUT(pk, posG[j], 2 * NR + 2 * j);
                            PLANE_PUT(pk, (posG[j] >> 32), 2 * NR + 2 * j + 1);
                        }
                        if (lane < 4 * NR) pst[st_lane + st_off_round] = (uint32_t)pk;
                        if (lane == 0) {
                            if (step_ll != LL_NONE) sh->cold[wave][0] = (uint32_t)step_ll;
                            if (step_nl != LL_NONE) sh->cold[wave][1] = (uint32_t)step_nl;
                        }
                        double se = ss0;
                        for (int w = 0; w < WG_WAVES; w++) se += exd[w];
                        ex_ss = rfl(se);
                        ex_valid = true;
                        ssf = rfl((float)ex_ss);
                        if (!open_round()) fx = 1u;
                    }
                    f = fx;
                    fw = fx | (1u << 8);
                    t4 = make_float4(tlo, thi, ssf, 0.f);
#ifdef NFC_EX_PRINTF
                    if (tid == 0) ex_trips += trips;
                    ex_tx += clock64() - ex_tb;
#endif
                }
            }
            if (f) {
                why = (f > 5u) ? 2u : f;   // (codes of several waves may be or-ed together: the aid names one at most)
                good_run = false;
            } else {
                tlo = rfl(t4.x);
                thi = rfl(t4.y);
                ssf = rfl(t4.z);
                cur_sup = (int)(fw >> 8);
                if (!was_exact) ex_valid = false;   // (the round stands by its check: the exact state of the round before is history)
            }
        }
    }
#ifdef NFC_EX_PRINTF
    ex_ta = clock64();
#endif
    // (the whole chunk's plane words are in the LDS staging -- every round regular, nothing flushed yet)
    const bool staged_all = bulk && st_cnt > 0 && st_base == m_chunk && (uint32_t)st_cnt * (uint32_t)WG_ROUND == n1 - m_chunk;
    flush_planes();
    // (where the whole chunk's plane words are still in the LDS staging the summary reads them THERE: it need not wait for the stores
    // that have just been asked for -- every workgroup of the launch is at this point at about the same time, and the wait was part of
    // every chunk's last microseconds: 0.1407 -> 0.1397 ms per launch, five alternating rounds)
    if (!staged_all) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
    }
    // neg-plane word W (64 samples from sample 64 W on) of this chunk
    auto neg_word = [&](const uint32_t W) __attribute__((always_inline)) -> unsigned long long {
        if (!staged_all) return neg_p[W];
        const uint32_t rel = W * 64u - m_chunk, r = rel / (uint32_t)WG_ROUND, in_round = rel - r * (uint32_t)WG_ROUND;
        const uint32_t wv = in_round / STEPN, wd_i = (in_round - wv * STEPN) >> 6;
        const uint32_t o = r * PST_ROUND + wv * (uint32_t)(2 * NR) + 2u * wd_i;
        return (unsigned long long)pst[o] | ((unsigned long long)pst[o + 1u] << 32);
    };

    // ---------------- the chunk's summary ----------------
    // LOW bookkeeping at the chunk's end: the last non-LOW sample and the last LOW sample lie in its last steps (a LOW sample
    // further back is out of every HIGH sample's reach: any key that is not live stands for it, certify_block).  Regular rounds
    // keep no such bookkeeping: the neg-plane words this wave stored for its last two steps say it.
    int chunk_nl = LL_NONE, chunk_kl = KEY_NONE;
    {
        bool tail_hit = false;   // an aligned block of LOW samples in this wave's last two regular steps
        // last LOW / non-LOW sample of this wave's steps in the general form (kept exactly over the chunk's last rounds)
        int my_ll = (int)rfl(sh->cold[wave][0]), my_nl = (int)rfl(sh->cold[wave][1]);
        if (good_run && hot_done > 0) {
            const int which = (lane >= NR) ? 1 : 0;   // lanes 0 .. NR - 1 the last regular step's words, NR .. 2 NR - 1 the one a round before
            const int wi = lane - which * NR;
            const uint32_t sb = hot_last - (uint32_t)which * (uint32_t)WG_ROUND;
            unsigned long long wd = 0ull;
            const bool have = lane < 2 * NR && (which == 0 || hot_done > 1);
            if (have) wd = neg_word((sb >> 6) + (uint32_t)wi);
            const int rb = (int)sb + 64 * wi;
            int ll = (have && wd) ? rb + last_set(wd) : LL_NONE;
            int nl = (have && ~wd) ? rb + last_set(~wd) : LL_NONE;
            unsigned long long t = wd;
#pragma unroll
            for (int f = 0; f < 6; f++) t &= t >> A.fold_sh[f];
            tail_hit = __ballot((t & A.selmask) != 0ull) != 0ull;
            my_ll = max(my_ll, wave_max_i32(ll));
            nl = wave_max_i32(nl);
            if (nl == LL_NONE && my_nl == LL_NONE) {
                // (rare: both of them are LOW throughout -- a loss of signal.  The wave looks further back through the words it
                // stored, four at a time, for its last sample that is not LOW.)
                for (int k = 2; k < hot_done && nl == LL_NONE; k++) {
                    const uint32_t sb2 = hot_last - (uint32_t)k * (uint32_t)WG_ROUND;
                    unsigned long long w2 = ~0ull;
                    if (lane < NR) w2 = neg_word((sb2 >> 6) + (uint32_t)lane);
                    nl = wave_max_i32((lane < NR && ~w2) ? (int)sb2 + 64 * lane + last_set(~w2) : LL_NONE);
                }
            }
            my_nl = max(my_nl, nl);
        }
        if (lane == 0) {
            sh->fin[wave][0] = my_ll;
            sh->fin[wave][1] = my_nl;
            sh->fin[wave][2] = lz_base;
            sh->fin[wave][3] = tail_hit ? 1 : 0;
        }
        wg_barrier();
        int ll = LL_NONE, nl = LL_NONE, lz = LL_NONE, th = 0;
        for (int w = 0; w < WG_WAVES; w++) {
            ll = max(ll, rfl(sh->fin[w][0]));
            nl = max(nl, rfl(sh->fin[w][1]));
            lz = max(lz, rfl(sh->fin[w][2]));
            th |= rfl(sh->fin[w][3]);
        }
        chunk_nl = nl;
        if (ll != LL_NONE) chunk_kl = 2 * ll + 1;
        else if (lz != LL_NONE) chunk_kl = 2 * (lz + (int)STEPN - 1) + 1;
        // the key of the chunk's last LOW sample is good unless its run ended on a time-out -- which nobody has measured: if
        // that sample is still in reach of the next chunk and its run may have been longer than max_len, the chunk gives up
        // (round 5: measured, then.  The run's first sample is looked for in the neg-plane words the chunk has stored -- they are all in
        // place behind the barrier above --, up to 64 words back per trip; a run that began before the chunk is measured against the
        // state the chunk starts from, like the runs its steps meet.  A loss of signal across a chunk's end cost the batch a re-run
        // of that chunk by ONE wave otherwise: 0.65 ms for a chunk of 98 304 samples, 0.29 -> 0.93 ms per batch on the stress capture
        // whenever a cut happens to fall into one of its 100 losses of signal.)
        if (good_run && ll != LL_NONE && ((int)n1 - ll) <= mx + 1 && th) {
            int s1 = LL_NONE;   // the last sample before ll that is not LOW
            bool found = false;
            const int w_top = ll >> 6, w_bot = (int)(m_chunk >> 6);
            for (int trip = 0; trip < 4 && !found; trip++) {
                const int wi = w_top - 64 * trip - lane;
                unsigned long long nonlow = 0ull;
                if (wi >= w_bot) {
                    nonlow = ~neg_word((uint32_t)wi);
                    if (wi == w_top) nonlow &= (1ull << (ll & 63)) - 1ull;   // (below ll in its own word)
                }
                const int cand = wave_max_i32(nonlow ? wi * 64 + last_set(nonlow) : LL_NONE);
                if (cand != LL_NONE) {
                    s1 = cand;
                    found = true;
                } else if (w_top - 64 * trip - 63 <= w_bot) {   // the words reach the chunk's first sample: the run came in with the chunk
                    s1 = nl_in;
                    found = nl_in != LL_NONE;
                    break;
                }
            }
            if (found) {
                const int koff = ll - (s1 + 1);
                if (koff > 0 && (koff % mx) == 0) chunk_kl = 2 * ll;   // its run ended on a time-out: the key is not good
            } else {
                good_run = false;
                why = 3u;
            }
        }
    }

#ifdef NFC_EX_PRINTF
    if (EX && tid == 0) printf("chunk %u: %d of %d rounds exact (codes %x, first %d last %d, %d trips) good %d why %u; cycles: prologue %llu rounds %llu (in place %llu) summary-so-far %llu\n", c, ex_count, n_rounds, ex_codes, ex_first, ex_last, ex_trips, (int)good_run, why, ex_t1 - ex_t0, ex_ta - ex_t1, ex_tx, clock64() - ex_ta);
#endif
    if (A.dbg_clk) clk2 = clock64();
    const uint32_t all_robust = good_run ? 1u : 0u;
    if (c == 0 && tid == 0 && A.mode == 0) {   // chunk 0 has no certification of its own: its verdict travels here
        A.cert[0] = good_run ? 1 : 0;
        if (!good_run) atomicAdd(&A.sum->n_fail, 1u);
    }
    {
        // fold the raw-bit extremes of the rounds into the exponent guard
        uint32_t v[8], g[WG_WAVES][8];
        v[0] = min(wave_min_u32(vmin), rfl(sh->cold[wave][2]));
        v[1] = max(wave_max_u32(vmax), rfl(sh->cold[wave][3]));
        // pass 0 fills the buffer the version byte names (0: prepare_batch zeroes them), a re-run the other one
        const int vb_new = (A.mode == 1) ? 1 - (int)A.ver[c] : (int)A.ver[c];
        float *ro = (vb_new ? A.ring_out[1] : A.ring_out[0]) + (size_t)c * L;
        uint32_t *to = (vb_new ? A.touched[1] : A.touched[0]) + (size_t)c * A.twords;
        uint32_t untouched = 0;
        for (int sbase = 64 * wave; sbase < A.twords * 32; sbase += 256) {
            const int s = sbase + lane;
            const float rv = (s < L) ? ring[s] : 0.f;
            const bool t = (s < L) && !(__float_as_uint(rv) >> 31);
            const unsigned long long bal = __ballot(t);
            if (s < L) {
                __builtin_nontemporal_store(fabsf(rv), ro + s);
                if (!t) untouched++;
            }
            const int w = sbase >> 5;
            if (lane == 0) {
                to[w] = (uint32_t)bal;
                if (w + 1 < A.twords) to[w + 1] = (uint32_t)(bal >> 32);
            }
        }
        v[2] = (uint32_t)wave_sum_f32((float)untouched);
        wg_gather(v, 3, g);
        vmin = min(min(g[0][0], g[1][0]), min(g[2][0], g[3][0]));
        vmax = max(max(g[0][1], g[1][1]), max(g[2][1], g[3][1]));
        untouched = g[0][2] + g[1][2] + g[2][2] + g[3][2];
        if (vmax != 0u) {
            emax = max(emax, (vmax >> 31) ? 255u : max((vmax >> 23) & 0xFFu, 1u));
            if (vmin != 0xFFFFFFFFu) emin = min(emin, max((vmin >> 23) & 0xFFu, 1u));
        }
        const uint32_t vtop = __float_as_uint(fmaxf(__uint_as_float(vmax), ssf * (1.0f + eps + RND_SUM)) * 1.0009765625f);
        const uint32_t flags = good_run ? 0u : (4u | (why << 4));
        if (tid == 0) {
            ChunkInfo ci;
            ci.ss_out = (double)ssf;   // informational
            ci.low_key = chunk_kl;
            ci.last_nonlow = chunk_nl;
            ci.emin = emin;
            ci.emax = emax;
            ci.flags = flags;
            ci.n_untouched = untouched;
            (vb_new ? A.info[1] : A.info[0])[c] = ci;
            A.gmin[c] = (uint8_t)emin;
            A.gmax[c] = (uint8_t)emax;
            A.gflags[c] = (uint8_t)(flags | (untouched ? 2u : 0u));
            A.gvtop[c] = vtop;
            RunMeta mt;
            mt.min_ss = min_ss;
            mt.eps = eps;
            mt.nl_in = nl_in;
            mt.kl_in = kl_in;
            mt.all_robust = all_robust;
            mt.pad = 0;
            A.meta[c] = mt;
        }
    }
    if (A.dbg_clk && tid == 0) {
        A.dbg_clk[4 * (size_t)c + 0] = clk0;
        A.dbg_clk[4 * (size_t)c + 1] = clk1;
        A.dbg_clk[4 * (size_t)c + 2] = clk2;
        A.dbg_clk[4 * (size_t)c + 3] = clock64();
    }
#ifdef NFC_WG_PROF
    if (A.dbg_clk && lane == 0) {   // behind the stamps of all chunks: per wave, ticks spent waiting for samples / at the two barriers
        unsigned long long *pf = A.dbg_clk + 4 * (size_t)A.nchunks + 4 * ((size_t)c * WG_WAVES + wave);
        pf[0] = pf_take;
        pf[1] = pf_b1;
        pf[2] = pf_b2;
        pf[3] = pf_rounds;
    }
#endif
}

}  // namespace nfc
