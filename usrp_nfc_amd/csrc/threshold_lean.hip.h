// threshold_lean.hip.h -- pass 0 of the threshold stage (transition_sink.py:55-82) as a lean, purely optimistic kernel.
//
// Same decomposition as k_threshold (threshold.hip.h): one wavefront per time chunk, 4 rows of 64 samples per step (lane l
// holds samples l, 64+l, 128+l, 192+l), the ring of the last L accepted samples in LDS with "untouched" in the sign bit.
// Two things differ.
//
// (1) WHEN the bound of the window sum's drift is established.  k_threshold reduces B = sum |x - prev| over a step BEFORE it
// classifies the step.  Here a SUPERSTEP of PF steps is classified against thresholds fixed at its start,
//
//     M = G + (eps + RND) * ss        G: drift allowance (a guess), eps: certification margin of the speculated incoming
//                                     ring, RND: rounding of the f32 sum tracking
//     LOW   <=  x < (ss - M) * lo / L      not LOW  <=  x > (ss + M) * lo / L
//     HIGH  <=  x > (ss + M) * hi / L      not HIGH <=  x < (ss - M) * hi / L
//
// while every lane only ACCUMULATES |x - prev| and (x - prev) of the samples it accepts; one pair of wave reductions per
// superstep gives B and D.  If B <= G the classifications are the reference's: by induction over the samples of the
// superstep -- while the drift so far is <= G every classification made is exact, so the accepted set is exact, so the drift
// after the next sample is a partial sum of accepted (x - prev), in magnitude <= B <= G.  Then ss += D.
//
// (2) WHERE the work is done.  Measured on MI355X (tools/ubench/issue_rate.hip, 5 waves per SIMD): a scalar instruction costs
// a SIMD 4.2 cycles, a v_cmp into a scalar pair 4.5, a plain vector instruction 2.8 -- a formulation on wave masks (ballots
// combined by s_and / s_or, exec juggling around every conditional store) is the expensive one.  So the common steps stay in
// the lanes: two pre-tests (min / max of the lane's four samples against the bands) pick one of three straight-line forms --
// nothing classifies / only LOW samples / only HIGH samples with no LOW sample in reach -- whose conditional work is v_cndmask
// on the ONE compare the planes need anyway; "a sample inside a band" and "a LOW run that may reach max_len" are tracked as
// per-lane minima and looked at once per superstep.  Everything else (LOW and HIGH in one step, the stream's first stable
// sample, a batch's ragged end, a chunk that starts inside a LOW run) takes the general step on wave masks.
//
// Nothing is ever repaired in place: when a check fails the wave GIVES UP -- it flags its chunk (RunMeta.all_robust = 0;
// chunk 0: cert[0] = 0 and the failure count) and the host re-runs that chunk from the exact state with k_threshold (mode 1),
// exactly as it does for a chunk whose speculation cannot be certified.  The result that stands is always one whose every
// step was proven.
#pragma once
#include "threshold.hip.h"

#include <type_traits>

namespace nfc {

// The raw samples of the steps ahead must sit in registers whose loads the compiler neither waits for nor moves.  A
// compiler-issued load whose result lives across the loop's back edge is copied there, behind a wait for EVERY load in flight;
// an asm load into a compiler-allocated register ("+v") fares no better -- the allocator copies that register between the load
// and its hand-counted wait whenever the control flow gets interesting, i.e. reads it while it is still being loaded.  So the
// samples land in ACCUMULATOR registers named literally (a0 ..: gfx950 loads straight into them; the compiler itself never
// touches them unless it spills -- tools/audit_lean_isa.py checks that it does not) and reach the compiler's registers only
// through the statement that first waits for them: lean_take<K> = s_waitcnt vmcnt(N) + v_accvgpr_read.
// Step k of a superstep owns a[8 k .. 8 k + 7] (IQ: four pairs; the one-dword kinds use the first four).
// (non-temporal: the samples are read once -- threshold_wg.hip.h, NFC_WG_LDPOL)
#define LEAN_LOAD4(OP, R0, R1, R2, R3, STRIDE, ...)                                                                                         \
    asm volatile(OP " " R0 ", %0, off nt\n\t" OP " " R1 ", %0, off offset:%1 nt\n\t" OP " " R2 ", %0, off offset:%2 nt\n\t" OP " " R3 ", %0, off offset:%3 nt" \
                 :                                                                                                                          \
                 : "v"(p), "n"(STRIDE), "n"(2 * STRIDE), "n"(3 * STRIDE)                                                                    \
                 : "memory", __VA_ARGS__)
#define LEAN_CLOB0 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7"
#define LEAN_CLOB1 "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15"
#define LEAN_CLOB2 "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23"
#define LEAN_CLOB3 "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
template <int KIND, int K>
__device__ __forceinline__ void lean_load_step(const void *p) {   // the 256 samples at p (lane's own address) into step K's registers
    static_assert(K >= 0 && K < 4, "four steps of registers");
    if constexpr (KIND == IN_IQ_F32) {
        if constexpr (K == 0) LEAN_LOAD4("global_load_dwordx2", "a[0:1]", "a[2:3]", "a[4:5]", "a[6:7]", 512, LEAN_CLOB0);
        if constexpr (K == 1) LEAN_LOAD4("global_load_dwordx2", "a[8:9]", "a[10:11]", "a[12:13]", "a[14:15]", 512, LEAN_CLOB1);
        if constexpr (K == 2) LEAN_LOAD4("global_load_dwordx2", "a[16:17]", "a[18:19]", "a[20:21]", "a[22:23]", 512, LEAN_CLOB2);
        if constexpr (K == 3) LEAN_LOAD4("global_load_dwordx2", "a[24:25]", "a[26:27]", "a[28:29]", "a[30:31]", 512, LEAN_CLOB3);
    } else if constexpr (KIND == IN_I16_SQ) {
        if constexpr (K == 0) LEAN_LOAD4("global_load_sshort", "a0", "a1", "a2", "a3", 128, LEAN_CLOB0);
        if constexpr (K == 1) LEAN_LOAD4("global_load_sshort", "a8", "a9", "a10", "a11", 128, LEAN_CLOB1);
        if constexpr (K == 2) LEAN_LOAD4("global_load_sshort", "a16", "a17", "a18", "a19", 128, LEAN_CLOB2);
        if constexpr (K == 3) LEAN_LOAD4("global_load_sshort", "a24", "a25", "a26", "a27", 128, LEAN_CLOB3);
    } else {
        if constexpr (K == 0) LEAN_LOAD4("global_load_dword", "a0", "a1", "a2", "a3", 256, LEAN_CLOB0);
        if constexpr (K == 1) LEAN_LOAD4("global_load_dword", "a8", "a9", "a10", "a11", 256, LEAN_CLOB1);
        if constexpr (K == 2) LEAN_LOAD4("global_load_dword", "a16", "a17", "a18", "a19", 256, LEAN_CLOB2);
        if constexpr (K == 3) LEAN_LOAD4("global_load_dword", "a24", "a25", "a26", "a27", 256, LEAN_CLOB3);
    }
}
#define LEAN_TAKE8(R0, R1, R2, R3, R4, R5, R6, R7)                                                                                          \
    asm volatile("s_waitcnt vmcnt(%8)\n\tv_accvgpr_read_b32 %0, " R0 "\n\tv_accvgpr_read_b32 %1, " R1 "\n\tv_accvgpr_read_b32 %2, " R2             \
                 "\n\tv_accvgpr_read_b32 %3, " R3 "\n\tv_accvgpr_read_b32 %4, " R4 "\n\tv_accvgpr_read_b32 %5, " R5 "\n\tv_accvgpr_read_b32 %6, " R6 \
                 "\n\tv_accvgpr_read_b32 %7, " R7                                                                                            \
                 : "=v"(w[0]), "=v"(w[1]), "=v"(w[2]), "=v"(w[3]), "=v"(w[4]), "=v"(w[5]), "=v"(w[6]), "=v"(w[7])                            \
                 : "n"(N)                                                                                                                   \
                 : "memory")
#define LEAN_TAKE4(R0, R1, R2, R3)                                                                                                          \
    asm volatile("s_waitcnt vmcnt(%4)\n\tv_accvgpr_read_b32 %0, " R0 "\n\tv_accvgpr_read_b32 %1, " R1 "\n\tv_accvgpr_read_b32 %2, " R2             \
                 "\n\tv_accvgpr_read_b32 %3, " R3                                                                                            \
                 : "=v"(w[0]), "=v"(w[1]), "=v"(w[2]), "=v"(w[3])                                                                            \
                 : "n"(N)                                                                                                                   \
                 : "memory")
// Waits until at most N vector-memory operations are in flight, then hands step K's samples over as envelopes.
template <int KIND, int K, int N>
__device__ __forceinline__ void lean_take(float (&x)[4], float i16_scale) {
    if constexpr (KIND == IN_IQ_F32) {
        float w[8];
        if constexpr (K == 0) LEAN_TAKE8("a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
        if constexpr (K == 1) LEAN_TAKE8("a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
        if constexpr (K == 2) LEAN_TAKE8("a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
        if constexpr (K == 3) LEAN_TAKE8("a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float a = w[2 * j] * w[2 * j], b = w[2 * j + 1] * w[2 * j + 1];   // gnuradio complex_to_mag_squared: two products, one sum
            x[j] = a + b;
        }
    } else {
        float w[4];
        if constexpr (K == 0) LEAN_TAKE4("a0", "a1", "a2", "a3");
        if constexpr (K == 1) LEAN_TAKE4("a8", "a9", "a10", "a11");
        if constexpr (K == 2) LEAN_TAKE4("a16", "a17", "a18", "a19");
        if constexpr (K == 3) LEAN_TAKE4("a24", "a25", "a26", "a27");
#pragma unroll
        for (int j = 0; j < 4; j++) {
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
template <int KIND> struct LeanRaw { static constexpr int BYTES = (KIND == IN_IQ_F32) ? 8 : (KIND == IN_I16_SQ) ? 2 : 4; };
typedef __attribute__((address_space(3))) float lean_lds_f;
typedef __attribute__((address_space(1))) uint32_t lean_g_u32;   // (an address computed from integers must not become a flat access)
typedef uint32_t lean_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) lean_u32x4 lean_g_u128;
typedef __attribute__((address_space(3))) uint32_t lean_lds_u32;
typedef __attribute__((address_space(3))) lean_u32x4 lean_lds_u128;
__device__ __forceinline__ uint32_t lean_dpp_shl8(uint32_t v) {   // lane l <- lane l + 8 of its row of 16 (0 where that leaves the row)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x108, 0xF, 0xF, true);
}

// PF: steps per superstep = how many steps ahead the raw samples are asked for (a step's registers are refilled, for the
// step PF later, as soon as its envelopes are taken: PF * 2 KB in flight per wave for IQ input).
// BLK16: a LOW run longer than max_len covers an aligned block of 16 samples (max_len 46 .. 93: the default's 50) -- the block test of
// the LOW-only form is then two DPP instructions per row; otherwise it works on the row's mask (A.blk).
template <int KIND, int PF, bool BLK16>
__global__ __launch_bounds__(256) void k_threshold_lean(ThrArgs A) {
    static_assert(PF >= 2 && PF <= 4, "planes of a superstep leave in one store of 16 lanes per step");
    constexpr int NR = 4;
    constexpr uint32_t STEPN = 64u * NR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    const uint32_t c = blockIdx.x * wpb + wave;
    if (c >= (uint32_t)A.nchunks) return;
    float *ring = (float *)(smem + (size_t)wave * (size_t)A.Lpad * 4);
    lean_lds_f *const rl = (lean_lds_f *)ring;
    const int L = A.L;
    const int mx = A.mx;
    uint32_t m_chunk, chunk_len;
    chunk_span(A, c, m_chunk, chunk_len);
    const uint32_t n1 = min(A.n, m_chunk + chunk_len);
    const uint32_t m_start = max(m_chunk, A.skip);
    const Carry cr = *A.carry;
    uint64_t *const neg_p = A.neg, *const pos_p = A.pos;   // (by value: a per-lane choice between two kernel-argument FIELDS would be a vector load)
    // this lane's plane for a round's collective store (lane 16 k + i: step k; i < 8 the neg plane, dword i, else the pos plane,
    // dword i - 8) and for a single step's (lanes 0 .. 7 neg, 8 .. 15 pos): chosen once, as integers
    const uintptr_t plane_of_lane = (lane & 8) ? (uintptr_t)pos_p : (uintptr_t)neg_p;
    const float i16s = A.i16_scale;
    const float gfac = A.gfac, gfloor = A.gfloor;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    unsigned long long clk0 = 0, clk1 = 0, clk2 = 0;
    if (A.dbg_clk) clk0 = clock64();
    uint32_t emin = 255u, emax = 0u;
    int w_nl, w_kl;
    float ssf;
    float eps = 0.f;
    // (measured: a form of this prologue that keeps the whole window in registers -- one pass instead of seven over LDS, at
    // 100 instead of 96 registers -- was 4 % SLOWER on the kernel: all waves run it at once, behind the same burst of loads)
    double ss0;
    chunk_incoming<KIND, true>(A, c, lane, ring, cr, m_chunk, ss0, w_nl, w_kl, eps);
    ssf = (float)ss0;
    const uint32_t vtop0 = chunk_save_in<true>(A, c, lane, ring, nullptr, emin, emax);
    ssf = rfl(ssf);
    w_nl = rfl(w_nl);
    w_kl = rfl(w_kl);
    const int nl_in = w_nl, kl_in = w_kl;

    if (A.dbg_clk) clk1 = clock64();
    bool good_run = A.fast_ok != 0;   // false: the wave gave up
    // (raw envelopes, IN_ENV_F32: a negative sample, an infinity or a NaN -- raw bits 0x7F800000 and up -- in the state the chunk
    // starts from or among its samples makes the wave give up: the sign bit of a ring value and the raw-bit ordering are taken;
    // threshold_wg.hip.h says the same)
    if constexpr (KIND == IN_ENV_F32) good_run = good_run && !__ballot(vtop0 >= 0x7F800000u);
    uint32_t why = good_run ? 0u : 1u;   // (debugging aid: 1 parameters / sums out of range, 2 a sample inside a band, 3 LOW run, 4 allowance, 5 first stable sample)
    float min_ss = 3.0e38f;
    int chunk_nl = LL_NONE, chunk_kl = KEY_NONE;
    uint32_t vmin = 0xFFFFFFFFu, vmax = vtop0;
    uint32_t slot_step = (A.g0modL + m_chunk) % (uint32_t)L;
    const float etaD = 1.0f - 9.5367431640625e-07f;  // 1 - 2^-20
    const float slD = 1.0f - 3.814697265625e-06f, slU = 1.0f + 3.814697265625e-06f;
    const float loLf = (float)A.lo_L, hiLf = (float)A.hi_L;   // f32 thresholds carry 2^-18 of slack (slD / slU)
    constexpr int RB = LeanRaw<KIND>::BYTES;
    const char *const in_lane = (const char *)A.in + (size_t)lane * RB;
    // any step, synchronously (compiler loads): lanes past the batch's end read nothing
    auto fetch_env = [&](uint32_t b, float (&x)[NR]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const uint32_t m = b + 64u * j + lane;
            x[j] = (m < A.n) ? envelope_at<KIND>(A.in, (size_t)m, A.i16_scale) : 0.f;
        }
    };
    int steps_since_sync = 0;
    uint32_t last_whole = 0;   // base of the chunk's last whole step
    // steps wholly inside the fill stretch (chunk 0 of a stream's first batches) classify nothing
    uint32_t base = m_chunk;
    while (base + STEPN <= m_start && base < n1) {
        const uint32_t w = (base >> 6) + (uint32_t)(lane & (NR - 1));
        if (lane < 2 * NR && (size_t)w * 64 < A.n) *(__attribute__((address_space(1))) uint64_t *)((lane < NR ? (uintptr_t)neg_p : (uintptr_t)pos_p) + 8 * (uintptr_t)w) = 0ull;
        base += STEPN;
        slot_step += STEPN;
        slot_step = (slot_step >= (uint32_t)L) ? slot_step - (uint32_t)L : slot_step;
    }
    float G = ssf * 0.00390625f;   // (masked first step: 2^-8 of the sum; whole steps: a bound taken from the samples, below)
    float Bneed = 0.f;             // the largest B a superstep of whole steps has needed (or is expected to need)
    float b_acc = 0.f, dl_acc = 0.f;
    float tlo_dn = 0.f, tlo_up = 0.f, thi_dn = 0.f, thi_up = 0.f;
    // deferred per-lane checks of the straight-line step forms (looked at when the superstep closes)
    uint32_t amb_lo = 0xFFFFFFFFu, amb_hi = 0xFFFFFFFFu;   // min of (bits(x) - bits(band bottom)): inside the band iff <= its width
    uint32_t lrun = 0x7F7FFFFFu;                           // min over rows of max(x[l], x[l + 8]) at the lanes that start a 16-block (raw bits: x >= 0)
    // LOW bookkeeping of the straight-line forms, made explicit (w_nl, w_kl) only when the general step or the summary needs it
    uint32_t lz_base = 0;                        // base of the latest straight-line step that had LOW samples: its masks are still in the
                                                 // plane words of its round (pk) or of the round before (pk_prev)
    bool lz_set = false, hot_since = false;      // such a step / any straight-line step since w_nl, w_kl were last explicit
    const int ssl_min = (mx + 1 + (int)STEPN - 1) / (int)STEPN;   // steps without LOW samples after which no LOW sample is in reach of a HIGH one
    int steps_since_low = ((w_kl & 1) && ((int)m_chunk - (w_kl >> 1)) <= mx + 1) ? 0 : ssl_min;
    bool force_general = ((int)m_chunk - 1 - w_nl) > 0;   // the chunk starts inside a LOW run: its first step checks the carried length
    int pk = 0, pk_prev = 0;   // a round's plane words: lanes 16 k .. 16 k + 7 the neg plane of step k (dwords), + 8 .. + 15 the pos plane
    uint32_t pk_base = 0, pkp_base = 0xFFFFFFFFu;   // the bases of those rounds' first steps

    // thresholds of a superstep from the tracked sum and the allowance; false: the banded test cannot serve
    auto open_superstep = [&]() __attribute__((always_inline)) -> bool {
        if (steps_since_sync >= 256) {   // bound the rounding the f32 sum accumulates: re-derive it from the ring
            double part = 0;
#pragma unroll 8
            for (int s2 = lane; s2 < L; s2 += 64) part += (double)fabsf(ring[s2]);
            ssf = (float)rfl(wave_sum_f64(part) + cr.delta);
            steps_since_sync = 0;
        }
        const float M = rfl(G + (eps + RND_SUM) * ssf);
        const float dn = (ssf - M) * slD, up = (ssf + M) * slU;
        tlo_dn = rfl(dn * loLf);
        tlo_up = rfl(up * loLf);
        thi_dn = rfl(dn * hiLf);
        thi_up = rfl(up * hiLf);
        if (!(ssf > 1e-30f && ssf < 1e30f && M < 0.25f * ssf && tlo_dn > 1e-30f && thi_up < 1e30f)) { why = 1u; return false; }
        min_ss = fminf(min_ss, ssf * (etaD - RND_SUM));
        vmax = max(vmax, __float_as_uint(up));   // every window sum of the superstep lies below ssf + M
        vmin = min(vmin, __float_as_uint(tlo_dn));   // what the straight-line forms accept lies inside [tlo_dn, thi_up]
        return true;
    };
    // was the allowance enough for what the lanes accumulated, did no sample sit inside a band, can no LOW run have reached
    // max_len?  then the sum moves on and the next allowance is set
    auto close_superstep = [&](bool whole) __attribute__((always_inline)) -> bool {
        const float B = wave_sum_f32(b_acc) * 1.001f;
        const float D = wave_sum_f32(dl_acc);
        b_acc = 0.f;
        dl_acc = 0.f;
        if (!(B <= G)) { why = 4u; return false; }
        const uint32_t wlo = __float_as_uint(tlo_up) - __float_as_uint(tlo_dn), whi = __float_as_uint(thi_up) - __float_as_uint(thi_dn);
        const unsigned long long inband = __ballot(amb_lo <= wlo || amb_hi <= whi);
        const unsigned long long longlow = __ballot(lrun <= __float_as_uint(tlo_up)) & 0x0001000100010001ull;
        amb_lo = 0xFFFFFFFFu;
        amb_hi = 0xFFFFFFFFu;
        lrun = 0x7F7FFFFFu;
        if (inband) { why = 2u; return false; }
        if (longlow) { why = 3u; return false; }
        ssf = rfl(ssf + D);
        if (whole) {
            Bneed = fmaxf(Bneed, B);
            G = rfl(fminf(fmaxf(gfac * Bneed, ssf * gfloor), ssf * 0.125f));
        }
        return true;
    };
    // w_nl / w_kl (last non-LOW index, key of the last LOW sample) as of the step that starts at `base`, from what the
    // straight-line steps left behind
    auto materialize = [&]() __attribute__((always_inline)) -> bool {
        // (new values gathered in locals and assigned once, by selects: look-alike blocks of stores to these by-reference
        // captures get merged into stores through a chosen pointer, which pins them in scratch memory)
        int nkl = w_kl, nnl = w_nl;
        bool set_kl = false, set_nl = false, bad = false;
        if (lz_set) {
            int src = pk;
            uint32_t sb = pk_base;
            if (lz_base < pk_base) {
                src = pk_prev;
                sb = pkp_base;
            }
            set_kl = true;
            if (lz_base >= sb && lz_base - sb < (uint32_t)PF * STEPN) {
                const int l0 = (int)((lz_base - sb) >> 8) * 16;
                unsigned long long lm[NR];
#pragma unroll
                for (int j = 0; j < NR; j++)
                    lm[j] = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(src, l0 + 2 * j) |
                            ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(src, l0 + 2 * j + 1) << 32);
                int ll = LL_NONE;
#pragma unroll
                for (int j = 0; j < NR; j++) ll = lm[j] ? (int)(lz_base + 64u * j) + last_set(lm[j]) : ll;
                nkl = 2 * ll + 1;   // (a LOW sample of a straight-line step never ends on a time-out: its key is good)
                if (steps_since_low == 0) {   // that step was the latest one
                    int nl = LL_NONE;
#pragma unroll
                    for (int j = 0; j < NR; j++) nl = (~lm[j]) ? (int)(lz_base + 64u * j) + last_set(~lm[j]) : nl;
                    bad = nl == LL_NONE;   // (256 LOW samples: the block test will have seen it)
                    nnl = nl;
                    set_nl = true;
                }
            } else {
                // its round is gone: more than PF steps (>= 512 samples) back, out of every HIGH sample's reach (max_len <= 500
                // where this kernel runs).  Any key that is not live stands for it (k_certify, resolve_low_state: a key only
                // matters while live): the step's last sample.
                nkl = 2 * (int)(lz_base + STEPN - 1) + 1;
            }
        }
        if (hot_since && (steps_since_low > 0 || !lz_set)) {   // the latest step had no LOW sample
            nnl = (int)base - 1;
            set_nl = true;
        }
        w_kl = nkl;
        chunk_kl = set_kl ? nkl : chunk_kl;
        w_nl = nnl;
        chunk_nl = set_nl ? nnl : chunk_nl;
        lz_set = false;
        hot_since = false;
        if (bad) why = 3u;
        return !bad;
    };
    // plane words of a single step stored at once (masked steps, outside the superstep's collective store)
    auto store_planes_now = [&](const unsigned long long (&lowm)[NR], const unsigned long long (&posm)[NR]) __attribute__((always_inline)) {
        int q = 0;
#pragma unroll
        for (int k = 0; k < NR; k++) {
            PLANE_PUT(q, lowm[k], 2 * k);
            PLANE_PUT(q, (lowm[k] >> 32), 2 * k + 1);
            PLANE_PUT(q, posm[k], 2 * NR + 2 * k);
            PLANE_PUT(q, (posm[k] >> 32), 2 * NR + 2 * k + 1);
        }
        const int h = lane & (2 * NR - 1);
        const uint32_t w = (base >> 6) + (uint32_t)(h >> 1);
        lean_g_u32 *dst = (lean_g_u32 *)(plane_of_lane + 4 * (2 * (uintptr_t)(base >> 6) + (uintptr_t)h));
        if (lane < 4 * NR && (size_t)w * 64 < A.n) *dst = (uint32_t)q;
    };

    // ---------------- the general step, on wave masks ----------------
    // kslot >= 0: step kslot of a round of whole steps (plane words into pk); kslot < 0: a step on its own (plane words stored at
    // once).  masked: lanes outside [m_start, n1) are not samples (the stream's first stable sample, the batch's ragged end).
    // x: the step's envelopes.  Instantiated ONCE in the kernel (its call site is the slow arm of the chunk loop below): the
    // straight-line forms' registers are not to pay for it.
    auto general_step = [&](float (&x)[NR], const bool masked, const int kslot) __attribute__((always_inline)) -> bool {
        if (!materialize()) return false;
        float prev[NR];
        uint32_t slot[NR];
#pragma unroll
        for (int j = 0; j < NR; j++) {
            uint32_t s = slot_step + 64u * j + lane;
            s = (s >= (uint32_t)L) ? s - (uint32_t)L : s;
            slot[j] = s;
            prev[j] = fabsf(ring[s]);
        }
        unsigned long long unt[NR] = {0, 0, 0, 0}, am[NR] = {~0ull, ~0ull, ~0ull, ~0ull};
        if (masked) {
            // a lane that is not a sample repeats the value its slot holds (no drift; if "accepted" the slot keeps its
            // value) and the touched flag such a store sets is taken back after the step
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
        unsigned long long lowm[NR], posm[NR], good = ~0ull, anylow = 0;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const unsigned long long lo1 = __ballot(x[j] < tlo_dn), lo0 = __ballot(x[j] > tlo_up);
            const unsigned long long hi1 = __ballot(x[j] > thi_up), hi0 = __ballot(x[j] < thi_dn);
            unsigned long long g = (lo1 | lo0) & (hi1 | hi0);
            lowm[j] = lo1;
            posm[j] = hi1;
            g |= ~am[j];
            lowm[j] &= am[j];
            posm[j] &= am[j];
            good &= g;
            anylow |= lowm[j];
        }
        bool ok = (good == ~0ull);
        if (!ok) why = 2u;
        if (ok && anylow) {
            // every LOW sample must sit at run position <= max_len (then none ends on a time-out):
            // a longer run covers an aligned block of LOW samples, or continues the carried run
            const int lead = (lowm[0] == ~0ull) ? 64 : (__ffsll((long long)~lowm[0]) - 1);
            const int carry_run = (int)base - 1 - w_nl;
            unsigned long long pre = 0;
#pragma unroll
            for (int j = 0; j < NR; j++) pre |= lowm[j] & (lowm[j] >> A.probe_mid) & (lowm[j] >> A.probe_end);
            unsigned long long hit = 0;
            if (pre & A.selmask) {
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    unsigned long long t = lowm[j];
#pragma unroll
                    for (int f = 0; f < 6; f++) t &= t >> A.fold_sh[f];
                    hit |= t & A.selmask;
                }
            }
            if (hit || ((carry_run > 0) && (carry_run + lead > mx))) { ok = false; why = 3u; }
            if (masked && base < m_start) { ok = false; why = 5u; }   // (a LOW run across the first stable sample: leave it to the exact kernel)
        }
        if (!ok) return false;
        int before = (w_kl & 1) ? (w_kl >> 1) : LL_NONE;  // last LOW before the row (every key here is good)
#pragma unroll
        for (int j = 0; j < NR; j++) {
            // HIGH is ignored within max_len + 1 samples after a LOW sample
            const int rb = (int)(base + 64u * j);
            const unsigned long long below = lowm[j] & lane_lt;
            const int lastlow = below ? rb + last_set(below) : before;
            const bool ps = (x[j] > thi_up) && ((rb + lane - lastlow) > mx + 1);
            const bool a = !(x[j] < tlo_dn) && !ps;
            const float t = x[j] - prev[j];
            if (a) {
                b_acc += fabsf(t);
                dl_acc += t;
                ring[slot[j]] = x[j];
            }
            const uint32_t xb = __float_as_uint(x[j]);
            vmin = min(vmin, (a && xb != 0u) ? xb : 0xFFFFFFFFu);
            vmax = max(vmax, a ? xb : 0u);
            posm[j] = __ballot(ps);
            posm[j] &= am[j];
            before = lowm[j] ? rb + last_set(lowm[j]) : before;
        }
        int step_nl = LL_NONE, step_ll = LL_NONE;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const int rb = (int)(base + 64u * j);
            const unsigned long long nonlow = ~lowm[j] & am[j];
            if ((unt[j] >> lane) & 1ull) ring[slot[j]] = __uint_as_float(__float_as_uint(x[j]) | 0x80000000u);
            step_ll = lowm[j] ? rb + last_set(lowm[j]) : step_ll;
            step_nl = nonlow ? rb + last_set(nonlow) : step_nl;
        }
        // (selects, not two look-alike blocks of stores: merged by the optimiser into stores through a chosen POINTER, those would
        // pin the four variables -- captured by reference -- in scratch memory)
        w_kl = (step_ll != LL_NONE) ? 2 * step_ll + 1 : w_kl;
        chunk_kl = (step_ll != LL_NONE) ? w_kl : chunk_kl;
        w_nl = (step_nl != LL_NONE) ? step_nl : w_nl;
        chunk_nl = (step_nl != LL_NONE) ? step_nl : chunk_nl;
        steps_since_low = anylow ? 0 : steps_since_low + 1;
        if (kslot < 0) {
            store_planes_now(lowm, posm);
        } else {
#pragma unroll
            for (int k = 0; k < NR; k++) {   // (the lane of a v_writelane may come from a scalar register)
                PLANE_PUT_AT(pk, lowm[k], 16 * kslot + 2 * k);
                PLANE_PUT_AT(pk, (lowm[k] >> 32), 16 * kslot + 2 * k + 1);
                PLANE_PUT_AT(pk, posm[k], 16 * kslot + 8 + 2 * k);
                PLANE_PUT_AT(pk, (posm[k] >> 32), 16 * kslot + 8 + 2 * k + 1);
            }
        }
        return true;
    };

    // ---------------- step KS of a superstep of whole steps ----------------
    // -> 0 done, 1 the wave gives up, 2 the general form has to take this step (its envelopes are left in gx, nothing else of
    // the step has happened but the refill of its registers)
    float gx[NR] = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](auto ks_tag) __attribute__((always_inline)) -> int {
        constexpr int KS = decltype(ks_tag)::value;
        float x[NR];
        // this step's samples have landed (the loads of the PF - 1 steps after it may be in flight) ...
        lean_take<KIND, KS, (PF - 1) * NR>(x, i16s);
        // ... and its registers take the step PF later.  Unconditionally -- a branch around the loads would not be worth its
        // scalar instructions: past the chunk's last whole step the address is clamped to it and the values are never used.
        lean_load_step<KIND, KS>(in_lane + (size_t)min(base + (uint32_t)PF * STEPN, last_whole) * RB);

        // can anything be LOW / HIGH (or inside those bands) at all?
        // (envelopes are >= 0: their raw bits order like their values, and integer min / max need no canonicalising of NaNs)
        const float xmin = __uint_as_float(min(min(__float_as_uint(x[0]), __float_as_uint(x[1])), min(__float_as_uint(x[2]), __float_as_uint(x[3]))));
        const float xmax = __uint_as_float(max(max(__float_as_uint(x[0]), __float_as_uint(x[1])), max(__float_as_uint(x[2]), __float_as_uint(x[3]))));
        const unsigned long long lowany = __ballot(!(xmin > tlo_up)), highany = __ballot(!(xmax < thi_dn));
        const bool lowp = lowany != 0ull, highp = highany != 0ull;
        bool ok = true;
        if constexpr (KIND == IN_ENV_F32) {
            if (__ballot(__float_as_uint(xmax) >= 0x7F800000u)) { why = 1u; return 1; }
        }
        // which form: 0 nothing classifies, 1 only LOW, 2 only HIGH with no LOW sample in reach, 3 the general step
        if (__builtin_expect((lowany && highany) || force_general || (highany && steps_since_low < ssl_min), 0)) {
            force_general = false;
#pragma unroll
            for (int j = 0; j < NR; j++) gx[j] = x[j];
            return 2;
        }
        {
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
            if (!lowp && !highp) {
                // nothing classifies: every sample is accepted
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const float t = x[j] - fabsf(praw[j]);
                    b_acc += fabsf(t);
                    dl_acc += t;
                    *pa[j] = x[j];
                }
                steps_since_low++;
            } else if (lowp) {
                // LOW samples only: rejected ones keep their slot (value and sign bit)
                unsigned long long lw[NR];
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const bool lo = x[j] < tlo_dn;
                    lw[j] = __ballot(lo);
                    const float val = lo ? praw[j] : x[j];
                    const float ts = fabsf(val) - fabsf(praw[j]);   // 0 for a rejected sample
                    b_acc += fabsf(ts);
                    dl_acc += ts;
                    *pa[j] = val;
                }
                // a sample inside the LOW band?  (looked at when the superstep closes)
                amb_lo = min(min(amb_lo, __float_as_uint(x[0]) - __float_as_uint(tlo_dn)),
                             min(__float_as_uint(x[1]) - __float_as_uint(tlo_dn), __float_as_uint(x[2]) - __float_as_uint(tlo_dn)));
                amb_lo = min(amb_lo, __float_as_uint(x[3]) - __float_as_uint(tlo_dn));
                // a LOW run longer than max_len covers an aligned block of b samples (b = blk)
                if constexpr (BLK16) {
                    // ... whose first and ninth sample are then LOW: per 16-lane row of the wave, in the lanes
#pragma unroll
                    for (int j = 0; j < NR; j++) lrun = min(lrun, max(__float_as_uint(x[j]), lean_dpp_shl8(__float_as_uint(x[j]))));
                } else if (A.blk == 64) {
                    if ((lw[0] == ~0ull) || (lw[1] == ~0ull) || (lw[2] == ~0ull) || (lw[3] == ~0ull)) { ok = false; why = 3u; }
                } else {
                    unsigned long long hit = 0;
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        unsigned long long t = lw[j];
#pragma unroll
                        for (int f = 0; f < 6; f++) t &= t >> A.fold_sh[f];
                        hit |= t & A.selmask;
                    }
                    if (hit) { ok = false; why = 3u; }
                }
                PLANE_PUT8(pk, lw, 16 * KS);
                lz_base = base;
                lz_set = true;
                steps_since_low = 0;
            } else {
                // HIGH samples only, no LOW sample in reach: all of them are rejected (transition_sink.py:71-74)
                unsigned long long hw[NR];
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const bool hi = x[j] > thi_up;
                    hw[j] = __ballot(hi);
                    const float val = hi ? praw[j] : x[j];
                    const float ts = fabsf(val) - fabsf(praw[j]);
                    b_acc += fabsf(ts);
                    dl_acc += ts;
                    *pa[j] = val;
                }
                PLANE_PUT8(pk, hw, 16 * KS + 8);
                amb_hi = min(min(amb_hi, __float_as_uint(x[0]) - __float_as_uint(thi_dn)),
                             min(__float_as_uint(x[1]) - __float_as_uint(thi_dn), __float_as_uint(x[2]) - __float_as_uint(thi_dn)));
                amb_hi = min(amb_hi, __float_as_uint(x[3]) - __float_as_uint(thi_dn));
                steps_since_low++;
            }
            hot_since = true;
        }
        slot_step += STEPN;
        slot_step = (slot_step >= (uint32_t)L) ? slot_step - (uint32_t)L : slot_step;
        steps_since_sync++;
        base += STEPN;
        return ok ? 0 : 1;
    };
    // ---------------- the chunk ----------------
    // Rounds of PF whole steps out of the registers ahead (step k of a round lives in a[8 k ..]), `rounds` rounds to a superstep;
    // a step that no round can hold -- the one with the stream's first stable sample, fewer than PF whole steps before the
    // chunk's end (chunk lengths are cut to whole rounds: only a batch's last chunk has them), the batch's ragged end -- is a
    // superstep of its own on synchronously loaded samples.  One loop, so that the general form has a single call site.
    const int rounds = max(1, A.ksteps);
    int k = PF;          // next step of the round in progress (PF: between rounds)
    int rd = 0;          // rounds done in the superstep in progress
    bool primed = false, in_round = false, in_super = false;
    uintptr_t pl_addr = 0;   // this lane's dword of a round's plane store
    while (good_run) {
        int gen = -2;    // the general form's job this trip: -2 none, -1 a step on its own, k >= 0 step k of the round
        if (k == PF) {   // between rounds
            if (in_round) {   // the round that just ended: its plane words leave
                if (lane < 16 * PF) *(lean_g_u32 *)pl_addr = (uint32_t)pk;
                pl_addr += 32 * PF;
                in_round = false;
                rd++;
            }
            const bool more = base >= m_start && base + (uint32_t)PF * STEPN <= n1;
            if (in_super && (rd >= rounds || !more)) {
                if (!close_superstep(true)) { good_run = false; break; }
                in_super = false;
                rd = 0;
            }
            if (base >= n1) break;
            if (!more) {
                gen = -1;
            } else {
                if (!primed) {
                    last_whole = base + (n1 - base - STEPN) / STEPN * STEPN;
                    lean_load_step<KIND, 0>(in_lane + (size_t)base * RB);
                    lean_load_step<KIND, 1>(in_lane + (size_t)(base + STEPN) * RB);
                    if constexpr (PF > 2) lean_load_step<KIND, 2>(in_lane + (size_t)(base + 2 * STEPN) * RB);
                    if constexpr (PF > 3) lean_load_step<KIND, 3>(in_lane + (size_t)(base + 3 * STEPN) * RB);
                    // first allowance: sum |x - prev| over the samples of the first round that look acceptable, per such sample,
                    // for a superstep of acceptable samples only (a guess like any other allowance: the superstep's own B decides)
                    const float wlo = ssf * loLf * 0.5f, whi = ssf * hiLf * 1.02f;
                    float b0 = 0.f, n0 = 0.f;
                    uint32_t sl = slot_step;
                    auto look = [&](auto kt) {
                        float xv[NR];
                        lean_take<KIND, decltype(kt)::value, 0>(xv, i16s);   // (reading leaves the registers as they are: the round takes them again)
#pragma unroll
                        for (int j = 0; j < NR; j++) {
                            uint32_t q = sl + 64u * j + lane;
                            q = (q >= (uint32_t)L) ? q - (uint32_t)L : q;
                            const bool in = xv[j] > wlo && xv[j] < whi;
                            b0 += in ? fabsf(xv[j] - fabsf(ring[q])) : 0.f;
                            n0 += in ? 1.f : 0.f;
                        }
                        sl += STEPN;
                        sl = (sl >= (uint32_t)L) ? sl - (uint32_t)L : sl;
                    };
                    look(std::integral_constant<int, 0>{});
                    look(std::integral_constant<int, 1>{});
                    if constexpr (PF > 2) look(std::integral_constant<int, 2>{});
                    if constexpr (PF > 3) look(std::integral_constant<int, 3>{});
                    b0 = wave_sum_f32(b0);
                    n0 = wave_sum_f32(n0);
                    Bneed = ((n0 >= 64.f) ? b0 / n0 * (float)(PF * STEPN) : ssf * 0.001953125f * (float)PF) * (float)rounds;
                    G = rfl(fminf(fmaxf(gfac * Bneed, ssf * gfloor), ssf * 0.125f));
                    pl_addr = plane_of_lane + 4 * (2 * (uintptr_t)(base >> 6) + 8 * (uintptr_t)(lane >> 4) + (uintptr_t)(lane & 7));
                    pk_base = base;
                    primed = true;
                }
                if (!in_super) {
                    if (!open_superstep()) { good_run = false; break; }
                    in_super = true;
                }
                pk_prev = pk;
                pkp_base = pk_base;
                pk = 0;
                pk_base = base;
                in_round = true;
                k = 0;
            }
        }
        if (gen == -2) {
            int r = 0;
            auto stepk = [&](auto kt) -> int {   // (steps beyond the round's length are never instantiated)
                if constexpr (decltype(kt)::value < PF) return step(kt);
                else return 0;
            };
            switch (k) {
            case 0:
                r = stepk(std::integral_constant<int, 0>{});
                if (r) break;
                k = 1;
                [[fallthrough]];
            case 1:
                r = stepk(std::integral_constant<int, 1>{});
                if (r) break;
                k = 2;
                if (PF == 2) break;
                [[fallthrough]];
            case 2:
                r = stepk(std::integral_constant<int, 2>{});
                if (r) break;
                k = 3;
                if (PF == 3) break;
                [[fallthrough]];
            case 3:
                r = stepk(std::integral_constant<int, 3>{});
                if (r) break;
                k = 4;
                break;
            default:
                break;
            }
            if (r == 1) { good_run = false; break; }
            if (r == 0) continue;   // the round is through
            gen = k;
        }
        // ---- the slow arm: one step in the general form ----
        const bool own = gen < 0;
        if (own) {
            primed = false;   // (the registers ahead belong to another base from here on)
            fetch_env(base, gx);
            G = rfl(fminf(fmaxf(G, ssf * 0.00390625f), ssf * 0.125f));
            if (!open_superstep()) { good_run = false; break; }
        }
        if (!general_step(gx, own, gen)) { good_run = false; break; }
        slot_step += STEPN;
        slot_step = (slot_step >= (uint32_t)L) ? slot_step - (uint32_t)L : slot_step;
        steps_since_sync++;
        base += STEPN;
        if (own) {
            if (!close_superstep(false)) { good_run = false; break; }
        } else {
            k = gen + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (nothing of the rounds' loads is left in flight)
    if (good_run) good_run = materialize();

    if (A.dbg_clk) clk2 = clock64();
    const uint32_t all_robust = good_run ? 1u : 0u;
    if (c == 0 && lane == 0) {   // chunk 0 has no certification of its own: its verdict travels here
        A.cert[0] = good_run ? 1 : 0;
        if (!good_run) atomicAdd(&A.sum->n_fail, 1u);
    }
    chunk_publish<true, true>(A, c, lane, ring, nullptr, emin, emax, vmin, vmax, ssf, eps, good_run ? 0u : (4u | (why << 4)), chunk_kl, chunk_nl,
                        (double)ssf, min_ss, nl_in, kl_in, all_robust);
    if (A.dbg_clk && lane == 0) {
        A.dbg_clk[4 * (size_t)c + 0] = clk0;
        A.dbg_clk[4 * (size_t)c + 1] = clk1;
        A.dbg_clk[4 * (size_t)c + 2] = clk2;
        A.dbg_clk[4 * (size_t)c + 3] = clock64();
    }
}

}  // namespace nfc
