// threshold_lean.hip.h -- pass 0 of the threshold stage (transition_sink.py:55-82) as a lean, purely optimistic kernel.
//
// Same decomposition as k_threshold (threshold.hip.h): one wavefront per time chunk, NR rows of 64 samples per step,
// the ring of the last L accepted samples in LDS with "untouched" in the sign bit.  What differs is WHEN the bound of
// the window sum's drift is established.  k_threshold reduces B = sum |x - prev| over a step BEFORE it classifies the
// step (two wave reductions and a dozen wave-uniform float operations on every step's critical path).  Here a
// SUPERSTEP of K steps is classified against thresholds fixed at its start:
//
//     M = G + (eps + RND) * ss        G: drift allowance (guessed from the last superstep's B),
//                                     eps: certification margin of the speculated incoming ring, RND: f32 sum tracking
//     LOW   <=  x < (ss - M) * lo / L      not LOW  <=  x > (ss + M) * lo / L
//     HIGH  <=  x > (ss + M) * hi / L      not HIGH <=  x < (ss - M) * hi / L
//
// while every lane only ACCUMULATES |x - prev| and (x - prev) of the samples it accepts.  After the K steps one pair of
// wave reductions gives B and D.  If B <= G the classifications are the reference's: by induction over the samples of the
// superstep -- while the drift so far is <= G every classification made is exact, so the accepted set is exact, so the
// drift after the next sample is a partial sum of accepted (x - prev), in magnitude <= B <= G.  Then ss += D.
//
// Nothing is ever repaired in place: a sample inside a band, a LOW run that may reach max_len, B > G, parameters the
// banded test cannot serve -- the wave GIVES UP: it flags its chunk (RunMeta.all_robust = 0; chunk 0: cert[0] = 0 and the
// failure count), and the host re-runs that chunk from the exact state with k_threshold (mode 1), exactly as it does for
// a chunk whose speculation cannot be certified.  So this kernel has no fp64, no exact path and no retry structure in its
// loop; the result that stands is always one whose every step was proven.
#pragma once
#include "threshold.hip.h"

#include <type_traits>

namespace nfc {

// The raw samples of the steps ahead sit in registers whose loads the compiler must neither wait for nor move: they are
// issued by asm statements (a compiler-issued load whose result lives across the loop's back edge is copied there, behind a
// wait for EVERY load in flight) and counted by hand -- `lean_wait<N>` names every register of the step it releases, so no
// read of them can be scheduled above it (cdna_hip_programming.md 5.7, form (ii)).
typedef float lean_v2f __attribute__((ext_vector_type(2)));
template <int KIND> struct LeanRaw { using T = float; static constexpr int BYTES = 4; };
template <> struct LeanRaw<IN_IQ_F32> { using T = lean_v2f; static constexpr int BYTES = 8; };
template <> struct LeanRaw<IN_I16_SQ> { using T = int; static constexpr int BYTES = 2; };
template <int KIND, int OFF>
__device__ __forceinline__ void lean_load(typename LeanRaw<KIND>::T &q, const void *p) {
    if constexpr (KIND == IN_IQ_F32) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "+v"(q) : "v"(p), "n"(OFF) : "memory");
    else if constexpr (KIND == IN_I16_SQ) asm volatile("global_load_sshort %0, %1, off offset:%2" : "+v"(q) : "v"(p), "n"(OFF) : "memory");
    else asm volatile("global_load_dword %0, %1, off offset:%2" : "+v"(q) : "v"(p), "n"(OFF) : "memory");
}
template <int KIND>
__device__ __forceinline__ float lean_env(typename LeanRaw<KIND>::T v, float i16_scale) {
    if constexpr (KIND == IN_IQ_F32) {
        const float a = v.x * v.x, b = v.y * v.y;
        return a + b;
    } else if constexpr (KIND == IN_ENV_F32) {
        return v;
    } else if constexpr (KIND == IN_REAL_F32_SQ) {
        return v * v;
    } else {
        const float s = (float)v * i16_scale;
        return s * s;
    }
}
template <int N, class T>
__device__ __forceinline__ void lean_wait(T (&q)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "n"(N) : "memory");
}

// PF: steps per superstep = how many steps ahead the raw samples are asked for (a step's registers are refilled, for the
// step PF later, as soon as its envelopes are taken: PF * NR * 512 bytes in flight per wave).
template <int KIND, int NR, int PF>
__global__ __launch_bounds__(256) void k_threshold_lean(ThrArgs A) {
    constexpr uint32_t STEPN = 64u * NR;
    constexpr bool SIGN_T = (KIND != IN_ENV_F32);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    const uint32_t c = blockIdx.x * wpb + wave;
    if (c >= (uint32_t)A.nchunks) return;
    const size_t lds_wave = SIGN_T ? (size_t)A.Lpad * 4 : (size_t)A.Lpad * 5;
    float *ring = (float *)(smem + (size_t)wave * lds_wave);
    unsigned char *tch = (unsigned char *)(ring + A.Lpad);
    auto mark = [&](uint32_t sl) { if constexpr (!SIGN_T) tch[sl] = 1; };
    const int L = A.L;
    const int mx = A.mx;
    const uint32_t m_chunk = c * (uint32_t)A.C;
    const uint32_t n1 = min(A.n, m_chunk + (uint32_t)A.C);
    const uint32_t m_start = max(m_chunk, A.skip);
    const Carry cr = *A.carry;
    uint64_t *const neg_p = A.neg, *const pos_p = A.pos;   // (by value: a per-lane choice between two kernel-argument FIELDS would be a vector load)
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    uint32_t emin = 255u, emax = 0u;
    int w_nl, w_kl;
    double ss0;
    float eps = 0.f;
    chunk_incoming<KIND>(A, c, lane, ring, cr, m_chunk, ss0, w_nl, w_kl, eps);
    ss0 = rfl(ss0);
    w_nl = rfl(w_nl);
    w_kl = rfl(w_kl);
    const int nl_in = w_nl, kl_in = w_kl;
    const uint32_t vtop0 = chunk_save_in<SIGN_T>(A, c, lane, ring, tch, emin, emax);

    bool good_run = A.fast_ok != 0;   // false: the wave gave up
    uint32_t why = good_run ? 0u : 1u;   // (debugging aid: 1 parameters / sums out of range, 2 a sample inside a band, 3 LOW run, 4 allowance, 5 first stable sample)
    float min_ss = 3.0e38f;
    int chunk_nl = LL_NONE, chunk_kl = KEY_NONE;
    uint32_t vmin = 0xFFFFFFFFu, vmax = vtop0;
    uint32_t slot_step = (A.g0modL + m_chunk) % (uint32_t)L;
    const float etaD = 1.0f - 9.5367431640625e-07f;  // 1 - 2^-20
    const float slD = 1.0f - 3.814697265625e-06f, slU = 1.0f + 3.814697265625e-06f;
    const float loLf = (float)A.lo_L, hiLf = (float)A.hi_L;   // f32 thresholds carry 2^-18 of slack (slD / slU)
    static_assert(NR == 4, "lean_wait names four registers");
    using Raw = typename LeanRaw<KIND>::T;
    constexpr int RB = LeanRaw<KIND>::BYTES;
    Raw r[PF][NR];
#pragma unroll
    for (int k = 0; k < PF; k++)
#pragma unroll
        for (int j = 0; j < NR; j++) r[k][j] = Raw(0);
    const char *const in_lane = (const char *)A.in + (size_t)lane * RB;
    // a whole step's samples into q (asm loads: counted by hand, see above)
    auto fetch_whole = [&](uint32_t b, Raw (&q)[NR]) {
        const char *p = in_lane + (size_t)b * RB;
        lean_load<KIND, 0 * 64 * RB>(q[0], p);
        lean_load<KIND, 1 * 64 * RB>(q[1], p);
        lean_load<KIND, 2 * 64 * RB>(q[2], p);
        lean_load<KIND, 3 * 64 * RB>(q[3], p);
    };
    // any step, synchronously (compiler loads; no asm load may be in flight): lanes past the batch's end read nothing
    auto fetch = [&](uint32_t b, Raw (&q)[NR]) {
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const uint32_t m = b + 64u * j + lane;
            if constexpr (KIND == IN_IQ_F32) {
                const float2 v = (m < A.n) ? ((const float2 *)A.in)[m] : make_float2(0.f, 0.f);
                q[j].x = v.x;
                q[j].y = v.y;
            } else if constexpr (KIND == IN_I16_SQ) {
                q[j] = (m < A.n) ? (int)((const int16_t *)A.in)[m] : 0;
            } else {
                q[j] = (m < A.n) ? ((const float *)A.in)[m] : 0.f;
            }
        }
    };
    float ssf = (float)ss0;
    int steps_since_sync = 0;
    uint32_t last_whole = 0;   // base of the chunk's last whole step
    // steps wholly inside the fill stretch (chunk 0 of a stream's first batches) classify nothing
    uint32_t base = m_chunk;
    while (base + STEPN <= m_start && base < n1) {
        const uint32_t w = (base >> 6) + (uint32_t)(lane & (NR - 1));
        if (lane < 2 * NR && (size_t)w * 64 < A.n) (lane < NR ? neg_p : pos_p)[w] = 0ull;
        base += STEPN;
        slot_step += STEPN;
        slot_step = (slot_step >= (uint32_t)L) ? slot_step - (uint32_t)L : slot_step;
    }
    float G = ssf * 0.00390625f;   // (masked first step: 2^-8 of the sum; whole steps: a bound taken from the samples, below)
    float b_acc = 0.f, dl_acc = 0.f;
    float tlo_dn = 0.f, tlo_up = 0.f, thi_dn = 0.f, thi_up = 0.f;

    // thresholds of a superstep from the tracked sum and the allowance; false: the banded test cannot serve
    auto open_superstep = [&]() -> bool {
        if (steps_since_sync >= 256) {   // bound the rounding the f32 sum accumulates: re-derive it from the ring
            double part = 0;
#pragma unroll 8
            for (int s2 = lane; s2 < L; s2 += 64) part += (double)(SIGN_T ? fabsf(ring[s2]) : ring[s2]);
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
        return true;
    };
    // was the allowance enough for what the lanes accumulated?  then the sum moves on and the next allowance is set
    // (the next allowance is what this superstep needed PER ACCEPTED SAMPLE, for a superstep of accepted samples only:
    // inside a frame a third or a half of the samples are rejected, and the first idle superstep after it is not)
    uint32_t nrej = 0;   // rejected samples of the superstep
    auto close_superstep = [&](int kdone, int knext) -> bool {
        const float B = wave_sum_f32(b_acc) * 1.001f;
        const float D = wave_sum_f32(dl_acc);
        b_acc = 0.f;
        dl_acc = 0.f;
        if (!(B <= G)) { why = 4u; return false; }
        ssf = rfl(ssf + D);
        const uint32_t nacc = (uint32_t)kdone * STEPN - nrej;
        nrej = 0;
        if (knext > 0 && nacc >= STEPN / 2) {   // (knext 0: a masked step says nothing about the noise level)
            const float per = B / (float)nacc;
            G = rfl(fminf(fmaxf(A.gfac * per * (float)((uint32_t)knext * STEPN), ssf * A.gfloor), ssf * 0.125f));
        }
        return true;
    };
    // one step of NR rows.  MASKED: lanes outside [m_start, n1) are not samples (the stream's first stable sample, the
    // batch's ragged end): at most two steps of a chunk, kept out of the hot loop's register allocation.
    auto step = [&](auto masked_tag, Raw (&rq)[NR]) -> bool {
        constexpr bool MASKED = decltype(masked_tag)::value;
        float x[NR], prev[NR];
        uint32_t slot[NR];
        if constexpr (!MASKED) lean_wait<(PF - 1) * NR>(rq);   // this step's samples have landed (the loads of the PF - 1 steps after it may be in flight)
#pragma unroll
        for (int j = 0; j < NR; j++) x[j] = lean_env<KIND>(rq[j], A.i16_scale);
        if constexpr (!MASKED) {
            // these registers now take the step PF later.  Unconditionally -- a branch around the loads would cost the loop its
            // counted waits (the compiler then drains every load in flight at each use): past the chunk's last whole step the
            // address is clamped to it and the values are never used
            // the envelopes are taken BEFORE the registers are handed to the loads (an envelope computed later would make the
            // compiler keep a copy of the raw sample -- taken before the wait above, i.e. of a register still being loaded)
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
            fetch_whole(min(base + (uint32_t)PF * STEPN, last_whole), rq);
        }
        const bool nowrap = slot_step + STEPN <= (uint32_t)L;
        if (nowrap) {
            const float *rp = ring + slot_step + lane;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                slot[j] = slot_step + 64u * j + lane;
                prev[j] = SIGN_T ? fabsf(rp[64 * j]) : rp[64 * j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                uint32_t s = slot_step + 64u * j + lane;
                s = (s >= (uint32_t)L) ? s - (uint32_t)L : s;
                slot[j] = s;
                prev[j] = SIGN_T ? fabsf(ring[s]) : ring[s];
            }
        }
        unsigned long long unt[NR], am[NR];
        if constexpr (MASKED) {
            // a lane that is not a sample repeats the value its slot holds (no drift; if "accepted" the slot keeps its
            // value) and the touched flag such a store sets is taken back after the step
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const uint32_t m = base + 64u * j + lane;
                const bool inact = (m < m_start) || (m >= n1);
                bool untouched;
                if constexpr (SIGN_T) untouched = (__float_as_uint(ring[slot[j]]) >> 31) != 0u;
                else untouched = tch[slot[j]] == 0;
                unt[j] = __ballot(inact && untouched);
                am[j] = __ballot(!inact);
                if (inact) x[j] = prev[j];
            }
        }
        unsigned long long lowm[NR], posm[NR], good = ~0ull, anylow = 0, anyhi = 0;
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const unsigned long long lo1 = __ballot(x[j] < tlo_dn), lo0 = __ballot(x[j] > tlo_up);
            const unsigned long long hi1 = __ballot(x[j] > thi_up), hi0 = __ballot(x[j] < thi_dn);
            unsigned long long g = (lo1 | lo0) & (hi1 | hi0);
            lowm[j] = lo1;
            posm[j] = hi1;
            if constexpr (MASKED) {
                g |= ~am[j];
                lowm[j] &= am[j];
                posm[j] &= am[j];
            }
            good &= g;
            anylow |= lowm[j];
            anyhi |= posm[j];
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
            if (MASKED && base < m_start) { ok = false; why = 5u; }   // (a LOW run across the first stable sample: leave it to the exact kernel)
        }
        if (__builtin_expect(!ok, 0)) return false;
        const bool key_live = (w_kl & 1) && ((int)base - (w_kl >> 1)) <= mx + 1;
        if (anyhi == 0) {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const float t = x[j] - prev[j];
                if (!(x[j] < tlo_dn)) {
                    b_acc += fabsf(t);
                    dl_acc += t;
                    ring[slot[j]] = x[j];
                    mark(slot[j]);
                }
            }
            vmin = min(vmin, __float_as_uint(tlo_dn));
        } else if (anylow == 0 && !key_live) {
            // HIGH samples with no LOW sample within reach: all of them are rejected (transition_sink.py:71-74)
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const float t = x[j] - prev[j];
                if (!(x[j] > thi_up)) {
                    b_acc += fabsf(t);
                    dl_acc += t;
                    ring[slot[j]] = x[j];
                    mark(slot[j]);
                }
            }
            vmin = min(vmin, __float_as_uint(tlo_dn));
        } else {
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
                    mark(slot[j]);
                }
                const uint32_t xb = __float_as_uint(x[j]);
                vmin = min(vmin, (a && xb != 0u) ? xb : 0xFFFFFFFFu);
                vmax = max(vmax, a ? xb : 0u);
                posm[j] = __ballot(ps);
                if constexpr (MASKED) posm[j] &= am[j];
                before = lowm[j] ? rb + last_set(lowm[j]) : before;
            }
        }
        int step_nl = (int)(base + STEPN) - 1, step_ll = LL_NONE;   // nothing LOW: the last sample is the last non-LOW
        if constexpr (MASKED) {
            step_nl = LL_NONE;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int rb = (int)(base + 64u * j);
                const unsigned long long nonlow = ~lowm[j] & am[j];
                step_ll = lowm[j] ? rb + last_set(lowm[j]) : step_ll;
                step_nl = nonlow ? rb + last_set(nonlow) : step_nl;
                if ((unt[j] >> lane) & 1ull) {
                    if constexpr (SIGN_T) ring[slot[j]] = __uint_as_float(__float_as_uint(x[j]) | 0x80000000u);
                    else tch[slot[j]] = 0;
                }
            }
        } else if (anylow) {
            step_nl = LL_NONE;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const int rb = (int)(base + 64u * j);
                const unsigned long long nonlow = ~lowm[j];
                step_ll = lowm[j] ? rb + last_set(lowm[j]) : step_ll;
                step_nl = nonlow ? rb + last_set(nonlow) : step_nl;
            }
        }
        if (anylow | anyhi) {
#pragma unroll
            for (int j = 0; j < NR; j++) nrej += (uint32_t)__popcll(lowm[j]) + (uint32_t)__popcll(posm[j]);
        }
        if (step_ll != LL_NONE) {
            w_kl = 2 * step_ll + 1;
            chunk_kl = w_kl;
        }
        if (step_nl != LL_NONE) {
            w_nl = step_nl;
            chunk_nl = step_nl;
        }
        {   // the step's NR words per plane (see k_threshold): only non-zero masks cost their v_writelane
            int pk = 0;
            if (anylow) {
#pragma unroll
                for (int k = 0; k < NR; k++) {
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(pk) : "s"((uint32_t)lowm[k]), "n"(2 * k));
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(pk) : "s"((uint32_t)(lowm[k] >> 32)), "n"(2 * k + 1));
                }
            }
            if (anyhi) {
#pragma unroll
                for (int k = 0; k < NR; k++) {
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(pk) : "s"((uint32_t)posm[k]), "n"(2 * NR + 2 * k));
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(pk) : "s"((uint32_t)(posm[k] >> 32)), "n"(2 * NR + 2 * k + 1));
                }
            }
            const int h = lane & (2 * NR - 1);
            const uint32_t w = (base >> 6) + (uint32_t)(h >> 1);
            uint32_t *dst = (uint32_t *)(lane < 2 * NR ? neg_p : pos_p) + 2 * (size_t)(base >> 6) + h;
            if (lane < 4 * NR && (size_t)w * 64 < A.n) *dst = (uint32_t)pk;
        }
        slot_step += STEPN;
        slot_step = (slot_step >= (uint32_t)L) ? slot_step - (uint32_t)L : slot_step;
        steps_since_sync++;
        base += STEPN;
        return true;
    };
    using MaskedT = std::integral_constant<bool, true>;
    using WholeT = std::integral_constant<bool, false>;

    // (1) the step that holds the stream's first stable sample (chunk 0 of its first batches), a superstep of its own
    if (good_run && base < n1 && base < m_start) {
        fetch(base, r[0]);
        good_run = open_superstep() && step(MaskedT{}, r[0]) && close_superstep(1, 0);
    }
    // (2) whole steps, PF to a superstep; step k of a superstep lives in r[k]
    if (good_run && base + STEPN <= n1) {
        last_whole = base + (n1 - base - STEPN) / STEPN * STEPN;
#pragma unroll
        for (int k = 0; k < PF; k++) fetch_whole(min(base + (uint32_t)k * STEPN, last_whole), r[k]);
#pragma unroll
        for (int k = 0; k < PF; k++) lean_wait<0>(r[k]);   // (the first allowance reads them all)
        {   // first allowance: sum |x - prev| over the samples of the first superstep that look acceptable (a guess like any
            // other allowance: the superstep's own B decides)
            const float wlo = ssf * loLf * 0.5f, whi = ssf * hiLf * 1.02f;
            float b0 = 0.f;
            uint32_t sl = slot_step;
#pragma unroll
            for (int k = 0; k < PF; k++) {
                if (base + (uint32_t)(k + 1) * STEPN <= n1) {
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        const float xv = lean_env<KIND>(r[k][j], A.i16_scale);
                        uint32_t q = sl + 64u * j + lane;
                        q = (q >= (uint32_t)L) ? q - (uint32_t)L : q;
                        const float pv = SIGN_T ? fabsf(ring[q]) : ring[q];
                        b0 += (xv > wlo && xv < whi) ? fabsf(xv - pv) : 0.f;
                    }
                    sl += STEPN;
                    sl = (sl >= (uint32_t)L) ? sl - (uint32_t)L : sl;
                }
            }
            G = rfl(fminf(fmaxf(A.gfac * wave_sum_f32(b0), ssf * A.gfloor), ssf * 0.125f));
        }
    }
    while (good_run && base + (uint32_t)PF * STEPN <= n1) {
        if (!open_superstep()) { good_run = false; break; }
        bool okk = true;
#pragma unroll
        for (int k = 0; k < PF; k++) okk = okk && step(WholeT{}, r[k]);
        if (!okk || !close_superstep(PF, PF)) { good_run = false; break; }
    }
#pragma unroll
    for (int k = 0; k < PF; k++) lean_wait<0>(r[k]);   // nothing of the loop's may still be landing in registers the code below reuses
    // (2b) fewer than PF whole steps left (chunk lengths are cut to multiples of PF steps: only a batch's last chunk has
    // them): one at a time, each loading its own samples
    while (good_run && base + STEPN <= n1) {
        fetch(base, r[0]);
        G = rfl(fminf(fmaxf(G, ssf * A.gfloor), ssf * 0.125f));
        if (!(open_superstep() && step(MaskedT{}, r[0]) && close_superstep(1, 0))) { good_run = false; break; }
    }
    // (3) the batch's ragged end
    if (good_run && base < n1) {
        fetch(base, r[0]);
        G = rfl(fminf(fmaxf(G, ssf * 0.00390625f), ssf * 0.125f));
        good_run = open_superstep() && step(MaskedT{}, r[0]) && close_superstep(1, 0);
    }
    const uint32_t all_robust = good_run ? 1u : 0u;
    if (c == 0 && lane == 0) {   // chunk 0 has no certification of its own: its verdict travels here
        A.cert[0] = good_run ? 1 : 0;
        if (!good_run) atomicAdd(&A.sum->n_fail, 1u);
    }
    chunk_publish<SIGN_T>(A, c, lane, ring, tch, emin, emax, vmin, vmax, ssf, eps, good_run ? 0u : (4u | (why << 4)), chunk_kl, chunk_nl, (double)ssf,
                          min_ss, nl_in, kl_in, all_robust);
}

}  // namespace nfc
