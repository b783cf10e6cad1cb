// threshold.hip.h -- envelope + gated running-mean threshold (transition_sink.py:37-82)
// as time-chunked CDNA4 kernels.
//
// The reference classifies sample n from ratio = x[n]*L/ss[n], where ss is the sum
// of a ring of the last L *accepted* samples; rejected samples leave their ring
// slot untouched, so ss[n] depends on every earlier classification.  Here the
// stream is cut into time chunks of C samples, one wavefront per chunk:
//
//   * inside a chunk the wave walks 256 samples per step (4 contiguous samples
//     per lane, 16-byte coalesced loads).  It guesses the accept mask from the
//     step's starting sum, gets every sample's exact sum with a wave prefix
//     scan of the accepted (x - prev) deltas, reclassifies, and repeats until the
//     mask is a fixed point -- which is unique and equals the sequential result;
//   * across chunks the incoming ring is first speculated (pass 0: the last L
//     raw samples with rejected-looking ones replaced by a level estimate), then
//     resolved exactly from the predecessors' published (touched, value)
//     summaries by look-back (pass 1: verify).  A chunk whose summary changed
//     triggers re-evaluation of the chunks that can see it; when nothing changes
//     the result is the reference's, by induction from chunk 0.
//
// All sums are fp64 and exact while the window's exponent spread fits 53 bits
// (tracked per chunk); otherwise the host runs k_threshold_seq, a one-lane
// restatement of the loop, so results stay bit-identical in every case.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nfc {

constexpr int STEP = 256;            // samples per wave step
constexpr int LL_NONE = -(1 << 30);  // "no LOW sample seen" (batch-local index)
constexpr int MAX_FIX_ITERS = 300;

enum : int { IN_IQ_F32 = 0, IN_ENV_F32 = 1, IN_REAL_F32_SQ = 2, IN_I16_SQ = 3 };

// Carried stream state (device resident; mirrored to the host after each batch).
struct Carry {
    double ss;         // transition_sink._sum
    double delta;      // ss - S(ring): the part of ss that is not the exact ring sum (0 unless a sum was inexact)
    int32_t filled;    // transition_sink._filled
    int32_t stable;
    int32_t ss_emin;   // guard: lowest set bit of ss/delta on the f32 exponent-field scale, 255 if zero
    int32_t ss_emax;
    int32_t ring_emin; // guard over the carried ring values
    int32_t ring_emax;
    int32_t inexact;   // a sequential sum rounded at least once
    int32_t pad;
};

// "HIGH is ignored" (cur_state == 2, transition_sink.py:71) at sample n  <=>  the last LOW sample m < n
// did not end on a run-length timeout (transition_sink.py:95-99 resets the state) and n - m <= max_len + 1.
// Tracked as two running maxima: the last non-LOW index (gives a LOW run's start) and
// key = 2*m + good for the last LOW sample.
constexpr int KEY_NONE = INT32_MIN;  // even: good = 0

struct ChunkInfo {
    double ss_out;
    int32_t low_key;     // key of the last LOW sample in the chunk (batch-local index), KEY_NONE if none
    int32_t last_nonlow; // batch-local index of the last non-LOW sample in the chunk, LL_NONE if none
    uint32_t emin;       // guard: min/max f32 exponent field over ring-in and accepted samples
    uint32_t emax;
    uint32_t flags;      // 1 = fix-point iteration cap hit (internal error)
    uint32_t n_untouched;
};

struct ThrArgs {
    const void *in;
    uint32_t n;       // samples in the batch
    uint32_t skip;    // leading samples consumed by the fill phase
    uint32_t g0modL;  // (global index of batch sample 0) mod L
    int32_t L, Lpad, mx, C, nchunks;
    double lo, hi, hi_plus, lo_a, lo_b, hi_a, hi_b;
    int32_t bands_ok;
    float i16_scale;
    const float *ring_carry;
    const Carry *carry;
    int32_t nl0, kl0;      // carried last-non-LOW index and LOW key at the batch start (batch-local, <= -1)
    float *ring_out[2];
    uint32_t *touched[2];  // [nchunks][twords]
    ChunkInfo *info[2];
    const uint8_t *ver;    // which buffer holds chunk c's current summary
    uint8_t *changed;      // out: summary differs from the current one
    uint8_t *gmin, *gmax, *gflags;  // out: guard exponents / flags of the chunk's latest evaluation
    uint8_t *val;          // 2-bit codes, 4 samples per byte: 0 accepted, 1 HIGH, 2 LOW
    const uint32_t *list;  // chunks to run (nullptr: all)
    uint32_t nlist;
    int32_t mode;          // 0 speculate, 1 resolve exactly
    int32_t twords;        // u32 words per touched bitmap
};

__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    int lo = __shfl_up(__double2loint(v), d, 64), hi = __shfl_up(__double2hiint(v), d, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int l) {
    int lo = __shfl(__double2loint(v), l, 64), hi = __shfl(__double2hiint(v), l, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        int lo = __shfl_xor(__double2loint(v), d, 64), hi = __shfl_xor(__double2hiint(v), d, 64);
        v += __hiloint2double(hi, lo);
    }
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

// Envelope of one sample (gnuradio complex_to_mag_squared; compiled with
// -ffp-contract=off so the products and the sum round separately).
template <int KIND>
__device__ __forceinline__ float envelope_at(const void *in, size_t m, float i16_scale) {
    if (KIND == IN_IQ_F32) {
        const float2 v = ((const float2 *)in)[m];
        const float a = v.x * v.x, b = v.y * v.y;
        return a + b;
    } else if (KIND == IN_ENV_F32) {
        return ((const float *)in)[m];
    } else if (KIND == IN_REAL_F32_SQ) {
        const float s = ((const float *)in)[m];
        return s * s;
    } else {
        const float s = (float)((const int16_t *)in)[m] * i16_scale;
        return s * s;
    }
}

// Four consecutive samples starting at m (m % 4 == 0, fully inside the buffer).
template <int KIND>
__device__ __forceinline__ void load4(const void *in, size_t m, float i16_scale, float x[4]) {
    if (KIND == IN_IQ_F32) {
        const float4 *p = (const float4 *)in + (m >> 1);
        const float4 a = p[0], b = p[1];
        const float a0 = a.x * a.x, a1 = a.y * a.y, a2 = a.z * a.z, a3 = a.w * a.w;
        const float b0 = b.x * b.x, b1 = b.y * b.y, b2 = b.z * b.z, b3 = b.w * b.w;
        x[0] = a0 + a1;
        x[1] = a2 + a3;
        x[2] = b0 + b1;
        x[3] = b2 + b3;
    } else if (KIND == IN_ENV_F32) {
        const float4 a = ((const float4 *)in)[m >> 2];
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    } else if (KIND == IN_REAL_F32_SQ) {
        const float4 a = ((const float4 *)in)[m >> 2];
        x[0] = a.x * a.x; x[1] = a.y * a.y; x[2] = a.z * a.z; x[3] = a.w * a.w;
    } else {
        const short4 a = ((const short4 *)in)[m >> 2];
        const float s0 = (float)a.x * i16_scale, s1 = (float)a.y * i16_scale;
        const float s2 = (float)a.z * i16_scale, s3 = (float)a.w * i16_scale;
        x[0] = s0 * s0; x[1] = s1 * s1; x[2] = s2 * s2; x[3] = s3 * s3;
    }
}

__device__ __forceinline__ uint32_t f32_expfield(float v) { return (__float_as_uint(v) >> 23) & 0xFFu; }

// transition_sink.py:59-77 for one sample, given the exact running sum.
__device__ __forceinline__ void classify_one(const ThrArgs &A, double x64, double ss, bool &low, bool &high) {
    const double p = x64 * (double)A.L;
    bool amb = true;
    if (A.bands_ok && ss > 1e-150 && ss < 1e150) {
        amb = false;
        if (p < A.lo_a * ss) low = true;
        else if (p > A.lo_b * ss) low = false;
        else amb = true;
        if (p > A.hi_b * ss) high = true;
        else if (p < A.hi_a * ss) high = false;
        else amb = true;
    }
    if (amb) {
        double ratio;
        if (ss == 0) ratio = (x64 == 0) ? 1.0 : A.hi_plus;
        else ratio = p / ss;
        low = A.lo > ratio;
        high = ratio > A.hi;
    }
}

// Where a chunk's look-back ends: newest predecessor that saw a non-LOW sample / a LOW sample.
__device__ __forceinline__ void resolve_low_state(const ThrArgs &A, int c, int &nl, int &kl) {
    nl = LL_NONE;
    kl = KEY_NONE;
    bool have_nl = false, have_kl = false;
    for (int cc = c - 1; cc >= 0 && !(have_nl && have_kl); cc--) {
        const ChunkInfo ci = A.info[A.ver[cc]][cc];
        if (!have_nl && ci.last_nonlow != LL_NONE) { nl = ci.last_nonlow; have_nl = true; }
        if (!have_kl && ci.low_key != KEY_NONE) { kl = ci.low_key; have_kl = true; }
    }
    if (!have_nl) nl = A.nl0;
    if (!have_kl) kl = A.kl0;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_threshold(ThrArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const uint32_t slotid = blockIdx.x * wpb + wave;
    uint32_t c;
    if (A.list) {
        if (slotid >= A.nlist) return;
        c = A.list[slotid];
    } else {
        if (slotid >= (uint32_t)A.nchunks) return;
        c = slotid;
    }
    float *ring = (float *)(smem + (size_t)wave * ((size_t)A.Lpad * 5));
    unsigned char *tch = (unsigned char *)(ring + A.Lpad);
    const int L = A.L;
    const uint32_t m_chunk = c * (uint32_t)A.C;
    const uint32_t n1 = min(A.n, m_chunk + (uint32_t)A.C);
    const uint32_t m_start = max(m_chunk, A.skip);
    const Carry cr = *A.carry;

    uint32_t emin = 255u, emax = 0u;
    int w_nl, w_kl;  // last non-LOW index / key of the last LOW sample, before the current step
    double ss0;

    // ---------------- incoming state ----------------
    if (c == 0) {
        for (int s = lane; s < L; s += 64) ring[s] = A.ring_carry[s];
        ss0 = cr.ss;
        w_nl = A.nl0;
        w_kl = A.kl0;
    } else if (A.mode == 1) {
        // exact: latest predecessor that accepted a sample into the slot, else the carried ring
        for (int s = lane; s < L; s += 64) {
            int cc = (int)c - 1;
            float v;
            for (;;) {
                if (cc < 0) { v = A.ring_carry[s]; break; }
                const int vb = A.ver[cc];
                const uint32_t w = A.touched[vb][(size_t)cc * A.twords + (s >> 5)];
                if ((w >> (s & 31)) & 1u) { v = A.ring_out[vb][(size_t)cc * L + s]; break; }
                cc--;
            }
            ring[s] = v;
        }
        double part = 0;
        for (int s = lane; s < L; s += 64) part += (double)ring[s];
        ss0 = wave_sum_f64(part) + cr.delta;
        resolve_low_state(A, (int)c, w_nl, w_kl);
    } else {
        // speculate: the L samples before the chunk, rejected-looking ones replaced by a level estimate
        const uint32_t w0 = m_chunk - (uint32_t)L;
        float mxv = 0.f;
        for (int i = lane; i < L; i += 64) {
            const uint32_t m = w0 + i;
            const float x = envelope_at<KIND>(A.in, m, A.i16_scale);
            const uint32_t slot = (A.g0modL + m) % (uint32_t)L;
            ring[slot] = x;
            mxv = fmaxf(mxv, x);
        }
        mxv = wave_max_f32(mxv);
        const float half = 0.5f * mxv;
        double sa = 0, na = 0;
        for (int s = lane; s < L; s += 64) {
            const float x = ring[s];
            if (x >= half) { sa += (double)x; na += 1.0; }
        }
        sa = wave_sum_f64(sa);
        na = wave_sum_f64(na);
        const float ca = (na > 0) ? (float)(sa / na) : mxv;
        double sb = 0, nb = 0;
        for (int s = lane; s < L; s += 64) {
            const float x = ring[s];
            if (x >= half && x <= ca) { sb += (double)x; nb += 1.0; }
        }
        sb = wave_sum_f64(sb);
        nb = wave_sum_f64(nb);
        const float c0 = (nb > 0) ? (float)(sb / nb) : ca;
        const float tlo = (float)A.lo * c0, thi = (float)A.hi * c0;
        int ll = LL_NONE, nl = LL_NONE;
        for (int i = lane; i < L; i += 64) {
            const uint32_t m = w0 + i;
            const uint32_t slot = (A.g0modL + m) % (uint32_t)L;
            const float x = ring[slot];
            if (x < tlo) ll = max(ll, (int)m);
            else nl = max(nl, (int)m);
            if (!(x >= tlo && x <= thi)) ring[slot] = c0;
        }
        ll = wave_max_i32(ll);
        w_nl = wave_max_i32(nl);
        w_kl = (ll == LL_NONE) ? KEY_NONE : 2 * ll + 1;
        double part = 0;
        for (int s = lane; s < L; s += 64) part += (double)ring[s];
        ss0 = wave_sum_f64(part) + cr.delta;
    }
    for (int s = lane; s < A.Lpad; s += 64) tch[s] = 0;
    for (int s = lane; s < L; s += 64) {
        const float v = ring[s];
        if (v != 0.f) {
            const uint32_t e = max(f32_expfield(v), 1u);
            emin = min(emin, e);
            emax = max(emax, e);
        }
    }

    // ---------------- the chunk, 256 samples per step ----------------
    uint32_t flags = 0;
    int chunk_nl = LL_NONE, chunk_kl = KEY_NONE;
    uint32_t slot_step = (A.g0modL + m_chunk) % (uint32_t)L;
    const int mx = A.mx;
    for (uint32_t base = m_chunk; base < n1; base += STEP) {
        const uint32_t m0 = base + 4u * lane;
        float x[4];
        if (base + STEP <= A.n) {
            load4<KIND>(A.in, m0, A.i16_scale, x);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) x[j] = (m0 + j < A.n) ? envelope_at<KIND>(A.in, m0 + j, A.i16_scale) : 0.f;
        }
        bool act[4];
        uint32_t slot[4];
        float prev[4];
        double x64[4];
        uint32_t s0 = slot_step + 4u * lane;
        s0 = (s0 >= (uint32_t)L) ? s0 % (uint32_t)L : s0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t m = m0 + j;
            act[j] = (m >= m_start) && (m < n1);
            uint32_t s = s0 + j;
            s = (s >= (uint32_t)L) ? s - (uint32_t)L : s;
            slot[j] = s;
            prev[j] = ring[s];
            x64[j] = (double)x[j];
        }
        bool low[4], high[4], acc[4];
        int val[4], key[4];
        double ssj[4] = {ss0, ss0, ss0, ss0};
        int lane_nl = LL_NONE, lane_kl = KEY_NONE;
        int iter = 0;
        for (;;) {
            // ratio classification with the current per-sample sums (transition_sink.py:59-71)
            lane_nl = LL_NONE;
            bool anyl = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bool lw = false, hg = false;
                if (act[j]) classify_one(A, x64[j], ssj[j], lw, hg);
                low[j] = lw;
                high[j] = hg;
                anyl |= lw;
                if (act[j] && !lw) lane_nl = (int)(m0 + j);
            }
            const bool step_low = __any(anyl);
            const bool key_live = (w_kl & 1) && ((int)base - (w_kl >> 1)) <= mx + 1;
            bool st2[4] = {false, false, false, false};
            lane_kl = KEY_NONE;
            if (step_low || key_live) {
                // (1) last non-LOW index before this lane
                int inc = lane_nl;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int up = __shfl_up(inc, d, 64);
                    if (lane >= d) inc = max(inc, up);
                }
                int nl = __shfl_up(inc, 1, 64);
                if (lane == 0) nl = LL_NONE;
                nl = max(nl, w_nl);
                // keys of this lane's LOW samples: position in their LOW run decides "ended on a timeout"
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int m = (int)(m0 + j);
                    key[j] = KEY_NONE;
                    if (low[j]) {
                        const int p = m - nl;   // 1-based position in the LOW run
                        const bool bad = (p > mx) && ((p - 1) % mx == 0);
                        key[j] = 2 * m + (bad ? 0 : 1);
                        lane_kl = key[j];
                    } else if (act[j]) {
                        nl = m;
                    }
                }
                // (2) key of the last LOW sample before this lane
                int kinc = lane_kl;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int up = __shfl_up(kinc, d, 64);
                    if (lane >= d) kinc = max(kinc, up);
                }
                int kl = __shfl_up(kinc, 1, 64);
                if (lane == 0) kl = KEY_NONE;
                kl = max(kl, w_kl);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int m = (int)(m0 + j);
                    st2[j] = (kl & 1) && (m - (kl >> 1)) <= mx + 1;
                    if (low[j]) kl = key[j];
                }
            }
            bool same = true;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int v = 0;
                if (low[j]) v = -1;
                else if (high[j] && !st2[j]) v = 1;
                const bool a = act[j] && (v == 0);
                if (iter > 0 && a != acc[j]) same = false;
                acc[j] = a;
                val[j] = v;
            }
            if (iter > 0 && __all(same)) break;
            if (iter >= MAX_FIX_ITERS) { flags |= 1u; break; }
            // exact running sums under this accept mask (transition_sink.py:80-82)
            double l[4];
            double run = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (acc[j]) run += (x64[j] - (double)prev[j]);
                l[j] = run;
            }
            double incs = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double up = shfl_up_f64(incs, d);
                if (lane >= d) incs += up;
            }
            double ex = shfl_up_f64(incs, 1);
            if (lane == 0) ex = 0;
            const double b = ss0 + ex;
            ssj[0] = b;
            ssj[1] = b + l[0];
            ssj[2] = b + l[1];
            ssj[3] = b + l[2];
            iter++;
        }
        // commit the step
        double run = 0;
        uint32_t code = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (acc[j]) {
                run += (x64[j] - (double)prev[j]);
                ring[slot[j]] = x[j];
                tch[slot[j]] = 1;
                if (x[j] != 0.f) {
                    const uint32_t e = max(f32_expfield(x[j]), 1u);
                    emin = min(emin, e);
                    emax = max(emax, e);
                }
            }
            code |= (uint32_t)(val[j] == 1 ? 1u : (val[j] == -1 ? 2u : 0u)) << (2 * j);
        }
        ss0 += wave_sum_f64(run);
        const int step_nl = wave_max_i32(lane_nl);
        const int step_kl = wave_max_i32(lane_kl);
        w_nl = max(w_nl, step_nl);
        w_kl = max(w_kl, step_kl);
        chunk_nl = max(chunk_nl, step_nl);
        chunk_kl = max(chunk_kl, step_kl);
        if (m0 < A.n) A.val[m0 >> 2] = (unsigned char)code;
        slot_step += STEP;
        slot_step = (slot_step >= (uint32_t)L) ? slot_step % (uint32_t)L : slot_step;
    }

    // ---------------- publish the summary ----------------
    const int vb_old = A.ver[c];
    const int vb_new = (A.mode == 1) ? (1 - vb_old) : vb_old;  // pass 0 fills buffer ver[c] directly
    float *ro = A.ring_out[vb_new] + (size_t)c * L;
    uint32_t *to = A.touched[vb_new] + (size_t)c * A.twords;
    bool diff = false;
    uint32_t untouched = 0;
    const float *ro_old = A.ring_out[vb_old] + (size_t)c * L;
    const uint32_t *to_old = A.touched[vb_old] + (size_t)c * A.twords;
    for (int sbase = 0; sbase < A.twords * 32; sbase += 64) {
        const int s = sbase + lane;
        const bool t = (s < L) && tch[s];
        const float v = (s < L) ? ring[s] : 0.f;
        const unsigned long long bal = __ballot(t);
        if (s < L) {
            ro[s] = v;
            if (!t) untouched++;
        }
        const int w = sbase >> 5;
        if (A.mode == 1) {
            if (s < L && t && __float_as_uint(ro_old[s]) != __float_as_uint(v)) diff = true;
            if (lane == 0) {
                if (to_old[w] != (uint32_t)bal) diff = true;
                if (w + 1 < A.twords && to_old[w + 1] != (uint32_t)(bal >> 32)) diff = true;
            }
        }
        if (lane == 0) {
            to[w] = (uint32_t)bal;
            if (w + 1 < A.twords) to[w + 1] = (uint32_t)(bal >> 32);
        }
    }
    emin = wave_min_u32(emin);
    emax = wave_max_u32(emax);
    untouched = (uint32_t)wave_sum_f64((double)untouched);
    flags = wave_max_u32(flags);
    if (A.mode == 1) {
        const ChunkInfo old = A.info[vb_old][c];
        if (old.low_key != chunk_kl || old.last_nonlow != chunk_nl) diff = true;
        const bool any = __any(diff);
        if (lane == 0) A.changed[c] = any ? 1 : 0;
    } else if (lane == 0) {
        A.changed[c] = 1;
    }
    if (lane == 0) {
        ChunkInfo ci;
        ci.ss_out = ss0;
        ci.low_key = chunk_kl;
        ci.last_nonlow = chunk_nl;
        ci.emin = emin;
        ci.emax = emax;
        ci.flags = flags;
        ci.n_untouched = untouched;
        A.info[vb_new][c] = ci;
        A.gmin[c] = (uint8_t)emin;
        A.gmax[c] = (uint8_t)emax;
        A.gflags[c] = (uint8_t)(flags | (untouched ? 2u : 0u));
    }
}

// ---------------------------------------------------------------------------
// Fill phase (transition_sink.py:109-125): copy the first L samples into the ring,
// then sum them in the reference's order.  One wave; the sum is one lane.
// ---------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(64) void k_fill(const void *in, uint32_t n, float i16_scale, int L, float *ring, Carry *carry) {
    const int lane = threadIdx.x;
    const int filled = carry->filled;
    const int can = min((int)n, L - filled);
    for (int i = lane; i < can; i += 64) ring[filled + i] = envelope_at<KIND>(in, (size_t)i, i16_scale);
    __syncthreads();
    if (lane == 0) {
        carry->filled = filled + can;
        if (filled + can == L) {
            double s = 0, err = 0;
            for (int i = 0; i < L; i++) {
                const double v = (double)ring[i];
                const double t = s + v;            // transition_sink.py:122, sequential
                const double bv = t - s;           // TwoSum residue: was the addition exact?
                err += fabs((s - (t - bv)) + (v - bv));
                s = t;
            }
            carry->ss = s;
            carry->stable = 1;
            if (err != 0.0) carry->inexact = 1;
        }
    }
}

// Lowest / highest set bit of a double on the f32 exponent-field scale (bit value 2^(field-127)).
__device__ __forceinline__ void f64_bit_span(double v, int &elow, int &ehigh) {
    if (v == 0) { elow = 255; ehigh = 0; return; }
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int e = (int)((b >> 52) & 0x7FF);
    unsigned long long m = b & 0xFFFFFFFFFFFFFull;
    if (e) m |= 1ull << 52;
    const int tz = __ffsll((long long)m) - 1;
    const int unb = (e ? e : 1) - 1023;   // exponent of the hidden-bit position
    elow = unb - 52 + tz + 127;
    ehigh = unb + 127;
    if (e == 0x7FF) { elow = -4000; ehigh = 4000; }   // inf/nan: never provably exact
}

// Per batch: S(ring), delta = ss - S(ring), and the guard span of the carried values.
__global__ __launch_bounds__(64) void k_prepare(const float *ring, int L, Carry *carry) {
    const int lane = threadIdx.x;
    double part = 0;
    uint32_t emin = 255u, emax = 0u;
    for (int s = lane; s < L; s += 64) {
        const float v = ring[s];
        part += (double)v;
        if (v != 0.f) {
            const uint32_t e = max(f32_expfield(v), 1u);
            emin = min(emin, e);
            emax = max(emax, e);
        }
    }
    const double S = wave_sum_f64(part);
    emin = wave_min_u32(emin);
    emax = wave_max_u32(emax);
    if (lane == 0) {
        const double ss = carry->ss;
        const double delta = ss - S;
        carry->delta = delta;
        int el, eh, el2, eh2;
        f64_bit_span(ss, el, eh);
        f64_bit_span(delta, el2, eh2);
        carry->ss_emin = min(el, el2);
        carry->ss_emax = max(eh, eh2);
        carry->ring_emin = (int)emin;
        carry->ring_emax = (int)emax;
    }
}

// After the passes converged: the ring at the end of the batch (look-back over all
// chunks) and its sum become the carried state.
struct FinArgs {
    int32_t L, nchunks, twords;
    const float *ring_carry;
    float *ring_next;
    float *ring_out[2];
    uint32_t *touched[2];
    const uint8_t *ver;
    Carry *carry;
};
__global__ __launch_bounds__(64) void k_finalize_state(FinArgs A) {
    const int lane = threadIdx.x;
    double part = 0;
    for (int s = lane; s < A.L; s += 64) {
        int cc = A.nchunks - 1;
        float v;
        for (;;) {
            if (cc < 0) { v = A.ring_carry[s]; break; }
            const int vb = A.ver[cc];
            const uint32_t w = A.touched[vb][(size_t)cc * A.twords + (s >> 5)];
            if ((w >> (s & 31)) & 1u) { v = A.ring_out[vb][(size_t)cc * A.L + s]; break; }
            cc--;
        }
        A.ring_next[s] = v;
        part += (double)v;
    }
    const double S = wave_sum_f64(part);
    if (lane == 0) A.carry->ss = S + A.carry->delta;
}

// ---------------------------------------------------------------------------
// Literal sequential restatement of transition_sink.py:55-99 (classification part)
// on ONE lane.  Used when the fp64 window sums cannot be proven exact (so the
// summation order matters) and under NFC_FLAG_FORCE_SEQUENTIAL.  Slow by
// construction; it carries the reference's own state variables.
// ---------------------------------------------------------------------------
struct SeqArgs {
    const void *in;
    uint32_t n, skip, g0modL;
    int32_t L, mx;
    double lo, hi, hi_plus;
    float i16_scale;
    float *ring;   // carried ring, updated in place
    Carry *carry;
    int32_t state, last_bit, dur;   // transition_sink._current_state/_last_bit/_dur at the batch start
    uint8_t *val;
};
template <int KIND>
__global__ __launch_bounds__(64) void k_threshold_seq(SeqArgs A) {
    if (threadIdx.x != 0) return;
    double ss = A.carry->ss;
    double err = 0;
    int state = A.state, last_bit = A.last_bit, dur = A.dur;
    uint32_t slot = (A.g0modL + A.skip) % (uint32_t)A.L;
    uint32_t code = 0;
    const double Ld = (double)A.L;
    for (uint32_t m = 0; m < A.n; m++) {
        int v = 0;
        if (m >= A.skip) {
            const float xf = envelope_at<KIND>(A.in, m, A.i16_scale);
            const double bit = (double)xf;
            const double prev = (double)A.ring[slot];
            double ratio;
            if (ss == 0) ratio = (bit == 0) ? 1.0 : A.hi_plus;
            else ratio = bit * Ld / ss;
            double cur;
            if (A.lo > ratio) { v = -1; cur = prev; state = 2; }
            else if (state != 2 && ratio > A.hi) { v = 1; cur = prev; state = 1; }
            else { v = 0; cur = bit; A.ring[slot] = xf; }
            const double dlt = cur - prev;
            const double t = ss + dlt;
            const double bv = t - ss;
            err += fabs((ss - (t - bv)) + (dlt - bv));
            ss = t;
            slot++;
            if (slot == (uint32_t)A.L) slot = 0;
            if (v == last_bit) dur++;
            else { dur = 1; last_bit = v; }
            if (dur > A.mx) { dur = 1; state = 0; }
        }
        code |= (uint32_t)(v == 1 ? 1u : (v == -1 ? 2u : 0u)) << (2 * (m & 3));
        if ((m & 3) == 3 || m + 1 == A.n) {
            A.val[m >> 2] = (unsigned char)code;
            code = 0;
        }
    }
    A.carry->ss = ss;
    if (err != 0.0) A.carry->inexact = 1;
}

}  // namespace nfc
