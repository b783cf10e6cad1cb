// threshold.hip.h -- envelope + gated running-mean threshold (transition_sink.py:37-82)
// as time-chunked CDNA4 kernels.
//
// The reference classifies sample n from ratio = x[n]*L/ss[n], where ss is the sum of a
// ring of the last L *accepted* samples; rejected samples leave their ring slot untouched,
// so ss[n] depends on every earlier classification.  Here the stream is cut into time
// chunks of C samples, one wavefront per chunk, walking 256 samples per step (lane l holds
// samples l, 64+l, 128+l, 192+l of the step, so a __ballot is 64 consecutive samples):
//
//   * fast path of a step: the drift of ss inside the step is bounded by B = sum |x - prev|
//     over the samples that can be accepted; a sample whose x lies outside the lo/hi bands
//     widened by B (and by the certification margin, below) has the same classification for
//     every possible ss, so no per-sample sum is needed -- two wave reductions and a few
//     ballots per 256 samples;
//   * exact path (any sample inside a band, long LOW runs, odd parameters): per 64-sample
//     row, the accept mask is iterated to its unique fixed point with a wave prefix scan of
//     the accepted (x - prev) deltas -- this equals the sequential result;
//   * across chunks the incoming ring is first speculated (pass 0: the previous L samples
//     with rejected-looking ones replaced by a level estimate, classification margins
//     widened by eps*ss).  k_certify then resolves every chunk's true incoming ring from the
//     predecessors' (touched, value) summaries by look-back and proves the speculation
//     harmless when the L1 distance between the two rings is below the margin; the few
//     chunks that cannot be proven are re-run from the exact state, and chunks that can see
//     a re-run chunk are certified again.  When nothing is left the result is the
//     reference's, by induction from chunk 0.
//
// All sums are fp64 and exact while the window's exponent spread fits 53 bits (tracked per
// chunk); otherwise the host runs k_threshold_seq, a one-lane literal restatement of the
// loop, so results stay bit-identical in every case.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace nfc {

constexpr int STEP = 256;            // smallest step (4 rows of 64 samples); the wide variant walks 512
constexpr int LL_NONE = -(1 << 30);  // "no such sample" (batch-local index)
constexpr int MAX_FIX_ITERS = 80;
constexpr float RND_SUM = 2.44140625e-04f;   // 2^-12: margin for the accumulated rounding of the f32 running sum

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
    int32_t fin_valid; // ss_fin holds the sum at the end of the last batch (applied by the next batch's preparation)
    double ss_fin;     // The end-of-batch sum does NOT overwrite ss: a batch whose sums cannot be proven exact is replayed
                       // by the sequential kernel from the batch-start ss after the parallel attempt has already finalized.
    int32_t low_nl, low_kl;   // the "HIGH ignored" bookkeeping (below) at the end of the batch, in the NEXT batch's sample
                              // indices (<= -1): what a batch enqueued before this one's results are read starts from
                              // (ThrArgs.low_src; the synchronous path derives the same from the edge stage's carried values)
};
__host__ __device__ inline void carry_apply_fin(Carry &c) {
    if (c.fin_valid) {
        c.ss = c.ss_fin;
        c.fin_valid = 0;
    }
}

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

// What a chunk's latest evaluation assumed, for k_certify.
struct RunMeta {
    float min_ss;        // smallest step-start sum seen (rounded down)
    float eps;           // relative margin every fast-path step carried (0: evaluated from the exact state)
    int32_t nl_in, kl_in;
    uint32_t all_robust; // every step took the fast path (or eps == 0)
    uint32_t pad;
};

struct CertSummary;
struct ThrArgs {
    void *gring;                // GRING kernels: per chunk a ring row (Lpad floats, + Lpad touched bytes for raw envelopes)
    const void *in;
    uint32_t n;       // samples in the batch
    uint32_t skip;    // leading samples consumed by the fill phase
    uint32_t g0modL;  // (global index of batch sample 0) mod L
    int32_t L, Lpad, mx, C, nchunks;
    double lo, hi, hi_plus, lo_a, lo_b, hi_a, hi_b;
    int32_t bands_ok;
    int32_t fast_ok;       // lo, hi > 0 and sane: the banded fast path is usable
    float i16_scale;
    float eps;             // certification margin of pass 0
    const float *ring_carry;
    const Carry *carry;
    int32_t nl0, kl0;      // carried last-non-LOW index and LOW key at the batch start (batch-local, <= -1) ...
    const Carry *low_src;  // ... or, when not null, where the previous batch's kernels left them (Carry.low_nl / low_kl)
    double lo_L, hi_L;     // lo / L, hi / L (the fast path's thresholds carry 2^-20 of slack)
    int32_t fold_sh[6];    // long-LOW-run detector: shifts of the six folds (0 = fold disabled) ...
    int32_t probe_mid, probe_end;   // ... pre-filter probes: b/2 and b-1 for aligned blocks of b samples
    uint64_t selmask;      // ... and the bit that survives the folds in every aligned block
    float *ring_out[2];
    uint32_t *touched[2];  // [nchunks][twords]
    ChunkInfo *info[2];
    const uint8_t *ver;    // which buffer holds chunk c's current summary
    float *ring_in;        // [nchunks][L] the ring each chunk's latest evaluation started from
    RunMeta *meta;
    uint8_t *gmin, *gmax, *gflags;  // out: guard exponents / flags of the chunk's latest evaluation
    uint32_t *gvtop;                // out: raw bits (f32) of an upper bound of every window sum of the chunk
    uint64_t *neg, *pos;   // classification bit planes, 64 samples per word: LOW / HIGH
    const uint32_t *list;  // chunks to run (nullptr: all)
    uint32_t nlist;
    int32_t mode;          // 0 speculate, 1 resolve exactly
    int32_t twords;        // u32 words per touched bitmap
    int32_t off;           // chunk c covers samples [c*C - off, (c+1)*C - off)  (register-ring kernel: ring-aligned)
    int32_t nrows;         // ceil(L / 64)
    // k_threshold_lean (threshold_lean.hip.h)
    uint8_t *cert;         // per-chunk verdict bytes: chunk 0's is written by the threshold kernel itself
    CertSummary *sum;      // n_fail takes chunk 0's failure
    int32_t ksteps;        // steps per superstep
    float gfac, gfloor;    // drift allowance = max(gfac * (largest B needed so far), gfloor * ss)
    int32_t blk;           // a LOW run longer than max_len covers an aligned block of blk samples (a power of two)
    int32_t ver_zero;              // every version byte is 0 (the certification right behind pass 0): its launches need not load them
    int32_t wg_stage_rounds;       // k_threshold_wg: rounds of plane words its LDS staging holds per plane (threshold_wg.hip.h; launch_wg sets it)
    unsigned long long *dbg_clk;   // debugging aid (NFC_DEBUG_CLK): per chunk four s_memtime stamps -- start, incoming state ready, loop done, end
    // Chunks of UNEQUAL length by dispatch row (round 5; chunk_span below).  The k-th workgroup a CU is given is the k-th slowest
    // (measured: the lives of k_threshold_wg's workgroups fall in four steps by block index -- 0.965, 0.985, 1.010, 1.042 of the mean
    // for blocks 0-255, 256-511, ... --, and the launch is as long as its slowest workgroup): the chunks of row r are row_len[r] samples
    // long, row r begins at sample row_start[r] and holds row_div chunks; rows past the third go on like the third.  With every
    // row_len = C and row_start[r] = r * row_div * C this is the equal cut, c * C.
    uint32_t row_len[4], row_start[4], row_div;
    int C_max;   // the longest row's chunk length (the sizing of a workgroup's plane staging: launch_wg)
};
// chunk c covers the batch's samples [m_chunk, m_chunk + len)
__device__ __forceinline__ void chunk_span(const ThrArgs &A, uint32_t c, uint32_t &m_chunk, uint32_t &len) {
    const uint32_t row = min(c / A.row_div, 3u);
    len = A.row_len[row];
    m_chunk = A.row_start[row] + (c - row * A.row_div) * len;
}

// ---------------------------------------------------------------------------
// cross-lane helpers
// ---------------------------------------------------------------------------
// Wave-uniform values the compiler cannot prove uniform (wave index, chunk id, running sum, ...) are
// moved to scalar registers explicitly: branches on them become scalar branches and the wave masks
// (__ballot results) stay in SGPRs instead of being rebuilt lane by lane.
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float rfl(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double rfl(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ bool uni(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
// DPP row_shr / row_bcast reductions (gfx9 family): result valid in lane 63, broadcast by readlane.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i32(int ident, int v) {
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += __int_as_float(dpp_i32<0x111, 0xF>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i32<0x112, 0xF>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i32<0x114, 0xF>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i32<0x118, 0xF>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i32<0x142, 0xA>(0, __float_as_int(v)));
    v += __int_as_float(dpp_i32<0x143, 0xC>(0, __float_as_int(v)));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = dpp_i32<CTRL, ROWMASK>(0, __double2loint(v));
    const int hi = dpp_i32<CTRL, ROWMASK>(0, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<0x111, 0xF>(v);
    v += dpp_f64<0x112, 0xF>(v);
    v += dpp_f64<0x114, 0xF>(v);
    v += dpp_f64<0x118, 0xF>(v);
    v += dpp_f64<0x142, 0xA>(v);
    v += dpp_f64<0x143, 0xC>(v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// max / min across the wave on the same DPP moves (a __shfl_xor is an LDS permute: ~60 cycles a step, six steps in a row).  All 64
// lanes must be active, as for the sums above.
template <class F>
__device__ __forceinline__ int wave_reduce_i32(int v, int ident, F f) {
    v = f(v, dpp_i32<0x111, 0xF>(ident, v));
    v = f(v, dpp_i32<0x112, 0xF>(ident, v));
    v = f(v, dpp_i32<0x114, 0xF>(ident, v));
    v = f(v, dpp_i32<0x118, 0xF>(ident, v));
    v = f(v, dpp_i32<0x142, 0xA>(ident, v));
    v = f(v, dpp_i32<0x143, 0xC>(ident, v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_max_f32(float v) {   // (v >= 0 or any finite value: the identity is -inf)
    return __int_as_float(wave_reduce_i32(__float_as_int(v), (int)0xFF800000u,
                                          [](int a, int b) { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); }));
}
__device__ __forceinline__ int wave_max_i32(int v) {
    return wave_reduce_i32(v, INT32_MIN, [](int a, int b) { return max(a, b); });
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    return (uint32_t)wave_reduce_i32((int)v, -1, [](int a, int b) { return (int)min((uint32_t)a, (uint32_t)b); });
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    return (uint32_t)wave_reduce_i32((int)v, 0, [](int a, int b) { return (int)max((uint32_t)a, (uint32_t)b); });
}
// ordered inclusive scans on the same moves, and the value of the lane below (`ident` in lane 0)
__device__ __forceinline__ int wave_scan_max_i32(int v, int ident) {
    v = max(v, dpp_i32<0x111, 0xF>(ident, v));
    v = max(v, dpp_i32<0x112, 0xF>(ident, v));
    v = max(v, dpp_i32<0x114, 0xF>(ident, v));
    v = max(v, dpp_i32<0x118, 0xF>(ident, v));
    v = max(v, dpp_i32<0x142, 0xA>(ident, v));
    v = max(v, dpp_i32<0x143, 0xC>(ident, v));
    return v;
}
__device__ __forceinline__ int wave_below_i32(int v, int ident) { return dpp_i32<0x138, 0xF>(ident, v); }   // wave_shr:1
__device__ __forceinline__ double wave_scan_sum_f64(double v) {
    v += dpp_f64<0x111, 0xF>(v);
    v += dpp_f64<0x112, 0xF>(v);
    v += dpp_f64<0x114, 0xF>(v);
    v += dpp_f64<0x118, 0xF>(v);
    v += dpp_f64<0x142, 0xA>(v);
    v += dpp_f64<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ double wave_below_f64(double v) { return dpp_f64<0x138, 0xF>(v); }   // (0 in lane 0)
// One dword of a wave mask into lane LANE of pk.  An asm statement because there is no builtin for it in this toolchain -- and
// therefore with its own wait states: on gfx950 a scalar register written by a vector instruction (the v_cmp behind a ballot)
// must not be read by a vector instruction within the next two issue slots, and inside an asm statement nobody pads that
// (measured: the low half of a fresh mask arrived as the previous row's).
#define PLANE_PUT(pk, dword_of_mask, LANE) asm volatile("s_nop 2\n\tv_writelane_b32 %0, %1, %2" : "+v"(pk) : "s"((uint32_t)(dword_of_mask)), "n"(LANE))
// the same with the lane in a scalar register
// (the lane select travels in M0 -- one SGPR per VALU instruction on the constant bus -- which is saved and put back here)
#define PLANE_PUT_AT(pk, dword_of_mask, lane)                                                                                    \
    do {                                                                                                                         \
        uint32_t m0_keep_;                                                                                                       \
        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"             \
                     : "+v"(pk), "=&s"(m0_keep_)                                                                                 \
                     : "s"((uint32_t)(dword_of_mask)), "s"(__builtin_amdgcn_readfirstlane(lane)));                               \
    } while (0)
// the eight dwords of four masks into lanes LANE0 .. LANE0 + 7 of pk, behind one pad
#define PLANE_PUT8(pk, m, LANE0)                                                                                                             \
    asm volatile("s_nop 2\n\tv_writelane_b32 %0, %1, %9\n\tv_writelane_b32 %0, %2, %9+1\n\tv_writelane_b32 %0, %3, %9+2\n\tv_writelane_b32 %0, %4, %9+3" \
                 "\n\tv_writelane_b32 %0, %5, %9+4\n\tv_writelane_b32 %0, %6, %9+5\n\tv_writelane_b32 %0, %7, %9+6\n\tv_writelane_b32 %0, %8, %9+7"       \
                 : "+v"(pk)                                                                                                                  \
                 : "s"((uint32_t)(m)[0]), "s"((uint32_t)((m)[0] >> 32)), "s"((uint32_t)(m)[1]), "s"((uint32_t)((m)[1] >> 32)),               \
                   "s"((uint32_t)(m)[2]), "s"((uint32_t)((m)[2] >> 32)), "s"((uint32_t)(m)[3]), "s"((uint32_t)((m)[3] >> 32)), "n"(LANE0))
__device__ __forceinline__ int last_set(unsigned long long m) { return 63 - __clzll((long long)m); }  // m != 0

// int16 PCM -> float as GNU Radio's wavfile_source does it (gr-blocks wavfile_source_impl.cc: 16-bit samples are DIVIDED by
// 0x7FFF): i16_scale < 0 asks for exactly that -- fl(v / 32767), obtained without a division as fma(v', 2^-15 + 2^-30, v') with
// v' = v * 2^-15 (exact): equal to the IEEE quotient for every one of the 65536 inputs (tests/test_host_abi.py checks the
// identity exhaustively).  i16_scale > 0: the plain product fl(v * i16_scale).
__host__ __device__ __forceinline__ float i16_to_float(int v, float i16_scale) {
    if (i16_scale < 0.f) {
        const float vp = (float)v * 3.0517578125e-05f;
        return fmaf(vp, 3.0517578125e-05f + 9.31322574615478515625e-10f, vp);
    }
    return (float)v * i16_scale;
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
        const float s = i16_to_float((int)((const int16_t *)in)[m], i16_scale);
        return s * s;
    }
}

// The same in two halves, so that a load can stay in flight across loop iterations: the raw sample
// is fetched steps ahead and only turned into the envelope when its step begins.
template <int KIND> struct RawOf { using T = float; };
template <> struct RawOf<IN_IQ_F32> { using T = float2; };
template <> struct RawOf<IN_I16_SQ> { using T = int16_t; };
template <int KIND>
__device__ __forceinline__ typename RawOf<KIND>::T load_raw(const void *in, size_t m) {
    return ((const typename RawOf<KIND>::T *)in)[m];
}
template <int KIND>
__device__ __forceinline__ float env_of(typename RawOf<KIND>::T v, float i16_scale) {
    if constexpr (KIND == IN_IQ_F32) {
        const float a = v.x * v.x, b = v.y * v.y;
        return a + b;
    } else if constexpr (KIND == IN_ENV_F32) {
        return v;
    } else if constexpr (KIND == IN_REAL_F32_SQ) {
        return v * v;
    } else {
        const float s = i16_to_float((int)v, i16_scale);
        return s * s;
    }
}
template <int KIND>
__device__ __forceinline__ typename RawOf<KIND>::T raw_zero() {
    if constexpr (KIND == IN_IQ_F32) return make_float2(0.f, 0.f);
    else if constexpr (KIND == IN_I16_SQ) return (int16_t)0;
    else return 0.f;
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

// LOW bookkeeping at a chunk start.  A LOW key only matters within max_len + 1 samples and a chunk is longer
// than that, so only the predecessor's key can be live; the last non-LOW index needs a deeper look only when
// whole chunks were LOW.
__device__ __forceinline__ int carried_nl(const ThrArgs &A) { return A.low_src ? A.low_src->low_nl : A.nl0; }
__device__ __forceinline__ int carried_kl(const ThrArgs &A) { return A.low_src ? A.low_src->low_kl : A.kl0; }
__device__ __forceinline__ void resolve_low_state(const ThrArgs &A, int c, int &nl, int &kl) {
    const ChunkInfo p = A.info[A.ver[c - 1]][c - 1];
    kl = p.low_key;
    nl = p.last_nonlow;
    for (int cc = c - 2; cc >= 0 && nl == LL_NONE; cc--) nl = A.info[A.ver[cc]][cc].last_nonlow;
    if (nl == LL_NONE) nl = carried_nl(A);
    if (c == 1 && kl == KEY_NONE) kl = carried_kl(A);
}

// exact incoming ring value of slot s for chunk c: latest predecessor that accepted a sample into it
// (*from_carry, when asked for: no chunk before c accepted a sample into the slot -- the value is the incoming ring's)
__device__ __forceinline__ float resolve_slot(const ThrArgs &A, int c, int s, bool *from_carry = nullptr) {
    if (from_carry) *from_carry = false;
    if (c >= 1) {   // common case: the predecessor accepted a sample into the slot; both loads issue at once
        const int vb = A.ver[c - 1];
        const uint32_t w = A.touched[vb][(size_t)(c - 1) * A.twords + (s >> 5)];
        const float v = A.ring_out[vb][(size_t)(c - 1) * A.L + s];
        if ((w >> (s & 31)) & 1u) return v;
    }
    for (int cc = c - 2; cc >= 0; cc--) {
        const int vb = A.ver[cc];
        const uint32_t w = A.touched[vb][(size_t)cc * A.twords + (s >> 5)];
        if ((w >> (s & 31)) & 1u) return A.ring_out[vb][(size_t)cc * A.L + s];
    }
    if (from_carry) *from_carry = true;
    return A.ring_carry[s];
}

#ifdef NFC_GEN_PROF
__device__ unsigned long long g_row_iters[2];   // (a profiling build: rows walked by row_exact, iterations they took)
#endif
// One 64-sample row (lane l = sample m), exact: iterate the accept mask to its fixed point.
// Updates ss0, w_nl, w_kl; returns the classification through low/pos ballots.
__device__ __forceinline__ bool row_exact(const ThrArgs &A, int lane, int m, bool act, float x, float prev, double &ss0,
                                          int &w_nl, int &w_kl, uint32_t &emin, uint32_t &emax, uint32_t &flags,
                                          unsigned long long &lowm, unsigned long long &posm) {
    const int mx = A.mx;
    const double x64 = (double)x;
    const int rb = m - lane;   // the row's first sample
    const unsigned long long lane_lt = (1ull << lane) - 1ull, actm = __ballot(act);
    double ss = ss0, incs = 0;
    bool low = false, acc = false, sum_stands = false;
    int val = 0, key = KEY_NONE;
    unsigned long long low_seen = 0;
    bool st2 = false;
    int row_iters = 0;
    (void)row_iters;
    for (int iter = 0;; iter++) {
        bool lw = false, hg = false;
        if (act) classify_one(A, x64, ss, lw, hg);
        low = lw;
        // "HIGH is ignored" depends on the row's LOW samples (and the carried bookkeeping) only: done again only when the LOW mask
        // moved since the iteration before (rows that hover at the HIGH threshold keep theirs).  On the row's MASKS, not on scans
        // (round 4: two six-step DPP scans of dependent moves per iteration before -- a lone wave pays every one of their
        // latencies): the last sample before a lane that is not LOW / that is LOW is the highest bit below the lane.
        const unsigned long long low_now = __ballot(lw);
        if (iter == 0 || low_now != low_seen) {
            low_seen = low_now;
            const unsigned long long bn = actm & ~low_now & lane_lt;
            const int nl = bn ? rb + last_set(bn) : w_nl;   // (a sample of the row lies behind everything carried into it)
            key = KEY_NONE;
            if (lw) {
                const int p = m - nl;  // 1-based position in the LOW run
                const bool bad = (p > mx) && ((p - 1) % mx == 0);
                key = 2 * m + (bad ? 0 : 1);
            }
            const unsigned long long bl = low_now & lane_lt;
            const int kq = __shfl(key, bl ? last_set(bl) : 0, 64);   // the key of the last LOW sample below this lane
            const int kl = bl ? max(kq, w_kl) : w_kl;
            st2 = (kl & 1) && (m - (kl >> 1)) <= mx + 1;
        }
        int v = 0;
        if (lw) v = -1;
        else if (hg && !st2) v = 1;
        const bool a = act && (v == 0);
        const bool same = (iter > 0) && (a == acc);
        acc = a;
        val = v;
        row_iters = iter;
        if (iter > 0 && __all(same)) { sum_stands = true; break; }   // (incs was scanned over exactly this accept mask)
        if (iter >= MAX_FIX_ITERS) { flags |= 1u; break; }   // (iter is uniform: a scalar branch)
        incs = wave_scan_sum_f64(acc ? (x64 - (double)prev) : 0.0);
        ss = ss0 + wave_below_f64(incs);
    }
    if (acc && x != 0.f) {
        const uint32_t e = max(f32_expfield(x), 1u);
        emin = min(emin, e);
        emax = max(emax, e);
    }
#ifdef NFC_GEN_PROF
    if (lane == 0) {
        atomicAdd(&g_row_iters[0], 1ull);
        atomicAdd(&g_row_iters[1], (unsigned long long)(row_iters + 1));
    }
#endif
    // the row's accepted differences: the last scan's total (lane 63 of its inclusive prefix), the LOW bookkeeping off the masks
    double total;
    if (sum_stands) total = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(incs), 63), __builtin_amdgcn_readlane(__double2loint(incs), 63));
    else total = wave_sum_f64(acc ? (x64 - (double)prev) : 0.0);
    ss0 = rfl(ss0 + total);
    const unsigned long long lowf = __ballot(low), nonlowf = actm & ~lowf;
    w_nl = nonlowf ? rb + last_set(nonlowf) : w_nl;
    if (lowf) w_kl = max(w_kl, __builtin_amdgcn_readlane(key, last_set(lowf)));
    lowm = __ballot(low);
    posm = __ballot(val == 1);
    return acc;   // the caller stores x into the ring slot
}

// The state a chunk starts from: chunk 0 the carried (exact) one; mode 1 the exact one resolved from the predecessors'
// summaries; otherwise speculated from the L samples before the chunk.
// PASS0: the caller only ever runs pass 0 (no resolved states, every summary in buffer 0): the look-back code and its
// dynamically indexed buffer pairs (which the compiler would copy to scratch memory) stay out of that kernel.
template <int KIND, bool PASS0 = false>
__device__ __forceinline__ void chunk_incoming(const ThrArgs &A, uint32_t c, int lane, float *ring, const Carry &cr, uint32_t m_chunk,
                                               double &ss0, int &w_nl, int &w_kl, float &eps) {
    const int L = A.L;
    if (c == 0) {
        #pragma unroll 8
        for (int s = lane; s < L; s += 64) ring[s] = A.ring_carry[s];
        ss0 = cr.ss;
        w_nl = carried_nl(A);
        w_kl = carried_kl(A);
    } else if (!PASS0 && A.mode == 1) {
        // (four slots per round, their loads in flight together: the predecessor's touched word and value decide almost every slot)
        {
            const int vb = A.ver[c - 1];
            const uint32_t *tw = A.touched[vb] + (size_t)(c - 1) * A.twords;
            const float *rv = A.ring_out[vb] + (size_t)(c - 1) * L;
            for (int s0 = 0; s0 < L; s0 += 256) {
                uint32_t w[4];
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int s = s0 + 64 * k + lane;
                    w[k] = (s < L) ? tw[s >> 5] : 0u;
                    v[k] = (s < L) ? rv[s] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int s = s0 + 64 * k + lane;
                    if (s < L) ring[s] = ((w[k] >> (s & 31)) & 1u) ? v[k] : resolve_slot(A, (int)c, s);
                }
            }
        }
        double part = 0;
        #pragma unroll 8
        for (int s = lane; s < L; s += 64) part += (double)ring[s];
        ss0 = wave_sum_f64(part) + cr.delta;
        resolve_low_state(A, (int)c, w_nl, w_kl);
    } else {
        // speculate: the L samples before the chunk, rejected-looking ones replaced by a level estimate
        eps = A.eps;
        const uint32_t w0 = m_chunk - (uint32_t)L;
        const uint32_t slot0 = (A.g0modL + w0) % (uint32_t)L;   // ring slot of sample w0 (uniform)
        float mxv = 0.f;
        for (int i0 = 0; i0 < L; i0 += 512) {   // eight independent loads in flight per lane
            typename RawOf<KIND>::T rw[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + 64 * k + lane;
                rw[k] = (i < L) ? load_raw<KIND>(A.in, (size_t)w0 + i) : raw_zero<KIND>();
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + 64 * k + lane;
                if (i < L) {
                    const float x = env_of<KIND>(rw[k], A.i16_scale);
                    uint32_t slot = slot0 + (uint32_t)i;
                    slot = (slot >= (uint32_t)L) ? slot - (uint32_t)L : slot;
                    ring[slot] = x;
                    mxv = fmaxf(mxv, x);
                }
            }
        }
        mxv = wave_max_f32(mxv);
        const float half = 0.5f * mxv;
        float sa = 0.f, na = 0.f;
        #pragma unroll 8
        for (int s = lane; s < L; s += 64) {
            const float x = ring[s];
            if (x >= half) { sa += x; na += 1.f; }
        }
        sa = wave_sum_f32(sa);
        na = wave_sum_f32(na);
        const float ca = (na > 0.f) ? sa / na : mxv;
        float sb = 0.f, nb = 0.f;
        #pragma unroll 8
        for (int s = lane; s < L; s += 64) {
            const float x = ring[s];
            if (x >= half && x <= ca) { sb += x; nb += 1.f; }
        }
        sb = wave_sum_f32(sb);
        nb = wave_sum_f32(nb);
        const float c0 = (nb > 0.f) ? sb / nb : ca;
        const float tlo = (float)A.lo * c0, thi = (float)A.hi * c0;
        // ring slot s last saw sample m = w0 + ((s - slot0) mod L)
        int ll = LL_NONE, nl = LL_NONE;
        double part = 0;
        #pragma unroll 8
        for (int s = lane; s < L; s += 64) {
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
        ll = wave_max_i32(ll);
        w_nl = wave_max_i32(nl);
        w_kl = (ll == LL_NONE) ? KEY_NONE : 2 * ll + 1;
        ss0 = wave_sum_f64(part) + cr.delta;
    }
}

// Keeps the ring the evaluation starts from (for k_certify), marks every slot untouched, folds the exponent guard;
// returns the raw bits of the largest incoming value.
template <bool SIGN_T>
__device__ __forceinline__ uint32_t chunk_save_in(const ThrArgs &A, uint32_t c, int lane, float *ring, unsigned char *tch, uint32_t &emin,
                                                  uint32_t &emax) {
    const int L = A.L;
    float *rin = A.ring_in + (size_t)c * L;
    if constexpr (!SIGN_T)
        for (int s = lane; s < A.Lpad; s += 64) tch[s] = 0;
    uint32_t vtop0 = 0u;   // raw bits of the largest incoming ring value (envelopes are >= 0: bits order like values)
    #pragma unroll 8
    for (int s = lane; s < L; s += 64) {
        const float v = ring[s];
        rin[s] = v;
        vtop0 = max(vtop0, __float_as_uint(v));
        if constexpr (SIGN_T) ring[s] = __uint_as_float(__float_as_uint(v) | 0x80000000u);
        if (v != 0.f) {
            const uint32_t e = max(f32_expfield(v), 1u);
            emin = min(emin, e);
            emax = max(emax, e);
        }
    }

    return vtop0;
}

// The chunk's summary: ring at its end + touched map, LOW bookkeeping, guard, what the evaluation assumed.
template <bool SIGN_T, bool PASS0 = false>
__device__ __forceinline__ void chunk_publish(const ThrArgs &A, uint32_t c, int lane, const float *ring, const unsigned char *tch, uint32_t emin,
                                              uint32_t emax, uint32_t vmin, uint32_t vmax, float ssf, float eps, uint32_t flags, int chunk_kl,
                                              int chunk_nl, double ss_out, float min_ss, int nl_in, int kl_in, uint32_t all_robust) {
    const int L = A.L;
    // fold the raw-bit extremes of the fast path into the exponent guard
    if (vmax != 0u) {
        emax = max(emax, (vmax >> 31) ? 255u : max((vmax >> 23) & 0xFFu, 1u));
        if (vmin != 0xFFFFFFFFu) emin = min(emin, max((vmin >> 23) & 0xFFu, 1u));
    }

    // ---------------- publish the summary ----------------
    // pass 0 fills buffer ver[c] directly -- which is buffer 0 (prepare_batch zeroes the version bytes)
    const int vb_new = PASS0 ? 0 : ((A.mode == 1) ? (1 - (int)A.ver[c]) : (int)A.ver[c]);
    float *ro = (PASS0 ? A.ring_out[0] : A.ring_out[vb_new]) + (size_t)c * L;
    uint32_t *to = (PASS0 ? A.touched[0] : A.touched[vb_new]) + (size_t)c * A.twords;
    uint32_t untouched = 0;
    for (int sbase = 0; sbase < A.twords * 32; sbase += 64) {
        const int s = sbase + lane;
        const float rv = (s < L) ? ring[s] : 0.f;
        const bool t = (s < L) && (SIGN_T ? !(__float_as_uint(rv) >> 31) : (tch[s] != 0));
        const unsigned long long bal = __ballot(t);
        if (s < L) {
            ro[s] = SIGN_T ? fabsf(rv) : rv;
            if (!t) untouched++;
        }
        const int w = sbase >> 5;
        if (lane == 0) {
            to[w] = (uint32_t)bal;
            if (w + 1 < A.twords) to[w + 1] = (uint32_t)(bal >> 32);
        }
    }
    emin = wave_min_u32(emin);
    emax = wave_max_u32(emax);
    // every window sum of the chunk lies below this: each step folded its own bound (fast: tracked sum + margin, which
    // covers drift, speculation and rounding; exact: sum + total variation of the step)
    const uint32_t vtop = wave_max_u32(__float_as_uint(fmaxf(__uint_as_float(wave_max_u32(vmax)), ssf * (1.0f + eps + RND_SUM)) * 1.0009765625f));
    untouched = (uint32_t)wave_sum_f32((float)untouched);
    flags = wave_max_u32(flags);
    if (lane == 0) {
        ChunkInfo ci;
        ci.ss_out = ss_out;   // informational
        ci.low_key = chunk_kl;
        ci.last_nonlow = chunk_nl;
        ci.emin = emin;
        ci.emax = emax;
        ci.flags = flags;
        ci.n_untouched = untouched;
        (PASS0 ? A.info[0] : A.info[vb_new])[c] = ci;
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

// GRING: the ring lives in global memory (one row of A.gring per chunk) instead of LDS -- for windows whose LDS ring would
// leave a SIMD with one or two waves; its old values are then asked for one step ahead, like the input.
#ifdef NFC_GEN_PROF
__device__ unsigned long long g_gen_prof[8192 * 8];
#define GP_T() clock64()
#else
#define GP_T() 0ull
#endif
template <int KIND, int NR, bool GRING>
__global__ __launch_bounds__(256) void k_threshold(ThrArgs A) {
    constexpr uint32_t STEPN = 64u * NR;   // samples per step: NR rows of 64
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = rfl((int)(threadIdx.x >> 6));
    const int wpb = blockDim.x >> 6;
    const uint32_t slotid = blockIdx.x * wpb + wave;
    uint32_t c;
    if (A.list) {
        if (slotid >= A.nlist) return;
        c = rfl(A.list[slotid]);
    } else {
        if (slotid >= (uint32_t)A.nchunks) return;
        c = slotid;
    }
    // Envelopes are >= 0, so the ring keeps "not yet accepted into during this chunk" in the SIGN bit of a slot
    // (-|v|: untouched); the first accepted sample stores +x.  (A raw-envelope input may be negative: that
    // kind keeps a byte map after the ring.)
    constexpr bool SIGN_T = (KIND != IN_ENV_F32);
    const size_t lds_wave = SIGN_T ? (size_t)A.Lpad * 4 : (size_t)A.Lpad * 5;
    float *ring = GRING ? (float *)((unsigned char *)A.gring + (size_t)c * lds_wave) : (float *)(smem + (size_t)wave * lds_wave);
    unsigned char *tch = (unsigned char *)(ring + A.Lpad);
    auto mark = [&](uint32_t sl) { if constexpr (!SIGN_T) tch[sl] = 1; };
    const int L = A.L;
    const int mx = A.mx;
    uint32_t m_chunk, chunk_len;   // (this kernel runs with A.off == 0)
    chunk_span(A, c, m_chunk, chunk_len);
    const uint32_t n1 = min(A.n, m_chunk + chunk_len);
    const uint32_t m_start = max(m_chunk, A.skip);
    const Carry cr = *A.carry;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    uint32_t emin = 255u, emax = 0u;
    int w_nl, w_kl;  // last non-LOW index / key of the last LOW sample, before the current row
    double ss0;
    float eps = 0.f;  // margin of this evaluation

    // ---------------- incoming state ----------------
    unsigned long long gp0 = GP_T(), gp_fast = 0, gp_exact = 0, gp_exsum = 0, gp_nex = 0, gp_t = 0;
    (void)gp0; (void)gp_fast; (void)gp_exact; (void)gp_exsum; (void)gp_nex; (void)gp_t;
    chunk_incoming<KIND>(A, c, lane, ring, cr, m_chunk, ss0, w_nl, w_kl, eps);
    ss0 = rfl(ss0);
    w_nl = rfl(w_nl);
    w_kl = rfl(w_kl);
    const int nl_in = w_nl, kl_in = w_kl;
    const uint32_t vtop0 = chunk_save_in<SIGN_T>(A, c, lane, ring, tch, emin, emax);
    const unsigned long long gp1 = GP_T();
    (void)gp1;

    // ---------------- the chunk, NR rows of 64 samples per step ----------------
    uint32_t flags = 0;
    uint32_t all_robust = 1;
    float min_ss = 3.0e38f;
    int chunk_nl = LL_NONE, chunk_kl = KEY_NONE;
    uint32_t vmin = 0xFFFFFFFFu, vmax = vtop0;   // raw bits (positive floats order like uints): smallest accepted value; largest value OR window-sum bound
    uint32_t slot_step = (A.g0modL + m_chunk) % (uint32_t)L;
    const float etaD = 1.0f - 9.5367431640625e-07f;  // 1 - 2^-20
    // the raw samples of the next step are in flight while the current step is classified
    using Raw = typename RawOf<KIND>::T;
    Raw r1[NR];
    auto fetch = [&](uint32_t b, Raw (&r)[NR]) {
        if (b + STEPN <= A.n) {
#pragma unroll
            for (int j = 0; j < NR; j++) r[j] = load_raw<KIND>(A.in, (size_t)b + 64u * j + lane);
        } else {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const uint32_t m = b + 64u * j + lane;
                r[j] = (m < A.n) ? load_raw<KIND>(A.in, m) : raw_zero<KIND>();
            }
        }
    };
    fetch(m_chunk, r1);
    // Every step of a chunk is whole except the batch's last one (masked, see `tail`) and the stretch before the
    // first stable sample, which takes the exact row path (chunk lengths are multiples of the step).
    const float loLf = (float)A.lo_L, hiLf = (float)A.hi_L;   // thresholds in f32 carry 2^-18 of slack (>> 4 roundings)
    // The fast path never needs the exact sum: its classifications hold for every sum within the step's margin,
    // so it tracks the sum in f32 (ssf) and pays for the accumulated rounding with RND * ss of extra margin
    // (each step adds < 2^-22 relative; ssf is re-derived from the ring at least every 256 steps).  The exact
    // fp64 sum is re-derived from the ring -- S(ring) + delta -- whenever the exact path needs it.
    const float RND = RND_SUM;
    float ssf = (float)ss0;
    bool ss0_valid = true;
    int steps_since_sync = 0;
    auto exact_sum = [&]() {
        double part = 0;
        #pragma unroll 8
        for (int s2 = lane; s2 < L; s2 += 64) part += (double)(SIGN_T ? fabsf(ring[s2]) : ring[s2]);
        return rfl(wave_sum_f64(part) + cr.delta);
    };
    const float slD = 1.0f - 3.814697265625e-06f, slU = 1.0f + 3.814697265625e-06f;
    float pn[NR];   // GRING: the next step's old ring values, in flight
    bool have_pn = false;
    for (uint32_t base = m_chunk; base < n1; base += STEPN) {
        float x[NR], prev[NR];
        uint32_t slot[NR];
        const bool full = (base >= m_start);                // uniform; false only inside the fill stretch of chunk 0
        const bool tail = (base + STEPN > n1);              // uniform; the batch's last, ragged step
#pragma unroll
        for (int j = 0; j < NR; j++) {
            x[j] = env_of<KIND>(r1[j], A.i16_scale);
        }
        if (base + STEPN < n1) fetch(base + STEPN, r1);   // one step (~3 us of work) ahead covers the HBM latency; two measured the same
        if (slot_step + STEPN <= (uint32_t)L) {   // the step does not wrap the ring: immediate offsets
            const float *rp = ring + slot_step + lane;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                slot[j] = slot_step + 64u * j + lane;
                if (!GRING || !have_pn) prev[j] = SIGN_T ? fabsf(rp[64 * j]) : rp[64 * j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NR; j++) {
                uint32_t s = slot_step + 64u * j + lane;
                s = (s >= (uint32_t)L) ? s - (uint32_t)L : s;
                slot[j] = s;
                if (!GRING || !have_pn) prev[j] = SIGN_T ? fabsf(ring[s]) : ring[s];
            }
        }
        if constexpr (GRING) {
            // the ring is in global memory: this step's old values were asked for a step ago, the next step's are asked for
            // now (slots no step in between writes: the window holds at least two steps)
            if (have_pn) {
#pragma unroll
                for (int j = 0; j < NR; j++) prev[j] = SIGN_T ? fabsf(pn[j]) : pn[j];
            }
            have_pn = base + STEPN < n1;
            if (have_pn) {
                uint32_t ns = slot_step + STEPN;
                ns = (ns >= (uint32_t)L) ? ns - (uint32_t)L : ns;
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    uint32_t s = ns + 64u * j + lane;
                    s = (s >= (uint32_t)L) ? s - (uint32_t)L : s;
                    pn[j] = ring[s];
                }
            }
        }
        unsigned long long unt[NR];   // tail only: lanes past the end whose slot had not been touched
        if (tail) {
            // Lanes past the batch's end ride along as samples that cannot matter: each repeats the value its slot
            // holds (no drift; if "accepted" the slot keeps its value), and the touched flag such a store sets
            // is taken back after the step.
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const bool inact = base + 64u * j + lane >= n1;
                bool untouched;
                if constexpr (SIGN_T) untouched = (__float_as_uint(ring[slot[j]]) >> 31) != 0u;
                else untouched = tch[slot[j]] == 0;
                unt[j] = __ballot(inact && untouched);
                if (inact) x[j] = prev[j];
            }
        }
        if (!ss0_valid && steps_since_sync >= 256) {
            ss0 = exact_sum();
            ssf = (float)ss0;
            ss0_valid = true;
            steps_since_sync = 0;
        }
        min_ss = fminf(min_ss, ssf * (etaD - RND));

        // ---- fast path: classification that holds for every sum the step can see ----
        // Straight-line code on wave masks (a v_cmp IS the ballot); whether the step may commit is decided once.
        unsigned long long lowm[NR], posm[NR];
        bool fast = false;
        if (full && A.fast_ok && ssf > 1e-30f && ssf < 1e30f) {
            // Samples that are LOW for any sum within 30 % of ss0 never enter the ring; the others bound the
            // drift of the sum inside the step: B = sum |x - prev|.  (By induction over the samples: while the
            // drift so far is below 25 % no surely-LOW sample is accepted, so the drift stays below B.)
            const float t_sure = ssf * loLf * 0.70f, t_maylow = ssf * loLf * 1.30f;
            float b = 0.f;
            unsigned long long maylow = 0;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                b += (x[j] < t_sure) ? 0.f : fabsf(x[j] - prev[j]);
                maylow |= __ballot(x[j] < t_maylow);
            }
            b = wave_sum_f32(b) * 1.001f;
            float bt = b + (eps + RND) * ssf;
            bool ok = bt < 0.25f * ssf;
            float thi_up = (ssf + bt) * slU * hiLf;
            // Second look: a sample that is HIGH for every sum within that bound is rejected unless a LOW sample
            // put the state machine into state 2 -- impossible while no sample of the step can be LOW and the
            // carried LOW sample is out of reach.  Leaving those samples out tightens the bound.
            const bool key_live = (w_kl & 1) && ((int)base - (w_kl >> 1)) <= mx + 1;
            if (!key_live && maylow == 0) {
                unsigned long long h1 = 0;
#pragma unroll
                for (int j = 0; j < NR; j++) h1 |= __ballot(x[j] > thi_up);
                if (h1) {
                    float b2 = 0.f;
#pragma unroll
                    for (int j = 0; j < NR; j++) b2 += (x[j] > thi_up) ? 0.f : fabsf(x[j] - prev[j]);
                    b2 = wave_sum_f32(b2) * 1.001f;
                    bt = fminf(bt, b2 + (eps + RND) * ssf);
                    thi_up = (ssf + bt) * slU * hiLf;
                }
            }
            const float dn = (ssf - bt) * slD, up = (ssf + bt) * slU;
            // every sum inside the step lies below ssf + bt (if the step commits on this path); vmax carries the bound of
            // the window sums (which dominates every single value: they are non-negative)
            vmax = max(vmax, __float_as_uint(up));
            const float tlo_dn = dn * loLf, tlo_up = up * loLf, thi_dn = dn * hiLf;
            unsigned long long him[NR], amb = 0, anyhi = 0, anylow = 0;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const unsigned long long lo1 = __ballot(x[j] < tlo_dn), lo0 = __ballot(x[j] > tlo_up);
                const unsigned long long hi1 = __ballot(x[j] > thi_up), hi0 = __ballot(x[j] < thi_dn);
                amb |= ~((lo1 | lo0) & (hi1 | hi0));
                lowm[j] = lo1;
                him[j] = hi1;
                anylow |= lo1;
                anyhi |= hi1;
            }
            ok = __all(ok && (tlo_dn > 1e-30f) && (thi_up < 1e30f)) && (amb == 0);
            if (ok && anylow) {
                // every LOW sample must sit at run position <= max_len (then none ends on a time-out):
                // a longer run covers an aligned block of LOW samples, or continues the carried run
                const int lead = (lowm[0] == ~0ull) ? 64 : (__ffsll((long long)~lowm[0]) - 1);
                const int carry_run = (int)base - 1 - w_nl;
                // an all-LOW aligned block has its first, middle and last sample LOW: three probes filter first
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
                if (hit || ((carry_run > 0) && (carry_run + lead > mx))) ok = false;
            }
            if (ok) {
                fast = true;
                float dl = 0.f;
                if (anyhi == 0) {
                    // nothing HIGH: a sample is accepted unless it is LOW; accepted values lie inside the bands,
                    // so their exponent range comes from the thresholds
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        const bool a = !(x[j] < tlo_dn);
                        dl += a ? (x[j] - prev[j]) : 0.f;
                        if (a) {
                            ring[slot[j]] = x[j];
                            mark(slot[j]);
                        }
                        posm[j] = 0ull;
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
                        dl += a ? (x[j] - prev[j]) : 0.f;
                        if (a) {
                            ring[slot[j]] = x[j];
                            mark(slot[j]);
                        }
                        const uint32_t xb = __float_as_uint(x[j]);
                        vmin = min(vmin, (a && xb != 0u) ? xb : 0xFFFFFFFFu);
                        vmax = max(vmax, a ? xb : 0u);
                        posm[j] = __ballot(ps);
                        before = lowm[j] ? rb + last_set(lowm[j]) : before;
                    }
                }
                int step_nl = (int)(base + STEPN) - 1, step_ll = LL_NONE;   // nothing LOW: the last sample is the last non-LOW
                if (anylow) {
                    step_nl = LL_NONE;
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        const int rb = (int)(base + 64u * j);
                        const unsigned long long nonlow = ~lowm[j];
                        step_ll = lowm[j] ? rb + last_set(lowm[j]) : step_ll;
                        step_nl = nonlow ? rb + last_set(nonlow) : step_nl;
                    }
                }
                if (tail) {   // lanes past the end are not samples
                    step_nl = LL_NONE;
                    step_ll = LL_NONE;
#pragma unroll
                    for (int j = 0; j < NR; j++) {
                        const int rb = (int)(base + 64u * j);
                        const unsigned long long am = __ballot(base + 64u * j + lane < n1);
                        lowm[j] &= am;
                        posm[j] &= am;
                        const unsigned long long nonlow = ~lowm[j] & am;
                        step_ll = lowm[j] ? rb + last_set(lowm[j]) : step_ll;
                        step_nl = nonlow ? rb + last_set(nonlow) : step_nl;
                        if ((unt[j] >> lane) & 1ull) {
                            if constexpr (SIGN_T) ring[slot[j]] = __uint_as_float(__float_as_uint(x[j]) | 0x80000000u);
                            else tch[slot[j]] = 0;
                        }
                    }
                }
                ssf += wave_sum_f32(dl);
                ss0_valid = false;
                steps_since_sync++;
                if (step_ll != LL_NONE) {
                    w_kl = 2 * step_ll + 1;
                    chunk_kl = w_kl;
                }
                if (step_nl != LL_NONE) {
                    w_nl = step_nl;
                    chunk_nl = step_nl;
                }
            }
        }
        if (__builtin_expect(!fast, 0)) {   // (the hint keeps the cold path out of the loop's layout: measured 4 % on the kernel)
            // ---- exact path, one 64-sample row at a time ----
            gp_t = GP_T();
            gp_nex++;
            if (eps > 0.f) all_robust = 0;
            if (!ss0_valid) {
                ss0 = exact_sum();
                ss0_valid = true;
                steps_since_sync = 0;
            }
            gp_exsum += GP_T() - gp_t;
            float xs[NR], pv[NR];
            float bx = 0.f;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                xs[j] = x[j];
                pv[j] = prev[j];
                bx += fabsf(x[j] - prev[j]);
            }
            vmax = max(vmax, __float_as_uint((float)ss0 * slU + wave_sum_f32(bx) * 1.001f));   // no sum inside the step can exceed this
#pragma unroll 1
            for (int j = 0; j < NR; j++) {
                const int m = (int)(base + 64u * j) + lane;
                const bool aj = ((uint32_t)m >= m_start) && ((uint32_t)m < n1);
                const int nl_b = w_nl, kl_b = w_kl;
                unsigned long long lm, pm;
                float xj = xs[0], pj = pv[0];
                uint32_t sj = slot[0];
#pragma unroll
                for (int k = 1; k < NR; k++) {
                    xj = (j == k) ? xs[k] : xj;
                    pj = (j == k) ? pv[k] : pj;
                    sj = (j == k) ? slot[k] : sj;
                }
                if (row_exact(A, lane, m, aj, xj, pj, ss0, w_nl, w_kl, emin, emax, flags, lm, pm)) {
                    ring[sj] = xj;
                    mark(sj);
                    vmax = max(vmax, __float_as_uint(xj));
                }
#pragma unroll
                for (int k = 0; k < NR; k++) {
                    lowm[k] = (j == k) ? lm : lowm[k];
                    posm[k] = (j == k) ? pm : posm[k];
                }
                if (w_nl != nl_b) chunk_nl = w_nl;
                if (w_kl != kl_b) chunk_kl = w_kl;
            }
            ssf = (float)ss0;
            gp_exact += GP_T() - gp_t;
        }
        {   // the step's NR words per plane: the masks are scalar pairs, v_writelane puts their halves into lanes 0 .. 2 NR - 1
            // (neg) and 2 NR .. 4 NR - 1 (pos) of ONE register, and those lanes store a dword each (the builtin, not an asm
            // statement: a mask fresh from a v_cmp needs wait states before v_writelane may read it, and only the compiler pads them)
            int pk = 0;
#pragma unroll
            for (int k = 0; k < NR; k++) {
                PLANE_PUT(pk, lowm[k], 2 * k);
                PLANE_PUT(pk, (lowm[k] >> 32), 2 * k + 1);
                PLANE_PUT(pk, posm[k], 2 * NR + 2 * k);
                PLANE_PUT(pk, (posm[k] >> 32), 2 * NR + 2 * k + 1);
            }
            const int h = lane & (2 * NR - 1);                      // dword within the plane's NR words
            const uint32_t w = (base >> 6) + (uint32_t)(h >> 1);    // its word
            uint32_t *dst = (uint32_t *)(lane < 2 * NR ? A.neg : A.pos) + 2 * (size_t)(base >> 6) + h;
            if (lane < 4 * NR && (size_t)w * 64 < A.n) *dst = (uint32_t)pk;
        }
        slot_step += STEPN;
        slot_step = (slot_step >= (uint32_t)L) ? slot_step - (uint32_t)L : slot_step;
    }
    const unsigned long long gp2 = GP_T();
    (void)gp2;
    chunk_publish<SIGN_T>(A, c, lane, ring, tch, emin, emax, vmin, vmax, ssf, eps, flags, chunk_kl, chunk_nl,
                          ss0_valid ? ss0 : (double)ssf, min_ss, nl_in, kl_in, all_robust);
#ifdef NFC_GEN_PROF
    if (lane == 0 && c < 8192u) {
        unsigned long long *o = g_gen_prof + (size_t)c * 8;
        o[0] = gp1 - gp0;          // incoming state
        o[1] = gp2 - gp1;          // the loop
        o[2] = gp_exact;           // ... of which in exact steps
        o[3] = gp_exsum;           // ... of which re-deriving the sum
        o[4] = gp_nex;             // exact steps
        o[5] = GP_T() - gp2;       // summary
        o[6] = (n1 - m_chunk) / STEPN;
        o[7] = (unsigned long long)A.mode + 1;
    }
#endif
}

// ---------------------------------------------------------------------------
// Certification: is what a chunk's latest evaluation assumed about its incoming state
// close enough to the truth (as resolved from the current summaries) that its result stands?
//   evaluated from a speculated state (eps > 0): every step was on the fast path and
//       sum |true ring - assumed ring| <= eps * (smallest step-start sum)
//   evaluated from a resolved state (eps == 0): the resolved state is still bitwise the same
// plus equivalent LOW bookkeeping at the chunk start.
// ---------------------------------------------------------------------------
struct CertInfo {
    float d, allowed;
    uint32_t all_robust, low_ok;
};
// What the host needs to know after the first certification, small enough to travel in the mirrored state block:
// how many chunks failed, and the batch-wide exponent guard of the fp64 sums.
struct CertSummary {
    uint32_t n_fail;            // chunks whose certification failed (atomic; zeroed by k_fill)
    uint32_t eminmax;           // exponent fields over every chunk's accepted values: emin | emax << 16
    uint32_t worst;             // f32 bits (atomic max): the largest L1 distance between a chunk's speculated and true incoming window, over the
                                // chunk's smallest window sum -- how much of the margin eps the speculation used (the host sizes eps by it)
    uint32_t flagged;           // some chunk met a value it cannot vouch for
    uint32_t vtop;              // raw bits of an upper bound of every ring value of the batch
    uint32_t n_carried;         // window slots that no chunk of the batch accepted a sample into (their value is the incoming ring's)
};
constexpr int FIN_BLOCK = 256;
// The ring at the end of the batch (look-back over all chunks) and its sum become the carried state; with
// `sum` the guard summary of the batch is folded too.  One workgroup of FIN_BLOCK threads.
__device__ __forceinline__ void finalize_state(const ThrArgs &A, float *ring_next, Carry *carry, CertSummary *sum) {
    __shared__ double s_part[FIN_BLOCK / 64];
    __shared__ uint32_t s_mn[FIN_BLOCK / 64], s_mx[FIN_BLOCK / 64], s_fl[FIN_BLOCK / 64], s_vt[FIN_BLOCK / 64], s_cr[FIN_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double part = 0;
    uint32_t carried = 0;
    // a thread's slots in rounds of eight: the common case (the last chunk accepted a sample into the slot) is two
    // loads per slot, all sixteen in flight at once; only slots it left untouched walk further back
    const int last = A.nchunks - 1;
    const int vb = A.ver_zero ? 0 : A.ver[last];
    const uint32_t *tw = A.touched[vb] + (size_t)last * A.twords;
    const float *ro = A.ring_out[vb] + (size_t)last * A.L;
    for (int s0 = tid; s0 < A.L; s0 += 8 * FIN_BLOCK) {
        uint32_t w[8];
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int s = s0 + k * FIN_BLOCK;
            const bool in = s < A.L;
            w[k] = in ? tw[s >> 5] : 0u;
            v[k] = in ? ro[s] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int s = s0 + k * FIN_BLOCK;
            if (s < A.L) {
                if (!((w[k] >> (s & 31)) & 1u)) {   // look further back
                    bool from_carry;   // (one look-back answers both questions)
                    v[k] = resolve_slot(A, last, s, &from_carry);
                    if (from_carry) carried++;
                }
                ring_next[s] = v[k];
                part += (double)v[k];
            }
        }
    }
    uint32_t mn = 255u, mx = 0u, fl = 0u, vt = 0u;
    carried = (uint32_t)wave_sum_f32((float)carried);
    if (sum)
        for (int k = tid; k < A.nchunks; k += FIN_BLOCK) {
            mn = min(mn, (uint32_t)A.gmin[k]);
            mx = max(mx, (uint32_t)A.gmax[k]);
            fl |= (uint32_t)A.gflags[k] & 1u;
            vt = max(vt, A.gvtop[k]);
        }
    part = wave_sum_f64(part);
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    fl = wave_max_u32(fl);
    vt = wave_max_u32(vt);
    if (lane == 0) {
        s_vt[wave] = vt;
        s_part[wave] = part;
        s_mn[wave] = mn;
        s_mx[wave] = mx;
        s_fl[wave] = fl;
        s_cr[wave] = carried;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t cr_all = 0;
        for (int w = 0; w < FIN_BLOCK / 64; w++) cr_all += s_cr[w];
        double S = 0;
        for (int w = 0; w < FIN_BLOCK / 64; w++) {
            S += s_part[w];
            mn = min(mn, s_mn[w]);
            mx = max(mx, s_mx[w]);
            fl |= s_fl[w];
            vt = max(vt, s_vt[w]);
        }
        carry->ss_fin = S + carry->delta;
        carry->fin_valid = 1;
        // the LOW bookkeeping a chunk after the last one would start from, rebased to the next batch's sample 0.  (Only the
        // run's phase modulo max_len and "longer than max_len + 1" matter of a last-non-LOW index far back: it is brought
        // within 2^29 in whole multiples of max_len, so that batch after batch of LOW samples cannot overflow it.)
        int nl, kl;
        resolve_low_state(A, A.nchunks, nl, kl);
        long long rel = (long long)nl - (long long)A.n;
        const long long floor_ = -(1ll << 29);
        if (rel < floor_) rel += ((floor_ - rel + A.mx - 1) / A.mx) * (long long)A.mx;
        carry->low_nl = (int32_t)rel;
        carry->low_kl = (kl == KEY_NONE || (long long)(kl >> 1) - (long long)A.n < floor_) ? KEY_NONE : kl - 2 * (int32_t)A.n;
        if (sum) {
            sum->eminmax = mn | (mx << 16);
            sum->flagged = fl;
            sum->vtop = vt;
        }
        if (A.sum) A.sum->n_carried = cr_all;   // (whichever launch resolves the end-of-batch window says it)
    }
}

// One extra workgroup (the last) resolves the end-of-batch ring meanwhile: if every chunk certifies, that is
// the carried state of the next batch (otherwise k_finalize_state runs again after the re-runs).
// (bid of nblocks: the certification may share a launch with another stage's workgroups, see k_certify_and_count)
__device__ __forceinline__ void certify_block(const ThrArgs &A, uint8_t *cert, CertInfo *dbg, float *ring_next, Carry *carry,
                                              CertSummary *sum, uint32_t bid, uint32_t nblocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (ring_next && bid == nblocks - 1) {
        finalize_state(A, ring_next, carry, sum);
        return;
    }
    // a chunk per WORKGROUP (round 4; a chunk per wave before): the window's slots are one round of loads for 256 threads at the
    // bench's window instead of four dependent rounds for 64 -- the launch was as long as its longest chain of loads
    const uint32_t slotid = bid;
    if (slotid >= A.nlist) return;
    const uint32_t c = A.list ? A.list[slotid] : slotid + 1;
    if (c == 0) {   // (chunk 0 re-run from the carried state: nothing it could disagree with -- unless the re-run itself gave up)
        if (threadIdx.x == 0) {
            const bool ok0 = !(A.gflags[0] & 4u);
            cert[0] = ok0 ? 1 : 0;
            if (!ok0 && sum) atomicAdd(&sum->n_fail, 1u);
        }
        return;
    }
    __shared__ float s_d[4];
    const int L = A.L;
    const int tid = (int)threadIdx.x;
    const float *rin = A.ring_in + (size_t)c * L;
    const RunMeta mt = A.meta[c];
    if (mt.eps > 0.f && !mt.all_robust && !dbg) {
        // a speculated evaluation that gave up, or left the fast path: it cannot stand whatever its window turns out to be -- no
        // look-back for it (a stream hovering at a threshold: every chunk of pass 0, 106 -> 10 us of this launch)
        if (tid == 0) {
            cert[c] = 0;
            if (sum) atomicAdd(&sum->n_fail, 1u);
        }
        return;
    }
    const int vb = A.ver_zero ? 0 : A.ver[c - 1];   // (a load the two below would have to wait for)
    const uint32_t *tw = A.touched[vb] + (size_t)(c - 1) * A.twords;
    const float *ro = A.ring_out[vb] + (size_t)(c - 1) * L;
    float d = 0.f;
    bool differ = false;
    constexpr int CK = 8;    // slots per thread and round: every load of a round is in flight at once
    for (int s0 = 0; s0 < L; s0 += 256 * CK) {
        float t[CK], u[CK];
        uint32_t w[CK];
#pragma unroll
        for (int k = 0; k < CK; k++) {
            const int s = s0 + 256 * k + tid;
            const bool in = s < L;
            t[k] = in ? ro[s] : 0.f;
            u[k] = in ? rin[s] : 0.f;
            w[k] = in ? tw[s >> 5] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < CK; k++) {
            const int s = s0 + 256 * k + tid;
            if (s < L) {
                if (!((w[k] >> (s & 31)) & 1u)) t[k] = resolve_slot(A, (int)c, s);   // rare: look further back
                d += fabsf(t[k] - u[k]);
                if (__float_as_uint(t[k]) != __float_as_uint(u[k])) differ = true;
            }
        }
    }
    d = wave_sum_f32(d);
    if (lane == 0) s_d[wave] = d;
    const int any_differ = __syncthreads_or(differ ? 1 : 0);   // (its barrier also publishes the waves' sums)
    d = ((s_d[0] + s_d[1]) + (s_d[2] + s_d[3])) * 1.001f;
    if (tid != 0) return;
    int nl, kl;
    resolve_low_state(A, (int)c, nl, kl);
    uint32_t m0u, len0;
    chunk_span(A, c, m0u, len0);
    const int m0 = (int)m0u - A.off;
    const int mx = A.mx;
    auto live = [&](int k) { return (k & 1) && (m0 - (k >> 1)) <= mx + 1; };
    const bool low_ok = (nl == mt.nl_in) && ((kl == mt.kl_in) || (!live(kl) && !live(mt.kl_in)));
    bool ok;
    if (mt.eps > 0.f) ok = mt.all_robust && (d <= mt.eps * mt.min_ss * 0.999f) && low_ok;
    else ok = !any_differ && (nl == mt.nl_in) && (kl == mt.kl_in) && !(A.gflags[c] & 4u);   // (flag 4: a re-run by k_threshold_wg that gave up)
    // (how much of the margin the speculation used: a diagnostic -- non-negative floats order like their bits)
    // (looked at first: four thousand chunks of a finely cut batch would otherwise queue at one address)
    if (sum && mt.eps > 0.f && mt.all_robust && mt.min_ss > 0.f) {
        const uint32_t wv = __float_as_uint(d / mt.min_ss);
        if (wv > __hip_atomic_load(&sum->worst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&sum->worst, wv);
    }
    cert[c] = ok ? 1 : 0;
    if (!ok && sum) atomicAdd(&sum->n_fail, 1u);
    if (dbg) dbg[c] = CertInfo{d, mt.eps * mt.min_ss, mt.all_robust, (uint32_t)low_ok};
}
// The first certification of a batch depends on the threshold kernel only and nothing of the later stages depends on it, so it shares
// a launch with the edge stage's writer: `blocks` extra workgroups certify (the last of them resolves the end-of-batch state).
struct CertLaunch {
    ThrArgs A;
    uint8_t *cert;
    float *ring_next;
    Carry *carry;
    CertSummary *sum;
    uint32_t blocks;
};

// the certification's grid: a workgroup per pending chunk, and one more that resolves the end-of-batch state
inline uint32_t cert_grid(uint32_t pending) { return pending + 1u; }
__global__ __launch_bounds__(256) void k_certify(ThrArgs A, uint8_t *cert, CertInfo *dbg, float *ring_next, Carry *carry,
                                                 CertSummary *sum) {
    // (the workgroup that resolves the end-of-batch state -- one chain of look-backs, as long as the rest of the launch is wide -- goes first)
    const uint32_t nb = gridDim.x, bid = ring_next ? (blockIdx.x == 0 ? nb - 1u : blockIdx.x - 1u) : blockIdx.x;
    certify_block(A, cert, dbg, ring_next, carry, sum, bid, nb);
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

// Block-wide reductions for the one-workgroup kernels below (FILL_BLOCK threads).
constexpr int FILL_BLOCK = 512;
struct FillRed {
    double part[FILL_BLOCK / 64];
    uint32_t mn[FILL_BLOCK / 64], mx[FILL_BLOCK / 64];
};
__device__ __forceinline__ void fill_reduce(FillRed &r, double &part, uint32_t &emin, uint32_t &emax) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    part = wave_sum_f64(part);
    emin = wave_min_u32(emin);
    emax = wave_max_u32(emax);
    __syncthreads();   // r may still be read from an earlier use
    if (lane == 0) {
        r.part[wave] = part;
        r.mn[wave] = emin;
        r.mx[wave] = emax;
    }
    __syncthreads();
    part = 0;
    for (int w = 0; w < FILL_BLOCK / 64; w++) {
        part += r.part[w];
        emin = min(emin, r.mn[w]);
        emax = max(emax, r.mx[w]);
    }
}

// Per batch: S(ring), delta = ss - S(ring), and the guard span of the carried values; also resets the
// summary version bytes and the certification summary.  Runs as the tail of k_fill (one launch per batch for both).
__device__ __forceinline__ void prepare_batch_finish(Carry *carry, CertSummary *sum, double S, uint32_t emin, uint32_t emax);
__device__ __forceinline__ void prepare_batch(const float *ring, int L, Carry *carry, uint8_t *ver, int nchunks, CertSummary *sum,
                                              FillRed &red) {
    const int tid = threadIdx.x;
    for (int i = tid; i < nchunks; i += FILL_BLOCK) ver[i] = 0;
    double part = 0;
    uint32_t emin = 255u, emax = 0u;
    for (int s = tid; s < L; s += FILL_BLOCK) {
        const float v = ring[s];
        part += (double)v;
        if (v != 0.f) {
            const uint32_t e = max(f32_expfield(v), 1u);
            emin = min(emin, e);
            emax = max(emax, e);
        }
    }
    fill_reduce(red, part, emin, emax);
    prepare_batch_finish(carry, sum, part, emin, emax);
}
// (the tail of the above, for a caller that has just summed the same window itself)
// (on a Carry wherever it lives: the state block, or a copy in registers that is stored once)
__device__ __forceinline__ void batch_guard(Carry &cr, double S, uint32_t emin, uint32_t emax) {
    carry_apply_fin(cr);
    const double ss = cr.ss;
    const double delta = ss - S;
    cr.delta = delta;
    int el, eh, el2, eh2;
    f64_bit_span(ss, el, eh);
    f64_bit_span(delta, el2, eh2);
    cr.ss_emin = min(el, el2);
    cr.ss_emax = max(eh, eh2);
    cr.ring_emin = (int)emin;
    cr.ring_emax = (int)emax;
}
__device__ __forceinline__ void prepare_batch_finish(Carry *carry, CertSummary *sum, double S, uint32_t emin, uint32_t emax) {
    if (threadIdx.x == 0) {
        batch_guard(*carry, S, emin, emax);
        if (sum) *sum = CertSummary{0u, 255u, 0u, 0u, 0u, 0u};
    }
}

struct EdgeCarryInit {
    int32_t *dst;   // -> EdgeCarry {state, last_bit, dur}
    int32_t dur0;   // av_window % max_len
};

// ---------------------------------------------------------------------------
// Fill phase (transition_sink.py:109-125): copy the first L samples into the ring,
// then sum them in the reference's order.  One workgroup; the ordered sum, when the
// exponent spread cannot prove every order equal, is one lane reading LDS.
// ---------------------------------------------------------------------------
// A reset / primed / restored stream state that has not reached the device yet rides along with the batch's first
// launch: the head of the state block (carried values of all stages) by value, and the window's fill value.
struct StateInit {
    int32_t apply, fill_ring, ring_len, n_words;
    float fill;
    uint32_t *dst;        // the device state block
    uint32_t words[32];
};

template <int KIND>
__global__ __launch_bounds__(FILL_BLOCK) void k_fill(const void *in, uint32_t n, float i16_scale, int L, float *ring, Carry *carry,
                                                     EdgeCarryInit eci, uint8_t *ver, int nchunks, CertSummary *sum, StateInit init,
                                                     uint32_t *seq_dst, uint32_t seq, int fresh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ FillRed red;
    float *lr = (float *)smem;
    const int tid = threadIdx.x;
    if (tid == 0 && seq_dst) *seq_dst = seq;   // this batch's stamp in the state block (the host checks it in its mirror)
    if (fresh) {
        // A reset stream whose first batch holds a whole window (the host knows both: the state rides in `init`, and its carry says
        // "nothing filled"): nothing of the state block is read back.  This launch stands in front of the batch's every other one
        // and was a chain of dependent round trips -- the state stored, read again for `stable` / `filled`, the samples, the carry
        // stored field by field and read again for the batch's preparation: 7.1 us of a 250 us batch.  Here the samples are asked
        // for at once, the carry is finished in registers and stored once.
        static_assert(sizeof(Carry) % 4 == 0 && sizeof(Carry) <= sizeof(init.words), "the carry heads the state block");
        constexpr int CW = (int)(sizeof(Carry) / 4);
        if (tid >= CW && tid < init.n_words) init.dst[tid] = init.words[tid];   // (the carry's words: thread 0, below)
        double part = 0;
        uint32_t emin = 255u, emax = 0u;
        for (int i0 = tid; i0 < L; i0 += 4 * FILL_BLOCK) {   // (four samples per thread in flight: the stores to the ring may alias the input for all the compiler knows)
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = (i0 + k * FILL_BLOCK < L) ? envelope_at<KIND>(in, (size_t)(i0 + k * FILL_BLOCK), i16_scale) : 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + k * FILL_BLOCK;
                if (i >= L) break;
                ring[i] = v[k];
                lr[i] = v[k];
                part += (double)v[k];
                if (v[k] != 0.f) {
                    const uint32_t e = max(f32_expfield(v[k]), 1u);
                    emin = min(emin, e);
                    emax = max(emax, e);
                }
            }
        }
        if (init.fill_ring)   // (the padding behind the window, as the general path leaves it)
            for (int i = L + tid; i < init.ring_len; i += FILL_BLOCK) ring[i] = 0.f;
        for (int i = tid; i < nchunks; i += FILL_BLOCK) ver[i] = 0;
        fill_reduce(red, part, emin, emax);   // (its barriers also publish lr)
        if (tid == 0) {
            const double S = part;
            int lg = 0;
            while ((1 << lg) < L) lg++;
            const bool exact = (emax < 255u) && ((int)emax + 2 + lg - ((int)emin - 23) <= 52);
            double s = S, err = 0;
            if (!exact) {
                s = 0;
                for (int i = 0; i < L; i++) {
                    const double v = (double)lr[i];
                    const double t = s + v;            // transition_sink.py:122, sequential
                    const double bv = t - s;           // TwoSum residue: was the addition exact?
                    err += fabs((s - (t - bv)) + (v - bv));
                    s = t;
                }
            }
            Carry nc;
            uint32_t cw[CW];
#pragma unroll
            for (int k = 0; k < CW; k++) cw[k] = init.words[k];
            memcpy(&nc, cw, sizeof nc);
            nc.filled = L;
            nc.ss = s;
            nc.stable = 1;
            if (err != 0.0) nc.inexact = 1;
            batch_guard(nc, S, emin, emax);
            *carry = nc;
            // work has just been rebound to work_stable: _dur = length % max (transition_sink.py:123)
            eci.dst[0] = 0;
            eci.dst[1] = 0;
            eci.dst[2] = eci.dur0;
            if (sum) *sum = CertSummary{0u, 255u, 0u, 0u, 0u, 0u};
        }
        return;
    }
    if (init.apply) {
        if (tid < init.n_words) init.dst[tid] = init.words[tid];
        if (init.fill_ring)
            for (int i = tid; i < init.ring_len; i += FILL_BLOCK) ring[i] = i < L ? init.fill : 0.f;
        __syncthreads();
    }
    if (carry->stable) {   // nothing to fill: only the per-batch preparation
        prepare_batch(ring, L, carry, ver, nchunks, sum, red);
        return;
    }
    const int filled = carry->filled;
    const int can = min((int)n, L - filled);
    // If the exponent spread of the window proves every partial sum exact, any order gives the reference's
    // sum; otherwise add in the reference's order on one lane and note whether a rounding happened.
    double part = 0;
    uint32_t emin = 255u, emax = 0u;
    auto take = [&](int i, float v) {
        lr[i] = v;
        part += (double)v;
        if (v != 0.f) {
            const uint32_t e = max(f32_expfield(v), 1u);
            emin = min(emin, e);
            emax = max(emax, e);
        }
    };
    if (filled == 0 && can == L) {
        // a whole window from this batch (a fresh stream's first batch): the samples go to the ring and into the sums in one pass
        // -- no second trip to the ring behind a barrier (this launch is a chain of dependent round trips: 7.9 us of a batch)
        for (int i = tid; i < L; i += FILL_BLOCK) {
            const float v = envelope_at<KIND>(in, (size_t)i, i16_scale);
            ring[i] = v;
            take(i, v);
        }
    } else {
        for (int i = tid; i < can; i += FILL_BLOCK) ring[filled + i] = envelope_at<KIND>(in, (size_t)i, i16_scale);
        __syncthreads();
        if (filled + can != L) {
            if (tid == 0) carry->filled = filled + can;
            return;
        }
        for (int i = tid; i < L; i += FILL_BLOCK) take(i, ring[i]);
    }
    fill_reduce(red, part, emin, emax);   // (its barriers also publish lr)
    const double S = part;
    int lg = 0;
    while ((1 << lg) < L) lg++;
    const bool exact = (emax < 255u) && ((int)emax + 2 + lg - ((int)emin - 23) <= 52);
    if (tid == 0) {
        double s = S;
        double err = 0;
        if (!exact) {
            s = 0;
            for (int i = 0; i < L; i++) {
                const double v = (double)lr[i];
                const double t = s + v;            // transition_sink.py:122, sequential
                const double bv = t - s;           // TwoSum residue: was the addition exact?
                err += fabs((s - (t - bv)) + (v - bv));
                s = t;
            }
        }
        carry->filled = L;
        carry->ss = s;
        carry->stable = 1;
        if (err != 0.0) carry->inexact = 1;
        // work has just been rebound to work_stable: _dur = length % max (transition_sink.py:123)
        eci.dst[0] = 0;
        eci.dst[1] = 0;
        eci.dst[2] = eci.dur0;
    }
    // the per-batch preparation: the window was summed just now (S, emin, emax are every thread's), only the resets are left
    for (int i = tid; i < nchunks; i += FILL_BLOCK) ver[i] = 0;
    prepare_batch_finish(carry, sum, S, emin, emax);
}

// After the passes converged: the ring at the end of the batch (look-back over all
// chunks) and its sum become the carried state.
__global__ __launch_bounds__(FIN_BLOCK) void k_finalize_state(ThrArgs A, float *ring_next, Carry *carry) {
    finalize_state(A, ring_next, carry, nullptr);
}

// ---------------------------------------------------------------------------
// Literal sequential restatement of transition_sink.py:55-99 (classification part)
// on ONE lane.  Used when the fp64 window sums cannot be proven exact (so the
// summation order matters), for windows shorter than one step, and under
// NFC_FLAG_FORCE_SEQUENTIAL.  Slow by construction; it carries the reference's own
// state variables.
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
    uint64_t *neg, *pos;
    int32_t *out;                   // (state, last_bit, dur) after the last sample
};
template <int KIND>
__global__ __launch_bounds__(64) void k_threshold_seq(SeqArgs A) {
    if (threadIdx.x != 0) return;
    double ss = A.carry->ss;
    double err = 0;
    int state = A.state, last_bit = A.last_bit, dur = A.dur;
    uint32_t slot = (A.g0modL + A.skip) % (uint32_t)A.L;
    unsigned long long wn = 0, wp = 0;
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
        if (v == -1) wn |= 1ull << (m & 63);
        if (v == 1) wp |= 1ull << (m & 63);
        if ((m & 63) == 63 || m + 1 == A.n) {
            A.neg[m >> 6] = wn;
            A.pos[m >> 6] = wp;
            wn = wp = 0;
        }
    }
    if (A.out) {
        A.out[0] = state;
        A.out[1] = last_bit;
        A.out[2] = dur;
    }
    A.carry->ss = ss;
    A.carry->fin_valid = 0;   // (a parallel attempt at this batch may have left its end-of-batch sum: void)
    if (err != 0.0) A.carry->inexact = 1;
}

}  // namespace nfc
