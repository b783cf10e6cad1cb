// threshold_rr.hip.h -- the threshold kernel with the averaging ring held in REGISTERS.
//
// Same algorithm, same summaries and same certification as k_threshold (threshold.hip.h), for windows of
// 256 <= L <= 2048 samples (the reference's 2000 included).  Time chunks are aligned to the ring: chunk c
// starts where the ring slot is 0, so slot 64*r + lane lives in register R[r] of that lane for the whole
// chunk, and one ring period is a fully unrolled run over rows r = 0 .. 31 (static register indices).
// No LDS at all: the previous value of a sample's slot is a register read, accepting it is one v_cndmask,
// the touched map is one 32-bit mask per lane, and the classification bit planes leave through a scalar
// bit accumulator because a row no longer starts on a 64-sample boundary of the batch.
#pragma once
#include <type_traits>

#include "threshold.hip.h"

namespace nfc {

constexpr int RR_ROWS = 32;

// Appends row masks to the classification bit planes at arbitrary sample offsets.  Interior words are
// stored whole; the first and last word of a chunk are shared with the neighbouring chunks, so there the
// chunk clears exactly its own bit positions and ORs its bits in (atomics: the neighbour does the same
// to the other positions of that word).
struct PlaneWriter {
    uint64_t *neg, *pos;
    unsigned long long an, ap;   // bits not yet stored (low `fill` bits valid)
    int fill;
    int first_fill;              // >= 0: the next store is the chunk's first word, whose low first_fill bits are foreign
    long long word;              // index of the word the accumulator starts in

    __device__ __forceinline__ void init(uint64_t *n, uint64_t *p, long long sample) {
        neg = n; pos = p; an = ap = 0;
        word = sample >> 6;
        fill = (int)(sample & 63);
        first_fill = fill;
    }
    __device__ __forceinline__ void merge(unsigned long long keep, int lane) {
        if (lane == 0) {
            atomicAnd((unsigned long long *)&neg[word], keep);
            atomicAnd((unsigned long long *)&pos[word], keep);
            atomicOr((unsigned long long *)&neg[word], an & ~keep);
            atomicOr((unsigned long long *)&pos[word], ap & ~keep);
        }
    }
    __device__ __forceinline__ void put(unsigned long long mn, unsigned long long mp, int nbits, int lane) {
        // nbits in 1..64, masks hold nothing above nbits
        const int room = 64 - fill;
        an |= mn << fill;
        ap |= mp << fill;
        if (nbits >= room) {
            if (first_fill > 0) merge((1ull << first_fill) - 1ull, lane);
            else if (lane == 0) {
                neg[word] = an;
                pos[word] = ap;
            }
            first_fill = 0;
            word++;
            an = room < 64 ? (mn >> room) : 0ull;
            ap = room < 64 ? (mp >> room) : 0ull;
            fill = nbits - room;
        } else {
            fill += nbits;
        }
    }
    __device__ __forceinline__ void finish(int lane) {
        if (fill > 0) {
            unsigned long long keep = ~0ull << fill;                      // positions after the chunk's last sample
            if (first_fill > 0) keep |= (1ull << first_fill) - 1ull;     // (chunk inside one word)
            merge(keep, lane);
        }
    }
};

template <int KIND>
__global__ __launch_bounds__(256) void k_threshold_rr(ThrArgs A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t slotid = blockIdx.x * (blockDim.x >> 6) + wave;
    uint32_t c;
    if (A.list) {
        if (slotid >= A.nlist) return;
        c = A.list[slotid];
    } else {
        if (slotid >= (uint32_t)A.nchunks) return;
        c = slotid;
    }
    const int L = A.L, mx = A.mx, nrows = A.nrows;
    const int p0 = (int)(c * (uint32_t)A.C) - A.off;              // sample whose ring slot is 0 (may be < 0 for chunk 0)
    const int n1 = (int)min((long long)A.n, (long long)p0 + A.C);  // end of the chunk
    const int m_start = max(max(p0, 0), (int)A.skip);
    const Carry cr = *A.carry;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    float R[RR_ROWS];
    uint32_t T = 0;   // bit r: slot 64 r + lane accepted a sample in this chunk
    uint32_t emin = 255u, emax = 0u;
    int w_nl, w_kl;
    double ss0;
    float eps = 0.f;

    // ---------------- incoming ring ----------------
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < RR_ROWS; r++) {
            const int s = 64 * r + lane;
            R[r] = (s < L) ? A.ring_carry[s] : 0.f;
        }
        ss0 = cr.ss;
        w_nl = A.nl0;
        w_kl = A.kl0;
    } else if (A.mode == 1) {
        double part = 0;
#pragma unroll
        for (int r = 0; r < RR_ROWS; r++) {
            const int s = 64 * r + lane;
            R[r] = (s < L) ? resolve_slot(A, (int)c, s) : 0.f;
            part += (double)R[r];
        }
        ss0 = wave_sum_f64(part) + cr.delta;
        resolve_low_state(A, (int)c, w_nl, w_kl);
    } else {
        // speculate: slot s last saw sample p0 - L + s; rejected-looking samples are replaced by a level estimate
        eps = A.eps;
        float mxv = 0.f;
#pragma unroll
        for (int r = 0; r < RR_ROWS; r++) {
            const int s = 64 * r + lane;
            R[r] = (s < L) ? envelope_at<KIND>(A.in, (size_t)(p0 - L + s), A.i16_scale) : 0.f;
            mxv = fmaxf(mxv, R[r]);
        }
        mxv = wave_max_f32(mxv);
        const float half = 0.5f * mxv;
        double sa = 0, na = 0;
#pragma unroll
        for (int r = 0; r < RR_ROWS; r++) {
            const bool in = (64 * r + lane < L) && (R[r] >= half);
            sa += in ? (double)R[r] : 0.0;
            na += in ? 1.0 : 0.0;
        }
        sa = wave_sum_f64(sa);
        na = wave_sum_f64(na);
        const float ca = (na > 0) ? (float)(sa / na) : mxv;
        double sb = 0, nb = 0;
#pragma unroll
        for (int r = 0; r < RR_ROWS; r++) {
            const bool in = (64 * r + lane < L) && (R[r] >= half) && (R[r] <= ca);
            sb += in ? (double)R[r] : 0.0;
            nb += in ? 1.0 : 0.0;
        }
        sb = wave_sum_f64(sb);
        nb = wave_sum_f64(nb);
        const float c0 = (nb > 0) ? (float)(sb / nb) : ca;
        const float tlo = (float)A.lo * c0, thi = (float)A.hi * c0;
        int ll = LL_NONE, nl = LL_NONE;
        double part = 0;
#pragma unroll
        for (int r = 0; r < RR_ROWS; r++) {
            const int s = 64 * r + lane;
            if (s < L) {
                const int m = p0 - L + s;
                const float x = R[r];
                if (x < tlo) ll = max(ll, m);
                else nl = max(nl, m);
                if (!(x >= tlo && x <= thi)) R[r] = c0;
                part += (double)R[r];
            }
        }
        ll = wave_max_i32(ll);
        w_nl = wave_max_i32(nl);
        w_kl = (ll == LL_NONE) ? KEY_NONE : 2 * ll + 1;
        ss0 = wave_sum_f64(part) + cr.delta;
    }
    const int nl_in = w_nl, kl_in = w_kl;
    {
        float *rin = A.ring_in + (size_t)c * L;
        uint32_t vmn = 0xFFFFFFFFu, vmx = 0u;
#pragma unroll
        for (int r = 0; r < RR_ROWS; r++) {
            const int s = 64 * r + lane;
            if (s < L) {
                rin[s] = R[r];
                const uint32_t b = __float_as_uint(R[r]);
                vmn = min(vmn, b != 0u ? b : 0xFFFFFFFFu);
                vmx = max(vmx, b);
            }
        }
        if (vmx != 0u) {
            emax = max(emax, (vmx >> 31) ? 255u : max((vmx >> 23) & 0xFFu, 1u));
            if (vmn != 0xFFFFFFFFu) emin = min(emin, max((vmn >> 23) & 0xFFu, 1u));
        }
    }

    // ---------------- the chunk: ring periods of `nrows` rows, four rows per group ----------------
    uint32_t flags = 0, all_robust = 1;
    float min_ss = 3.0e38f;
    int chunk_nl = LL_NONE, chunk_kl = KEY_NONE;
    uint32_t vmin = 0xFFFFFFFFu, vmax = 0u;
    const float etaD = 1.0f - 9.5367431640625e-07f, etaU = 1.0f + 9.5367431640625e-07f;  // 1 -+ 2^-20
    PlaneWriter pw;
    pw.init(A.neg, A.pos, (long long)max(p0, 0));
    // software prefetch: the envelope of the NEXT group's samples is loaded while this one is classified
    float xn[4];
    auto fetch = [&](int gb, int gs) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int m = gb + 64 * j + lane;
            xn[j] = (64 * j + lane < gs && m >= 0 && m < (int)A.n) ? envelope_at<KIND>(A.in, (size_t)m, A.i16_scale) : 0.f;
        }
    };
    fetch(p0, min(256, L));

    for (int pbase = p0; pbase < n1; pbase += L) {
        if (pbase + L <= 0) {           // chunk 0: a period that ends before the batch begins
            fetch(pbase + L, min(256, L));
            continue;
        }
        auto group = [&](auto gc) {
            constexpr int G = decltype(gc)::value;
            const int gbase = pbase + 256 * G;          // first sample of the group
            const int gslots = min(256, L - 256 * G);    // ring slots in the group (<= 0: beyond the ring)
            if (gslots <= 0) return;                     // beyond the ring: the previous group already fetched the next period
            if (gbase >= n1 || gbase + gslots <= 0) {    // outside the chunk / wholly before the batch: keep the fetch chain going
                const int nslots = L - 256 * (G + 1);
                if (nslots > 0) fetch(gbase + 256, min(256, nslots));
                else fetch(pbase + L, min(256, L));
                return;
            }
            float x[4], prev[4];
            bool act[4];
            const bool full = (gbase >= m_start) && (gbase + 256 <= n1) && (gslots == 256);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int m = gbase + 64 * j + lane;
                act[j] = full || ((m >= m_start) && (m < n1) && (64 * j + lane < gslots));
                x[j] = xn[j];
                prev[j] = R[4 * G + j];
            }
            {   // next group: the following four rows of this period, or the first rows of the next period
                const int nslots = L - 256 * (G + 1);
                if (nslots > 0) fetch(gbase + 256, min(256, nslots));
                else fetch(pbase + L, min(256, L));
            }
            min_ss = fminf(min_ss, (float)ss0 * etaD);
            bool fast = A.fast_ok && ss0 > 1e-30 && ss0 < 1e30;
            unsigned long long lowm[4], posm[4];
            if (fast) {
                const float t_sure = (float)(A.lo_L * 0.70 * ss0) * etaD;
                const float t_maylow = (float)(A.lo_L * 1.30 * ss0) * etaU;
                float b = 0.f;
                bool maylow = false;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    b += (act[j] && !(x[j] < t_sure)) ? fabsf(x[j] - prev[j]) : 0.f;
                    maylow |= act[j] && (x[j] < t_maylow);
                }
                b = wave_sum_f32(b) * 1.001f;
                double bt = (double)b + (double)eps * ss0;
                fast = bt < 0.25 * ss0;
                double s_dn = ss0 - bt, s_up = ss0 + bt;
                float thi_up = (float)(A.hi_L * s_up) * etaU;
                const bool key_live = (w_kl & 1) && (gbase - (w_kl >> 1)) <= mx + 1;
                if (fast && !key_live && !__any(maylow)) {
                    bool h1 = false;
#pragma unroll
                    for (int j = 0; j < 4; j++) h1 |= act[j] && (x[j] > thi_up);
                    if (__any(h1)) {
                        float b2 = 0.f;
#pragma unroll
                        for (int j = 0; j < 4; j++) b2 += (act[j] && !(x[j] > thi_up)) ? fabsf(x[j] - prev[j]) : 0.f;
                        b2 = wave_sum_f32(b2) * 1.001f;
                        bt = (double)b2 + (double)eps * ss0;
                        s_dn = ss0 - bt;
                        s_up = ss0 + bt;
                        thi_up = (float)(A.hi_L * s_up) * etaU;
                    }
                }
                const float tlo_dn = (float)(A.lo_L * s_dn) * etaD, tlo_up = (float)(A.lo_L * s_up) * etaU;
                const float thi_dn = (float)(A.hi_L * s_dn) * etaD;
                fast = fast && (tlo_dn > 1e-30f) && (thi_up < 1e30f);
                bool lw[4], hg[4], amb = false, anyhg = false;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    lw[j] = act[j] && (x[j] < tlo_dn);
                    hg[j] = act[j] && (x[j] > thi_up);
                    const bool inband = !((x[j] < tlo_dn) || (x[j] > tlo_up)) || !((x[j] < thi_dn) || (x[j] > thi_up));
                    amb |= act[j] && inband;
                    anyhg |= hg[j];
                    lowm[j] = __ballot(lw[j]);
                }
                if (__any(amb)) fast = false;
                if (fast) {
                    const int lead = (lowm[0] == ~0ull) ? 64 : (__ffsll((long long)~lowm[0]) - 1);
                    const int carry_run = gbase - 1 - w_nl;
                    unsigned long long hit = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        unsigned long long t = lowm[j];
#pragma unroll
                        for (int f = 0; f < 6; f++) t &= t >> A.fold_sh[f];
                        hit |= t & A.selmask;
                    }
                    // rows of the register ring are 64-sample aligned to the RING, the fold's blocks to the row: same thing
                    if (hit || ((carry_run > 0) && (carry_run + lead > mx))) fast = false;
                }
                if (fast) {
                    double dl = 0;
                    const bool need_st2 = __any(anyhg);
                    int before = (w_kl & 1) ? (w_kl >> 1) : LL_NONE;
                    int step_nl = LL_NONE, step_ll = LL_NONE;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        bool ps = false;
                        if (need_st2) {
                            const int m = gbase + 64 * j + lane;
                            const unsigned long long below = lowm[j] & lane_lt;
                            const int lastlow = below ? gbase + 64 * j + last_set(below) : before;
                            ps = hg[j] && ((m - lastlow) > mx + 1);
                        }
                        const bool a = act[j] && !lw[j] && !ps;
                        dl += a ? ((double)x[j] - (double)prev[j]) : 0.0;
                        R[4 * G + j] = a ? x[j] : prev[j];
                        T |= a ? (1u << (4 * G + j)) : 0u;
                        const uint32_t xb = __float_as_uint(x[j]);
                        vmin = min(vmin, (a && xb != 0u) ? xb : 0xFFFFFFFFu);
                        vmax = max(vmax, a ? xb : 0u);
                        posm[j] = need_st2 ? __ballot(ps) : 0ull;
                        const unsigned long long actm = full ? ~0ull : __ballot(act[j]);
                        const unsigned long long nonlow = actm & ~lowm[j];
                        if (lowm[j]) {
                            before = gbase + 64 * j + last_set(lowm[j]);
                            step_ll = before;
                        }
                        if (nonlow) step_nl = gbase + 64 * j + last_set(nonlow);
                    }
                    ss0 += wave_sum_f64(dl);
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
            if (!fast) {
                if (eps > 0.f) all_robust = 0;
                float nr[4] = {prev[0], prev[1], prev[2], prev[3]};
                uint32_t tb = 0;
#pragma unroll 1
                for (int j = 0; j < 4; j++) {
                    const int m = gbase + 64 * j + lane;
                    const int nl_b = w_nl, kl_b = w_kl;
                    unsigned long long lm, pm;
                    const float xj = j == 0 ? x[0] : j == 1 ? x[1] : j == 2 ? x[2] : x[3];
                    const float pj = j == 0 ? prev[0] : j == 1 ? prev[1] : j == 2 ? prev[2] : prev[3];
                    const bool aj = j == 0 ? act[0] : j == 1 ? act[1] : j == 2 ? act[2] : act[3];
                    const bool took = row_exact(A, lane, m, aj, xj, pj, ss0, w_nl, w_kl, emin, emax, flags, lm, pm);
                    if (j == 0) { lowm[0] = lm; posm[0] = pm; if (took) { nr[0] = xj; tb |= 1u; } }
                    else if (j == 1) { lowm[1] = lm; posm[1] = pm; if (took) { nr[1] = xj; tb |= 2u; } }
                    else if (j == 2) { lowm[2] = lm; posm[2] = pm; if (took) { nr[2] = xj; tb |= 4u; } }
                    else { lowm[3] = lm; posm[3] = pm; if (took) { nr[3] = xj; tb |= 8u; } }
                    if (w_nl != nl_b) chunk_nl = w_nl;
                    if (w_kl != kl_b) chunk_kl = w_kl;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) R[4 * G + j] = nr[j];
                T |= tb << (4 * G);
            }
            // classification bits of the group's samples that lie inside [max(p0,0), n1)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int s0 = gbase + 64 * j;                       // first sample of the row
                int nb = min(64, gslots - 64 * j);              // ring slots in the row
                if (nb <= 0) break;
                nb = min(nb, n1 - s0);                          // the chunk may end inside the row
                if (nb <= 0) break;
                unsigned long long mn = lowm[j], mp = posm[j];
                if (s0 < 0) {                                   // chunk 0 begins inside the row
                    const int cut = -s0;
                    if (cut >= nb) continue;
                    mn >>= cut;
                    mp >>= cut;
                    nb -= cut;
                }
                if (nb < 64) {
                    const unsigned long long keep = (1ull << nb) - 1ull;
                    mn &= keep;
                    mp &= keep;
                }
                pw.put(mn, mp, nb, lane);
            }
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
        group(std::integral_constant<int, 4>{});
        group(std::integral_constant<int, 5>{});
        group(std::integral_constant<int, 6>{});
        group(std::integral_constant<int, 7>{});
    }
    pw.finish(lane);
    if (vmax != 0u) {
        emax = max(emax, (vmax >> 31) ? 255u : max((vmax >> 23) & 0xFFu, 1u));
        if (vmin != 0xFFFFFFFFu) emin = min(emin, max((vmin >> 23) & 0xFFu, 1u));
    }

    // ---------------- publish the summary ----------------
    const int vb_old = A.ver[c];
    const int vb_new = (A.mode == 1) ? (1 - vb_old) : vb_old;
    float *ro = A.ring_out[vb_new] + (size_t)c * L;
    uint32_t *to = A.touched[vb_new] + (size_t)c * A.twords;
    uint32_t untouched = 0;
#pragma unroll
    for (int r = 0; r < RR_ROWS; r++) {
        if (r < nrows) {
            const int s = 64 * r + lane;
            const bool t = (s < L) && ((T >> r) & 1u);
            const unsigned long long bal = __ballot(t);
            if (s < L) {
                ro[s] = R[r];
                if (!t) untouched++;
            }
            if (lane == 0) {
                to[2 * r] = (uint32_t)bal;
                if (2 * r + 1 < A.twords) to[2 * r + 1] = (uint32_t)(bal >> 32);
            }
        }
    }
    emin = wave_min_u32(emin);
    emax = wave_max_u32(emax);
    untouched = (uint32_t)wave_sum_f32((float)untouched);
    flags = wave_max_u32(flags);
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
        A.gvtop[c] = 0x7F800000u;   // (this experimental kernel does not bound its window sums: the guard then asks for the sequential replay)
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

}  // namespace nfc
