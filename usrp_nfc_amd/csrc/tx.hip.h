// tx.hip.h -- SURVEY.md section 8 row f4: the transmit side as a device-side signal generator.
//
//   encode_bits   miller_encoder / manchester_encoder / binary_src.encoder (miller.py:200-233, manchester.py:64-79,
//                 binary_src.py:17-20): bits -> (level, microseconds) runs.  Host, a few hundred entries per frame.
//   k_tx_render   binary_src.work (binary_src.py:64-103): every run becomes int(dur * samp_rate / 1e6) samples of its
//                 level as complex64 (level + 0j), and -- multiplier.py:18-22 -- the stream is multiplied by a complex
//                 carrier A exp(j 2 pi f k / samp_rate).  One elementwise, write-only pass: 8 B per sample out.
//
// The reference's carrier comes from GNU Radio's sig_source_c (a fixed-point NCO with a sine table), which is not under
// /root/reference: parity at that boundary is unpinned.  The arithmetic here is stated instead: phase of
// sample k = (k * inc) mod 2^64 with inc = floor(frac(f / samp_rate) * 2^64); the top 24 bits of the phase select the
// angle, cos / sin by sincospif in fp32.  With the carrier off the output is exactly (level, 0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/nfc_amd.h"

namespace nfc {

// ---- encoders (host) ------------------------------------------------------------------------------------------
// utilities.PulseLength (utilities.py:16-23), with the reference's own expressions
constexpr double TX_FULL = 9.44, TX_ZERO = 3.00;
constexpr double TX_HALF = TX_FULL / 2, TX_ZERO_REM = TX_FULL - TX_ZERO, TX_ONE_REM = TX_HALF - TX_ZERO;

inline void tx_encode_same(const uint8_t *bits, size_t n, std::vector<nfc_tx_run> &out) {   // binary_src.py:17-20
    for (size_t i = 0; i < n; i++) out.push_back(nfc_tx_run{(int32_t)bits[i], 0, TX_FULL});
}

inline void tx_encode_manchester(const uint8_t *bits, size_t n, std::vector<nfc_tx_run> &out) {   // manchester.py:64-79
    out.push_back(nfc_tx_run{1, 0, TX_HALF});
    out.push_back(nfc_tx_run{0, 0, TX_HALF});
    int last = 0;
    for (size_t i = 0; i < n; i++) {
        const int bit = bits[i];
        if (bit == last) {   // the half that ends the previous bit and the one that starts this one merge
            out.back() = nfc_tx_run{bit, 0, TX_FULL};
            last = 1 - last;
            out.push_back(nfc_tx_run{last, 0, TX_HALF});
        } else {
            out.push_back(nfc_tx_run{1 - last, 0, TX_HALF});
            out.push_back(nfc_tx_run{last, 0, TX_HALF});
        }
    }
}

inline void tx_encode_miller(const uint8_t *bits, size_t n, std::vector<nfc_tx_run> &out) {   // miller.py:200-233
    const nfc_tx_run one[3] = {{1, 0, TX_HALF}, {0, 0, TX_ZERO}, {1, 0, TX_ONE_REM}};
    const nfc_tx_run zero0[2] = {{0, 0, TX_ZERO}, {1, 0, TX_ZERO_REM}};
    const nfc_tx_run zero1[1] = {{1, 0, TX_FULL}};
    out.push_back(zero0[0]);   // start of frame
    out.push_back(zero0[1]);
    int last_bit = 0;
    for (size_t i = 0; i <= n; i++) {   // one more zero signifies the end
        const int bit = i < n ? bits[i] : 0;
        const nfc_tx_run *cur = one;
        int len = 3;
        if (bit == 0) {
            if (last_bit == 0) { cur = zero0; len = 2; }
            else { cur = zero1; len = 1; }
        }
        last_bit = bit;
        int k = 0;
        if (cur[0].level == out.back().level) {   // same level as the last pulse: one longer pulse
            out.back().dur_us = cur[0].dur_us + out.back().dur_us;
            k = 1;
        }
        for (; k < len; k++) out.push_back(cur[k]);
    }
}

// ---- renderer (device) -------------------------------------------------------------------------------------------
constexpr int TX_BLOCK = 256;
constexpr int TX_PER_THREAD = 4;                     // complex samples per thread: 32 contiguous bytes (64 per thread measured half the store rate)
constexpr int TX_TILE = TX_BLOCK * TX_PER_THREAD;
constexpr int TX_LDS_RUNS = 1024;

struct TxArgs {
    const uint64_t *ends;      // per run: index one past its last sample (runs of zero samples repeat the previous end)
    const int8_t *levels;
    const uint32_t *tile_first;   // per tile of TX_TILE samples: the run its first sample falls in (host-made; n_tiles + 1 entries)
    uint32_t n_runs;
    uint64_t n_samples, first_index;
    int32_t carrier;
    uint64_t phase_inc;
    float amp;
    float2 *out;
};

// first run whose end lies beyond sample s, in [lo, hi)
template <class P>
__device__ __forceinline__ uint32_t tx_find_run(P ends, uint32_t lo, uint32_t hi, uint64_t s) {
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ends[mid] > s) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

__global__ __launch_bounds__(TX_BLOCK) void k_tx_render(TxArgs A) {
    __shared__ uint64_t s_ends[TX_LDS_RUNS];
    __shared__ int8_t s_lvl[TX_LDS_RUNS];
    const uint64_t t0 = (uint64_t)blockIdx.x * TX_TILE;
    if (t0 >= A.n_samples) return;
    const uint64_t t1 = min(A.n_samples, t0 + TX_TILE);
    const uint32_t r0 = A.tile_first[blockIdx.x], nr = A.tile_first[blockIdx.x + 1] - r0 + 1;   // the runs this tile's samples fall in
    const bool staged = nr <= (uint32_t)TX_LDS_RUNS;   // (nearly always: a tile of 2048 samples seldom holds more runs)
    if (staged) {
        for (uint32_t i = threadIdx.x; i < nr; i += TX_BLOCK) {
            s_ends[i] = A.ends[r0 + i];
            s_lvl[i] = A.levels[r0 + i];
        }
        __syncthreads();
    }
    const uint64_t s0 = t0 + (uint64_t)threadIdx.x * TX_PER_THREAD;
    if (s0 >= t1) return;
    float2 v[TX_PER_THREAD];
    auto sample = [&](uint64_t s, float lvl) {
        float re = lvl, im = 0.f;
        if (A.carrier) {
            const uint64_t ph = (A.first_index + s) * A.phase_inc;
            const float turn = (float)(uint32_t)(ph >> 40) * 5.9604644775390625e-08f;   // top 24 bits: [0, 1) exactly
            float sn, cs;
            sincospif(2.0f * turn, &sn, &cs);
            re = lvl * (A.amp * cs);
            im = lvl * (A.amp * sn);
        }
        return make_float2(re, im);
    };
    if (staged) {   // the tile's runs from LDS
        uint32_t r = tx_find_run(s_ends, 0u, nr, s0);
        uint64_t end = s_ends[r];
        float lvl = (float)s_lvl[r];
#pragma unroll
        for (int k = 0; k < TX_PER_THREAD; k++) {
            const uint64_t s = s0 + k;
            while (s >= end && r + 1 < nr) {   // (runs of zero samples are stepped over)
                r++;
                end = s_ends[r];
                lvl = (float)s_lvl[r];
            }
            v[k] = sample(s, lvl);
        }
    } else {
        uint32_t r = tx_find_run(A.ends, r0, r0 + nr, s0);
        uint64_t end = A.ends[r];
        float lvl = (float)A.levels[r];
#pragma unroll
        for (int k = 0; k < TX_PER_THREAD; k++) {
            const uint64_t s = s0 + k;
            while (s >= end && r + 1 < A.n_runs) {
                r++;
                end = A.ends[r];
                lvl = (float)A.levels[r];
            }
            v[k] = sample(s, lvl);
        }
    }
    if (s0 + TX_PER_THREAD <= t1) {
        float4 *o = (float4 *)(A.out + s0);   // 32-byte aligned: s0 is a multiple of TX_PER_THREAD samples
#pragma unroll
        for (int k = 0; k < TX_PER_THREAD / 2; k++)   // (plain stores: nontemporal ones measured 2.2x slower here)
            o[k] = make_float4(v[2 * k].x, v[2 * k].y, v[2 * k + 1].x, v[2 * k + 1].y);
    } else {
#pragma unroll
        for (int k = 0; k < TX_PER_THREAD; k++)
            if (s0 + k < t1) A.out[s0 + k] = v[k];
    }
}

}  // namespace nfc
