// edges.hip.h -- run-length / edge timing (transition_sink.py:84-99) from the
// per-sample classification codes, as a run-level data-parallel computation.
//
// The reference walks samples keeping (_last_bit, _dur, _current_state).  Grouping
// samples into maximal runs of equal val makes every emission a closed form of at
// most the two previous runs (see DESIGN.md "edge stage"):
//   * a run start emits ((v, d*factor), t) with d = max_len if the state after the
//     previous run is 0 else the previous run's length folded by the time-outs;
//   * inside a run of length l, time-outs fire at positions j*max_len + 1.
// Run 0 is virtual: it continues the previous batch (start = first stable sample -
// carried _dur, value = carried _last_bit, state = carried _current_state).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nfc_amd.h"

namespace nfc {

// carried transition_sink variables (device resident, host mirrored)
struct EdgeCarry {
    int32_t state, last_bit, dur;
    int32_t pad;
};

__host__ __device__ __forceinline__ int code_to_val(uint32_t c) { return c == 1u ? 1 : (c == 2u ? -1 : 0); }
__host__ __device__ __forceinline__ uint32_t val_to_code(int v) { return v == 1 ? 1u : (v == -1 ? 2u : 0u); }

struct RunView {
    const uint32_t *starts;  // starts[k-1] = first sample of real run k (k >= 1)
    const uint64_t *neg, *pos;  // classification bit planes (LOW / HIGH), 64 samples per word
    uint32_t nruns;          // real runs
    uint32_t n;              // samples in the batch
    int32_t skip;            // first stable sample
    int32_t mx;
    int32_t dur_in, last_bit_in, state_in;

    __device__ __forceinline__ long long start(uint32_t k) const {
        return k == 0 ? (long long)skip - dur_in : (long long)starts[k - 1];
    }
    __device__ __forceinline__ long long end(uint32_t k) const { return k < nruns ? (long long)starts[k] : (long long)n; }
    __device__ __forceinline__ int value(uint32_t k) const {
        if (k == 0) return last_bit_in;
        const uint32_t s = starts[k - 1];
        if ((neg[s >> 6] >> (s & 63)) & 1ull) return -1;
        return (int)((pos[s >> 6] >> (s & 63)) & 1ull);
    }
    __device__ __forceinline__ bool virt_empty() const { return nruns >= 1 && (int32_t)starts[0] == skip; }
    // state after the first sample of run k
    __device__ __forceinline__ int state_in_run(uint32_t k) const {
        const int v = value(k);
        if (v == -1) return 2;
        if (v == 1) return 1;
        if (k == 0) return state_in;
        return state_after(k - 1);
    }
    // state after the last sample of run k (transition_sink.py:95-99 resets it on a time-out)
    __device__ int state_after(uint32_t k) const {
        if (k == 0 && virt_empty()) return state_in;
        const long long l = end(k) - start(k);
        const int v = value(k);
        if (v == 0) {
            if (l >= (long long)mx + 1) return 0;
            if (k == 0) return state_in;
            // a zero run follows a non-zero run (or the virtual run), whose state needs no further look-back
            return state_after_nonzero(k - 1);
        }
        return (l > 1 && (l - 1) % mx == 0) ? 0 : (v == -1 ? 2 : 1);
    }
    __device__ __forceinline__ int state_after_nonzero(uint32_t k) const {
        if (k == 0 && virt_empty()) return state_in;
        const long long l = end(k) - start(k);
        const int v = value(k);
        if (v == 0) {  // only the virtual run can be a zero run here
            return (l >= (long long)mx + 1) ? 0 : state_in;
        }
        return (l > 1 && (l - 1) % mx == 0) ? 0 : (v == -1 ? 2 : 1);
    }
    __device__ __forceinline__ uint32_t emissions(uint32_t k) const {
        const long long l = end(k) - start(k);
        const uint32_t nt = l > 0 ? (uint32_t)((l - 1) / mx) : 0u;
        return (k >= 1 ? 1u : 0u) + nt;
    }
    // e-th emission of run k (0-based; for k >= 1 emission 0 is the run start)
    __device__ void emission(uint32_t k, uint32_t e, uint64_t g0, nfc_edge &out) const {
        const int v = value(k);
        if (k >= 1 && e == 0) {
            const int lb = value(k - 1);
            const int ps = state_after(k - 1);
            const int cs = (v == -1) ? 2 : (v == 1 ? 1 : ps);
            const long long lp = end(k - 1) - start(k - 1);
            const int durp = lp <= 0 ? 0 : (int)((lp - 1) % mx) + 1;
            out.idx = g0 + (uint64_t)start(k);
            out.d = (ps == 0) ? mx : durp;                   // transition_sink.py:87
            out.v = (int8_t)(cs == 2 ? lb + 1 : lb);        // transition_sink.py:88
            out.t = (int8_t)(cs - 1);
            out.pad = 0;
            return;
        }
        const uint32_t j = (k >= 1) ? e : e + 1;  // time-out number, 1-based
        int cs;
        if (v == -1) cs = 2;
        else if (v == 1) cs = 1;
        else cs = (j == 1) ? state_in_run(k) : 0;
        out.idx = g0 + (uint64_t)(start(k) + (long long)j * mx);
        out.d = mx;                                          // transition_sink.py:97
        out.v = (int8_t)(cs == 2 ? v + 1 : v);
        out.t = (int8_t)(cs - 1);
        out.pad = 0;
    }
};

// ---- run starts -------------------------------------------------------------
// 64 samples per word in two bit planes; a sample whose (neg, pos) differs from its predecessor's starts a run.
struct ChangeMask {
    const uint64_t *neg, *pos;
    uint32_t n, skip;
    int32_t last_bit_in;  // carried _last_bit
    __device__ __forceinline__ uint64_t mask(size_t w) const {
        const uint64_t ng = neg[w], ps = pos[w];
        uint64_t pn, pp;
        if (w == 0) {
            pn = last_bit_in == -1 ? 1ull : 0ull;
            pp = last_bit_in == 1 ? 1ull : 0ull;
        } else {
            pn = neg[w - 1] >> 63;
            pp = pos[w - 1] >> 63;
        }
        uint64_t m = (ng ^ ((ng << 1) | pn)) | (ps ^ ((ps << 1) | pp));
        // keep samples whose index is in [skip, n)
        const long long first = (long long)w * 64;
        const long long lo = (long long)skip - first, hi = (long long)n - first;
        if (lo > 0) m &= (lo >= 64) ? 0ull : (~0ull << lo);
        if (hi < 64) m &= (hi <= 0) ? 0ull : (~0ull >> (64 - hi));
        return m;
    }
};
struct LoadChangeCount {
    ChangeMask cm;
    __device__ __forceinline__ uint32_t operator()(size_t w) const { return (uint32_t)__popcll(cm.mask(w)); }
};
struct StoreRunStarts {
    ChangeMask cm;
    uint32_t *starts;
    __device__ __forceinline__ void operator()(size_t w, uint32_t excl, uint32_t /*cnt*/) const {
        uint64_t m = cm.mask(w);
        uint32_t k = excl;
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            starts[k++] = (uint32_t)(w * 64 + b);
            m &= m - 1;
        }
    }
};

// ---- emission counts per run, then one thread per edge ---------------------------
struct LoadEmissionCount {
    RunView rv;
    __device__ __forceinline__ uint32_t operator()(size_t k) const { return rv.emissions((uint32_t)k); }
};
struct StoreEmissionOffset {
    uint32_t *offs;
    __device__ __forceinline__ void operator()(size_t k, uint32_t excl, uint32_t) const { offs[k] = excl; }
};

constexpr int EDGE_ITEMS = 4;
__global__ __launch_bounds__(256) void k_write_edges(RunView rv, const uint32_t *offs, uint32_t nedges, uint64_t g0,
                                                      nfc_edge *edges) {
    const uint32_t e0 = (blockIdx.x * 256u + threadIdx.x) * EDGE_ITEMS;
    if (e0 >= nedges) return;
    // last run whose offset <= e0  (offs has nruns+1 entries, ascending; empty runs share an offset)
    uint32_t lo = 0, hi = rv.nruns + 1;  // search in [lo, hi)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offs[mid] <= e0) lo = mid; else hi = mid;
    }
    uint32_t k = lo;
    uint32_t cnt = rv.emissions(k);
    uint32_t base = offs[k];
    for (int i = 0; i < EDGE_ITEMS; i++) {
        const uint32_t e = e0 + i;
        if (e >= nedges) break;
        while (e - base >= cnt) {  // advance to the run that owns edge e
            k++;
            base = offs[k];
            cnt = rv.emissions(k);
        }
        nfc_edge out;
        rv.emission(k, e - base, g0, out);
        edges[e] = out;
    }
}

// carried (_last_bit, _dur, _current_state) after the batch
__global__ void k_edge_carry(RunView rv, EdgeCarry *carry) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if ((uint32_t)rv.skip >= rv.n) return;  // nothing but fill samples: unchanged
    const uint32_t k = rv.nruns;
    const long long l = rv.end(k) - rv.start(k);
    carry->last_bit = rv.value(k);
    carry->dur = l <= 0 ? 0 : (int)((l - 1) % rv.mx) + 1;
    carry->state = rv.state_after(k);
}

}  // namespace nfc
