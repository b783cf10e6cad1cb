// chunk_cut.h -- how a batch is cut into the threshold stage's time chunks when the cut goes by dispatch row (host only, no device
// code: host_threshold.h decides WHETHER, this says HOW; nfc_plan_row_cut exposes it to the tests, which check it without a GPU).
//
// A launch of k_threshold_wg that is one full wave of resident workgroups -- R per CU on `cus` CUs -- ends with its slowest workgroup,
// and the k-th workgroup a CU is handed is the k-th slowest.  Row r (workgroups r * cus .. (r + 1) * cus - 1) gets
// chunks of C * f[r] samples, whole rounds of rs samples, the LAST row what is left of the batch over cus chunks (rounded up to whole
// rounds: its last chunks may be short or missing).  chunk_span (threshold.hip.h) reads the table on the device.
#pragma once
#include <stdint.h>

#include <algorithm>

namespace nfc {

struct RowCut {
    uint32_t row_len[4];    // chunk length of row r (rows past R - 1 go on like row R - 1)
    uint32_t row_start[4];  // first sample of row r
    uint32_t row_div;       // chunks per row
    uint32_t nch;           // chunks that begin inside the batch
    bool by_row;            // false: the equal cut (row_len = C everywhere), because the rows' cut does not apply
};

// the equal cut as a table: chunk c covers [c * C, (c + 1) * C)
inline RowCut equal_cut(uint32_t n, uint32_t C, uint32_t cus) {
    RowCut t;
    t.row_div = std::max(1u, cus);
    for (uint32_t r = 0; r < 4; r++) {
        t.row_len[r] = C;
        t.row_start[r] = (uint32_t)std::min<uint64_t>((uint64_t)r * t.row_div * (uint64_t)C, 0xFFFFFFFFull);
    }
    t.nch = (uint32_t)(((uint64_t)n + C - 1) / C);
    t.by_row = false;
    return t;
}

// n: samples of the batch; C: the equal cut's chunk length (whole rounds); rs: samples per round; R: rows (2 .. 4); f: R - 1 factors;
// max_len: the longest chunk that may be cut (0: any -- where a chunk's plane words wait in LDS, what the LDS holds)
inline RowCut plan_row_cut(uint32_t n, uint32_t C, uint32_t rs, uint32_t cus, uint32_t R, const double *f, uint32_t max_len) {
    RowCut t = equal_cut(n, C, cus);
    if (!(cus >= 1 && rs >= 1 && C >= rs && C % rs == 0 && R >= 2 && R <= 4 && t.nch > (R - 1) * cus && t.nch <= R * cus)) return t;
    uint32_t len[4], start[4];
    uint64_t at = 0;
    for (uint32_t r = 0; r + 1 < R; r++) {
        if (!(f[r] > 0.5 && f[r] < 1.5)) return t;
        len[r] = std::max(rs, (uint32_t)((double)C * f[r] / rs + 0.5) * rs);
        if (max_len && len[r] > max_len) return t;
        start[r] = (uint32_t)at;
        at += (uint64_t)len[r] * cus;
    }
    if (at >= (uint64_t)n) return t;
    const uint32_t last = (uint32_t)((((uint64_t)n - at + cus - 1) / cus + rs - 1) / rs * rs);
    if (last < rs || last > C) return t;   // (the last row is the shortest)
    const uint32_t nch = (R - 1) * cus + (uint32_t)(((uint64_t)n - at + last - 1) / last);
    if (nch > R * cus) return t;
    for (uint32_t r = R - 1; r < 4; r++) {   // (rows past the last go on like it: none of their chunks exist)
        start[r] = (uint32_t)std::min<uint64_t>(at + (uint64_t)(r - (R - 1)) * cus * last, 0xFFFFFFFFull);
        len[r] = last;
    }
    for (uint32_t r = 0; r < 4; r++) {
        t.row_len[r] = len[r];
        t.row_start[r] = start[r];
    }
    t.nch = nch;
    t.by_row = true;
    return t;
}

}  // namespace nfc
