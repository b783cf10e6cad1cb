// decoder_tables.h -- host-side construction of the duration LUTs that turn the
// Modified-Miller and Manchester symbol decoders into finite-state transducers.
//
// Every duration the threshold stage emits is d * factor with an integer
// d in [1, max_len] (transition_sink.py:87,97), so each fp64 predicate of the
// reference decoders (miller.py:153-197, manchester.py:30-61) is a function of
// (state, cur, d) only.  The tables are evaluated here with the reference's own
// fp64 expressions (same constants, same operation order) and uploaded once per
// context; the decode kernels never touch floating point.
//
// State packing
//   Miller:      stage (miller.py:14-17) | has_started << 2 | prev << 3      (16 states)
//   Manchester:  prev_set | (prev + 1) << 1, prev in -1..2                   (8 states)
// Per (cur+1, d):
//   map word: next state for every current state, 4 bits each (u64 Miller, u32 Manchester)
//   out word per (state, cur+1, d): nout | sym0 << 2 | sym1 << 5  (u8)
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace nfc {

// utilities.py:7-23
enum : int { E_NONE = 0, E_TOO_SHORT = 2, E_TOO_LONG = 3, E_ENCODING = 4, E_INTERNAL = 5, E_WRONG_DUR = 6 };
constexpr double kFull = 9.44;
constexpr double kZero = 3.00;
constexpr double kHalf = kFull / 2;
constexpr double kZeroRem = kFull - kZero;
constexpr double kOneRem = kHalf - kZero;
constexpr double kOneHalf = kFull + kHalf;

enum : int { MS_BEGIN = 0, MS_ZS0 = 1, MS_OS0 = 2, MS_OS1 = 3 };
constexpr int kMillerStates = 16;
constexpr int kManchStates = 8;

struct Step {
    int next;
    int nout;
    int out[2];
};

inline bool near(double dur, double av) { return std::fabs(dur - av) <= 1.5; }  // miller.py:25,62-63

// One transition through miller_decoder.process_transition's loop body.
inline Step miller_step(int state, int cur, double dur) {
    int stage = state & 3, started = (state >> 2) & 1, prev = (state >> 3) & 1;
    Step r{0, 0, {0, 0}};
    auto pack = [&]() { return stage | (started << 2) | (prev << 3); };
    auto reset = [&]() { started = 0; stage = MS_BEGIN; };                      // miller.py:65-67
    const double lo = kZero - 1.5, hi = 2 * kFull;                               // miller.py:26-27
    if (cur == 0 && std::fabs(dur - kZero) < kZero / 2) dur = kZero;             // miller.py:157-158
    int err = E_NONE;
    if ((dur < lo || dur > hi) && (stage == MS_ZS0 || stage == MS_OS1)) {        // miller.py:165-167
        r.out[r.nout++] = (stage == MS_ZS0) ? 0 : 1;                             // _cur_type of the stage
        err = E_TOO_LONG;
    } else if (dur < lo) {
        err = E_TOO_SHORT;
    } else if (dur > hi) {
        err = E_TOO_LONG;
    }
    if (err != E_NONE) {                                                         // miller.py:173-176
        r.out[r.nout++] = err;
        reset();
        r.next = pack();
        return r;
    }
    int rets[2], n = 0;
    switch (stage) {
    case MS_BEGIN:                                                               // miller.py:73-96
        if (cur == 0) {
            if (near(dur, kZero)) { stage = MS_ZS0; started = 1; }
            else rets[n++] = E_TOO_LONG;
        } else if (started) {
            int bit = prev == 0 ? (int)E_ENCODING : 0;
            if (near(dur, kHalf)) stage = MS_OS0;
            else if (near(dur, kFull)) rets[n++] = bit;
            else if (near(dur, kOneHalf)) { rets[n++] = bit; stage = MS_OS0; }
            else rets[n++] = E_WRONG_DUR;
        }
        break;
    case MS_ZS0:                                                                 // miller.py:98-112
        if (cur == 0) rets[n++] = E_ENCODING;
        else if (near(dur, kZeroRem)) { stage = MS_BEGIN; rets[n++] = 0; }
        else if (near(dur, kZeroRem + kHalf)) { stage = MS_OS0; rets[n++] = 0; }
        else rets[n++] = E_WRONG_DUR;
        break;
    case MS_OS0:                                                                 // miller.py:114-122
        if (cur != 0) rets[n++] = E_ENCODING;
        else if (!near(dur, kZero)) rets[n++] = E_WRONG_DUR;
        else stage = MS_OS1;
        break;
    default:                                                                     // miller.py:124-148
        if (cur != 1) rets[n++] = E_ENCODING;
        else if (near(dur, kOneRem)) { rets[n++] = 1; stage = MS_BEGIN; }
        else {
            rets[n++] = 1;
            stage = MS_BEGIN;
            dur -= kOneRem;
            if (near(dur, kFull)) rets[n++] = 0;
            else if (near(dur, kHalf)) stage = MS_OS0;
            else if (near(dur, kOneHalf)) { rets[n++] = 0; stage = MS_OS0; }
            else rets[n++] = E_WRONG_DUR;
        }
        break;
    }
    for (int i = 0; i < n; i++) {                                                // miller.py:191-197
        r.out[r.nout++] = rets[i];
        if (rets[i] > 1) { reset(); prev = 0; }
        else prev = rets[i];
    }
    r.next = pack();
    return r;
}

// One transition through manchester_decoder.process_transition's loop body.
inline Step manch_step(int state, int cur, double dur) {
    int prev_set = state & 1, prev = ((state >> 1) & 3) - 1;
    Step r{0, 0, {0, 0}};
    auto pack = [&]() { return prev_set | ((prev + 1) << 1); };
    const double lo = kHalf - 1, mid = kHalf + 1, hi = 2 * kHalf + 1;            // manchester.py:17-20
    int err = E_NONE;
    if (dur < lo) err = E_TOO_SHORT;
    else if (dur > hi) err = E_TOO_LONG;
    if (err != E_NONE) {                                                         // manchester.py:40-43
        prev_set = 0;
        prev = 0;
        r.out[r.nout++] = err;
        r.next = pack();
        return r;
    }
    bool dual = dur > mid;                                                       // manchester.py:44
    if (prev_set) {                                                              // manchester.py:48-54
        if (prev == cur || (prev != 0 && prev != 1)) {
            r.out[r.nout++] = E_INTERNAL;
            r.next = pack();
            return r;
        }
        r.out[r.nout++] = prev;
        prev_set = dual ? 1 : 0;
    } else {                                                                     // manchester.py:55-59
        if (dual) {
            r.out[r.nout++] = E_ENCODING;
            r.next = pack();
            return r;
        }
        prev_set = 1;
    }
    prev = cur;                                                                  // manchester.py:61
    r.next = pack();
    return r;
}

struct DecoderTables {
    int max_len = 0;
    // index (cur+1) * (max_len+1) + d
    std::vector<uint64_t> miller_map;
    std::vector<uint32_t> manch_map;
    // index ((cur+1) * (max_len+1) + d) * nstates + state
    std::vector<uint8_t> miller_out;
    std::vector<uint8_t> manch_out;
};

inline uint8_t pack_out(const Step &s) { return (uint8_t)(s.nout | (s.out[0] << 2) | (s.out[1] << 5)); }

// factor = 1e6 / samp_rate (transition_sink.py:21); dur = d * factor (transition_sink.py:89,97).
inline DecoderTables build_tables(double samp_rate, int max_len) {
    DecoderTables t;
    t.max_len = max_len;
    const double factor = 1e6 / samp_rate;
    const int nd = max_len + 1;
    t.miller_map.assign(4 * nd, 0);
    t.manch_map.assign(4 * nd, 0);
    t.miller_out.assign((size_t)4 * nd * kMillerStates, 0);
    t.manch_out.assign((size_t)4 * nd * kManchStates, 0);
    for (int c = 0; c < 4; c++) {
        for (int d = 0; d <= max_len; d++) {
            const double dur = d * factor;
            uint64_t mm = 0;
            for (int s = 0; s < kMillerStates; s++) {
                Step st = miller_step(s, c - 1, dur);
                mm |= (uint64_t)(st.next & 15) << (4 * s);
                t.miller_out[((size_t)c * nd + d) * kMillerStates + s] = pack_out(st);
            }
            t.miller_map[c * nd + d] = mm;
            uint32_t tm = 0;
            for (int s = 0; s < kManchStates; s++) {
                Step st = manch_step(s, c - 1, dur);
                tm |= (uint32_t)(st.next & 15) << (4 * s);
                t.manch_out[((size_t)c * nd + d) * kManchStates + s] = pack_out(st);
            }
            t.manch_map[c * nd + d] = tm;
        }
    }
    return t;
}

// ---- the Miller decoder's quotient machine ----------------------------------------------------------------------------
// Two states are EQUIVALENT when no sequence of edges tells them apart: the same symbols come out of both, for ever
// (Moore / Mealy minimisation over every row of the table: partition refinement on (out byte, class of the next state)).
// miller.py's `_prev` is read in one place only -- stage BEGINNING of a started frame (miller.py:81) -- and every path into that
// situation writes it first, so the 16 states (stage x has_started x prev) fall into 9 classes, and the states reachable from the
// initial one (has_started == False only with stage BEGINNING: reset() sets both, miller.py:65-67) into 6.  A class map is then 8
// bytes and the composition of two is two v_perm_b32; outputs are those of the 16-state machine by construction.
// q_of[state]: class 0..7, or 0xFF for a state outside the reachable classes (only nfc_set_state can produce one: such a batch
// takes the 16-state kernels); rep[class] / canon[state]: the lowest state of the class -- `prev` reads 0 wherever the decoder
// cannot see it (what every decode path publishes as the carried state, so that equal decoder states compare equal).
struct MillerQuotient {
    bool ok = false;
    int classes = 0;
    uint8_t q_of[16], rep[8], canon[16];
    std::vector<uint64_t> map;    // per row: 8 classes, one byte each (unused class ids follow class 0)
    std::vector<uint16_t> step;   // per (row, class): next class | out byte << 8
};
inline MillerQuotient miller_quotient(const DecoderTables &t) {
    MillerQuotient q;
    const size_t rows = t.miller_map.size();
    int cls[16] = {0};
    for (;;) {   // refine until stable
        int ncls[16], nid = 0;
        std::vector<std::vector<int>> sigs;
        for (int s = 0; s < 16; s++) {
            std::vector<int> sig{cls[s]};
            for (size_t r = 0; r < rows; r++) {
                sig.push_back(t.miller_out[r * 16 + s]);
                sig.push_back(cls[(t.miller_map[r] >> (4 * s)) & 15]);
            }
            int id = -1;
            for (int k = 0; k < nid; k++)
                if (sigs[k] == sig) id = k;
            if (id < 0) {
                id = nid++;
                sigs.push_back(sig);
            }
            ncls[s] = id;
        }
        bool same = true;
        for (int s = 0; s < 16; s++) same = same && ncls[s] == cls[s];
        for (int s = 0; s < 16; s++) cls[s] = ncls[s];
        if (same) break;
    }
    bool reach[16] = {false};   // from the initial state (miller.py:22,29: stage BEGINNING, not started, prev 0)
    reach[0] = true;
    for (bool grew = true; grew;) {
        grew = false;
        for (int s = 0; s < 16; s++)
            if (reach[s])
                for (size_t r = 0; r < rows; r++) {
                    const int nx = (int)((t.miller_map[r] >> (4 * s)) & 15);
                    if (!reach[nx]) reach[nx] = grew = true;
                }
    }
    int qid[16];
    for (int k = 0; k < 16; k++) qid[k] = -1;
    for (int s = 0; s < 16; s++)
        if (reach[s] && qid[cls[s]] < 0) {
            if (q.classes == 8) return q;   // (does not fit: the caller keeps the 16 states)
            q.rep[q.classes] = (uint8_t)s;   // (states ascend: the lowest of the class, prev == 0 where both are in it)
            qid[cls[s]] = q.classes++;
        }
    for (int s = 0; s < 16; s++) {
        q.q_of[s] = qid[cls[s]] >= 0 ? (uint8_t)qid[cls[s]] : (uint8_t)0xFF;
        q.canon[s] = qid[cls[s]] >= 0 ? q.rep[qid[cls[s]]] : (uint8_t)s;
    }
    for (int k = q.classes; k < 8; k++) q.rep[k] = q.rep[0];
    q.map.assign(rows, 0);
    q.step.assign(rows * 8, 0);
    for (size_t r = 0; r < rows; r++) {
        uint64_t m = 0;
        for (int k = 0; k < 8; k++) {
            const int s = q.rep[k < q.classes ? k : 0];
            const int nx = q.q_of[(t.miller_map[r] >> (4 * s)) & 15];   // (a reachable state's successor is reachable)
            m |= (uint64_t)nx << (8 * k);
            q.step[r * 8 + k] = (uint16_t)(nx | (t.miller_out[r * 16 + s] << 8));
        }
        q.map[r] = m;
    }
    q.ok = true;
    return q;
}

}  // namespace nfc
