// nfc_amd.hip -- context, batch orchestration and the C-ABI (include/nfc_amd.h) of the
// MI355X-native ISO-14443A IQ -> bit path.  gfx950 only; no CPU fallback.
//
// One nfc_push = one batch, enqueued without a host round trip (process_batch):
//   k_fill (first av_window samples, then the per-batch preparation)
//   -> k_threshold pass 0 (speculate) -> k_certify (+ end-of-batch state, verdict summary)
//      [-> re-runs from the exact state | k_threshold_seq over a prefix, then another attempt]      (run_threshold)
//   -> tile aggregates (first / last two changes, entries) -> k_write_edges                         (run_edges)
//   -> k_dec_reduce -> tile prefixes -> k_dec_apply -> symbol / bit / close offsets and framing states (one scan)
//   -> k_frame_write -> k_pkt_finish (both packet types in each launch; fills the host's mirror of the state) (run_decode)
//   (batches up to 2^18 samples: the three stages after the threshold stage in ONE launch, small.hip.h)
// then one wait; the edge / decode stages are repeated if the certification failed or a capacity estimate was short.
// Tile prefixes: folded by every tile's own workgroup while the tiles are few, by a prefix launch beyond (scan.hip.h).
// Outputs stay in HBM until read through nfc_read_*.  Host-only: the protocol layer of protocol.h (nfc_fsm_*), the
// encoders of tx.hip.h; its renderer (row f4) is a kernel of its own outside the batch.
//
// The host side in parts, included below into this one translation unit:
//   host_context.h    nfc_ctx, the mirrored state block, launch helpers
//   host_threshold.h  the threshold stage: plan, pass 0, certification rounds, re-runs, sequential fallback
//   host_stages.h     edge stage, decode + framing stage, short batches in one launch, process_batch
//   host_submit.h     batches submitted ahead (nfc_submit_device / nfc_wait)
// and, in this file, the C-ABI entry points.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <functional>
#include <vector>

#include "../../include/nfc_amd.h"
#include "launch_check.h"
#include "decode.hip.h"
#include "decoder_tables.h"
#include "edges.hip.h"
#include "protocol.h"
#include "scan.hip.h"
#include "small.hip.h"
#include "threshold.hip.h"
#ifdef NFC_TEST_HOOKS
#include "tail.hip.h"   // the fused tail: built, measured, not adopted -- the test build keeps it exact
#endif
#include "threshold_lean.hip.h"
#include "threshold_wg.hip.h"
#include "tx.hip.h"

using namespace nfc;

#include "chunk_cut.h"
#include "host_context.h"
#include "host_threshold.h"
#include "host_stages.h"
#include "host_submit.h"

// ===========================================================================
// C-ABI
// ===========================================================================
namespace {
// the instantiation of k_threshold_wg a context launches (for the occupancy query and the LDS attribute)
const void *wg_ex_kernel_of(int kind) {
    switch (kind) {
    case NFC_IN_IQ_F32: return (const void *)k_threshold_wg<IN_IQ_F32, 4, true>;
    case NFC_IN_ENV_F32: return (const void *)k_threshold_wg<IN_ENV_F32, 4, true>;
    case NFC_IN_REAL_F32_SQ: return (const void *)k_threshold_wg<IN_REAL_F32_SQ, 4, true>;
    default: return (const void *)k_threshold_wg<IN_I16_SQ, 4, true>;
    }
}
#ifdef NFC_TEST_HOOKS
const void *wg_flags_kernel_of(int kind, int nr) {
    switch (kind) {
    case NFC_IN_IQ_F32: return nr == 8 ? (const void *)k_threshold_wg<IN_IQ_F32, 8, false, true> : (const void *)k_threshold_wg<IN_IQ_F32, 4, false, true>;
    case NFC_IN_ENV_F32: return nr == 8 ? (const void *)k_threshold_wg<IN_ENV_F32, 8, false, true> : (const void *)k_threshold_wg<IN_ENV_F32, 4, false, true>;
    case NFC_IN_REAL_F32_SQ: return (const void *)k_threshold_wg<IN_REAL_F32_SQ, 4, false, true>;
    default: return (const void *)k_threshold_wg<IN_I16_SQ, 4, false, true>;
    }
}
#endif
const void *wg_kernel_of(int kind, int nr) {
    // (six instantiations for pass 0: four rows per step for every input kind, eight for the two kinds a long-window capture arrives in;
    // wg_ex_kernel_of above: the four that re-run chunks with failed rounds evaluated in place)
    switch (kind) {
    case NFC_IN_IQ_F32: return nr == 8 ? (const void *)k_threshold_wg<IN_IQ_F32, 8> : (const void *)k_threshold_wg<IN_IQ_F32, 4>;
    case NFC_IN_ENV_F32: return nr == 8 ? (const void *)k_threshold_wg<IN_ENV_F32, 8> : (const void *)k_threshold_wg<IN_ENV_F32, 4>;
    case NFC_IN_REAL_F32_SQ: return (const void *)k_threshold_wg<IN_REAL_F32_SQ, 4>;
    default: return (const void *)k_threshold_wg<IN_I16_SQ, 4>;
    }
}
}  // namespace

extern "C" {

int nfc_abi_version(void) { return NFC_AMD_ABI_VERSION; }

int nfc_plan_row_cut(uint32_t n, uint32_t C, uint32_t rs, uint32_t cus, uint32_t rows, const double *factors, uint32_t max_len, uint32_t out[10]) {
    if (!out || !rs || !C || !cus || (rows >= 2 && !factors)) return NFC_ERR_ARG;
    const RowCut t = (rows >= 2 && rows <= 4) ? plan_row_cut(n, C, rs, cus, rows, factors, max_len) : equal_cut(n, C, cus);
    for (int r = 0; r < 4; r++) {
        out[r] = t.row_len[r];
        out[4 + r] = t.row_start[r];
    }
    out[8] = t.row_div;
    out[9] = t.nch;
    return t.by_row ? 1 : 0;
}

int nfc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *nfc_last_error(const nfc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int nfc_create(const nfc_params *p, nfc_ctx **out) {
    if (!p || !out) return fail(nullptr, NFC_ERR_ARG, "null argument");
    *out = nullptr;
    if (!(p->samp_rate > 0)) return fail(nullptr, NFC_ERR_ARG, "samp_rate must be positive");
    if (p->av_window < 1 || p->av_window > 30000) return fail(nullptr, NFC_ERR_ARG, "av_window must be in [1, 30000]");
    if (p->max_len < 1 || p->max_len > 4000) return fail(nullptr, NFC_ERR_ARG, "max_len must be in [1, 4000]");
    if (p->input_kind < 0 || p->input_kind > 3) return fail(nullptr, NFC_ERR_ARG, "unknown input_kind");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, NFC_ERR_DEVICE, "no HIP device: this library has no CPU fallback");
    if (p->device < 0 || p->device >= ndev) return fail(nullptr, NFC_ERR_ARG, "device %d out of range (%d devices)", p->device, ndev);
    if (hipSetDevice(p->device) != hipSuccess) return fail(nullptr, NFC_ERR_DEVICE, "hipSetDevice failed");

    nfc_ctx *c = new nfc_ctx();
    c->P = *p;
    c->L = p->av_window;
    c->mx = p->max_len;
    c->factor = 1e6 / p->samp_rate;
    c->Lpad = (c->L + 15) & ~15;
    c->twords = (c->L + 31) / 32;
    int C = p->chunk_samples > 0 ? p->chunk_samples : 4096;   // smallest chunk of the adaptive rule (run_threshold)
    C = std::max(C, 2 * c->L);
    C = std::max(C, c->mx + 2);
    c->rows_per_step = 4;
    // The PRODUCT build reads no environment variable: which kernels a deployed process runs does not depend on what it inherited.
    // Every switch below -- the ones that make a context misbehave on purpose or talk, and the ones that select between kernel
    // forms that are all exact (the A/Bs of DESIGN.md, the tests that keep the less travelled forms exact) -- exists in the TEST
    // build only (-DNFC_TEST_HOOKS: usrp_nfc_amd/libnfc_amd_hooks.so, build.py), and is read here, once.  README.md lists them.
#ifdef NFC_TEST_HOOKS
#define NFC_ENV(name) getenv(name)
#else
#define NFC_ENV(name) ((const char *)nullptr)
#endif
    c->use_small = NFC_ENV("NFC_NO_SMALL") ? 0 : 1;
#ifdef NFC_TEST_HOOKS
    if (const char *e = getenv("NFC_TAIL")) c->tail_on = atoi(e) != 0;   // 1: the fused tail (tail.hip.h) instead of the five launches
    c->dbg_bad_launch = getenv("NFC_DEBUG_BAD_LAUNCH") != nullptr;
    c->dbg_redo_submitted = getenv("NFC_DEBUG_REDO_SUBMITTED") != nullptr;
    c->dbg_any = getenv("NFC_DEBUG") != nullptr;
    c->dbg_trace = getenv("NFC_TRACE") != nullptr;
    devbuf_trace() = getenv("NFC_TRACE_ALLOC") != nullptr;
    if (const char *e = getenv("NFC_DEBUG_CLK")) {
        c->dbg_clk = true;
        c->dbg_clk_path = e;
    }
#endif
    c->dbg_no_submit_ahead = NFC_ENV("NFC_NO_SUBMIT_AHEAD") != nullptr;
    if (const char *e = NFC_ENV("NFC_WG")) c->wg = atoi(e) != 0;
    if (const char *e = NFC_ENV("NFC_WG_ROUNDS")) c->wg_rounds = std::max(1, atoi(e));
    if (const char *e = NFC_ENV("NFC_WG_BULK")) c->wg_bulk = atoi(e) != 0;
    if (const char *e = NFC_ENV("NFC_WG_RERUN")) c->wg_rerun_lone = atoi(e) != 0;   // 0: lone failures are k_threshold's too (with NFC_WG_EX=0: it re-runs everything)
    if (const char *e = NFC_ENV("NFC_WG_LONE")) {   // max,div
        int a = 4, b = 64;
        if (sscanf(e, "%d,%d", &a, &b) >= 1) c->wg_lone_max = std::max(0, a), c->wg_lone_div = std::max(1, b);
    }
    if (const char *e = NFC_ENV("NFC_LEAN")) c->lean = atoi(e) != 0;
    c->lean_k = 0;        // chosen below from the occupancy the LDS ring allows, unless set here
    c->lean_rounds = 0;
    if (const char *e = NFC_ENV("NFC_LEAN_ROUNDS")) c->lean_rounds = std::max(1, atoi(e));
    if (const char *e = NFC_ENV("NFC_LEAN_GFAC")) c->lean_gfac = (float)atof(e);
    if (const char *e = NFC_ENV("NFC_LEAN_GMIN")) c->lean_gmin = (float)atof(e);
    if (const char *e = NFC_ENV("NFC_OWN_PREFIX_MAX")) c->own_prefix_max = (uint32_t)strtoul(e, nullptr, 10);
    if (const char *e = NFC_ENV("NFC_DEC_SPEC")) c->dec_spec = atoi(e) != 0;
    if (const char *e = NFC_ENV("NFC_SPIN_WAIT")) c->spin_wait = atoi(e) != 0;
    if (const char *e = NFC_ENV("NFC_WG_ROWBAL")) {   // 0: chunks of equal length (host_threshold.h: the cut by dispatch row); a,b,c: the rows' factors
        double f[3];
        f[1] = f[2] = 1.0;
        if (sscanf(e, "%lf,%lf,%lf", &f[0], &f[1], &f[2]) >= 2 && f[0] > 0.5 && f[0] < 1.5 && f[1] > 0.5 && f[1] < 1.5 && f[2] > 0.5 && f[2] < 1.5)
            memcpy(c->rowbal_f[0], f, sizeof f), c->rowbal_set = true;
        else c->wg_rowbal = atoi(e) != 0;
    }
    if (const char *e = NFC_ENV("NFC_DEC_RUNIN")) {   // run-in edges per decode tile: 512, 1024 or 2048
        const int v = atoi(e);
        c->dec_runin = v >= 2048 ? 8 : (v >= 1024 ? 4 : 2);
    }
    {
        c->rows_per_step = 4;   // (8-row steps measured slower: 126 VGPRs, four waves per SIMD)
        const int stp = 64 * c->rows_per_step;
        C = (C + stp - 1) / stp * stp;
    }
    c->C = C;
    c->C_min = C;
    c->lds_per_slot = (p->input_kind == NFC_IN_ENV_F32) ? 5 : 4;   // ring (+ touched byte map for raw envelopes)
    c->wpb = std::max(1, std::min(4, (int)(65536 / ((size_t)c->Lpad * c->lds_per_slot))));
    // An LDS ring beyond 12 KB leaves a SIMD with fewer than four waves: such windows keep the ring in global memory
    // (a delay line read one step ahead), and the registers set the occupancy again.
    c->gring_ok = (size_t)c->Lpad * c->lds_per_slot > 12 * 1024 && c->L >= 2 * STEP;
    if (const char *e = NFC_ENV("NFC_RING")) {
        if (strcmp(e, "global") == 0 && c->L >= 2 * STEP) c->gring_ok = c->gring_force = 1;
        else if (strcmp(e, "lds") == 0) c->gring_ok = 0;
    }
    c->hi_plus = p->hi_val + 0.1;  // transition_sink.py:63
    const double eps = std::ldexp(1.0, -48);
    auto band = [&](double v, double &a, double &b) {
        a = v - std::fabs(v) * eps;
        b = v + std::fabs(v) * eps;
    };
    band(p->lo_val, c->lo_a, c->lo_b);
    band(p->hi_val, c->hi_a, c->hi_b);
    auto sane = [](double v) { return v == 0 || (std::fabs(v) > 1e-100 && std::fabs(v) < 1e100); };
    c->bands_ok = sane(p->lo_val) && sane(p->hi_val) && std::isfinite(p->lo_val) && std::isfinite(p->hi_val);
    c->fast_ok = c->bands_ok && p->lo_val > 0 && p->hi_val > p->lo_val;
    {   // a LOW run longer than max_len covers an aligned block of b samples, b the largest power of two
        // with 3b - 2 <= max_len + 1: the fast path detects those runs by folding the LOW mask
        int b = 1;
        while (3 * (2 * b) - 2 <= c->mx + 1 && 2 * b <= 64) b *= 2;   // (3b - 2: rows of the register ring may be short)
        c->nfold = 0;
        while ((1 << c->nfold) < b) c->nfold++;
        c->selmask = 0;
        for (int k = 0; k < 64; k += b) c->selmask |= 1ull << k;
    }
    c->eps = 0.01f;  // certification margin of the speculative pass, relative to the window sum
    if (const char *e = NFC_ENV("NFC_EPS")) c->eps = (float)atof(e);
    c->i16_scale = p->i16_scale > 0.f ? p->i16_scale : -1.0f;   // (0: GNU Radio's wavfile_source normalisation, sample / 32767; threshold.hip.h: i16_to_float)
    static const size_t bps[4] = {8, 4, 4, 2};
    c->in_bytes_per_sample = bps[p->input_kind];
    memset(&c->h_carry, 0, sizeof c->h_carry);
    c->h_carry.ss_emin = 255;
    c->h_ecarry = EdgeCarry{0, 0, 1, 0};  // transition_sink.py:22-23,30
    memset(&c->h_dcarry, 0, sizeof c->h_dcarry);
    c->h_dcarry.mil_state = 0;             // stage BEGINNING, not started, prev 0 (miller.py:22,29)
    c->h_dcarry.man_state = (0 + 1) << 1;  // prev_set False, prev 0 (manchester.py:22-25)
    memset(&c->stats, 0, sizeof c->stats);

#define CRT(call)                                                                                      \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            int rc__ = fail(nullptr, NFC_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e__));   \
            nfc_destroy(c);                                                                            \
            return rc__;                                                                               \
        }                                                                                              \
    } while (0)
    {
        hipDeviceProp_t prop;
        CRT(hipGetDeviceProperties(&prop, p->device));
        const size_t lds_wave = (size_t)c->Lpad * c->lds_per_slot;
        int per_cu = (int)std::min<size_t>(20, (size_t)(160 * 1024) / (lds_wave * c->wpb) * c->wpb);   // LDS- and VGPR-bound
        c->n_cus = std::max(1, prop.multiProcessorCount);
        c->wave_slots = std::max(1, prop.multiProcessorCount * std::max(1, per_cu));
        // lean kernel: four steps ahead (104 registers: at most four waves per SIMD); a superstep long enough that its fixed cost
        // fades -- the drift allowance grows with its length relative to the window (about half the window at most: beyond, the
        // widened bands reach the loaded half bits of tag frames)
        // Measured (configs[1] / [2], 1e8 samples): two waves per SIMD with four steps ahead beat five waves with three --
        // 0.206 / 0.196 ms against 0.237 / 0.230 on the same box: a chunk's fixed cost (the window before it read and
        // summarised, its summary written) is paid half as often, and two waves already keep a SIMD's issue slots busy.
        if (!c->lean_k) c->lean_k = 4;
        c->lean_slots = std::min(c->wave_slots, prop.multiProcessorCount * 8);
        if (const char *e = NFC_ENV("NFC_LEAN_WAVES")) c->lean_slots = std::min(c->wave_slots, prop.multiProcessorCount * 4 * std::max(1, atoi(e)));
        if (!c->lean_rounds) c->lean_rounds = std::max(1, (int)(0.4 * c->L / (256.0 * c->lean_k)));
        c->wave_slots_g = prop.multiProcessorCount * 20;   // VGPR-bound: five waves per SIMD
        // LDS the lean kernel's resident waves hold per CU: a batch is only run ahead of its predecessor's edge / decode stages
        // (nfc_submit_device) while those stages' workgroups (25 KB each) still fit beside it
        c->lean_lds_per_cu = (size_t)((c->lean_slots + prop.multiProcessorCount - 1) / prop.multiProcessorCount) * lds_wave;
        c->ahead_lds_per_cu = c->lean_lds_per_cu;
        // the workgroup kernel: one chunk per 256-thread workgroup, as many resident per CU as LDS and registers admit
        c->wg_lds_base = (size_t)c->Lpad * 4 + WG_SHARED_BYTES;
        // rows per step: eight where that leaves a superstep of at least two rounds within 0.8 windows (measured: at av_window 10000
        // eight rows gain 2 % over four; at 2000 more rows with one-round supersteps lose to four rows with two), else four; a round
        // (four steps) must fit the window, max_len must lie within one step.  Eight rows are instantiated for fc32 IQ and the
        // float32 envelope -- what a capture at a rate that wants such a window arrives as.
        const bool nr8_kind = p->input_kind == NFC_IN_IQ_F32 || p->input_kind == NFC_IN_ENV_F32;
        c->wg_nr = (nr8_kind && 0.8 * c->L / (double)wg_round_samples(8) >= 1.5) ? 8 : 4;
        if (const char *e = NFC_ENV("NFC_WG_NR")) {
            const int v = atoi(e);
            if ((v == 4 || (v == 8 && nr8_kind)) && c->L >= wg_round_samples(v)) c->wg_nr = v;
        }
        // (+ the staging of the plane words: a ring of 2 FR rounds; a whole chunk's where the LDS has room, launch_wg)
        c->wg_lds = c->wg_lds_base + wg_stage_bytes(c->wg_nr, 2 * wg_flush_rounds(c->wg_nr));
        c->wg_ok = c->mx <= 64 * c->wg_nr - 2 && c->L >= wg_round_samples(c->wg_nr) && c->wg_lds <= 160 * 1024;
        if (c->wg_ok) {
            const void *kern = wg_kernel_of(p->input_kind, c->wg_nr);
#ifdef NFC_TEST_HOOKS
            if (const char *e = getenv("NFC_WG_FLAGS")) c->wg_flags = atoi(e) != 0;
            if (c->wg_flags) kern = wg_flags_kernel_of(p->input_kind, c->wg_nr);
#endif
            if (c->wg_lds > 64 * 1024) CRT(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->wg_lds));
            int per_cu_wg = 0;
            CRT(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_wg, kern, 256, c->wg_lds));
            const int per_cu_max = per_cu_wg;
            if (per_cu_wg < 1) c->wg_ok = 0;   // (the kernel does not fit a CU with this ring: the one-wave kernels)
            per_cu_wg = std::max(1, std::min(4, per_cu_wg));   // (measured: four resident workgroups per CU -- four waves per SIMD -- beat five and three)
            if (const char *e = NFC_ENV("NFC_WG_PER_CU")) per_cu_wg = std::max(1, std::min(per_cu_max, atoi(e)));
            c->wg_slots = prop.multiProcessorCount * per_cu_wg;
            // the most dynamic LDS a workgroup may ask for with per_cu_wg of them still resident per CU: what a chunk's planes may
            // take when they are kept until the chunk is done (one batch at a time only: launch_wg)
            c->wg_lds_bulk_max = 0;
            if (c->wg_bulk) {
                size_t cand = ((size_t)160 * 1024 / (size_t)per_cu_wg) & ~(size_t)1023;
                while (cand > c->wg_lds) {
                    int fit = 0;
                    if (cand > 64 * 1024) CRT(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cand));
                    CRT(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, kern, 256, cand));
                    if (fit >= per_cu_wg) break;
                    cand -= 1024;
                }
                if (cand > c->wg_lds) c->wg_lds_bulk_max = cand;
            }
            // Batches submitted ahead run beside the edge and decode stages of the batch before them, and those need registers to
            // be resident at all: four workgroups of this kernel per CU hold 4 x 96 of a SIMD's 512 registers and leave the stages
            // ONE wave per SIMD (measured: k_dec_apply 19 -> 102 us beside it, the stages' chain -- not this kernel -- then sets the
            // period).  Three per CU leave them two or three: 0.263 -> 0.241 ms per batch (two: 0.254).
            int per_cu_ahead = std::min(per_cu_wg, 3);
            // (... and LDS: the later stages' workgroups want 12-29 KB each beside it.  A long window's ring -- av_window 10 000: 40 KB, 52 KB
            // with the rest -- fills the CU at three per CU, so such streams take the synchronous path: host_submit.h, submit_fast_ok.
            // Measured in round 5, configs[3], with TWO per CU for batches submitted ahead (56 KB left; the kernel alone loses 3 % to it,
            // 1.589 -> 1.637 ms per launch; one per CU: 2.31): 1.87-2.02 ms per batch against 1.92 one batch at a time -- the kernel
            // stretches to 1.69-2.0 ms beside the other stages, which are 0.35 ms of a 1.95 ms step to begin with.  Not taken.)
            if (const char *e = NFC_ENV("NFC_WG_PER_CU_AHEAD")) per_cu_ahead = std::max(1, std::min(per_cu_max, atoi(e)));
            c->wg_slots_ahead = prop.multiProcessorCount * per_cu_ahead;
            // (what a batch submitted ahead holds of a CU's LDS is this kernel's, not the lean kernel's: host_submit.h, submit_fast_ok)
            if (c->wg_ok && c->wg && c->lean) c->ahead_lds_per_cu = c->wg_lds * (size_t)per_cu_ahead;
            // re-runs with failed rounds evaluated in place (k_threshold_wg<KIND, 4, true>): max_len within one four-row step, a round of
            // four of them within the window; up to a machine-full of failing chunks per round
            c->wg_ex_lds = c->wg_lds_base + wg_stage_bytes(4, 2 * wg_flush_rounds(4));
            c->wg_ex_ok = c->wg_ok && c->mx <= 64 * 4 - 2 && c->L >= wg_round_samples(4) && c->wg_ex_lds <= 160 * 1024;
            if (c->wg_ex_ok) {
                const void *kx = wg_ex_kernel_of(p->input_kind);
                if (c->wg_ex_lds > 64 * 1024) CRT(hipFuncSetAttribute(kx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->wg_ex_lds));
                int fit = 0;
                CRT(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, kx, 256, c->wg_ex_lds));
                if (fit < 1) c->wg_ex_ok = 0;
                c->wg_ex_max = prop.multiProcessorCount * std::max(1, std::min(4, fit));
            }
            if (const char *e = NFC_ENV("NFC_WG_EX")) {   // 0: never; N: up to N failing chunks per round
                const int v = atoi(e);
                if (v <= 0) c->wg_ex_ok = 0;
                else c->wg_ex_max = v;
            }
            // the longest superstep (rounds): the kernel lengthens and shortens its supersteps by the head-room it sees between the
            // samples and the thresholds; this caps them
            if (!c->wg_rounds) c->wg_rounds = 8;
            if (const char *e = NFC_ENV("NFC_CHUNK_ADAPT")) c->fine_adapt = atoi(e) != 0;
            if (const char *e = NFC_ENV("NFC_CHUNK_MULT")) c->fine_mult = std::max(1, std::min(16, atoi(e)));
        }
    }
    CRT(hipStreamCreateWithFlags(&c->own_st, hipStreamNonBlocking));
    c->st = c->own_st;
    for (auto &e : c->ev) CRT(hipEventCreate(&e));
    for (auto &e : c->kev) CRT(hipEventCreate(&e));
    CRT(hipStreamCreateWithFlags(&c->st_a, hipStreamNonBlocking));
    static_assert(NRING == 4 && NSUB == 3, "the buffer list of nfc_destroy names them");
    for (int b = 0; b < NSUB; b++) {
        CRT(hipEventCreateWithFlags(&c->ev_a[b], hipEventDisableTiming));
        CRT(hipEventCreateWithFlags(&c->ev_b[b], hipEventDisableTiming));
        CRT(hipEventCreate(&c->kev_sub[b][0]));
        CRT(hipEventCreate(&c->kev_sub[b][1]));
        CRT(hipHostMalloc((void **)&c->hs_a[b], sizeof(DevState), hipHostMallocDefault));
        memset(c->hs_a[b], 0, sizeof(DevState));
    }
    const size_t lds = (size_t)c->wpb * c->Lpad * c->lds_per_slot;
    if (lds > 160 * 1024) {
        nfc_destroy(c);
        return fail(nullptr, NFC_ERR_ARG, "av_window too large for one wave's LDS ring");
    }
    if (lds > 64 * 1024) {
        CRT(hipFuncSetAttribute((const void *)k_threshold<IN_IQ_F32, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold<IN_ENV_F32, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold<IN_REAL_F32_SQ, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold<IN_I16_SQ, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_IQ_F32, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_IQ_F32, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_ENV_F32, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_ENV_F32, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_REAL_F32_SQ, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_REAL_F32_SQ, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_I16_SQ, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_I16_SQ, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // decoder LUTs
    DecoderTables t = build_tables(p->samp_rate, c->mx);
    std::vector<uint8_t> milb(t.miller_map.size() * 16), manb(t.manch_map.size() * 8);
    for (size_t i = 0; i < t.miller_map.size(); i++)
        for (int s = 0; s < 16; s++) milb[i * 16 + s] = (uint8_t)((t.miller_map[i] >> (4 * s)) & 15u);
    for (size_t i = 0; i < t.manch_map.size(); i++)
        for (int s = 0; s < 8; s++) manb[i * 8 + s] = (uint8_t)((t.manch_map[i] >> (4 * s)) & 15u);
    // walking form: next state | out byte << 8 per (LUT row, state)
    std::vector<uint16_t> mils(milb.size()), mans(manb.size());
    for (size_t i = 0; i < milb.size(); i++) mils[i] = (uint16_t)(milb[i] | (t.miller_out[i] << 8));
    for (size_t i = 0; i < manb.size(); i++) mans[i] = (uint16_t)(manb[i] | (t.manch_out[i] << 8));
    // one row more in both map tables, the identity, at index 4 (max_len + 1): what an edge that is not routed to a decoder
    // looks up in the speculative decode's compositions (decode.hip.h: k_dec_spec)
    for (int s = 0; s < 16; s++) milb.push_back((uint8_t)s);
    for (int s = 0; s < 8; s++) manb.push_back((uint8_t)s);
    CRT(c->d_mil_map.ensure(milb.size()));
    CRT(c->d_man_map.ensure(manb.size()));
    CRT(c->d_mil_out.ensure(mils.size() * 2));
    CRT(c->d_man_out.ensure(mans.size() * 2));
    CRT(hipMemcpy(c->d_mil_map.p, milb.data(), milb.size(), hipMemcpyHostToDevice));
    CRT(hipMemcpy(c->d_man_map.p, manb.data(), manb.size(), hipMemcpyHostToDevice));
    CRT(hipMemcpy(c->d_mil_out.p, mils.data(), mils.size() * 2, hipMemcpyHostToDevice));
    CRT(hipMemcpy(c->d_man_out.p, mans.data(), mans.size() * 2, hipMemcpyHostToDevice));
    {   // the Miller decoder's quotient machine (decoder_tables.h): class maps of 8 bytes for the speculative decode
        const MillerQuotient q = miller_quotient(t);
        c->T.q_ok = q.ok ? 1 : 0;
        for (int k = 0; k < 16; k++) c->mil_q_of[k] = q.ok ? q.q_of[k] : (uint8_t)0xFF;
        uint8_t canon[16], rep[8] = {0};
        for (int k = 0; k < 16; k++) canon[k] = q.ok ? q.canon[k] : (uint8_t)k;
        if (q.ok) memcpy(rep, q.rep, 8);
        memcpy(c->T.canon, canon, 16);
        memcpy(c->T.q_rep, rep, 8);
        memcpy(c->mil_canon, canon, 16);
        if (q.ok) {
            std::vector<uint64_t> qm(q.map);
            qm.push_back(0x0706050403020100ull);   // the identity row at index 4 (max_len + 1)
            CRT(c->d_qmil_map.ensure(qm.size() * 8));
            CRT(c->d_qmil_step.ensure(q.step.size() * 2 + 16));
            CRT(hipMemcpy(c->d_qmil_map.p, qm.data(), qm.size() * 8, hipMemcpyHostToDevice));
            CRT(hipMemcpy(c->d_qmil_step.p, q.step.data(), q.step.size() * 2, hipMemcpyHostToDevice));
            c->T.qmil_map = c->d_qmil_map.as<uint2>();
            c->T.qmil_step = c->d_qmil_step.as<uint16_t>();
        }
    }
    c->T.mil_map = c->d_mil_map.as<uint4>();
    c->T.man_map = c->d_man_map.as<uint2>();
    c->T.mil_step = c->d_mil_out.as<uint16_t>();
    c->T.man_step = c->d_man_out.as<uint16_t>();
    c->T.nd = c->mx + 1;
    c->T.reader = p->enable_reader ? 1 : 0;
    c->T.tag = p->enable_tag ? 1 : 0;
    // carried state
    CRT(c->d_state.ensure(sizeof(DevState)));
    // (mapped AND coherent, explicitly: the host watches the stamp word while the stream runs -- host_threshold.h: wait_for_stamp)
    CRT(hipHostMalloc((void **)&c->hs, sizeof(DevState), hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->hs, 0, sizeof(DevState));
    CRT(hipHostGetDevicePointer(&c->hs_dev, c->hs, 0));
    for (int b = 0; b < NRING; b++) {
        CRT(c->d_ring[b].ensure((size_t)c->Lpad * 4));
        CRT(hipMemset(c->d_ring[b].p, 0, (size_t)c->Lpad * 4));
    }
    for (int b = 0; b < 2; b++) {
        CRT(c->d_pending[b][0].ensure(1024));
        CRT(c->d_pending[b][1].ensure(1024));
    }
    push_state(c, 1);
    CRT(hipStreamSynchronize(c->st));
#undef CRT
    *out = c;
    return NFC_OK;
}

void nfc_destroy(nfc_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->P.device);
    if (c->st_a) (void)hipStreamSynchronize(c->st_a);
    if (c->st && c->st == c->own_st) (void)hipStreamSynchronize(c->st);
    else (void)hipDeviceSynchronize();   // on a caller's stream (nfc_set_stream): the handle may be gone by now
    DevBuf *all[] = {&c->d_mil_map, &c->d_man_map, &c->d_mil_out, &c->d_man_out, &c->d_qmil_map, &c->d_qmil_step, &c->d_state,
                     &c->d_ring[0], &c->d_ring[1], &c->d_ring[2], &c->d_ring[3], &c->d_neg_alt[0], &c->d_pos_alt[0], &c->d_neg_alt[1], &c->d_pos_alt[1], &c->d_certinfo, &c->d_in, &c->d_neg, &c->d_pos, &c->d_ringin, &c->d_meta, &c->d_ringout[0], &c->d_ringout[1], &c->d_touched[0],
                     &c->d_touched[1], &c->d_info[0], &c->d_info[1], &c->d_ver, &c->d_cflags, &c->d_list, &c->d_ecode, &c->d_epos, &c->d_eidx, &c->d_states, &c->d_sym[0], &c->d_sym[1],
                     &c->d_bits[0], &c->d_bits[1], &c->d_pending[0][0], &c->d_pending[0][1],
                     &c->d_pending[1][0], &c->d_pending[1][1], &c->d_partials2, &c->d_close_end[0], &c->d_close_end[1], &c->d_close_idx[0], &c->d_close_idx[1],
                     &c->d_partials, &c->d_aggs, &c->d_faggs, &c->d_spec, &c->d_stage_bits[0], &c->d_stage_bits[1], &c->d_stage_cb[0], &c->d_stage_cb[1], &c->d_stage_ci[0], &c->d_stage_ci[1], &c->d_stage_q[0], &c->d_stage_q[1], &c->d_stage_own, &c->d_gring, &c->d_pack, &c->d_gvtop, &c->d_seqout, &c->d_tail_st, &c->d_tail_ticket, &c->d_bits_alt[0], &c->d_bits_alt[1]};
    for (DevBuf *b : all) b->release();
    if (c->hs) (void)hipHostFree(c->hs);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_edge_stage) (void)hipHostFree(c->h_edge_stage);
    if (c->h_pk_stage) (void)hipHostFree(c->h_pk_stage);
    if (c->h_cflags) (void)hipHostFree(c->h_cflags);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->kev)
        if (e) (void)hipEventDestroy(e);
    for (int b = 0; b < NSUB; b++) {
        if (c->ev_a[b]) (void)hipEventDestroy(c->ev_a[b]);
        if (c->ev_b[b]) (void)hipEventDestroy(c->ev_b[b]);
        if (c->kev_sub[b][0]) (void)hipEventDestroy(c->kev_sub[b][0]);
        if (c->kev_sub[b][1]) (void)hipEventDestroy(c->kev_sub[b][1]);
        if (c->hs_a[b]) (void)hipHostFree(c->hs_a[b]);
    }
    if (c->st_a) (void)hipStreamDestroy(c->st_a);
    if (c->own_st) (void)hipStreamDestroy(c->own_st);
    delete c;
}

int nfc_push_device(nfc_ctx *c, const void *dev_samples, size_t n) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    if (n && !dev_samples) return fail(c, NFC_ERR_ARG, "null input");
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    return process_batch(c, dev_samples, n);
}

int nfc_submit_device(nfc_ctx *c, const void *dev_samples, size_t n) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    if (n && !dev_samples) return fail(c, NFC_ERR_ARG, "null input");
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    return submit_batch(c, dev_samples, n);
}

int nfc_wait(nfc_ctx *c) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    return wait_batch(c);
}

int nfc_submitted(nfc_ctx *c) { return c ? c->sub_count : 0; }

#define NOSUB(c)                                                                                                                     \
    do {                                                                                                                             \
        if ((c)->sub_count) return fail((c), NFC_ERR_STATE, "batches submitted with nfc_submit_device are in flight: nfc_wait first"); \
    } while (0)

int nfc_push(nfc_ctx *c, const void *host_samples, size_t n) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    NOSUB(c);
    if (n && !host_samples) return fail(c, NFC_ERR_ARG, "null input");
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    const size_t bytes = n * c->in_bytes_per_sample;
    HIPCHK(c, c->d_in.ensure(bytes + 64));
    if (bytes) HIPCHK(c, hipMemcpyAsync(c->d_in.p, host_samples, bytes, hipMemcpyHostToDevice, c->st));
    return process_batch(c, c->d_in.p, n);
}

int nfc_push_edges(nfc_ctx *c, const nfc_edge *host_edges, size_t n64) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    NOSUB(c);
    if (n64 && !host_edges) return fail(c, NFC_ERR_ARG, "null input");
    if (n64 > 0xFFFFFF00ull) return fail(c, NFC_ERR_ARG, "too many edges for one call");
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    const uint32_t n = (uint32_t)n64;
    c->have_outputs = false;
    c->pk_ready[0] = c->pk_ready[1] = false;
    c->n_edges = 0;
    for (int t = 0; t < 2; t++) c->n_sym[t] = c->n_close[t] = c->n_bits[t] = 0;
    memset(&c->stats, 0, sizeof c->stats);
    c->last_n = 0;
    launch_error() = LaunchError{};
    flush_state(c);
    // the entries and their decoder codes (edges.hip.h: edge_code) as the edge stage would have left them
    std::vector<uint16_t> code(n + 8, 0);
    const int nd = c->mx + 1;
    for (uint32_t i = 0; i < n; i++) {
        const nfc_edge &e = host_edges[i];
        if (e.v < -1 || e.v > 2 || e.t < -1 || e.t > 1 || e.d < 0) return fail(c, NFC_ERR_ARG, "edge %u out of range", i);
        const int dd = e.d < nd ? e.d : nd - 1;
        code[i] = (uint16_t)(((e.v + 1) * nd + dd) | ((e.t + 1) << 14));
    }
    c->cap_edges = n;
    for (int t = 0; t < 2; t++) c->cap_sym[t] = (t == 1 ? 2u : 1u) * n + 16;
    HIPCHK(c, c->d_eidx.ensure(((size_t)n + 1) * 8));
    HIPCHK(c, c->d_ecode.ensure(((size_t)n + 8) * 2));
    HIPCHK(c, c->d_epos.ensure(8 * 4));
    c->h_pushed.assign(host_edges, host_edges + n);
    c->edges_from_host = true;   // (the packets are labelled with the caller's indices)
    std::vector<uint64_t> idx(n);
    for (uint32_t i = 0; i < n; i++) idx[i] = host_edges[i].idx;
    if (n) {
        HIPCHK(c, hipMemcpyAsync(c->d_eidx.p, idx.data(), (size_t)n * 8, hipMemcpyHostToDevice, c->st));
        HIPCHK(c, hipMemcpyAsync(c->d_ecode.p, code.data(), (size_t)n * 2, hipMemcpyHostToDevice, c->st));
    }
    HIPCHK(c, hipMemcpyAsync(dT(c) + TOT_EDGES, &n, 4, hipMemcpyHostToDevice, c->st));
    c->tail_now = false;
    const int rc = run_decode(c);   // k_dec_spec (or k_dec_reduce -> k_dec_apply) -> k_frame_write -> k_pkt_finish (mirrors the state block)
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->st));
    BATCHCHK(c, false);
    if (spec_failed(c)) {   // a tile's assumed decoder state was wrong: the stage again, in the form that assumes nothing
        note_respeculation(c);
        const int rc2 = run_decode(c, true);
        if (rc2) return rc2;
        HIPCHK(c, hipStreamSynchronize(c->st));
        BATCHCHK(c, false);
    }
    spec_batch_done(c);
    uint32_t ns[2];
    memcpy(ns, c->hs->totals + TOT_NSYM, 8);
    c->n_edges = n;
    c->n_sym[0] = ns[0];
    c->n_sym[1] = ns[1];
    const EdgeCarry keep = c->h_ecarry;
    adopt_mirror(c);
    c->h_ecarry = keep;
    uint64_t pk[2];
    memcpy(&pk[0], c->hs->totals + TOT_PKT0, 8);
    memcpy(&pk[1], c->hs->totals + TOT_PKT1, 8);
    for (int t = 0; t < 2; t++) {
        c->n_bits[t] = (uint32_t)pk[t];
        c->n_close[t] = (uint32_t)(pk[t] >> 32);
    }
    c->pend_cur = 1 - c->pend_cur;
    c->have_outputs = true;
    return NFC_OK;
}

int nfc_sync(nfc_ctx *c) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return NFC_OK;
}

int nfc_set_stream(nfc_ctx *c, void *stream) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    NOSUB(c);
    HIPCHK(c, hipStreamSynchronize(c->st));   // nothing of this context is left on the stream it leaves
    c->st = stream ? (hipStream_t)stream : c->own_st;
    return NFC_OK;
}

int nfc_get_counts(nfc_ctx *c, nfc_counts *out) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || !out) return NFC_ERR_ARG;
    if (!c->have_outputs) return fail(c, NFC_ERR_STATE, "no completed batch");
    memset(out, 0, sizeof *out);
    out->n_samples = c->last_n;
    out->n_edges = c->n_edges;
    for (int t = 0; t < 2; t++) {
        out->n_symbols[t] = c->n_sym[t];
        int rc = build_packets(c, t);
        if (rc) return rc;
        out->n_packets[t] = c->pk[t].size();
        uint64_t nb = 0;
        for (auto &p : c->pk[t]) nb += p.n_bits;
        out->n_packet_bits[t] = nb;
    }
    return NFC_OK;
}

static int read_range(nfc_ctx *c, const void *dev, size_t total, size_t esz, size_t first, void *out, size_t cap, size_t *n_out) {
    if (!c->have_outputs) return fail(c, NFC_ERR_STATE, "no completed batch");
    size_t n = 0;
    if (first < total) n = std::min(cap, total - first);
    if (n && !out) return fail(c, NFC_ERR_ARG, "null output");
    if (n) HIPCHK(c, hipMemcpy(out, (const char *)dev + first * esz, n * esz, hipMemcpyDeviceToHost));
    if (n_out) *n_out = n;
    return NFC_OK;
}

// The device keeps an entry as (sample position, code): 6 bytes (edges.hip.h).  Both readers fetch them through a pinned
// staging area in pieces, the copy of one piece under the unpacking of the one before.
extern "C++" {
namespace {
constexpr size_t EDGE_PIECE = 1u << 20;   // entries per piece
template <class Consume>
int fetch_entries(nfc_ctx *c, size_t first, size_t n, Consume consume) {   // consume(pos, code, count, offset)
    const size_t need = 2 * EDGE_PIECE * 6 + 64;
    if (c->h_edge_stage_cap < need) {
        devbuf_allocs()++;
        if (c->h_edge_stage) (void)hipHostFree(c->h_edge_stage);
        c->h_edge_stage = nullptr;
        c->h_edge_stage_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_edge_stage, need, hipHostMallocDefault));
        c->h_edge_stage_cap = need;
    }
    auto pos_of = [&](int b) { return (uint32_t *)(c->h_edge_stage + (size_t)b * EDGE_PIECE * 6); };
    auto code_of = [&](int b) { return (uint16_t *)(c->h_edge_stage + (size_t)b * EDGE_PIECE * 6 + EDGE_PIECE * 4); };
    auto request = [&](size_t off, int b) -> hipError_t {
        const size_t cnt = std::min(EDGE_PIECE, n - off);
        hipError_t e = hipMemcpyAsync(pos_of(b), c->d_epos.as<uint32_t>() + first + off, cnt * 4, hipMemcpyDeviceToHost, c->st);
        if (e == hipSuccess) e = hipMemcpyAsync(code_of(b), c->d_ecode.as<uint16_t>() + first + off, cnt * 2, hipMemcpyDeviceToHost, c->st);
        if (e == hipSuccess) e = hipEventRecord(c->ev[6 + b], c->st);
        return e;
    };
    HIPCHK(c, request(0, 0));
    int b = 0;
    for (size_t off = 0; off < n; off += EDGE_PIECE, b ^= 1) {
        if (off + EDGE_PIECE < n) HIPCHK(c, request(off + EDGE_PIECE, b ^ 1));
        HIPCHK(c, hipEventSynchronize(c->ev[6 + b]));
        consume(pos_of(b), code_of(b), std::min(EDGE_PIECE, n - off), off);
    }
    return NFC_OK;
}
int edge_range(nfc_ctx *c, size_t first, const void *out, size_t cap, size_t *n_out, size_t &n) {
    if (!c->have_outputs) return fail(c, NFC_ERR_STATE, "no completed batch");
    n = 0;
    if (first < c->n_edges) n = std::min(cap, (size_t)c->n_edges - first);
    if (n && !out) return fail(c, NFC_ERR_ARG, "null output");
    if (n_out) *n_out = n;
    return NFC_OK;
}
}  // namespace
}  // extern "C++"

int nfc_read_edges(nfc_ctx *c, size_t first, nfc_edge *out, size_t cap, size_t *n_out) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    size_t n;
    const int rc = edge_range(c, first, out, cap, n_out, n);
    if (rc || !n) return rc;
    if (c->edges_from_host) {
        memcpy(out, c->h_pushed.data() + first, n * sizeof(nfc_edge));
        return NFC_OK;
    }
    // the second half of a record -- d, v, t -- per LUT row (edges.hip.h: edge_code), built once
    static_assert(sizeof(nfc_edge) == 16, "record layout");
    const int nd = c->mx + 1;
    if (c->edge_lut.empty()) {
        c->edge_lut.assign((size_t)4 * nd, 0);
        for (int li = 0; li < 4 * nd; li++) {
            nfc_edge e;
            memset(&e, 0, sizeof e);
            e.d = li % nd;
            e.v = (int8_t)(li / nd - 1);
            memcpy(&c->edge_lut[li], (const char *)&e + 8, 8);
        }
    }
    const uint64_t g0 = c->last_g0;
    const uint64_t *lut = c->edge_lut.data();
    const size_t rows = c->edge_lut.size();
    return fetch_entries(c, first, n, [&](const uint32_t *pos, const uint16_t *code, size_t cnt, size_t off) {
        nfc_edge *o = out + off;
        for (size_t i = 0; i < cnt; i++) {
            const uint32_t li = code[i] & 0x3FFFu;
            nfc_edge e;
            e.idx = g0 + pos[i];
            const uint64_t hi = li < rows ? lut[li] : 0;
            memcpy((char *)&e + 8, &hi, 8);
            e.t = (int8_t)((int)(code[i] >> 14) - 1);
            o[i] = e;
        }
    });
}

int nfc_read_edges_compact(nfc_ctx *c, size_t first, uint32_t *pos_out, uint16_t *code_out, size_t cap, size_t *n_out) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    size_t n;
    int rc = edge_range(c, first, pos_out, cap, n_out, n);
    if (rc || !n) return rc;
    if (!code_out) return fail(c, NFC_ERR_ARG, "null output");
    if (c->edges_from_host) return fail(c, NFC_ERR_STATE, "the entries of nfc_push_edges carry the caller's indices: use nfc_read_edges");
    {   // outputs in PINNED host memory (nfc_host_alloc_pinned): the copy engine writes them in place -- no staging, no second pass on
        // the host (a streaming caller's read-back was the longest part of its turn per piece: bench.py, end_to_end)
        unsigned int f0 = 0, f1 = 0;
        if (hipHostGetFlags(&f0, pos_out) == hipSuccess && hipHostGetFlags(&f1, code_out) == hipSuccess) {
            HIPCHK(c, hipMemcpyAsync(pos_out, c->d_epos.as<uint32_t>() + first, n * 4, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipMemcpyAsync(code_out, c->d_ecode.as<uint16_t>() + first, n * 2, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipStreamSynchronize(c->st));
            return NFC_OK;
        }
        (void)hipGetLastError();   // (not pinned: hipHostGetFlags left its complaint behind)
    }
    return fetch_entries(c, first, n, [&](const uint32_t *pos, const uint16_t *code, size_t cnt, size_t off) {
        memcpy(pos_out + off, pos, cnt * 4);
        memcpy(code_out + off, code, cnt * 2);
    });
}

int nfc_read_symbols(nfc_ctx *c, int type, size_t first, uint8_t *out, size_t cap, size_t *n_out) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || type < 0 || type > 1) return NFC_ERR_ARG;
    if (c->have_outputs && c->sym_lazy) {
        if (int rc = materialize_symbols(c)) return rc;
    }
    return read_range(c, c->d_sym[type].p, c->n_sym[type], 1, first, out, cap, n_out);
}

int nfc_read_packets(nfc_ctx *c, int type, nfc_packet *out, size_t cap, size_t *n_out) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || type < 0 || type > 1) return NFC_ERR_ARG;
    if (!c->have_outputs) return fail(c, NFC_ERR_STATE, "no completed batch");
    int rc = build_packets(c, type);
    if (rc) return rc;
    const size_t n = std::min(cap, c->pk[type].size());
    if (n && !out) return fail(c, NFC_ERR_ARG, "null output");
    if (n) memcpy(out, c->pk[type].data(), n * sizeof(nfc_packet));
    if (n_out) *n_out = n;
    return NFC_OK;
}

int nfc_read_packet_bits(nfc_ctx *c, int type, size_t first, uint8_t *out, size_t cap, size_t *n_out) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || type < 0 || type > 1) return NFC_ERR_ARG;
    if (!c->bits_packed) return read_range(c, c->d_bits[type].p, c->n_bits[type], 1, first, out, cap, n_out);
    // the multi-launch stage leaves the bits packed, 32 to a word (decode.hip.h: k_frame_write): fetch the words, hand out a byte per bit
    if (!c->have_outputs) return fail(c, NFC_ERR_STATE, "no completed batch");
    size_t n = 0;
    if (first < c->n_bits[type]) n = std::min(cap, (size_t)c->n_bits[type] - first);
    if (n && !out) return fail(c, NFC_ERR_ARG, "null output");
    if (n) {
        const size_t w0 = first >> 5, w1 = (first + n + 31) >> 5;
        // (the words through pinned staging: a pageable destination makes the runtime stage the copy itself, synchronously)
        const size_t need = (w1 - w0) * 4 + 64;
        if (c->h_pk_stage_cap < need) {
            devbuf_allocs()++;
            if (c->h_pk_stage) (void)hipHostFree(c->h_pk_stage);
            c->h_pk_stage = nullptr;
            c->h_pk_stage_cap = 0;
            const size_t cap2 = need + need / 2 + 65536;
            HIPCHK(c, hipHostMalloc((void **)&c->h_pk_stage, cap2, hipHostMallocDefault));
            c->h_pk_stage_cap = cap2;
        }
        const uint32_t *w = (const uint32_t *)c->h_pk_stage;
        HIPCHK(c, hipMemcpyAsync(c->h_pk_stage, c->d_bits[type].as<uint32_t>() + w0, (w1 - w0) * 4, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        // a byte per bit, eight at a time: byte j of (b * 0x0101.. & 0x8040..01) is nonzero exactly when bit j of b is set
        auto spread8 = [](uint32_t b) -> uint64_t {
            const uint64_t x = ((uint64_t)(b & 0xFFu) * 0x0101010101010101ull) & 0x8040201008040201ull;
            return ((x + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
        };
        size_t i = 0;
        for (; i < n && ((first + i) & 7u); i++) {   // (up to a byte boundary of the stream)
            const size_t b = first + i - (w0 << 5);
            out[i] = (uint8_t)((w[b >> 5] >> (b & 31)) & 1u);
        }
        for (; i + 8 <= n; i += 8) {
            const size_t b = first + i - (w0 << 5);
            const uint64_t v = spread8(w[b >> 5] >> (b & 31));
            memcpy(out + i, &v, 8);
        }
        for (; i < n; i++) {
            const size_t b = first + i - (w0 << 5);
            out[i] = (uint8_t)((w[b >> 5] >> (b & 31)) & 1u);
        }
    }
    if (n_out) *n_out = n;
    return NFC_OK;
}

int nfc_read_val(nfc_ctx *c, size_t first, int8_t *out, size_t cap, size_t *n_out) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    if (!c->have_outputs) return fail(c, NFC_ERR_STATE, "no completed batch");
    size_t n = 0;
    if (first < c->last_n) n = std::min(cap, (size_t)c->last_n - first);
    if (n_out) *n_out = n;
    if (!n) return NFC_OK;
    if (c->last_skip >= c->last_n) {  // batch was all fill: nothing was classified
        memset(out, 0, n);
        return NFC_OK;
    }
    const size_t w0 = first / 64, w1 = (first + n + 63) / 64;
    std::vector<uint64_t> ng(w1 - w0), ps(w1 - w0);
    HIPCHK(c, hipMemcpy(ng.data(), c->d_neg.as<uint64_t>() + w0, (w1 - w0) * 8, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ps.data(), c->d_pos.as<uint64_t>() + w0, (w1 - w0) * 8, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        const size_t m = first + i;
        const int lo = (int)((ng[m / 64 - w0] >> (m & 63)) & 1), hi = (int)((ps[m / 64 - w0] >> (m & 63)) & 1);
        out[i] = (int8_t)(lo ? -1 : (hi ? 1 : 0));
    }
    return NFC_OK;
}

static void init_carried(nfc_ctx *c) {
    memset(&c->h_carry, 0, sizeof c->h_carry);
    c->h_carry.ss_emin = 255;
    c->h_ecarry = EdgeCarry{0, 0, 1, 0};  // transition_sink.py:22-23,30
    memset(&c->h_dcarry, 0, sizeof c->h_dcarry);
    c->h_dcarry.mil_state = 0;             // stage BEGINNING, not started, prev 0 (miller.py:22,29)
    c->h_dcarry.man_state = (0 + 1) << 1;  // prev_set False, prev 0 (manchester.py:22-25)
    c->nseen = 0;
}

static int upload_carried(nfc_ctx *c) {
    push_state(c);
    return NFC_OK;
}

int nfc_set_timing(nfc_ctx *c, int level) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || level < 0 || level > 2) return NFC_ERR_ARG;
    c->timing = level;
    return NFC_OK;
}

int nfc_reset(nfc_ctx *c) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c) return NFC_ERR_ARG;
    NOSUB(c);
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    init_carried(c);
    c->have_outputs = false;
    push_state(c, 0, true);   // carried values and a zeroed window in one launch
    return NFC_OK;
}

int nfc_prime(nfc_ctx *c, uint64_t start_index, float level) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || !(level >= 0.f) || !std::isfinite(level)) return NFC_ERR_ARG;
    NOSUB(c);
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    init_carried(c);
    c->nseen = start_index;
    c->h_carry.filled = c->L;
    c->h_carry.stable = 1;
    c->h_carry.ss = (double)level * (double)c->L;   // L equal addends: exact in any order (24 + 12 bits)
    c->have_outputs = false;
    push_state(c, 0, true, level);                  // the window itself is filled on the device
    return NFC_OK;
}

static void fill_state_header(const nfc_ctx *c, nfc_state_header *h) {
    memset(h, 0, sizeof *h);
    h->n_seen = c->nseen;
    h->ss = c->h_carry.ss;
    h->last_low = -1;
    h->filled = c->h_carry.filled;
    h->stable = c->h_carry.stable;
    h->cur_state = c->h_ecarry.state;
    h->last_bit = c->h_ecarry.last_bit;
    h->dur = c->h_ecarry.dur;
    h->miller_state = c->h_dcarry.mil_state;
    h->manch_state = c->h_dcarry.man_state;
    for (int t = 0; t < 2; t++) {
        h->pkt_started[t] = c->h_dcarry.pkt_started[t];
        h->n_pending_bits[t] = c->h_dcarry.pending[t];
    }
    h->av_window = c->L;
}

// [u32 length of what follows | 12 bytes zero | nfc_state_header | ring | pending bits]: the boundary state in device
// memory, byte for byte what nfc_get_state returns, for an exchange that goes GPU to GPU (RCCL all-gather)
__global__ void k_export_state(uint8_t *dst, uint32_t len, int fits, nfc_state_header h, const float *ring, int L, const uint8_t *p0,
                               uint32_t n0, const uint8_t *p1, uint32_t n1) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (tid < 4) ((uint32_t *)dst)[tid] = tid == 0 ? len : 0u;
    if (!fits) return;
    uint8_t *b = dst + 16;
    if (tid < sizeof(nfc_state_header)) b[tid] = ((const uint8_t *)&h)[tid];
    float *rd = (float *)(b + sizeof(nfc_state_header));
    for (uint32_t i = tid; i < (uint32_t)L; i += nth) rd[i] = ring[i];
    uint8_t *pd = b + sizeof(nfc_state_header) + (size_t)L * 4;
    for (uint32_t i = tid; i < n0; i += nth) pd[i] = p0[i];
    for (uint32_t i = tid; i < n1; i += nth) pd[n0 + i] = p1[i];
}

int nfc_export_state(nfc_ctx *c, void *device_dst, size_t cap, size_t *len_out) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || !device_dst || cap < 16 || ((uintptr_t)device_dst & 15u)) return NFC_ERR_ARG;
    NOSUB(c);
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    flush_state(c);   // (a reset / prime that has not reached the device yet)
    nfc_state_header h;
    fill_state_header(c, &h);   // the carried values are host-mirrored after every push: no wait needed here
    const size_t p0 = h.n_pending_bits[0], p1 = h.n_pending_bits[1];
    const size_t len = sizeof h + (size_t)c->L * 4 + p0 + p1;
    static_assert(sizeof(nfc_state_header) % 4 == 0, "ring must stay 4-byte aligned behind the header");
    NFC_LAUNCH(k_export_state, dim3(4), dim3(256), 0, c->st, (uint8_t *)device_dst, (uint32_t)len, (int)(16 + len <= cap), h,
                       c->d_ring[c->ring_cur].as<float>(), c->L, c->d_pending[0][c->pend_cur].as<uint8_t>(), (uint32_t)p0,
                       c->d_pending[1][c->pend_cur].as<uint8_t>(), (uint32_t)p1);
    if (len_out) *len_out = len;
    return NFC_OK;
}

int nfc_get_state(nfc_ctx *c, nfc_state_header *h, float *ring, size_t ring_cap, uint8_t *pending, size_t pending_cap) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || !h) return NFC_ERR_ARG;
    NOSUB(c);
    flush_state(c);
    HIPCHK(c, hipStreamSynchronize(c->st));
    fill_state_header(c, h);
    const size_t p0 = c->h_dcarry.pending[0], p1 = c->h_dcarry.pending[1];
    if (ring && ring_cap < (size_t)c->L) return fail(c, NFC_ERR_ARG, "ring buffer too small");
    if (pending && pending_cap < p0 + p1) return fail(c, NFC_ERR_ARG, "pending-bit buffer too small");
    // through a pinned staging buffer: asynchronous copies and a single wait
    const size_t need = (size_t)c->L * 4 + p0 + p1;
    if (c->h_stage_cap < need) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_cap = need + 4096;
        HIPCHK(c, hipHostMalloc((void **)&c->h_stage, c->h_stage_cap, hipHostMallocDefault));
    }
    uint8_t *st = c->h_stage;
    if (ring || (pending && p0 + p1)) {
        // gathered on the device (ring | pending bits of type 0 | of type 1), then one copy and one wait
        HIPCHK(c, c->d_pack.ensure(need + 16));
        NFC_LAUNCH(k_pack_state, dim3(4), dim3(256), 0, c->st, c->d_pack.as<uint8_t>(), c->d_ring[c->ring_cur].as<float>(), c->L,
                           c->d_pending[0][c->pend_cur].as<uint8_t>(), (uint32_t)p0, c->d_pending[1][c->pend_cur].as<uint8_t>(),
                           (uint32_t)p1);
        HIPCHK(c, hipMemcpyAsync(st, c->d_pack.p, need, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
    }
    if (ring) memcpy(ring, st, (size_t)c->L * 4);
    if (pending && p0 + p1) memcpy(pending, st + (size_t)c->L * 4, p0 + p1);
    return NFC_OK;
}

int nfc_set_state(nfc_ctx *c, const nfc_state_header *h, const float *ring, size_t ring_len, const uint8_t *pending,
                  size_t pending_len) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || !h || !ring) return NFC_ERR_ARG;
    NOSUB(c);
    if (h->av_window != c->L || ring_len != (size_t)c->L) return fail(c, NFC_ERR_ARG, "state was taken with another av_window");
    // (_dur is 0 after the fill and 1 .. max_len after a sample: transition_sink.py:95-99, 123; the entry codes rely on it)
    if (h->dur < 0 || h->dur > c->mx || h->last_bit < -1 || h->last_bit > 1 || h->cur_state < 0 || h->cur_state > 2)
        return fail(c, NFC_ERR_ARG, "edge-timing state out of range (dur %d, last_bit %d, cur_state %d)", h->dur, h->last_bit, h->cur_state);
    const size_t p0 = h->n_pending_bits[0], p1 = h->n_pending_bits[1];
    if (p0 + p1 != pending_len || ((p0 + p1) && !pending)) return fail(c, NFC_ERR_ARG, "pending bits do not match the header");
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    HIPCHK(c, hipStreamSynchronize(c->st));
    init_carried(c);
    c->nseen = h->n_seen;
    c->h_carry.ss = h->ss;
    c->h_carry.filled = h->filled;
    c->h_carry.stable = h->stable;
    c->h_ecarry.state = h->cur_state;
    c->h_ecarry.last_bit = h->last_bit;
    c->h_ecarry.dur = h->dur;
    c->h_dcarry.mil_state = (h->miller_state >= 0 && h->miller_state < 16) ? (int32_t)c->mil_canon[h->miller_state] : h->miller_state;
    c->h_dcarry.man_state = h->manch_state;
    for (int t = 0; t < 2; t++) {
        c->h_dcarry.pkt_started[t] = h->pkt_started[t];
        c->h_dcarry.pending[t] = h->n_pending_bits[t];
        HIPCHK(c, c->d_pending[t][c->pend_cur].ensure((size_t)h->n_pending_bits[t] + 16));
    }
    if (p0) HIPCHK(c, hipMemcpy(c->d_pending[0][c->pend_cur].p, pending, p0, hipMemcpyHostToDevice));
    if (p1) HIPCHK(c, hipMemcpy(c->d_pending[1][c->pend_cur].p, pending + p0, p1, hipMemcpyHostToDevice));
    c->dirty_fill_ring = false;   // (a pending reset / prime fill is superseded by the explicit window)
    HIPCHK(c, hipMemcpy(c->d_ring[c->ring_cur].p, ring, (size_t)c->L * 4, hipMemcpyHostToDevice));
    c->have_outputs = false;
    return upload_carried(c);
}

int nfc_get_stats(nfc_ctx *c, nfc_stats *out) {
    LaunchScope launch_scope_(c ? &c->launch_err : nullptr);
    if (!c || !out) return NFC_ERR_ARG;
    *out = c->stats;
    out->redone_total = c->stats_redo_submitted;
    out->decode_respeculated = c->decode_respeculated;
    // (noted when the batch was adopted -- process_batch / wait_batch -- from the mirror that belongs to THAT batch: by now the
    // device's summary may have been rewritten by a batch submitted behind it)
    out->ring_slots_carried = c->ring_carried;
    return NFC_OK;
}

// ---- row f1: packets -> bytes -> commands (host only, protocol.h) ----
int nfc_fsm_create(nfc_fsm **out) {
    if (!out) return NFC_ERR_ARG;
    *out = new nfc_fsm();
    return NFC_OK;
}
void nfc_fsm_destroy(nfc_fsm *f) { delete f; }
int nfc_fsm_reset(nfc_fsm *f) {
    if (!f) return NFC_ERR_ARG;
    *f = nfc_fsm();
    return NFC_OK;
}
int nfc_fsm_process(nfc_fsm *f, const uint8_t *bits, size_t n_bits, int packet_type, nfc_frame *out, uint8_t *bytes_out, size_t bytes_cap,
                    uint16_t *enc_out) {
    if (!f || !out || !bytes_out || (n_bits && !bits) || (packet_type != 0 && packet_type != 1)) return NFC_ERR_ARG;
    if (bytes_cap < n_bits / 9 + 1) return NFC_ERR_ARG;
    fsm_process(*f, bits, n_bits, packet_type, out, bytes_out, enc_out);
    return NFC_OK;
}
int nfc_fsm_process_outgoing(nfc_fsm *f, const uint8_t *bits, size_t n_bits, int cmd, uint8_t *bits_out) {
    if (!f || (n_bits && (!bits || !bits_out)) || cmd < 0 || cmd >= CMD_COUNT) return NFC_ERR_ARG;
    return fsm_process_outgoing(*f, bits, n_bits, cmd, bits_out);
}
int nfc_fsm_set_keys(nfc_fsm *f, const uint8_t key_a[6], const uint8_t key_b[6]) {
    if (!f || !key_a || !key_b) return NFC_ERR_ARG;
    memcpy(f->key_a, key_a, 6);
    memcpy(f->key_b, key_b, 6);
    return NFC_OK;
}
int nfc_fsm_process_packets(nfc_fsm *f, const nfc_packet *packets, size_t n_packets, const uint8_t *bits0, const uint8_t *bits1,
                            nfc_frame *frames_out, uint8_t *bytes_out, size_t bytes_cap, size_t *bytes_used, uint16_t *enc_out) {
    if (!f || (n_packets && (!packets || !frames_out || !bytes_out))) return NFC_ERR_ARG;
    size_t used = 0;
    for (size_t i = 0; i < n_packets; i++) {
        const nfc_packet &p = packets[i];
        if (p.type != 0 && p.type != 1) return NFC_ERR_ARG;
        const uint8_t *bits = p.type ? bits1 : bits0;
        if (p.n_bits && !bits) return NFC_ERR_ARG;
        if (used + p.n_bits / 9 + 1 > bytes_cap) return NFC_ERR_ARG;
        fsm_process(*f, bits ? bits + p.bit_off : nullptr, p.n_bits, p.type, &frames_out[i], bytes_out + used, enc_out ? enc_out + used : nullptr);
        frames_out[i].byte_off = (uint32_t)used;
        used += std::max<size_t>(frames_out[i].n_bytes, frames_out[i].n_enc);
    }
    if (bytes_used) *bytes_used = used;
    return NFC_OK;
}
int nfc_command_count(void) { return CMD_COUNT; }
int nfc_command_get(int cmd, nfc_command_info *out) {
    if (cmd < 0 || cmd >= CMD_COUNT || !out) return NFC_ERR_ARG;
    const CommandDef &c = COMMANDS[cmd];
    memset(out, 0, sizeof *out);
    strncpy(out->name, c.name, sizeof out->name - 1);
    out->stage = c.stage;
    out->type = c.type;
    out->crc = c.crc;
    out->n_header = c.n_header;
    out->n_extra = c.n_extra;
    out->xor_check = c.xor_check;
    out->header[0] = c.header[0];
    out->header[1] = c.header[1];
    return NFC_OK;
}
int nfc_crc_a(const uint8_t *data, size_t n, uint8_t out[2]) {
    if ((n && !data) || !out) return NFC_ERR_ARG;
    const uint16_t c = crc_a(data, n);
    out[0] = (uint8_t)(c & 0xFF);
    out[1] = (uint8_t)(c >> 8);
    return NFC_OK;
}

// ---- row f4: transmit side ---------------------------------------------------------------------------------------
int nfc_tx_encode(int encoding, const uint8_t *bits, size_t n_bits, nfc_tx_run *out, size_t cap, size_t *n_out) {
    if ((!bits && n_bits) || !n_out) return fail(nullptr, NFC_ERR_ARG, "nfc_tx_encode: null argument");
    for (size_t i = 0; i < n_bits; i++)
        if (bits[i] > 1) return fail(nullptr, NFC_ERR_ARG, "nfc_tx_encode: bit %zu is %u", i, (unsigned)bits[i]);
    std::vector<nfc_tx_run> v;
    v.reserve(3 * n_bits + 8);
    switch (encoding) {
    case NFC_TX_SAME: tx_encode_same(bits, n_bits, v); break;
    case NFC_TX_MANCHESTER: tx_encode_manchester(bits, n_bits, v); break;
    case NFC_TX_MILLER: tx_encode_miller(bits, n_bits, v); break;
    default: return fail(nullptr, NFC_ERR_ARG, "nfc_tx_encode: unknown encoding %d", encoding);
    }
    *n_out = v.size();
    if (v.size() > cap) return fail(nullptr, NFC_ERR_ARG, "nfc_tx_encode: %zu runs do not fit %zu", v.size(), cap);
    if (out) memcpy(out, v.data(), v.size() * sizeof(nfc_tx_run));
    return NFC_OK;
}

namespace {
// binary_src.py:83 -- dur = int(dur * mult) with mult = samp_rate / 1e6, in double; the marker level 2 produces nothing
int tx_run_samples(const nfc_tx_run &r, double mult, uint64_t *n) {
    *n = 0;
    if (r.level == 2) return NFC_OK;
    if (r.level != 0 && r.level != 1) return fail(nullptr, NFC_ERR_ARG, "tx run level %d", r.level);
    const double d = r.dur_us * mult;
    if (!(d >= 0) || d > 1e15) return fail(nullptr, NFC_ERR_ARG, "tx run duration %g us", r.dur_us);
    *n = (uint64_t)d;   // truncation, as int() does
    return NFC_OK;
}
}  // namespace

int nfc_tx_sample_count(const nfc_tx_run *runs, size_t n_runs, double samp_rate, uint64_t *n_samples) {
    if ((!runs && n_runs) || !n_samples || !(samp_rate > 0)) return fail(nullptr, NFC_ERR_ARG, "nfc_tx_sample_count: bad argument");
    const double mult = samp_rate / 1e6;   // binary_src.py:38
    uint64_t tot = 0;
    for (size_t i = 0; i < n_runs; i++) {
        uint64_t k;
        const int r = tx_run_samples(runs[i], mult, &k);
        if (r) return r;
        tot += k;
    }
    *n_samples = tot;
    return NFC_OK;
}

int nfc_tx_render_device(int device, const nfc_tx_run *runs, size_t n_runs, double samp_rate, int carrier, double freq, float amp,
                         uint64_t first_index, void *dev_out, size_t cap_samples, size_t *n_samples, float *kernel_ms) {
    if ((!runs && n_runs) || !n_samples || !(samp_rate > 0) || (!dev_out && cap_samples)) return fail(nullptr, NFC_ERR_ARG, "nfc_tx_render_device: bad argument");
    if (n_runs > 0xFFFFFFF0ull) return fail(nullptr, NFC_ERR_ARG, "nfc_tx_render_device: too many runs");
    if (((uintptr_t)dev_out & 31u) != 0) return fail(nullptr, NFC_ERR_ARG, "nfc_tx_render_device: output must be 32-byte aligned");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, NFC_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    const double mult = samp_rate / 1e6;
    std::vector<uint64_t> ends(n_runs);
    std::vector<int8_t> levels(n_runs);
    uint64_t tot = 0;
    for (size_t i = 0; i < n_runs; i++) {
        uint64_t k;
        const int r = tx_run_samples(runs[i], mult, &k);
        if (r) return r;
        tot += k;
        ends[i] = tot;
        levels[i] = (int8_t)runs[i].level;
    }
    *n_samples = (size_t)tot;
    if (tot > cap_samples) return fail(nullptr, NFC_ERR_ARG, "nfc_tx_render_device: %llu samples do not fit %zu", (unsigned long long)tot, cap_samples);
    if (kernel_ms) *kernel_ms = 0.f;
    if (tot == 0) return NFC_OK;
    TxArgs A;
    memset(&A, 0, sizeof A);
    void *d_ends = nullptr, *d_levels = nullptr, *d_first = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // per tile of the output, the run its first sample falls in (one walk over the runs; the last entry closes the last tile)
    const size_t n_tiles = (size_t)((tot + TX_TILE - 1) / TX_TILE);
    std::vector<uint32_t> first(n_tiles + 1);
    {
        size_t r = 0;
        for (size_t t = 0; t < n_tiles; t++) {
            const uint64_t s = (uint64_t)t * TX_TILE;
            while (ends[r] <= s) r++;
            first[t] = (uint32_t)r;
        }
        while (ends[r] <= tot - 1) r++;
        first[n_tiles] = (uint32_t)r;
    }
    int rc = NFC_OK;
    auto bad = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return false;
        rc = fail(nullptr, NFC_ERR_DEVICE, "nfc_tx_render_device: %s: %s", what, hipGetErrorString(e));
        return true;
    };
    do {
        if (bad(hipMalloc(&d_ends, n_runs * 8), "hipMalloc")) break;
        if (bad(hipMalloc(&d_levels, n_runs), "hipMalloc")) break;
        if (bad(hipMemcpy(d_ends, ends.data(), n_runs * 8, hipMemcpyHostToDevice), "upload")) break;
        if (bad(hipMemcpy(d_levels, levels.data(), n_runs, hipMemcpyHostToDevice), "upload")) break;
        if (bad(hipMalloc(&d_first, first.size() * 4), "hipMalloc")) break;
        if (bad(hipMemcpy(d_first, first.data(), first.size() * 4, hipMemcpyHostToDevice), "upload")) break;
        A.tile_first = (const uint32_t *)d_first;
        A.ends = (const uint64_t *)d_ends;
        A.levels = (const int8_t *)d_levels;
        A.n_runs = (uint32_t)n_runs;
        A.n_samples = tot;
        A.first_index = first_index;
        A.carrier = carrier ? 1 : 0;
        const double turns = freq / samp_rate, fr = turns - std::floor(turns);
        A.phase_inc = (uint64_t)(fr * 18446744073709551616.0);   // floor(frac(f / fs) * 2^64)
        A.amp = amp;
        A.out = (float2 *)dev_out;
        const unsigned blocks = (unsigned)n_tiles;
        if (kernel_ms) {
            if (bad(hipEventCreate(&e0), "event") || bad(hipEventCreate(&e1), "event")) break;
            NFC_LAUNCH_EXT(k_tx_render, dim3(blocks), dim3(TX_BLOCK), 0, nullptr, e0, e1, 0, A);
        } else {
            NFC_LAUNCH(k_tx_render, dim3(blocks), dim3(TX_BLOCK), 0, nullptr, A);
        }
        if (bad(hipGetLastError(), "launch")) break;
        if (bad(hipDeviceSynchronize(), "kernel")) break;
        if (kernel_ms && bad(hipEventElapsedTime(kernel_ms, e0, e1), "event")) break;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_ends) (void)hipFree(d_ends);
    if (d_levels) (void)hipFree(d_levels);
    if (d_first) (void)hipFree(d_first);
    return rc;
}

int nfc_device_alloc(int device, size_t bytes, void **out) {
    if (!out) return NFC_ERR_ARG;
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, NFC_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(nullptr, NFC_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return NFC_OK;
}

int nfc_device_free(int device, void *p) {
    if (hipSetDevice(device) != hipSuccess) return NFC_ERR_DEVICE;
    return hipFree(p) == hipSuccess ? NFC_OK : NFC_ERR_DEVICE;
}

int nfc_device_upload(int device, void *dst, const void *src, size_t bytes) {
    if (hipSetDevice(device) != hipSuccess) return NFC_ERR_DEVICE;
    return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? NFC_OK : NFC_ERR_DEVICE;
}

int nfc_device_download(int device, void *dst, const void *src, size_t bytes) {
    if (hipSetDevice(device) != hipSuccess) return NFC_ERR_DEVICE;
    return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? NFC_OK : NFC_ERR_DEVICE;
}

int nfc_stream_create(int device, void **stream_out) {
    if (!stream_out) return NFC_ERR_ARG;
    *stream_out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, NFC_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return fail(nullptr, NFC_ERR_DEVICE, "hipStreamCreate failed");
    *stream_out = (void *)st;
    return NFC_OK;
}

int nfc_stream_destroy(int device, void *stream) {
    if (hipSetDevice(device) != hipSuccess) return NFC_ERR_DEVICE;
    return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? NFC_OK : NFC_ERR_DEVICE;
}

int nfc_stream_sync(int device, void *stream) {
    if (hipSetDevice(device) != hipSuccess) return NFC_ERR_DEVICE;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? NFC_OK : NFC_ERR_DEVICE;
}

int nfc_device_download_async(int device, void *dst, const void *src, size_t bytes, void *stream) {
    if (hipSetDevice(device) != hipSuccess) return NFC_ERR_DEVICE;
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess ? NFC_OK : NFC_ERR_DEVICE;
}

int nfc_device_fill(int device, void *dst, int byte_value, size_t bytes) {
    if (hipSetDevice(device) != hipSuccess) return NFC_ERR_DEVICE;
    return hipMemset(dst, byte_value, bytes) == hipSuccess ? NFC_OK : NFC_ERR_DEVICE;
}

int nfc_host_alloc_pinned(size_t bytes, void **out) {
    if (!out) return NFC_ERR_ARG;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault) == hipSuccess ? NFC_OK : NFC_ERR_NOMEM;
}

int nfc_host_free_pinned(void *p) { return hipHostFree(p) == hipSuccess ? NFC_OK : NFC_ERR_DEVICE; }

float nfc_host_i16_to_float(int16_t pcm, float i16_scale) { return i16_to_float((int)pcm, i16_scale > 0.f ? i16_scale : -1.0f); }

int nfc_host_decode_steps(int type, const int8_t *cur, const double *dur_us, size_t n, int32_t *state, uint8_t *sym_out, size_t cap,
                          size_t *n_out) {
    if (!cur || !dur_us || !state || !n_out || type < 0 || type > 1) return NFC_ERR_ARG;
    // the packed states of decoder_tables.h; a fresh decoder: Miller stage BEGINNING, not started, prev 0 (miller.py:22,29) /
    // Manchester prev_set False, prev 0 (manchester.py:22-25).  The caller's 0 stands for "fresh" in both.
    int st = *state ? (*state - 1) : (type ? 0 : ((0 + 1) << 1));
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        const Step r = type ? miller_step(st, cur[i], dur_us[i]) : manch_step(st, cur[i], dur_us[i]);
        st = r.next;
        for (int j = 0; j < r.nout; j++) {
            if (k < cap && sym_out) sym_out[k] = (uint8_t)r.out[j];
            k++;
        }
    }
    *state = st + 1;
    *n_out = k;
    return (k <= cap) ? NFC_OK : NFC_ERR_ARG;
}

int nfc_host_decode_lut(const nfc_params *p, int type, const int8_t *cur, const int32_t *d, size_t n, uint8_t *sym_out,
                        size_t cap, size_t *n_out) {
    if (!p || !cur || !d || !n_out || type < 0 || type > 2) return NFC_ERR_ARG;
    if (!(p->samp_rate > 0) || p->max_len < 1) return NFC_ERR_ARG;
    const DecoderTables t = build_tables(p->samp_rate, p->max_len);
    const int nd = p->max_len + 1;
    int state = type ? 0 : ((0 + 1) << 1);
    MillerQuotient q;
    if (type == 2) {   // the Miller decoder's quotient machine, as the speculative decode kernel walks it (classes, 8 per row)
        q = miller_quotient(t);
        if (!q.ok) return NFC_ERR_INTERNAL;
        state = q.q_of[0];
    }
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (cur[i] < -1 || cur[i] > 2 || d[i] < 0 || d[i] > p->max_len) return NFC_ERR_ARG;
        const int li = (cur[i] + 1) * nd + d[i];
        uint8_t w;
        if (type == 2) {
            const uint16_t e = q.step[(size_t)li * 8 + state];
            if ((int)((q.map[li] >> (8 * state)) & 0xFF) != (e & 15)) return NFC_ERR_INTERNAL;   // (the two tables of the kernel agree)
            w = (uint8_t)(e >> 8);
            state = e & 15;
        } else if (type) {
            w = t.miller_out[(size_t)li * kMillerStates + state];
            state = (int)((t.miller_map[li] >> (4 * state)) & 15u);
        } else {
            w = t.manch_out[(size_t)li * kManchStates + state];
            state = (int)((t.manch_map[li] >> (4 * state)) & 15u);
        }
        const int no = w & 3;
        if (no >= 1) { if (k < cap) sym_out[k] = (w >> 2) & 7; k++; }
        if (no >= 2) { if (k < cap) sym_out[k] = (w >> 5) & 7; k++; }
    }
    *n_out = k;
    return NFC_OK;
}

int nfc_host_miller_classes(const nfc_params *p, uint8_t class_of[16], uint8_t canonical[16], int *n_classes) {
    if (!p || !class_of || !canonical || !n_classes || !(p->samp_rate > 0) || p->max_len < 1) return NFC_ERR_ARG;
    const MillerQuotient q = miller_quotient(build_tables(p->samp_rate, p->max_len));
    if (!q.ok) return NFC_ERR_INTERNAL;
    memcpy(class_of, q.q_of, 16);
    memcpy(canonical, q.canon, 16);
    *n_classes = q.classes;
    return NFC_OK;
}

}  // extern "C"

#ifdef NFC_TAIL_PROF
extern "C" int nfc_debug_tail_prof(unsigned long long *out, int reset) {
    const size_t bytes = sizeof(unsigned long long) * 4 * nfc::TP_WGS * 8;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nfc::g_tail_prof), bytes) != hipSuccess) return -1;
    if (reset) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(nfc::g_tail_prof)) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#ifdef NFC_TAIL_PROF
extern "C" int nfc_debug_lb_tries(int reset) {
    uint32_t v = 0, z = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(nfc::g_lb_tries), 4) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(nfc::g_lb_tries), &z, 4) != hipSuccess) return -1;
    return (int)v;
}
#endif
#ifdef NFC_GEN_PROF
extern "C" int nfc_debug_gen_prof(unsigned long long *out, int reset) {
    const size_t bytes = sizeof(unsigned long long) * 8192 * 8;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nfc::g_gen_prof), bytes) != hipSuccess) return -1;
    {
        unsigned long long ri[2] = {0, 0};
        if (hipMemcpyFromSymbol(ri, HIP_SYMBOL(nfc::g_row_iters), sizeof ri) == hipSuccess)
            fprintf(stderr, "[nfc] row_exact: %llu rows, %.2f iterations each\n", ri[0], ri[0] ? (double)ri[1] / (double)ri[0] : 0.0);
        void *q = nullptr;
        if (reset && hipGetSymbolAddress(&q, HIP_SYMBOL(nfc::g_row_iters)) == hipSuccess) (void)hipMemset(q, 0, sizeof ri);
    }
    if (reset) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(nfc::g_gen_prof)) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) return -1;
    }
    return 0;
}
#endif
