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
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
#include "threshold_lean.hip.h"
#include "tx.hip.h"

using namespace nfc;

namespace {

std::string g_create_error;

// The first certification of a batch and the edge stage's reduce pass both depend on k_threshold only, so they share
// a launch: the first cert_blocks workgroups certify (the last of them resolves the end-of-batch state), the others
// reduce their tile of the planes to its aggregate (edges.hip.h).
struct CertLaunch {
    ThrArgs A;
    uint8_t *cert;
    float *ring_next;
    Carry *carry;
    CertSummary *sum;
    uint32_t blocks;
};
__global__ __launch_bounds__(256) void k_certify_and_reduce(CertLaunch C, EdgeArgs E, size_t nwords, EdgeAgg *partials) {
    if (blockIdx.x < C.blocks) {
        certify_block(C.A, C.cert, nullptr, C.ring_next, C.carry, C.sum, blockIdx.x, C.blocks);
        return;
    }
    edge_reduce_block(E, nwords, blockIdx.x - C.blocks, partials);
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, bool keep = false, hipStream_t st = nullptr) {
        if (bytes <= cap) return hipSuccess;
        size_t ncap = std::max(bytes, cap + cap / 2);
        ncap = (ncap + 255) & ~(size_t)255;
        void *np = nullptr;
        hipError_t e = hipMalloc(&np, ncap);
        if (e != hipSuccess) return e;
        if (keep && p && cap) {
            e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { (void)hipFree(np); return e; }
        }
        if (p) (void)hipFree(p);
        p = np;
        cap = ncap;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const { return (T *)p; }
};

}  // namespace

// totals layout (device scalars inside DevState)
enum : int {
    TOT_RUNS = 0,       // u32
    TOT_EDGES = 8,      // u32
    TOT_DECMAP = 16,    // DecMaps (24 bytes)
    TOT_PKT0 = 72,      // PktCnt: per type, bits | closes << 32
    TOT_PKT1 = 80,
    TOT_LAST2 = 88,     // Last2 (8 bytes)
    TOT_NSYM = 96,      // u32[2]: symbols per packet type
    TOT_CERT = 104,     // CertSummary (16 bytes)
    TOT_FRAME = 128,    // FrameAgg (32 bytes): symbol counts, framing maps, bit / close counts of the batch
    TOT_BYTES = 160
};

// Everything the host mirrors after a batch, in one block so that one copy fetches it.
struct DevState {
    Carry carry;
    EdgeCarry ecarry;
    DecCarry dcarry;
    uint8_t totals[TOT_BYTES];
    uint32_t seq[4];   // seq[0]: the batch the block belongs to, stamped by the batch's first kernel (k_fill): a mirror that
                       // does not carry the current number was not written by this batch's kernels
};

constexpr int NRING = 4;   // window buffers: the carried one + one per batch that may be in flight (they rotate)
constexpr int NSUB = 3;    // batches that may be submitted and not yet waited for

struct nfc_ctx {
    nfc_params P;
    int L, mx, C, Lpad, wpb, twords;
    double factor;
    double hi_plus, lo_a, lo_b, hi_a, hi_b;
    int bands_ok, fast_ok, nfold, rows_per_step, C_min, wave_slots, lds_per_slot;
    int lean = 1, lean_k = 0, lean_rounds = 0, lean_slots = 0;   // pass 0 by k_threshold_lean (NFC_LEAN=0 turns it off), steps per superstep (NFC_LEAN_K)
    float lean_gfac = 1.3f, lean_gmin = 9.765625e-4f;   // drift allowance of the next superstep: max(gfac * B, gmin * ss)
    int gring = 0;   // this batch: the ring of a chunk in global memory instead of LDS
    int gring_ok = 0, gring_force = 0, wave_slots_g = 0;   // long windows qualify (NFC_RING=lds|global overrides the choice)
    DevBuf d_gring;
    uint32_t own_prefix_max = OWN_PREFIX_MAX_TILES;   // tile counts up to this need no prefix launches (NFC_OWN_PREFIX_MAX overrides)
    int use_small = 1;   // short batches take the one-launch edge / decode / framing kernel (NFC_NO_SMALL=1 turns it off)
    uint64_t selmask;
    float eps;
    float i16_scale;
    size_t in_bytes_per_sample;
    hipStream_t st = nullptr;
    hipStream_t own_st = nullptr;   // the stream the context created (st may be the caller's: nfc_set_stream)
    hipEvent_t ev[8] = {};
    bool state_dirty = false, dirty_fill_ring = false;   // host-side carried values not yet on the device (push_state)
    float dirty_fill = 0.f;
    Carry dirty_carry;
    EdgeCarry dirty_ecarry;
    DecCarry dirty_dcarry;
    bool cert_pending = false;   // the first certification waits to share a launch with the edge stage (k_certify_and_count)
    CertLaunch cert;
    uint32_t batch_seq = 0;   // stamped into the state block by every batch's first kernel, checked in the mirror
    int timing = 0;   // 0: no events, 1: the threshold kernels' own start / stop events, 2: + batch total and stages as stream markers (nfc_set_timing)
    hipEvent_t kev[2 * 6] = {};  // start/stop pairs around the first k_threshold launches of a batch
    int n_kev = 0;
    std::string err;

    // tables
    DevBuf d_mil_map, d_man_map, d_mil_out, d_man_out;
    DecTables T;

    // carried state
    DevBuf d_state, d_ring[NRING];   // the window: the carried one, the one the batch in work writes, and -- with batches submitted
                                     // ahead (nfc_submit_device) -- the ones THOSE write; they rotate
    DevState *hs = nullptr;        // pinned host mirror of d_state
    void *hs_dev = nullptr;        // the same memory as the device addresses it (kernels may fill the mirror themselves)
    uint8_t *h_stage = nullptr;    // pinned staging for nfc_get_state
    size_t h_stage_cap = 0;
    uint8_t *h_edge_stage = nullptr;   // pinned staging for nfc_read_edges / nfc_read_edges_compact (two pieces)
    size_t h_edge_stage_cap = 0;
    std::vector<uint64_t> edge_lut;    // per LUT row: the (d, v) half of an nfc_edge record
    uint8_t *h_cflags = nullptr;   // pinned mirror of the per-chunk flag sections
    size_t h_cflags_cap = 0;
    int ring_cur = 0;
    // ---- a batch submitted ahead (nfc_submit_device / nfc_wait): its threshold stage runs on st_a beside the edge and
    // decode stages of the batch before it on st
    hipStream_t st_a = nullptr;
    DevBuf d_neg_alt[NSUB - 1], d_pos_alt[NSUB - 1];   // planes of the batches whose edge stage is not enqueued yet (a set becomes
                                                       // d_neg / d_pos then, and the retired set takes its place in the pool)
    uint32_t alt_free = (1u << (NSUB - 1)) - 1u;       // which of them are free
    DevState *hs_a[NSUB] = {};                 // pinned snapshots of the state block taken right after a submitted batch's certification
    hipEvent_t ev_a[NSUB] = {}, ev_b[NSUB] = {};   // its threshold stage / its last stage done
    hipEvent_t kev_sub[NSUB][2] = {};          // start / stop of its threshold kernel (nfc_set_timing >= 1)
    struct Submitted {
        const void *d_in = nullptr;
        uint32_t n = 0, seq = 0, nch = 0, chunk = 0;
        uint64_t g0 = 0;
        int slot = 0, planes = -1, ring_in = 0, timing = 0;   // (timing: nfc_set_timing's level when the batch was submitted)
        bool fast = false, b_enqueued = false, timed = false;
    } sub[NSUB];
    int sub_count = 0;             // batches submitted and not yet waited for (sub[0] the oldest)
    uint32_t slot_next = 0;
    bool low_valid = false;        // Carry.low_nl / low_kl on the device describe the end of the last completed batch
    size_t lean_lds_per_cu = 0;
    uint32_t stamp_b = 0;          // the batch number the decode stage's last launch writes into the mirror (seq[1])
    bool in_wait = false;
    uint32_t dbg_fast_waits = 0;
    uint32_t stats_redo_submitted = 0;   // submitted batches that had to go through the synchronous path after all
    Carry h_carry;
    EdgeCarry h_ecarry;
    DecCarry h_dcarry;
    uint64_t nseen = 0;

    // batch buffers
    DevBuf d_certinfo;
    DevBuf d_in, d_neg, d_pos, d_ringout[2], d_touched[2], d_info[2], d_ringin, d_meta, d_ver, d_cflags, d_list;
    DevBuf d_ecode, d_epos, d_eidx;   // per entry: code, batch-local sample position (edges.hip.h); caller's own indices (nfc_push_edges)
    bool edges_from_host = false;
    std::vector<nfc_edge> h_pushed;   // the entries of the last nfc_push_edges, as nfc_read_edges hands them back
    DevBuf d_states, d_sym[2], d_bits[2], d_pending[2][2], d_close_end[2],
        d_close_idx[2];
    DevBuf d_partials, d_partials2, d_aggs, d_faggs;  // scan scratch
    DevBuf d_pack;                           // nfc_get_state staging
    DevBuf d_gvtop;                          // per chunk: bound of the ring values (guard of the fp64 sums)
    DevBuf d_seqout;                         // sequential kernel: edge-timing state after its last sample
    uint32_t cap_edges = 0, cap_sym[2] = {0, 0};   // capacity estimates of the edge / symbol buffers
    uint64_t cap_edges_floor = 0, cap_sym_floor[2] = {0, 0};   // raised when an estimate proved too small for this batch
    double edge_rate = 0.125;                      // entries per sample seen lately (peak-hold with slow decay)
    double sym_rate[2] = {1.0, 2.0};               // symbols per entry, per type (start at the upper bounds)
    int pend_cur = 0;                               // which half of d_pending holds the open packets' bits
    std::vector<uint8_t> h_ver;
    std::vector<uint32_t> h_list;

    // last batch
    const void *last_in = nullptr;
    uint32_t last_n = 0, last_skip = 0;
    uint64_t last_g0 = 0;
    uint32_t n_edges = 0;
    uint32_t n_sym[2] = {0, 0}, n_close[2] = {0, 0}, n_bits[2] = {0, 0};
    bool have_outputs = false;
    nfc_stats stats;
    // lazily built packet lists
    std::vector<nfc_packet> pk[2];
    bool pk_ready[2] = {false, false};
};

namespace {

int fail(nfc_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

inline Carry *dC(nfc_ctx *c) { return &((DevState *)c->d_state.p)->carry; }
inline EdgeCarry *dE(nfc_ctx *c) { return &((DevState *)c->d_state.p)->ecarry; }
inline DecCarry *dD(nfc_ctx *c) { return &((DevState *)c->d_state.p)->dcarry; }
inline uint8_t *dT(nfc_ctx *c) { return ((DevState *)c->d_state.p)->totals; }

// one copy brings the whole mirrored block to pinned host memory
inline hipError_t mirror_async(nfc_ctx *c) {
    return hipMemcpyAsync(c->hs, c->d_state.p, sizeof(DevState), hipMemcpyDeviceToHost, c->st);
}
inline void adopt_mirror(nfc_ctx *c) {
    c->h_carry = c->hs->carry;
    carry_apply_fin(c->h_carry);
    c->h_ecarry = c->hs->ecarry;
    c->h_dcarry = c->hs->dcarry;
}
// carried state set from the host without a copy engine round trip: the values travel as kernel arguments
__global__ void k_set_state(DevState *d, Carry a, EdgeCarry b, DecCarry e, int zero_totals, float *fill_ring, int ring_len, float fill) {
    if (fill_ring)
        for (int i = threadIdx.x; i < ring_len; i += blockDim.x) fill_ring[i] = fill;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    d->carry = a;
    d->ecarry = b;
    d->dcarry = e;
    if (zero_totals)
        for (int i = 0; i < TOT_BYTES; i++) d->totals[i] = 0;
}
inline void launch_set_state(nfc_ctx *c, int zero_totals, bool fill_ring, float fill) {
    NFC_LAUNCH(k_set_state, dim3(1), dim3(256), 0, c->st, (DevState *)c->d_state.p, c->h_carry, c->h_ecarry, c->h_dcarry,
                       zero_totals, fill_ring ? c->d_ring[c->ring_cur].as<float>() : (float *)nullptr, fill_ring && fill != 0.f ? c->L : c->Lpad,
                       fill);
}
// The host values become the device state lazily: with the next batch's first launch (k_fill takes them along), or
// right away when something reads the device state first (flush_state).
inline void push_state(nfc_ctx *c, int zero_totals = 0, bool fill_ring = false, float fill = 0.f) {
    c->low_valid = false;   // (the LOW bookkeeping a submitted batch would read on the device is not part of what the host sets)
    if (zero_totals) {
        launch_set_state(c, zero_totals, fill_ring, fill);
        c->state_dirty = false;
        return;
    }
    c->state_dirty = true;
    c->dirty_fill_ring = c->dirty_fill_ring || fill_ring;
    if (fill_ring) c->dirty_fill = fill;
    // the values as of NOW (process_batch advances the host mirrors before the batch's first launch)
    c->dirty_carry = c->h_carry;
    c->dirty_ecarry = c->h_ecarry;
    c->dirty_dcarry = c->h_dcarry;
}
inline void flush_state(nfc_ctx *c) {
    if (!c->state_dirty) return;
    NFC_LAUNCH(k_set_state, dim3(1), dim3(256), 0, c->st, (DevState *)c->d_state.p, c->dirty_carry, c->dirty_ecarry, c->dirty_dcarry, 0,
                       c->dirty_fill_ring ? c->d_ring[c->ring_cur].as<float>() : (float *)nullptr,
                       c->dirty_fill_ring && c->dirty_fill != 0.f ? c->L : c->Lpad, c->dirty_fill);
    c->state_dirty = c->dirty_fill_ring = false;
}
// ring | pending bits (type 0, type 1) as one contiguous byte vector (nfc_get_state)
__global__ void k_pack_state(uint8_t *dst, const float *ring, int L, const uint8_t *p0, uint32_t n0, const uint8_t *p1, uint32_t n1) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    float *rd = (float *)dst;
    for (uint32_t i = tid; i < (uint32_t)L; i += nth) rd[i] = ring[i];
    uint8_t *pd = dst + (size_t)L * 4;
    for (uint32_t i = tid; i < n0; i += nth) pd[i] = p0[i];
    for (uint32_t i = tid; i < n1; i += nth) pd[n0 + i] = p1[i];
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return fail((c), NFC_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

// A kernel launch of this batch was rejected by the runtime (launch_check.h), or -- with_mirror -- the host's mirror of the
// state block was not written by this batch's kernels: nothing the host would read next can be trusted.
int batch_ok(nfc_ctx *c, bool with_mirror) {
    LaunchError &le = launch_error();
    if (le.err != hipSuccess) {
        const LaunchError e = le;
        le = LaunchError{};
        return fail(c, NFC_ERR_DEVICE, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e.err), e.file, e.line);
    }
    if (with_mirror && c->hs->seq[0] != c->batch_seq)
        return fail(c, NFC_ERR_DEVICE, "state mirror is stale (batch %u, mirror %u): a kernel of this batch did not run", c->batch_seq, c->hs->seq[0]);
    return NFC_OK;
}
#define BATCHCHK(c, with_mirror)                                   \
    do {                                                           \
        if (int rc__ = batch_ok((c), (with_mirror))) return rc__;  \
    } while (0)

// Timed launches (nfc_set_timing >= 1) hand the kernel its own start / stop events (hipExtLaunchKernelGGL): the
// events take the kernel's begin and end, not the position of a marker in the stream, so they neither measure nor add
// inter-launch gaps.
template <int KIND>
void launch_threshold(nfc_ctx *c, const ThrArgs &A, uint32_t nwork, hipEvent_t e0, hipEvent_t e1) {
    const uint32_t wpb = c->gring ? 4u : (uint32_t)c->wpb;
    const uint32_t blocks = (nwork + wpb - 1) / wpb;
    const size_t lds = c->gring ? 0 : (size_t)wpb * c->Lpad * c->lds_per_slot;
    if (e0) {
        if (c->gring) NFC_LAUNCH_EXT((k_threshold<KIND, 4, true>), dim3(blocks), dim3(64 * wpb), lds, c->st, e0, e1, 0, A);
        else NFC_LAUNCH_EXT((k_threshold<KIND, 4, false>), dim3(blocks), dim3(64 * wpb), lds, c->st, e0, e1, 0, A);
        return;
    }
    if (c->gring) NFC_LAUNCH((k_threshold<KIND, 4, true>), dim3(blocks), dim3(64 * wpb), lds, c->st, A);
    else NFC_LAUNCH((k_threshold<KIND, 4, false>), dim3(blocks), dim3(64 * wpb), lds, c->st, A);
}
// Pass 0 with the LDS ring: the lean optimistic kernel (threshold_lean.hip.h); chunks it gives up on are re-run by k_threshold.
template <int KIND>
void launch_lean(nfc_ctx *c, const ThrArgs &A, uint32_t nwork, hipEvent_t e0, hipEvent_t e1) {
    const uint32_t wpb = (uint32_t)c->wpb;
    const uint32_t blocks = (nwork + wpb - 1) / wpb;
    // (NFC_DEBUG_BAD_LAUNCH: a dynamic-LDS request the runtime must reject -- the test of the launch checks)
    const size_t lds = (size_t)wpb * c->Lpad * c->lds_per_slot + (getenv("NFC_DEBUG_BAD_LAUNCH") ? (size_t)1 << 20 : 0);
    auto go = [&](auto kern) {
        if (e0) NFC_LAUNCH_EXT(kern, dim3(blocks), dim3(64 * wpb), lds, c->st, e0, e1, 0, A);
        else NFC_LAUNCH(kern, dim3(blocks), dim3(64 * wpb), lds, c->st, A);
    };
    const bool b16 = (1 << c->nfold) == 16;
    switch (c->lean_k) {
    case 2: if (b16) go(k_threshold_lean<KIND, 2, true>); else go(k_threshold_lean<KIND, 2, false>); break;
    default: if (b16) go(k_threshold_lean<KIND, 4, true>); else go(k_threshold_lean<KIND, 4, false>); break;
    }
}
void launch_threshold_kind(nfc_ctx *c, const ThrArgs &A, uint32_t nwork, bool lean = false, hipEvent_t *own_events = nullptr) {
    const bool timed = !own_events && c->timing >= 1 && c->n_kev < 6;
    hipEvent_t e0 = timed ? c->kev[2 * c->n_kev] : nullptr, e1 = timed ? c->kev[2 * c->n_kev + 1] : nullptr;
    if (timed) c->n_kev++;
    if (own_events) {
        e0 = own_events[0];
        e1 = own_events[1];
    }
    if (lean) {
        switch (c->P.input_kind) {
        case NFC_IN_IQ_F32: launch_lean<IN_IQ_F32>(c, A, nwork, e0, e1); break;
        case NFC_IN_REAL_F32_SQ: launch_lean<IN_REAL_F32_SQ>(c, A, nwork, e0, e1); break;
        default: launch_lean<IN_I16_SQ>(c, A, nwork, e0, e1); break;
        }
        return;
    }
    switch (c->P.input_kind) {
    case NFC_IN_IQ_F32: launch_threshold<IN_IQ_F32>(c, A, nwork, e0, e1); break;
    case NFC_IN_ENV_F32: launch_threshold<IN_ENV_F32>(c, A, nwork, e0, e1); break;
    case NFC_IN_REAL_F32_SQ: launch_threshold<IN_REAL_F32_SQ>(c, A, nwork, e0, e1); break;
    default: launch_threshold<IN_I16_SQ>(c, A, nwork, e0, e1); break;
    }
}
void launch_fill_kind(nfc_ctx *c, const void *in, uint32_t n, int nchunks, int ring_idx = -1) {
    float *ring = c->d_ring[ring_idx < 0 ? c->ring_cur : ring_idx].as<float>();
    Carry *cr = dC(c);
    EdgeCarryInit eci{(int32_t *)dE(c), c->L % c->mx};
    uint8_t *ver = c->d_ver.as<uint8_t>();
    CertSummary *sum = (CertSummary *)(dT(c) + TOT_CERT);
    StateInit init;
    memset(&init, 0, sizeof init);
    if (c->state_dirty) {
        static_assert(offsetof(DevState, totals) <= sizeof init.words && offsetof(DevState, totals) % 4 == 0, "state head fits");
        DevState h;
        h.carry = c->dirty_carry;
        h.ecarry = c->dirty_ecarry;
        h.dcarry = c->dirty_dcarry;
        init.apply = 1;
        init.n_words = (int32_t)(offsetof(DevState, totals) / 4);
        memcpy(init.words, &h, offsetof(DevState, totals));
        init.dst = (uint32_t *)c->d_state.p;
        init.fill_ring = c->dirty_fill_ring ? 1 : 0;
        init.fill = c->dirty_fill;
        init.ring_len = c->Lpad;
        c->state_dirty = c->dirty_fill_ring = false;
    }
    switch (c->P.input_kind) {
    case NFC_IN_IQ_F32: NFC_LAUNCH((k_fill<IN_IQ_F32>), dim3(1), dim3(FILL_BLOCK), (size_t)c->Lpad * 4, c->st, in, n, c->i16_scale, c->L, ring, cr, eci, ver, nchunks, sum, init, &((DevState *)c->d_state.p)->seq[0], c->batch_seq); break;
    case NFC_IN_ENV_F32: NFC_LAUNCH((k_fill<IN_ENV_F32>), dim3(1), dim3(FILL_BLOCK), (size_t)c->Lpad * 4, c->st, in, n, c->i16_scale, c->L, ring, cr, eci, ver, nchunks, sum, init, &((DevState *)c->d_state.p)->seq[0], c->batch_seq); break;
    case NFC_IN_REAL_F32_SQ: NFC_LAUNCH((k_fill<IN_REAL_F32_SQ>), dim3(1), dim3(FILL_BLOCK), (size_t)c->Lpad * 4, c->st, in, n, c->i16_scale, c->L, ring, cr, eci, ver, nchunks, sum, init, &((DevState *)c->d_state.p)->seq[0], c->batch_seq); break;
    default: NFC_LAUNCH((k_fill<IN_I16_SQ>), dim3(1), dim3(FILL_BLOCK), (size_t)c->Lpad * 4, c->st, in, n, c->i16_scale, c->L, ring, cr, eci, ver, nchunks, sum, init, &((DevState *)c->d_state.p)->seq[0], c->batch_seq); break;
    }
}
void launch_seq_kind(nfc_ctx *c, const SeqArgs &A) {
    switch (c->P.input_kind) {
    case NFC_IN_IQ_F32: NFC_LAUNCH((k_threshold_seq<IN_IQ_F32>), dim3(1), dim3(64), 0, c->st, A); break;
    case NFC_IN_ENV_F32: NFC_LAUNCH((k_threshold_seq<IN_ENV_F32>), dim3(1), dim3(64), 0, c->st, A); break;
    case NFC_IN_REAL_F32_SQ: NFC_LAUNCH((k_threshold_seq<IN_REAL_F32_SQ>), dim3(1), dim3(64), 0, c->st, A); break;
    default: NFC_LAUNCH((k_threshold_seq<IN_I16_SQ>), dim3(1), dim3(64), 0, c->st, A); break;
    }
}

int ceil_log2(int v) {
    int b = 0;
    while ((1 << b) < v) b++;
    return b;
}

double elapsed_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

// ---------------------------------------------------------------------------
// threshold stage
// ---------------------------------------------------------------------------
// `ahead`, when given, enqueues the stages that follow (edges, decode) behind the first certification WITHOUT
// waiting for its verdict: certification almost always succeeds, so the host round trip that reads the verdict
// overlaps those stages instead of idling the GPU.  *clean reports that the verdict let that work stand.
// One parallel attempt at the samples [base, n_all) of the batch (base a multiple of the step; the planes, the ring and
// the carried sums already hold everything before base).  *need_seq: the attempt cannot vouch for its sums (or the
// sequential kernel was asked for): nothing of it stands and the caller replays sequentially.
// Every fp64 sum of a batch is exact -- hence independent of the order it was added in -- when all
// operands are multiples of 2^low and no sum reaches 2^(low + 53).  Operands: ring values (24-bit mantissas)
// and the carried ss / delta (their lowest set bits); sums: the window sums (bounded by the kernel from the sums
// it tracked) and the carried ss itself.  hc: the carried values after the batch; emin / emax / vtop: what its chunks measured.
static bool sums_exact(const Carry &hc, int emin, int emax, uint32_t vtop) {
    int low = emin - 23;
    if (hc.ss_emin != 255) low = std::min(low, hc.ss_emin);
    float vtf;
    memcpy(&vtf, &vtop, 4);
    int high = 255 + 64;   // vtop: f32 bits of an upper bound of every window sum the batch saw
    if (std::isfinite(vtf) && vtf >= 0.f) high = vtf > 0.f ? std::ilogb((double)vtf) + 127 : 0;
    high = std::max(high, hc.ss_emax);
    return (emax < 255) && (high - low <= 52);
}

// What one parallel attempt at [base, n_all) needs before anything is launched: the chunking, room in every per-chunk
// buffer, and the kernels' argument block.  (Shared by the synchronous path and by a batch submitted ahead, which works
// on the other pair of planes, from the window the batch before it leaves, with the LOW bookkeeping read on the device.)
struct ThrPlan {
    uint32_t nch;
    bool lean_applies;
    uint8_t *d_cert, *d_gflags, *d_gmin, *d_gmax;
    const uint8_t *h_cert, *h_gflags, *h_gmin, *h_gmax;
};
static int thr_prepare(nfc_ctx *c, const void *d_in, uint32_t n, uint32_t n_all, uint32_t skip, uint32_t base, uint64_t nseen,
                       const EdgeCarry &ec, int ring_in, DevBuf &planes_neg, DevBuf &planes_pos, bool low_on_device, ThrArgs &A, ThrPlan &P) {
    const int L = c->L;
    // Chunk length for this batch: one wave per chunk, and a chunk's latency is what the launch takes, so
    // aim at one full round of resident waves (no second, half-empty round), never below the configured size.
    // Long windows: the ring of a chunk in global memory frees the LDS and brings the occupancy back to what the registers
    // allow -- worth it when the batch then fills the machine with chunks of many windows each (a chunk pays one window
    // of speculation and two windows of certification traffic): otherwise the LDS ring, with its fewer, longer chunks.
    // Measured on configs[3] (10 Msps, av_window 10000, 1e9 samples): the lean kernel on the 40 KB LDS ring -- ONE wave per SIMD,
    // 1024 chunks -- takes 2.3 ms per pass, k_threshold with the ring in global memory at five waves per SIMD 4.7 ms (the delay
    // line adds a read and a write per sample; a lone wave is bound by its own instruction stream, which the lean kernel
    // shortened).  So the global ring is only taken on request (NFC_RING=global) or where the lean kernel does not apply.
    const bool lean_applies = c->lean && c->P.input_kind != NFC_IN_ENV_F32 && c->mx <= 500;
    c->gring = c->gring_ok && (c->gring_force || (!lean_applies && (uint64_t)n >= (uint64_t)c->wave_slots_g * 8u * (uint64_t)c->L));
    if (!c->P.chunk_samples) {
        const uint64_t slots = (uint64_t)(c->gring ? c->wave_slots_g : (lean_applies ? c->lean_slots : c->wave_slots));
        // (the lean kernel walks whole supersteps of lean_k steps: a chunk that is not a multiple of them ends on slow single steps)
        const int stp = 64 * c->rows_per_step * ((c->lean && !c->gring) ? c->lean_k : 1);
        uint64_t want = ((uint64_t)n + slots - 1) / slots;
        want = (want + stp - 1) / stp * stp;
        c->C = (int)std::max<uint64_t>(want, (uint64_t)c->C_min);
    }
    const uint32_t off = 0u;   // (chunk c covers samples [c*C - off, (c+1)*C - off): the kernels here use off = 0)
    const uint32_t nch = (uint32_t)(((uint64_t)n + off + c->C - 1) / c->C);
    c->stats.n_chunks = nch;
    c->stats.chunk_samples = (uint32_t)c->C;
    const size_t nwords = ((size_t)n_all + 63) / 64 + 8;
    HIPCHK(c, planes_neg.ensure(nwords * 8));
    HIPCHK(c, planes_pos.ensure(nwords * 8));
    for (int b = 0; b < 2; b++) {
        HIPCHK(c, c->d_ringout[b].ensure((size_t)nch * L * sizeof(float)));
        HIPCHK(c, c->d_touched[b].ensure((size_t)nch * c->twords * sizeof(uint32_t)));
        HIPCHK(c, c->d_info[b].ensure((size_t)nch * sizeof(ChunkInfo)));
    }
    HIPCHK(c, c->d_ringin.ensure((size_t)nch * L * sizeof(float)));
    HIPCHK(c, c->d_meta.ensure((size_t)nch * sizeof(RunMeta)));
    HIPCHK(c, c->d_ver.ensure(nch));
    HIPCHK(c, c->d_cflags.ensure((size_t)4 * nch));   // sections: cert | gflags | gmin | gmax
    if (c->h_cflags_cap < (size_t)4 * nch) {
        if (c->h_cflags) (void)hipHostFree(c->h_cflags);
        c->h_cflags_cap = (size_t)4 * nch + 4096;
        HIPCHK(c, hipHostMalloc((void **)&c->h_cflags, c->h_cflags_cap, hipHostMallocDefault));
    }
    uint8_t *d_cert = c->d_cflags.as<uint8_t>(), *d_gflags = d_cert + nch, *d_gmin = d_cert + 2 * (size_t)nch,
            *d_gmax = d_cert + 3 * (size_t)nch;
    const uint8_t *h_cert = c->h_cflags, *h_gflags = c->h_cflags + nch, *h_gmin = c->h_cflags + 2 * (size_t)nch,
                  *h_gmax = c->h_cflags + 3 * (size_t)nch;
    HIPCHK(c, c->d_list.ensure((size_t)nch * 4));
    HIPCHK(c, c->d_gvtop.ensure((size_t)nch * 4));
    if (c->gring) HIPCHK(c, c->d_gring.ensure((size_t)nch * c->Lpad * c->lds_per_slot));
    c->h_ver.assign(nch, 0);

    // carried "HIGH ignored" bookkeeping from (_current_state, _last_bit, _dur) at the first stable sample
    const int s0 = (int)skip;
    int nl0, kl0;
    if (ec.last_bit == -1) {
        nl0 = s0 - ec.dur - 1;
        kl0 = 2 * (s0 - 1) + (ec.state == 2 ? 1 : 0);
    } else if (ec.state == 2) {
        nl0 = s0 - 1;
        kl0 = 2 * (s0 - ec.dur - 1) + 1;
    } else {
        nl0 = s0 - 1;
        kl0 = KEY_NONE;
    }

    memset(&A, 0, sizeof A);
    A.gring = c->d_gring.p;
    A.in = d_in;
    A.n = n;
    A.skip = skip;
    A.g0modL = (uint32_t)(nseen % (uint64_t)L);
    A.L = L;
    A.Lpad = c->Lpad;
    A.mx = c->mx;
    A.C = c->C;
    A.nchunks = (int)nch;
    A.lo = c->P.lo_val;
    A.hi = c->P.hi_val;
    A.hi_plus = c->hi_plus;
    A.lo_a = c->lo_a;
    A.lo_b = c->lo_b;
    A.hi_a = c->hi_a;
    A.hi_b = c->hi_b;
    A.bands_ok = c->bands_ok;
    A.fast_ok = c->fast_ok;
    A.i16_scale = c->i16_scale;
    A.eps = c->eps;
    A.ring_carry = c->d_ring[ring_in].as<float>();
    A.carry = dC(c);
    A.nl0 = nl0;
    A.kl0 = kl0;
    A.low_src = low_on_device ? dC(c) : nullptr;
    A.lo_L = c->P.lo_val / (double)L;
    A.hi_L = c->P.hi_val / (double)L;
    for (int f = 0; f < 6; f++) A.fold_sh[f] = (f < c->nfold) ? (1 << f) : 0;
    A.probe_mid = (1 << c->nfold) / 2;
    A.probe_end = (1 << c->nfold) - 1;
    A.selmask = c->selmask;
    for (int b = 0; b < 2; b++) {
        A.ring_out[b] = c->d_ringout[b].as<float>();
        A.touched[b] = c->d_touched[b].as<uint32_t>();
        A.info[b] = c->d_info[b].as<ChunkInfo>();
    }
    A.ver = c->d_ver.as<uint8_t>();
    A.ring_in = c->d_ringin.as<float>();
    A.meta = c->d_meta.as<RunMeta>();
    A.gmin = d_gmin;
    A.gmax = d_gmax;
    A.gflags = d_gflags;
    A.gvtop = c->d_gvtop.as<uint32_t>();
    A.neg = planes_neg.as<uint64_t>() + base / 64;
    A.pos = planes_pos.as<uint64_t>() + base / 64;
    A.twords = c->twords;
    A.off = (int32_t)off;
    A.nrows = (L + 63) / 64;
    P.nch = nch;
    P.lean_applies = lean_applies;
    P.d_cert = d_cert;
    P.d_gflags = d_gflags;
    P.d_gmin = d_gmin;
    P.d_gmax = d_gmax;
    P.h_cert = h_cert;
    P.h_gflags = h_gflags;
    P.h_gmin = h_gmin;
    P.h_gmax = h_gmax;
    return NFC_OK;
}

static int threshold_span(nfc_ctx *c, const void *d_in_all, uint32_t n_all, uint32_t skip_all, uint32_t base, const EdgeCarry &ec,
                          const std::function<int()> *ahead, bool *clean, bool *need_seq_out) {
    bool ran_ahead = false;
    *clean = false;
    const int L = c->L;
    const uint32_t n = n_all - base;
    const uint32_t skip = skip_all > base ? skip_all - base : 0u;
    const void *d_in = (const char *)d_in_all + (size_t)base * c->in_bytes_per_sample;
    const uint64_t nseen = c->nseen + base;
    uint32_t passes = 0;
    ThrArgs A;
    ThrPlan P;
    if (int rc = thr_prepare(c, d_in, n, n_all, skip, base, nseen, ec, c->ring_cur, c->d_neg, c->d_pos, false, A, P)) return rc;
    const uint32_t nch = P.nch;
    const bool lean_applies = P.lean_applies;
    uint8_t *d_cert = P.d_cert;
    const uint8_t *h_cert = P.h_cert, *h_gflags = P.h_gflags, *h_gmin = P.h_gmin, *h_gmax = P.h_gmax;

    // fill (if the window is not full yet) + per-batch preparation (delta, guard span, version bytes): one launch
    launch_fill_kind(c, d_in, n, (int)nch);

    // a 256-sample step must not wrap the ring onto itself: short windows take the sequential kernel
    const bool force_seq = (c->P.flags & NFC_FLAG_FORCE_SEQUENTIAL) != 0 || L < STEP;
    bool need_seq = force_seq;
    if (!force_seq) {
        // pass 0: every chunk from a speculated incoming state (chunk 0: the carried, exact one)
        A.list = nullptr;
        A.nlist = 0;
        A.mode = 0;
        // the lean kernel wherever it applies (LDS ring, more than one chunk: chunk 0's verdict travels with the certification)
        const bool lean = lean_applies && !c->gring && nch > 1;   // (raw envelopes may be negative: no sign bit to spare;
                                                                    // max_len beyond 500 samples: not exercised, left to k_threshold)
        A.cert = d_cert;
        A.sum = (CertSummary *)(dT(c) + TOT_CERT);
        A.ksteps = c->lean_rounds;
        A.gfac = c->lean_gfac;
        A.gfloor = c->lean_gmin;
        A.blk = 1 << c->nfold;
        const bool dbg_clk = lean && getenv("NFC_DEBUG_CLK") != nullptr;
        if (dbg_clk) {
            HIPCHK(c, c->d_certinfo.ensure((size_t)nch * 32));
            A.dbg_clk = c->d_certinfo.as<unsigned long long>();
        }
        launch_threshold_kind(c, A, nch, lean);
        if (dbg_clk) {
            std::vector<unsigned long long> h((size_t)nch * 4);
            HIPCHK(c, hipStreamSynchronize(c->st));
            HIPCHK(c, hipMemcpy(h.data(), c->d_certinfo.p, (size_t)nch * 32, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            double pro = 0, loop = 0, epi = 0, maxtot = 0, maxstart = 0;
            for (uint32_t k = 0; k < nch; k++) {
                t0 = std::min(t0, h[4 * k]);
                t1 = std::max(t1, h[4 * k + 3]);
            }
            for (uint32_t k = 0; k < nch; k++) {
                pro += (double)(h[4 * k + 1] - h[4 * k]);
                loop += (double)(h[4 * k + 2] - h[4 * k + 1]);
                epi += (double)(h[4 * k + 3] - h[4 * k + 2]);
                maxtot = std::max(maxtot, (double)(h[4 * k + 3] - h[4 * k]));
                maxstart = std::max(maxstart, (double)(h[4 * k] - t0));
            }
            fprintf(stderr, "[nfc] lean kernel, %u chunks, s_memtime ticks: first start .. last end %llu; per wave: incoming state %.0f, loop %.0f, "
                    "summary %.0f, longest wave %.0f, latest start %.0f\n", nch, t1 - t0, pro / nch, loop / nch, epi / nch, maxtot, maxstart);
            if (const char *path = getenv("NFC_DEBUG_CLK")) {
                if (path[0] == '/' || path[0] == '.' || path[0] == 'g') {   // a file name: the raw stamps, for tools/clk_hist.py
                    if (FILE *f = fopen(path, "wb")) {
                        fwrite(h.data(), 8, h.size(), f);
                        fclose(f);
                    }
                }
            }
            A.dbg_clk = nullptr;
        }
        c->stats.threshold_passes++;
        passes++;

        // certify; re-run what cannot be proven from the exact (look-back) state; certify again what can
        // see a re-run chunk.  Every round makes at least the first pending chunk final.
        const bool dbg = getenv("NFC_DEBUG") != nullptr;
        bool first_round = true;
        c->h_list.clear();
        for (uint32_t k = 1; k < nch; k++) c->h_list.push_back(k);
        int rounds = 0;
        bool have_summary = false;
        CertSummary summary{};
        uint8_t *tot = dT(c);
        while (!c->h_list.empty()) {
            const uint32_t np = (uint32_t)c->h_list.size();
            if (first_round) {
                A.list = nullptr;   // k_certify: chunks 1 .. nch-1
            } else {
                HIPCHK(c, hipMemcpyAsync(c->d_list.p, c->h_list.data(), (size_t)np * 4, hipMemcpyHostToDevice, c->st));
                A.list = c->d_list.as<uint32_t>();
            }
            A.nlist = np;
            if (dbg) HIPCHK(c, c->d_certinfo.ensure((size_t)nch * sizeof(CertInfo)));
            // the first round also resolves the end-of-batch state (last workgroup) and leaves its verdict as a
            // summary in the mirrored state block; later rounds (after re-runs) read the per-chunk flags
            CertSummary *d_sum = (CertSummary *)(tot + TOT_CERT);
            auto launch_certify = [&]() {
                NFC_LAUNCH(k_certify, dim3((np + 3) / 4 + 1), dim3(256), 0, c->st, A, d_cert,
                                   dbg ? c->d_certinfo.as<CertInfo>() : nullptr, c->d_ring[(c->ring_cur + 1) % NRING].as<float>(), dC(c),
                                   first_round ? d_sum : (CertSummary *)nullptr);
            };
            if (first_round && ahead && !dbg) {
                // the stages that follow are enqueued now; their first full-width kernel takes the certification along
                c->cert = CertLaunch{A, d_cert, c->d_ring[(c->ring_cur + 1) % NRING].as<float>(), dC(c), d_sum, (np + 3) / 4 + 1};
                c->cert_pending = true;
                const int rc = (*ahead)();
                if (c->cert_pending) {   // (a short batch's one-launch stage, or no edge stage at all: on its own then)
                    c->cert_pending = false;
                    launch_certify();
                    HIPCHK(c, mirror_async(c));
                }
                if (rc) return rc;
                ran_ahead = true;
            } else {
                launch_certify();
                if (!first_round) HIPCHK(c, hipMemcpyAsync(c->h_cflags, c->d_cflags.p, (size_t)4 * nch, hipMemcpyDeviceToHost, c->st));
                HIPCHK(c, mirror_async(c));
                if (first_round && ahead) {
                    const int rc = (*ahead)();
                    if (rc) return rc;
                    ran_ahead = true;
                }
            }
            HIPCHK(c, hipStreamSynchronize(c->st));
            BATCHCHK(c, true);   // (the verdict summary and the totals are about to be read out of the mirror)
            std::vector<uint32_t> failing;
            if (first_round) {
                memcpy(&summary, c->hs->totals + TOT_CERT, sizeof summary);
                if (summary.n_fail == 0 && !dbg) {
                    have_summary = true;
                    break;
                }
                HIPCHK(c, hipMemcpy(c->h_cflags, c->d_cflags.p, (size_t)4 * nch, hipMemcpyDeviceToHost));
            }
            if (first_round && lean && !h_cert[0]) failing.push_back(0);   // chunk 0 gave up: re-run from the carried state
            for (uint32_t k : c->h_list)
                if (!h_cert[k]) failing.push_back(k);
            if (getenv("NFC_TRACE")) {
                fprintf(stderr, "[nfc] round %d: n_fail %u, %zu failing of %u pending (cert[0] %d):", rounds, summary.n_fail, failing.size(), np, (int)h_cert[0]);
                for (size_t i = 0; i < failing.size() && i < 12; i++) fprintf(stderr, " %u", failing[i]);
                fprintf(stderr, "\n");
            }
            if (dbg) {
                std::vector<CertInfo> ci(nch);
                HIPCHK(c, hipMemcpy(ci.data(), c->d_certinfo.p, (size_t)nch * sizeof(CertInfo), hipMemcpyDeviceToHost));
                double worst = 0;
                for (uint32_t k : c->h_list) worst = std::max(worst, (double)ci[k].d / std::max(1e-30f, ci[k].allowed));
                fprintf(stderr, "[nfc] certify round %d: %u pending, %zu failing, worst d/allowed %.3f\n", rounds, np,
                        failing.size(), worst);
                if (first_round && lean) {
                    std::vector<ChunkInfo> inf(nch);
                    HIPCHK(c, hipMemcpy(inf.data(), c->d_info[0].p, (size_t)nch * sizeof(ChunkInfo), hipMemcpyDeviceToHost));
                    int why[8] = {0};
                    for (uint32_t k : failing) why[(inf[k].flags >> 4) & 7]++;
                    fprintf(stderr, "[nfc]   lean gave up: range %d, band %d, low run %d, allowance %d, first sample %d; not lean %d\n", why[1], why[2],
                            why[3], why[4], why[5], why[0]);
                }
                for (size_t i = 0; i < failing.size() && i < 8; i++) {
                    const CertInfo &x = ci[failing[i]];
                    fprintf(stderr, "[nfc]   chunk %u: d %.6g allowed %.6g all_robust %u low_ok %u\n", failing[i], x.d,
                            x.allowed, x.all_robust, x.low_ok);
                }
            }
            if (failing.empty()) break;
            first_round = false;
            HIPCHK(c, hipMemcpyAsync(c->d_list.p, failing.data(), failing.size() * 4, hipMemcpyHostToDevice, c->st));
            A.list = c->d_list.as<uint32_t>();
            A.nlist = (uint32_t)failing.size();
            A.mode = 1;
            launch_threshold_kind(c, A, A.nlist);
            c->stats.threshold_passes++;
            passes++;
            c->stats.chunks_rerun += A.nlist;
            HIPCHK(c, hipMemcpyAsync(c->h_cflags, c->d_cflags.p, (size_t)4 * nch, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipStreamSynchronize(c->st));
            std::vector<uint8_t> ran(nch, 0);
            for (uint32_t k : failing) {
                ran[k] = 1;
                c->h_ver[k] ^= 1;
            }
            HIPCHK(c, hipMemcpyAsync(c->d_ver.p, c->h_ver.data(), nch, hipMemcpyHostToDevice, c->st));
            // pending: the re-run chunks (a predecessor may have been re-run beside them) and every
            // chunk that can see one of them through predecessors that left ring slots untouched
            c->h_list.clear();
            bool vis = false;
            for (uint32_t k = 0; k < nch; k++) {
                if (vis || ran[k]) c->h_list.push_back(k);
                const bool full = !(h_gflags[k] & 2);
                vis = ran[k] || (vis && !full);
            }
            if (++rounds > (int)nch + 2) return fail(c, NFC_ERR_INTERNAL, "threshold passes did not converge");
        }

        // can every fp64 sum of this batch be proven exact?  (otherwise the summation order matters)
        int emin = 255, emax = 0;
        bool flagged = false;
        uint32_t vtop = 0;
        if (have_summary && passes == 1) {
            emin = (int)summary.emin;
            emax = (int)summary.emax;
            flagged = summary.flagged != 0;
            vtop = summary.vtop;
        } else {
            HIPCHK(c, hipMemcpyAsync(c->h_cflags, c->d_cflags.p, (size_t)4 * nch, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, mirror_async(c));
            HIPCHK(c, hipStreamSynchronize(c->st));
            BATCHCHK(c, true);
            std::vector<uint32_t> hv(nch);
            HIPCHK(c, hipMemcpy(hv.data(), c->d_gvtop.p, (size_t)nch * 4, hipMemcpyDeviceToHost));
            for (uint32_t k = 0; k < nch; k++) {
                emin = std::min(emin, (int)h_gmin[k]);
                emax = std::max(emax, (int)h_gmax[k]);
                if (h_gflags[k] & 1) flagged = true;
                vtop = std::max(vtop, hv[k]);
            }
        }
        c->h_carry = c->hs->carry;
        carry_apply_fin(c->h_carry);
        const bool exact = sums_exact(c->h_carry, emin, emax, vtop);
        if (dbg) fprintf(stderr, "[nfc] guard: emin %d emax %d ss_emin %d ss_emax %d flagged %d exact %d\n", emin, emax,
                         c->h_carry.ss_emin, c->h_carry.ss_emax, (int)flagged, (int)exact);
        if (!exact || flagged) need_seq = true;
    }

    *need_seq_out = need_seq;
    if (!need_seq) {
        // the end-of-batch ring was resolved beside the last certification unless chunks were re-run after it
        if (passes > 1 || nch == 1)
            NFC_LAUNCH(k_finalize_state, dim3(1), dim3(FIN_BLOCK), 0, c->st, A, c->d_ring[(c->ring_cur + 1) % NRING].as<float>(), dC(c));
        c->ring_cur = (c->ring_cur + 1) % NRING;
        *clean = ran_ahead && passes == 1;
    }
    return NFC_OK;
}

// The literal loop on one lane over the samples [base, base + len) of the batch (exact whatever the sums look like).
// ec_out (optional): (_current_state, _last_bit, _dur) after the last of them, for the attempt that follows.
static int sequential_span(nfc_ctx *c, const void *d_in_all, uint32_t skip_all, uint32_t base, uint32_t len, const EdgeCarry &ec,
                           EdgeCarry *ec_out) {
    SeqArgs S;
    memset(&S, 0, sizeof S);
    S.in = (const char *)d_in_all + (size_t)base * c->in_bytes_per_sample;
    S.n = len;
    S.skip = skip_all > base ? std::min(skip_all - base, len) : 0u;
    S.g0modL = (uint32_t)((c->nseen + base) % (uint64_t)c->L);
    S.L = c->L;
    S.mx = c->mx;
    S.lo = c->P.lo_val;
    S.hi = c->P.hi_val;
    S.hi_plus = c->hi_plus;
    S.i16_scale = c->i16_scale;
    S.ring = c->d_ring[c->ring_cur].as<float>();
    S.carry = dC(c);
    S.state = ec.state;
    S.last_bit = ec.last_bit;
    S.dur = ec.dur;
    S.neg = c->d_neg.as<uint64_t>() + base / 64;
    S.pos = c->d_pos.as<uint64_t>() + base / 64;
    HIPCHK(c, c->d_seqout.ensure(64));
    S.out = c->d_seqout.as<int32_t>();
    launch_seq_kind(c, S);
    c->stats.used_sequential = 1;
    if (ec_out) {
        int32_t o[3];
        HIPCHK(c, hipStreamSynchronize(c->st));
        HIPCHK(c, hipMemcpy(o, c->d_seqout.p, sizeof o, hipMemcpyDeviceToHost));
        ec_out->state = o[0];
        ec_out->last_bit = o[1];
        ec_out->dur = o[2];
    }
    return NFC_OK;
}

// The threshold stage of a batch.  Almost always one parallel attempt.  When an attempt cannot prove its fp64 sums
// exact -- typically a stream that starts inside a transaction: the fill phase stored pause-level samples, and while
// they sit in the window the reference's own running sum rounds -- the sequential kernel replays a PREFIX (a few
// windows, until those values have been overwritten) and the rest of the batch gets another parallel attempt.
int run_threshold(nfc_ctx *c, const void *d_in, uint32_t n, uint32_t skip, const std::function<int()> *ahead, bool *clean) {
    *clean = false;
    c->stats.threshold_passes = 0;
    c->stats.chunks_rerun = 0;
    c->stats.used_sequential = 0;
    const uint32_t span = (uint32_t)std::min<uint64_t>(((uint64_t)8 * c->L + STEP - 1) / STEP * STEP, 1u << 30);   // prefix per round
    EdgeCarry ec = c->h_ecarry;
    uint32_t base = 0;
    for (int round = 0;; round++) {
        bool need_seq = false, span_clean = false;
        const int rc = threshold_span(c, d_in, n, skip, base, ec, base == 0 ? ahead : nullptr, &span_clean, &need_seq);
        if (rc) return rc;
        if (!need_seq) {
            *clean = span_clean && base == 0;
            return NFC_OK;
        }
        const bool forced = (c->P.flags & NFC_FLAG_FORCE_SEQUENTIAL) != 0 || c->L < STEP;
        const uint32_t left = n - base;
        if (forced || round >= 6 || left <= 4 * span) return sequential_span(c, d_in, skip, base, left, ec, nullptr);
        const int rs = sequential_span(c, d_in, skip, base, span, ec, &ec);
        if (rs) return rs;
        base += span;
    }
}

// ---------------------------------------------------------------------------
// edge stage
// ---------------------------------------------------------------------------
// Both stages run without a host round trip: buffers and grids are sized from capacity estimates (last batch's
// counts with head-room), the true counts stay on the device, and the caller checks them after the batch's
// final sync -- on overflow the two stages are simply repeated with larger estimates (they are idempotent:
// carried values come in by value and go out to write-only slots).
int run_edges(nfc_ctx *c, uint32_t n, uint32_t skip, uint64_t g0) {
    uint8_t *tot = dT(c);
    const size_t nwords = ((size_t)n + 63) / 64;
    EdgeArgs E;
    E.neg = c->d_neg.as<uint64_t>();
    E.pos = c->d_pos.as<uint64_t>();
    E.n = n;
    E.skip = skip;
    E.mx = c->mx;
    E.dur_in = c->h_ecarry.dur;
    E.last_bit_in = c->h_ecarry.last_bit;
    E.state_in = c->h_ecarry.state;
    E.nd = c->mx + 1;
    E.g0 = g0;
    E.mx_magic = c->mx > 1 ? (uint32_t)(0x100000000ull / (uint64_t)c->mx) : 0xFFFFFFFFu;
    E.per_mask = 0;
    for (int b = 0; b < 64; b += c->mx) E.per_mask |= 1ull << b;
    const size_t tiles = edge_num_tiles(nwords);   // EW_WORDS words per tile in both launches of the stage
    HIPCHK(c, c->d_partials.ensure((tiles + 1) * sizeof(EdgeAgg)));
    EdgeAgg *parts = c->d_partials.as<EdgeAgg>();
    // launch 1: one aggregate per tile (first change, last two changes, entries it is sure of).  While the tiles are few,
    // each tile's workgroup of the writer folds its predecessors' aggregates itself (scan.hip.h: tile_prefix) and the
    // single-workgroup prefix launch is not needed.
    const bool own = tiles <= c->own_prefix_max;
    Last2 *last2_total = (Last2 *)(tot + TOT_LAST2);
    uint32_t *edges_total = (uint32_t *)(tot + TOT_EDGES);
    const EdgeAggOp op{E.mx, E.mx_magic};
    if (c->cert_pending && tiles) {
        c->cert_pending = false;
        NFC_LAUNCH(k_certify_and_reduce, dim3((unsigned)(c->cert.blocks + tiles)), dim3(256), 0, c->st, c->cert, E, nwords, parts);
    } else if (tiles) {
        NFC_LAUNCH(k_edge_reduce, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, E, nwords, parts);
    }
    if (!own || !tiles)
        scan_partials_with(c->st, op, tiles, nullptr, (uint32_t)EW_WORDS, parts, op.identity(), (EdgeAgg *)nullptr,
                           EdgeTotalEpilogue{E, edges_total, last2_total, dE(c)});
    const uint32_t cap = c->cap_edges;
    HIPCHK(c, c->d_epos.ensure(((size_t)cap + 8) * 4));
    HIPCHK(c, c->d_ecode.ensure(((size_t)cap + 8) * 2));
    if (tiles)
        NFC_LAUNCH(k_write_edges, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, E, nwords, parts, c->d_epos.as<uint32_t>(),
                   c->d_ecode.as<uint16_t>(), cap, own, edges_total, last2_total, dE(c));
    c->edges_from_host = false;
    return NFC_OK;
}

// ---------------------------------------------------------------------------
// decode + framing
// ---------------------------------------------------------------------------
// what both forms of the stage write through: symbol, bit and close arrays sized from the estimates; the open packets'
// bits of earlier batches go in front of this batch's (the other half of the double buffer takes the next ones)
int frame_out(nfc_ctx *c, FrameOut &P, const bool (&enabled)[2]) {
    memset(&P, 0, sizeof P);
    const int pn = 1 - c->pend_cur;
    P.epos = c->d_epos.as<uint32_t>();
    P.g0 = c->last_g0;
    P.idx64 = c->edges_from_host ? c->d_eidx.as<uint64_t>() : nullptr;
    for (int t = 0; t < 2; t++) {
        const uint32_t cs = c->cap_sym[t];
        HIPCHK(c, c->d_sym[t].ensure((size_t)cs + 16));
        P.sym[t] = c->d_sym[t].as<uint8_t>();
        P.cap_sym[t] = cs;
        P.started_in[t] = (uint32_t)c->h_dcarry.pkt_started[t];
        if (!enabled[t]) continue;   // no symbols of this type (background.py:17-25); its carry stays
        const uint32_t pend = c->h_dcarry.pending[t];
        HIPCHK(c, c->d_bits[t].ensure((size_t)pend + cs + 16));
        HIPCHK(c, c->d_pending[t][pn].ensure((size_t)pend + cs + 16));
        HIPCHK(c, c->d_close_end[t].ensure(((size_t)cs + 4) * 4));
        HIPCHK(c, c->d_close_idx[t].ensure(((size_t)cs + 4) * 8));
        P.bits[t] = c->d_bits[t].as<uint8_t>();
        P.close_end[t] = c->d_close_end[t].as<uint32_t>();
        P.close_idx[t] = c->d_close_idx[t].as<uint64_t>();
        P.cap_bits[t] = pend + cs;
        P.cap_close[t] = cs;
        P.pending[t] = c->d_pending[t][c->pend_cur].as<uint8_t>();
        P.pend[t] = pend;
    }
    return NFC_OK;
}

int run_decode(nfc_ctx *c) {
    uint8_t *tot = dT(c);
    const uint32_t ce = c->cap_edges;                      // capacity; the count is on the device
    const uint32_t *ne_dev = (const uint32_t *)(tot + TOT_EDGES);
    const size_t tiles = dec_num_tiles(ce);
    HIPCHK(c, c->d_states.ensure((size_t)ce + 16));   // one out-byte per edge
    HIPCHK(c, c->d_partials.ensure((tiles + 1) * sizeof(DecMaps)));
    HIPCHK(c, c->d_partials2.ensure((tiles + 1) * sizeof(FrameAgg)));
    HIPCHK(c, c->d_aggs.ensure((tiles * SCAN_BLOCK + 1) * sizeof(DecMaps)));
    HIPCHK(c, c->d_faggs.ensure((tiles * SCAN_BLOCK + 1) * sizeof(FramePk)));
    const bool enabled[2] = {c->T.tag != 0, c->T.reader != 0};
    FrameOut P;
    const int rf = frame_out(c, P, enabled);
    if (rf) return rf;

    // decoder states: tile maps -> tile prefixes -> every thread walks its edges.  What the walk emits stays per edge
    // (one byte); a tile's symbol counts, framing map and bit / close counts are the aggregates of ONE more scan, whose
    // prefixes place the symbols, the packet bits and the packet ends in a single pass.
    const uint16_t *ecode = c->d_ecode.as<uint16_t>();
    uint8_t *outw = c->d_states.as<uint8_t>();
    const uint32_t dec_state_in = (uint32_t)c->h_dcarry.mil_state | ((uint32_t)c->h_dcarry.man_state << 4);
    const bool lds_tables = 4 * c->T.nd <= DEC_LDS_ROWS;
    DecMaps *dparts = c->d_partials.as<DecMaps>(), *daggs = c->d_aggs.as<DecMaps>();
    FrameAgg *fparts = c->d_partials2.as<FrameAgg>();
    FrameAgg *frame_total = (FrameAgg *)(tot + TOT_FRAME);
    PktCnt *pk_total = (PktCnt *)(tot + TOT_PKT0);
    if (tiles) {
        if (lds_tables)
            NFC_LAUNCH(k_dec_reduce<true>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, dparts, daggs);
        else
            NFC_LAUNCH(k_dec_reduce<false>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, dparts, daggs);
    }
    const bool own = tiles <= c->own_prefix_max;   // (scan.hip.h: tile_prefix -- no prefix launches while the tiles are few)
    DecMaps *map_total = (DecMaps *)(tot + TOT_DECMAP);
    const DecCarryEpilogue epi{map_total, dec_state_in, dD(c), (uint32_t *)(tot + TOT_NSYM), pk_total,
                               {P.pend[0], P.pend[1]}, {P.started_in[0], P.started_in[1]}};
    if (!own) scan_partials<ComposeDec>(c->st, tiles, ne_dev, DEC_TILE, dparts, ComposeDec::identity_host(), map_total);
    if (tiles) {
        if (lds_tables)
            NFC_LAUNCH(k_dec_apply<true>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, dparts, daggs,
                               dec_state_in, outw, fparts, c->d_faggs.as<FramePk>(), own, map_total);
        else
            NFC_LAUNCH(k_dec_apply<false>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, dparts, daggs,
                               dec_state_in, outw, fparts, c->d_faggs.as<FramePk>(), own, map_total);
    }
    if (!own) scan_partials<FrameAggOp>(c->st, tiles, ne_dev, DEC_TILE, fparts, FrameAggOp::identity(), frame_total, epi);
    NFC_LAUNCH(k_frame_write, dim3((unsigned)std::max<size_t>(tiles, 1)), dim3(SCAN_BLOCK), 0, c->st, outw, (size_t)ce, ne_dev, fparts,
                       c->d_faggs.as<FramePk>(), P, own, frame_total, epi);
    const int pn = 1 - c->pend_cur;
    PktFinish F;
    memset(&F, 0, sizeof F);
    for (int t = 0; t < 2; t++) {
        F.enabled[t] = enabled[t] ? 1 : 0;
        F.bits[t] = P.bits[t];
        F.pending_next[t] = c->d_pending[t][pn].as<uint8_t>();
        F.close_end[t] = P.close_end[t];
        F.started_in[t] = (int32_t)P.started_in[t];
        F.pending_cap[t] = (uint32_t)std::min<size_t>(c->d_pending[t][pn].cap, 0xFFFFFFFFu);
        F.cap_bits[t] = P.cap_bits[t];
        F.cap_close[t] = P.cap_close[t];
    }
    F.totals = pk_total;
    F.frame_total = frame_total;
    F.carry = dD(c);
    static_assert(sizeof(DevState) % 4 == 0, "whole words");
    F.mirror_src = (const uint32_t *)c->d_state.p;   // the stage's last launch also fills the host's mirror of the state block
    F.mirror_dst = (uint32_t *)c->hs_dev;
    F.mirror_words = (uint32_t)(sizeof(DevState) / 4);
    F.stamp_word = (uint32_t)(offsetof(DevState, seq) / 4 + 1);
    F.stamp = c->stamp_b;
    NFC_LAUNCH(k_pkt_finish, dim3(1), dim3(256), 0, c->st, F);
    return NFC_OK;
}

// ---------------------------------------------------------------------------
// short batches: edges, decoders and framing in one launch (small.hip.h)
// ---------------------------------------------------------------------------
int run_small(nfc_ctx *c, uint32_t n, uint32_t skip, uint64_t g0) {
    uint8_t *tot = dT(c);
    SmallArgs A;
    memset(&A, 0, sizeof A);
    A.E.neg = c->d_neg.as<uint64_t>();
    A.E.pos = c->d_pos.as<uint64_t>();
    A.E.n = n;
    A.E.skip = skip;
    A.E.mx = c->mx;
    A.E.dur_in = c->h_ecarry.dur;
    A.E.last_bit_in = c->h_ecarry.last_bit;
    A.E.state_in = c->h_ecarry.state;
    A.E.nd = c->mx + 1;
    A.E.g0 = g0;
    A.E.mx_magic = c->mx > 1 ? (uint32_t)(0x100000000ull / (uint64_t)c->mx) : 0xFFFFFFFFu;
    for (int b = 0; b < 64; b += c->mx) A.E.per_mask |= 1ull << b;
    A.nwords = ((size_t)n + 63) / 64;
    const uint32_t ce = c->cap_edges;
    HIPCHK(c, c->d_epos.ensure(((size_t)ce + 8) * 4));
    HIPCHK(c, c->d_ecode.ensure(((size_t)ce + 8) * 2));
    c->edges_from_host = false;
    A.epos = c->d_epos.as<uint32_t>();
    A.ecode = c->d_ecode.as<uint16_t>();
    A.cap_edges = ce;
    A.T = c->T;
    A.dec_state_in = (uint32_t)c->h_dcarry.mil_state | ((uint32_t)c->h_dcarry.man_state << 4);
    const bool enabled[2] = {c->T.tag != 0, c->T.reader != 0};
    const int rf = frame_out(c, A.P, enabled);
    if (rf) return rf;
    const int pn = 1 - c->pend_cur;
    for (int t = 0; t < 2; t++) {
        A.enabled[t] = enabled[t] ? 1 : 0;
        A.pending_next[t] = c->d_pending[t][pn].as<uint8_t>();
        A.pending_cap[t] = (uint32_t)std::min<size_t>(c->d_pending[t][pn].cap, 0xFFFFFFFFu);
    }
    A.tot_last2 = (Last2 *)(tot + TOT_LAST2);
    A.tot_edges = (uint32_t *)(tot + TOT_EDGES);
    A.tot_decmap = (DecMaps *)(tot + TOT_DECMAP);
    A.tot_frame = (FrameAgg *)(tot + TOT_FRAME);
    A.tot_pk = (PktCnt *)(tot + TOT_PKT0);
    A.tot_nsym = (uint32_t *)(tot + TOT_NSYM);
    A.ecarry = dE(c);
    A.dcarry = dD(c);
    A.mirror_src = (const uint32_t *)c->d_state.p;
    A.mirror_dst = (uint32_t *)c->hs_dev;
    A.mirror_words = (uint32_t)(sizeof(DevState) / 4);
    A.stamp_word = (uint32_t)(offsetof(DevState, seq) / 4 + 1);
    A.stamp = c->stamp_b;
    NFC_LAUNCH(k_small_stage, dim3(1), dim3(SM_BLOCK), 0, c->st, A);
    return NFC_OK;
}

// capacity estimates from the densities of the previous batches (batches of very different lengths alternate
// when a capture is sharded: a short overlap, then the shard), with head-room
void size_capacities(nfc_ctx *c, uint32_t n) {
    const uint64_t ce = (uint64_t)((double)n * c->edge_rate * 1.25) + 65536;
    c->cap_edges = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ce, c->cap_edges_floor), 0xFFFFFF00u);
    for (int t = 0; t < 2; t++) {
        const uint64_t ub = (t == 1 ? 2ull : 1ull) * c->cap_edges + 16;   // <= 2 (Miller) / 1 (Manchester) symbols per edge
        const uint64_t cs = (uint64_t)((double)c->cap_edges * c->sym_rate[t] * 1.1) + 65536;
        c->cap_sym[t] = (uint32_t)std::min<uint64_t>(std::min(ub, std::max<uint64_t>(cs, c->cap_sym_floor[t])), 0xFFFFFF00u);
    }
}
// densities for the next batch's estimates
void update_estimates(nfc_ctx *c, uint32_t n) {
    c->cap_edges_floor = 0;
    c->cap_sym_floor[0] = c->cap_sym_floor[1] = 0;
    c->edge_rate = std::max({(double)c->n_edges / (double)n, c->edge_rate * 0.9, 1.0 / 64});
    for (int t = 0; t < 2; t++)
        if (c->n_edges) c->sym_rate[t] = std::max((double)c->n_sym[t] / (double)c->n_edges, c->sym_rate[t] * 0.9);
}

int process_batch(nfc_ctx *c, const void *d_in, size_t n64) {
    if (c->sub_count && !c->in_wait) return fail(c, NFC_ERR_STATE, "batches submitted with nfc_submit_device are in flight: nfc_wait for them first");
    c->low_valid = false;
    c->have_outputs = false;
    c->pk_ready[0] = c->pk_ready[1] = false;
    c->n_edges = 0;
    for (int t = 0; t < 2; t++) c->n_sym[t] = c->n_close[t] = c->n_bits[t] = 0;
    memset(&c->stats, 0, sizeof c->stats);
    c->n_kev = 0;
    if (n64 > (1ull << 30)) return fail(c, NFC_ERR_ARG, "batch of %zu samples exceeds 2^30; push it in pieces", n64);
    const uint32_t n = (uint32_t)n64;
    c->last_n = n;
    c->last_g0 = c->nseen;
    c->last_skip = 0;
    c->stats.bytes_in = (uint64_t)n * c->in_bytes_per_sample;
    if (n == 0) {
        c->have_outputs = true;
        return NFC_OK;
    }
    if (((uintptr_t)d_in & 15u) != 0) return fail(c, NFC_ERR_ARG, "device input must be 16-byte aligned");
    c->batch_seq++;
    c->stamp_b = c->batch_seq;
    launch_error() = LaunchError{};   // (a failure nobody reported belongs to an earlier call)
    if (c->timing >= 2) HIPCHK(c, hipEventRecord(c->ev[0], c->st));

    uint32_t skip = 0;
    bool fills = false;
    if (!c->h_carry.stable) {
        fills = true;
        skip = (uint32_t)std::min<uint64_t>(n, (uint64_t)(c->L - c->h_carry.filled));
        c->h_carry.filled += (int)skip;
        if (c->h_carry.filled == c->L) {
            c->h_carry.stable = 1;
            c->h_ecarry.state = 0;
            c->h_ecarry.last_bit = 0;
            c->h_ecarry.dur = c->L % c->mx;  // transition_sink.py:123 (k_fill sets the device copy)
        }
    }
    c->last_skip = skip;
    if (!c->h_carry.stable || skip == n) {
        // the whole batch went into the averaging window: no callback content (transition_sink.py:109-125)
        if (fills) {
            HIPCHK(c, c->d_ver.ensure(16));
            launch_fill_kind(c, d_in, n, 0);
        }
        HIPCHK(c, mirror_async(c));
        HIPCHK(c, hipStreamSynchronize(c->st));
        BATCHCHK(c, fills);
        c->h_carry = c->hs->carry;
        c->nseen += n;
        c->have_outputs = true;
        return NFC_OK;
    }

    if (c->timing >= 2) HIPCHK(c, hipEventRecord(c->ev[1], c->st));
    const bool want_edges = !(c->P.flags & NFC_FLAG_NO_EDGES);
    const EdgeCarry ecarry_in = c->h_ecarry;
    const uint64_t g0 = c->nseen;
    auto size_caps = [&]() { size_capacities(c, n); };
    bool ev3_done = false;
    auto edges_and_decode = [&]() -> int {
        size_caps();
        if (c->timing >= 2) HIPCHK(c, hipEventRecord(c->ev[2], c->st));
        if (c->use_small && n <= SM_MAX_SAMPLES) {   // a short batch: one launch for the three stages
            if (c->cert_pending) {   // the certification (and the end-of-batch state) first: the stage's launch is the last, and mirrors the state
                c->cert_pending = false;
                NFC_LAUNCH(k_certify, dim3(c->cert.blocks), dim3(256), 0, c->st, c->cert.A, c->cert.cert, (CertInfo *)nullptr,
                                   c->cert.ring_next, c->cert.carry, c->cert.sum);
            }
            const int r = run_small(c, n, skip, g0);
            if (r) return r;
            if (!ev3_done && c->timing >= 2) HIPCHK(c, hipEventRecord(c->ev[3], c->st));
            ev3_done = true;
            return NFC_OK;
        }
        int r = run_edges(c, n, skip, g0);
        if (r) return r;
        if (!ev3_done && c->timing >= 2) HIPCHK(c, hipEventRecord(c->ev[3], c->st));
        ev3_done = true;
        return run_decode(c);   // (its last launch mirrors the state block)
    };
    const std::function<int()> ahead = edges_and_decode;
    bool clean = false;
    int rc = run_threshold(c, d_in, n, skip, want_edges ? &ahead : nullptr, &clean);
    if (rc) return rc;
    if (want_edges) {
        for (int attempt = 0;; attempt++) {
            if (attempt > 0 || !clean) {
                rc = edges_and_decode();
                if (rc) return rc;
                HIPCHK(c, hipStreamSynchronize(c->st));
            }
            BATCHCHK(c, true);
            uint32_t ne, ns[2];
            memcpy(&ne, c->hs->totals + TOT_EDGES, 4);
            memcpy(ns, c->hs->totals + TOT_NSYM, 8);
            const bool fit = ne <= c->cap_edges && ns[0] + 2 <= c->cap_sym[0] && ns[1] + 2 <= c->cap_sym[1];
            if (fit) {
                c->n_edges = ne;
                c->n_sym[0] = ns[0];
                c->n_sym[1] = ns[1];
                break;
            }
            // a buffer was too small: the stages read carried values by value and wrote only write-only slots, so
            // they can simply run again with room for what was counted
            if (attempt >= 3) return fail(c, NFC_ERR_INTERNAL, "edge / symbol capacity did not settle");
            c->cap_edges_floor = (uint64_t)std::max(ne, c->cap_edges) * 5 / 4 + 65536;
            c->cap_sym_floor[0] = (uint64_t)ns[0] * 5 / 4 + 65536;
            c->cap_sym_floor[1] = (uint64_t)ns[1] * 5 / 4 + 65536;
        }
        update_estimates(c, n);
        c->pend_cur = 1 - c->pend_cur;
    } else {
        if (c->timing >= 2) {
            HIPCHK(c, hipEventRecord(c->ev[2], c->st));
            HIPCHK(c, hipEventRecord(c->ev[3], c->st));
        }
        HIPCHK(c, mirror_async(c));
        HIPCHK(c, hipStreamSynchronize(c->st));
        BATCHCHK(c, true);
    }
    if (c->timing >= 2) {
        HIPCHK(c, hipEventRecord(c->ev[4], c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
    }
    {
        adopt_mirror(c);
        if (c->P.flags & NFC_FLAG_NO_EDGES) c->h_ecarry = ecarry_in;
        else {
            uint64_t pk[2];
            memcpy(&pk[0], c->hs->totals + TOT_PKT0, 8);
            memcpy(&pk[1], c->hs->totals + TOT_PKT1, 8);
            for (int t = 0; t < 2; t++) {
                c->n_bits[t] = (uint32_t)pk[t];
                c->n_close[t] = (uint32_t)(pk[t] >> 32);
            }
        }
    }
    if (c->timing >= 2) {
        c->stats.ms_total = elapsed_ms(c->ev[0], c->ev[4]);
        c->stats.ms_threshold = elapsed_ms(c->ev[1], c->ev[2]);
        c->stats.ms_edges = elapsed_ms(c->ev[2], c->ev[3]);
        c->stats.ms_decode = elapsed_ms(c->ev[3], c->ev[4]);
    }
    for (int i = 0; i < c->n_kev; i++) c->stats.ms_threshold_kernel[i] = elapsed_ms(c->kev[2 * i], c->kev[2 * i + 1]);
    c->stats.n_threshold_timed = (uint32_t)c->n_kev;
    c->nseen += n;
    c->last_in = d_in;
    c->have_outputs = true;
    // (the end-of-batch LOW bookkeeping on the device is what a batch submitted ahead may start from: only when the whole
    // batch went through one parallel attempt, whose last certification workgroup or k_finalize_state wrote it)
    c->low_valid = skip == 0 && c->stats.used_sequential == 0 && c->h_carry.stable;
    return NFC_OK;
}

// ---------------------------------------------------------------------------
// batches submitted ahead (nfc_submit_device / nfc_wait)
// ---------------------------------------------------------------------------
// The threshold stage of batch k + 1 needs nothing the edge and decode stages of batch k produce -- the window, the sums
// and the LOW bookkeeping it starts from are left on the device by batch k's own threshold stage -- so it is enqueued on
// a second stream as soon as it is submitted and runs beside them (it leaves two thirds of a SIMD's issue slots idle; they
// are bound by exactly those).  Its edge and decode stages take the carried values of theirs BY VALUE from the host
// mirror, so they are enqueued when batch k has been waited for -- at the next submit / wait call, not in nfc_wait
// itself, so that the outputs of batch k stay readable in between.  Everything optimistic is checked in nfc_wait (the
// certification verdict, the exactness guard, the capacities, the stamps of both mirrors); a batch that fails any check
// is simply processed again by the synchronous path from the state before it (host mirrors + the third window buffer),
// and the batch submitted behind it, which started from a state that does not stand, is enqueued again.
// Not for long windows: with a 40 KB ring per wave the threshold kernel holds ALL of a CU's LDS, the other stages'
// workgroups cannot start beside it, and the two streams only get in each other's way (configs[3], 1e9 samples: 3.8 ms per
// batch submitted ahead against 2.7 ms one after the other) -- such batches take the synchronous path inside nfc_wait.
bool submit_fast_ok(const nfc_ctx *c, uint32_t n) {
    return c->h_carry.stable && !(c->P.flags & (NFC_FLAG_NO_EDGES | NFC_FLAG_FORCE_SEQUENTIAL)) && c->L >= STEP && c->timing < 2 &&
           !(c->use_small && n <= SM_MAX_SAMPLES) && n > 0 && c->lean_lds_per_cu <= 96 * 1024 && !getenv("NFC_DEBUG") && !getenv("NFC_DEBUG_CLK") && !getenv("NFC_NO_SUBMIT_AHEAD");
}

// the threshold stage of a submitted batch, on st_a, into a free set of planes; b.fast is cleared when the batch turns out
// not to qualify
int enqueue_threshold_ahead(nfc_ctx *c, nfc_ctx::Submitted &b) {
    if (b.planes < 0) {
        if (!c->alt_free) return fail(c, NFC_ERR_INTERNAL, "no free set of planes");
        b.planes = __builtin_ctz(c->alt_free);
        c->alt_free &= ~(1u << b.planes);
    }
    ThrArgs A;
    ThrPlan P;
    const EdgeCarry unused{0, 0, 0, 0};
    const nfc_stats keep_stats = c->stats;   // (the context's statistics are those of the last completed batch until this one is)
    const int rc_prep = thr_prepare(c, b.d_in, b.n, b.n, 0u, 0u, b.g0, unused, b.ring_in, c->d_neg_alt[b.planes], c->d_pos_alt[b.planes], true, A, P);
    c->stats = keep_stats;
    if (rc_prep) return rc_prep;
    if (!P.lean_applies || c->gring || P.nch < 2) {   // (the general kernel's passes keep the synchronous path)
        b.fast = false;
        c->alt_free |= 1u << b.planes;
        b.planes = -1;
        return NFC_OK;
    }
    b.nch = P.nch;
    b.chunk = (uint32_t)c->C;
    hipStream_t keep = c->st;
    c->st = c->st_a;
    c->batch_seq = b.seq;
    launch_fill_kind(c, b.d_in, b.n, (int)P.nch, b.ring_in);
    A.list = nullptr;
    A.nlist = 0;
    A.mode = 0;
    A.cert = P.d_cert;
    A.sum = (CertSummary *)(dT(c) + TOT_CERT);
    A.ksteps = c->lean_rounds;
    A.gfac = c->lean_gfac;
    A.gfloor = c->lean_gmin;
    A.blk = 1 << c->nfold;
    b.timed = b.timing >= 1;
    launch_threshold_kind(c, A, P.nch, true, b.timed ? c->kev_sub[b.slot] : nullptr);
    const uint32_t np = P.nch - 1;
    A.nlist = np;
    NFC_LAUNCH(k_certify, dim3((np + 3) / 4 + 1), dim3(256), 0, c->st, A, P.d_cert, (CertInfo *)nullptr,
               c->d_ring[(b.ring_in + 1) % NRING].as<float>(), dC(c), A.sum);
    hipError_t e = hipMemcpyAsync(c->hs_a[b.slot], c->d_state.p, sizeof(DevState), hipMemcpyDeviceToHost, c->st);
    if (e == hipSuccess) e = hipEventRecord(c->ev_a[b.slot], c->st);
    c->st = keep;
    if (e != hipSuccess) return fail(c, NFC_ERR_DEVICE, "submitting the threshold stage failed: %s", hipGetErrorString(e));
    return NFC_OK;
}

// The planes of the oldest submitted batch become the context's; the retired set goes back to the pool.
void take_planes(nfc_ctx *c, nfc_ctx::Submitted &b) {
    std::swap(c->d_neg, c->d_neg_alt[b.planes]);
    std::swap(c->d_pos, c->d_pos_alt[b.planes]);
    c->alt_free |= 1u << b.planes;
    b.planes = -1;
}

// its edge and decode stages, on st behind its threshold stage; from here on the context's per-batch fields are this batch's
int enqueue_stages_behind(nfc_ctx *c, nfc_ctx::Submitted &b) {
    HIPCHK(c, hipStreamWaitEvent(c->st, c->ev_a[b.slot], 0));
    if (b.planes >= 0) take_planes(c, b);
    c->have_outputs = false;
    c->pk_ready[0] = c->pk_ready[1] = false;
    c->n_edges = 0;
    for (int t = 0; t < 2; t++) c->n_sym[t] = c->n_close[t] = c->n_bits[t] = 0;
    memset(&c->stats, 0, sizeof c->stats);
    c->n_kev = 0;
    c->last_n = b.n;
    c->last_g0 = b.g0;
    c->last_skip = 0;
    c->stats.bytes_in = (uint64_t)b.n * c->in_bytes_per_sample;
    c->stats.n_chunks = b.nch;
    c->stats.chunk_samples = b.chunk;
    c->stats.threshold_passes = 1;
    c->stats.ran_ahead = 1;
    c->stamp_b = b.seq;
    c->cert_pending = false;
    size_capacities(c, b.n);
    int rc = run_edges(c, b.n, 0u, b.g0);
    if (!rc) rc = run_decode(c);   // (its last launch mirrors the state block and stamps it)
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ev_b[b.slot], c->st));
    b.b_enqueued = true;
    return NFC_OK;
}

// (Re)start every submitted batch from the context's current state: after the batch before them went through the
// synchronous path, what they were enqueued on -- if they were -- does not stand.
int restart_submitted(nfc_ctx *c) {
    uint64_t g0 = c->nseen;
    int ring = c->ring_cur;
    bool fast = c->low_valid && !c->state_dirty;
    for (int i = 0; i < c->sub_count; i++) {
        nfc_ctx::Submitted &nb = c->sub[i];
        nb.g0 = g0;
        nb.ring_in = ring;
        nb.b_enqueued = false;
        nb.fast = fast && submit_fast_ok(c, nb.n);
        if (nb.fast) {
            nb.seq = c->batch_seq + 1;
            if (int rc = enqueue_threshold_ahead(c, nb)) return rc;
        }
        if (!nb.fast && nb.planes >= 0) {
            c->alt_free |= 1u << nb.planes;
            nb.planes = -1;
        }
        fast = nb.fast;
        g0 += nb.n;
        ring = (ring + 1) % NRING;
    }
    return NFC_OK;
}

int submit_batch(nfc_ctx *c, const void *d_in, size_t n64) {
    if (c->sub_count == NSUB) return fail(c, NFC_ERR_STATE, "%d batches are in flight: nfc_wait for the oldest one first", NSUB);
    if (n64 > (1ull << 30)) return fail(c, NFC_ERR_ARG, "batch of %zu samples exceeds 2^30; push it in pieces", n64);
    if (n64 && ((uintptr_t)d_in & 15u) != 0) return fail(c, NFC_ERR_ARG, "device input must be 16-byte aligned");
    const uint32_t n = (uint32_t)n64;
    nfc_ctx::Submitted *oldest = c->sub_count ? &c->sub[0] : nullptr;
    nfc_ctx::Submitted *prev = c->sub_count ? &c->sub[c->sub_count - 1] : nullptr;
    nfc_ctx::Submitted b;
    b.d_in = d_in;
    b.n = n;
    b.slot = (int)(c->slot_next++ % (uint32_t)NSUB);
    b.timing = c->timing;
    if (prev) {
        b.g0 = prev->g0 + prev->n;
        b.ring_in = (prev->ring_in + 1) % NRING;
        b.fast = prev->fast && submit_fast_ok(c, n);   // (behind a batch that takes the synchronous path nothing is known yet)
    } else {
        b.g0 = c->nseen;
        b.ring_in = c->ring_cur;
        b.fast = c->low_valid && !c->state_dirty && submit_fast_ok(c, n);
    }
    // The oldest batch's edge / decode stages were held back while the outputs of the batch before it could be read; they
    // are enqueued now -- AFTER the new batch's threshold stage, whose stream is the one that must not run dry.  The planes
    // change hands first: the oldest batch's become the context's, the retired set is free for the new batch.
    const bool behind = oldest && oldest->fast && !oldest->b_enqueued;
    // (with a threshold stage already queued behind the running one that stream has work for a while: the held-back stages go first then)
    const bool stages_first = behind && c->sub_count >= 2;
    if (behind) take_planes(c, *oldest);
    if (stages_first)
        if (int rc = enqueue_stages_behind(c, *oldest)) return rc;
    if (b.fast) {
        b.seq = c->batch_seq + 1;
        if (!prev) launch_error() = LaunchError{};
        if (int rc = enqueue_threshold_ahead(c, b)) return rc;
    }
    if (behind && !stages_first)
        if (int rc = enqueue_stages_behind(c, *oldest)) return rc;
    c->sub[c->sub_count++] = b;
    return NFC_OK;
}

int wait_batch(nfc_ctx *c) {
    if (!c->sub_count) return fail(c, NFC_ERR_STATE, "nothing was submitted");
    nfc_ctx::Submitted b = c->sub[0];
    auto pop = [&]() {
        for (int i = 1; i < c->sub_count; i++) c->sub[i - 1] = c->sub[i];
        c->sub_count--;
    };
    auto abandon = [&]() {   // an error: nothing submitted stands
        c->sub_count = 0;
        c->alt_free = (1u << (NSUB - 1)) - 1u;
    };
    struct Scope {   // process_batch refuses to run beside submitted batches unless it is this function that calls it
        nfc_ctx *c;
        explicit Scope(nfc_ctx *c_) : c(c_) { c->in_wait = true; }
        ~Scope() { c->in_wait = false; }
    } scope(c);
    if (!b.fast) {
        pop();
        const int keep_timing = c->timing;
        c->timing = b.timing;
        const int rc = process_batch(c, b.d_in, b.n);
        c->timing = keep_timing;
        if (rc) {
            abandon();
            return rc;
        }
        return restart_submitted(c);   // the batches behind it can start now that their state is known
    }
    if (!b.b_enqueued) {
        if (int rc = enqueue_stages_behind(c, b)) {
            abandon();
            return rc;
        }
        c->sub[0] = b;
    }
    HIPCHK(c, hipEventSynchronize(c->ev_b[b.slot]));
    bool regular = true;
    const char *why = "";
    {
        LaunchError &le = launch_error();
        if (le.err != hipSuccess) {
            const LaunchError e = le;
            le = LaunchError{};
            abandon();
            return fail(c, NFC_ERR_DEVICE, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e.err), e.file, e.line);
        }
    }
    const DevState *sa = c->hs_a[b.slot];
    if (sa->seq[0] != b.seq || c->hs->seq[1] != b.seq) {
        abandon();
        return fail(c, NFC_ERR_DEVICE, "state mirror is stale (batch %u, mirrors %u / %u): a kernel of this batch did not run", b.seq, sa->seq[0], c->hs->seq[1]);
    }
    CertSummary summary;
    memcpy(&summary, sa->totals + TOT_CERT, sizeof summary);
    Carry after = sa->carry;
    carry_apply_fin(after);
    uint32_t ne, ns[2];
    memcpy(&ne, c->hs->totals + TOT_EDGES, 4);
    memcpy(ns, c->hs->totals + TOT_NSYM, 8);
    if (summary.n_fail != 0) regular = false, why = "a chunk was not certified";
    else if (summary.flagged || !sums_exact(after, (int)summary.emin, (int)summary.emax, summary.vtop)) regular = false, why = "sums not provably exact";
    else if (!(ne <= c->cap_edges && ns[0] + 2 <= c->cap_sym[0] && ns[1] + 2 <= c->cap_sym[1])) regular = false, why = "a capacity estimate was too small";
    if (getenv("NFC_DEBUG_REDO_SUBMITTED") && (c->dbg_fast_waits++ % 3u) == 2u) regular = false, why = "test hook";   // every third batch that ran ahead
    if (regular) {
        c->h_carry = after;
        c->h_ecarry = c->hs->ecarry;
        c->h_dcarry = c->hs->dcarry;
        c->n_edges = ne;
        c->n_sym[0] = ns[0];
        c->n_sym[1] = ns[1];
        update_estimates(c, b.n);
        c->pend_cur = 1 - c->pend_cur;
        uint64_t pk[2];
        memcpy(&pk[0], c->hs->totals + TOT_PKT0, 8);
        memcpy(&pk[1], c->hs->totals + TOT_PKT1, 8);
        for (int t = 0; t < 2; t++) {
            c->n_bits[t] = (uint32_t)pk[t];
            c->n_close[t] = (uint32_t)(pk[t] >> 32);
        }
        if (b.timed) {
            c->stats.ms_threshold_kernel[0] = elapsed_ms(c->kev_sub[b.slot][0], c->kev_sub[b.slot][1]);
            c->stats.n_threshold_timed = 1;
        }
        c->ring_cur = (b.ring_in + 1) % NRING;
        c->nseen = b.g0 + b.n;
        c->last_in = b.d_in;
        c->have_outputs = true;
        c->low_valid = true;
        pop();
        return NFC_OK;
    }
    // The optimistic result does not stand: everything in flight is drained, the batch goes through the synchronous path
    // from the state before it (the host mirrors were last adopted there; its window buffer was not written since), and
    // the batches behind it start again from what that leaves.
    if (getenv("NFC_TRACE")) fprintf(stderr, "[nfc] submitted batch %u processed again: %s\n", b.seq, why);
    HIPCHK(c, hipStreamSynchronize(c->st_a));
    HIPCHK(c, hipStreamSynchronize(c->st));
    c->stats_redo_submitted++;
    push_state(c);
    pop();
    const int keep_timing = c->timing;
    c->timing = b.timing;
    const int rc = process_batch(c, b.d_in, b.n);
    c->timing = keep_timing;
    if (rc) {
        abandon();
        return rc;
    }
    return restart_submitted(c);
}

int build_packets(nfc_ctx *c, int t) {
    if (c->pk_ready[t]) return NFC_OK;
    c->pk[t].clear();
    const uint32_t nc = c->n_close[t];
    if (nc) {
        std::vector<uint32_t> ends(nc);
        std::vector<uint64_t> idx(nc);
        HIPCHK(c, hipMemcpy(ends.data(), c->d_close_end[t].p, (size_t)nc * 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(idx.data(), c->d_close_idx[t].p, (size_t)nc * 8, hipMemcpyDeviceToHost));
        uint32_t prev = 0;
        for (uint32_t k = 0; k < nc; k++) {
            if (ends[k] > prev) {  // packets.py:97 -- empty lists never reach the fsm
                nfc_packet p;
                p.idx = idx[k];
                p.bit_off = prev;
                p.n_bits = ends[k] - prev;
                p.type = t;
                c->pk[t].push_back(p);
            }
            prev = ends[k];
        }
    }
    c->pk_ready[t] = true;
    return NFC_OK;
}

}  // namespace

// ===========================================================================
// C-ABI
// ===========================================================================
extern "C" {

int nfc_abi_version(void) { return NFC_AMD_ABI_VERSION; }

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
    c->use_small = getenv("NFC_NO_SMALL") ? 0 : 1;
    if (const char *e = getenv("NFC_LEAN")) c->lean = atoi(e) != 0;
    c->lean_k = 0;        // chosen below from the occupancy the LDS ring allows, unless set here
    c->lean_rounds = 0;
    if (const char *e = getenv("NFC_LEAN_K")) c->lean_k = atoi(e) <= 2 ? 2 : 4;   // (steps ahead: the two instantiations)
    if (const char *e = getenv("NFC_LEAN_ROUNDS")) c->lean_rounds = std::max(1, atoi(e));
    if (const char *e = getenv("NFC_LEAN_GFAC")) c->lean_gfac = (float)atof(e);
    if (const char *e = getenv("NFC_LEAN_GMIN")) c->lean_gmin = (float)atof(e);
    if (const char *e = getenv("NFC_OWN_PREFIX_MAX")) c->own_prefix_max = (uint32_t)strtoul(e, nullptr, 10);
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
    if (const char *e = getenv("NFC_RING")) {
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
        c->wave_slots = std::max(1, prop.multiProcessorCount * std::max(1, per_cu));
        // lean kernel: four steps ahead (104 registers: at most four waves per SIMD); a superstep long enough that its fixed cost
        // fades -- the drift allowance grows with its length relative to the window (about half the window at most: beyond, the
        // widened bands reach the loaded half bits of tag frames)
        // Measured (configs[1] / [2], 1e8 samples): two waves per SIMD with four steps ahead beat five waves with three --
        // 0.206 / 0.196 ms against 0.237 / 0.230 on the same box: a chunk's fixed cost (the window before it read and
        // summarised, its summary written) is paid half as often, and two waves already keep a SIMD's issue slots busy.
        if (!c->lean_k) c->lean_k = 4;
        c->lean_slots = std::min(c->wave_slots, prop.multiProcessorCount * 8);
        if (const char *e = getenv("NFC_LEAN_WAVES")) c->lean_slots = std::min(c->wave_slots, prop.multiProcessorCount * 4 * std::max(1, atoi(e)));
        if (!c->lean_rounds) c->lean_rounds = std::max(1, (int)(0.4 * c->L / (256.0 * c->lean_k)));
        c->wave_slots_g = prop.multiProcessorCount * 20;   // VGPR-bound: five waves per SIMD
        // LDS the lean kernel's resident waves hold per CU: a batch is only run ahead of its predecessor's edge / decode stages
        // (nfc_submit_device) while those stages' workgroups (25 KB each) still fit beside it
        c->lean_lds_per_cu = (size_t)((c->lean_slots + prop.multiProcessorCount - 1) / prop.multiProcessorCount) * lds_wave;
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
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_IQ_F32, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_IQ_F32, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_IQ_F32, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_IQ_F32, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_REAL_F32_SQ, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_REAL_F32_SQ, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_REAL_F32_SQ, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_REAL_F32_SQ, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_I16_SQ, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CRT(hipFuncSetAttribute((const void *)k_threshold_lean<IN_I16_SQ, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
    CRT(c->d_mil_map.ensure(milb.size()));
    CRT(c->d_man_map.ensure(manb.size()));
    CRT(c->d_mil_out.ensure(mils.size() * 2));
    CRT(c->d_man_out.ensure(mans.size() * 2));
    CRT(hipMemcpy(c->d_mil_map.p, milb.data(), milb.size(), hipMemcpyHostToDevice));
    CRT(hipMemcpy(c->d_man_map.p, manb.data(), manb.size(), hipMemcpyHostToDevice));
    CRT(hipMemcpy(c->d_mil_out.p, mils.data(), mils.size() * 2, hipMemcpyHostToDevice));
    CRT(hipMemcpy(c->d_man_out.p, mans.data(), mans.size() * 2, hipMemcpyHostToDevice));
    c->T.mil_map = c->d_mil_map.as<uint4>();
    c->T.man_map = c->d_man_map.as<uint2>();
    c->T.mil_step = c->d_mil_out.as<uint16_t>();
    c->T.man_step = c->d_man_out.as<uint16_t>();
    c->T.nd = c->mx + 1;
    c->T.reader = p->enable_reader ? 1 : 0;
    c->T.tag = p->enable_tag ? 1 : 0;
    // carried state
    CRT(c->d_state.ensure(sizeof(DevState)));
    CRT(hipHostMalloc((void **)&c->hs, sizeof(DevState), hipHostMallocMapped));
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
    DevBuf *all[] = {&c->d_mil_map, &c->d_man_map, &c->d_mil_out, &c->d_man_out, &c->d_state,
                     &c->d_ring[0], &c->d_ring[1], &c->d_ring[2], &c->d_ring[3], &c->d_neg_alt[0], &c->d_pos_alt[0], &c->d_neg_alt[1], &c->d_pos_alt[1], &c->d_certinfo, &c->d_in, &c->d_neg, &c->d_pos, &c->d_ringin, &c->d_meta, &c->d_ringout[0], &c->d_ringout[1], &c->d_touched[0],
                     &c->d_touched[1], &c->d_info[0], &c->d_info[1], &c->d_ver, &c->d_cflags, &c->d_list, &c->d_ecode, &c->d_epos, &c->d_eidx, &c->d_states, &c->d_sym[0], &c->d_sym[1],
                     &c->d_bits[0], &c->d_bits[1], &c->d_pending[0][0], &c->d_pending[0][1],
                     &c->d_pending[1][0], &c->d_pending[1][1], &c->d_partials2, &c->d_close_end[0], &c->d_close_end[1], &c->d_close_idx[0], &c->d_close_idx[1],
                     &c->d_partials, &c->d_aggs, &c->d_faggs, &c->d_gring, &c->d_pack, &c->d_gvtop, &c->d_seqout};
    for (DevBuf *b : all) b->release();
    if (c->hs) (void)hipHostFree(c->hs);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_edge_stage) (void)hipHostFree(c->h_edge_stage);
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
    if (!c) return NFC_ERR_ARG;
    if (n && !dev_samples) return fail(c, NFC_ERR_ARG, "null input");
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    return process_batch(c, dev_samples, n);
}

int nfc_submit_device(nfc_ctx *c, const void *dev_samples, size_t n) {
    if (!c) return NFC_ERR_ARG;
    if (n && !dev_samples) return fail(c, NFC_ERR_ARG, "null input");
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    return submit_batch(c, dev_samples, n);
}

int nfc_wait(nfc_ctx *c) {
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
    const int rc = run_decode(c);   // k_dec_reduce -> k_dec_apply -> k_frame_write -> k_pkt_finish (mirrors the state block)
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->st));
    BATCHCHK(c, false);
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
    if (!c) return NFC_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->st));
    return NFC_OK;
}

int nfc_set_stream(nfc_ctx *c, void *stream) {
    if (!c) return NFC_ERR_ARG;
    NOSUB(c);
    HIPCHK(c, hipStreamSynchronize(c->st));   // nothing of this context is left on the stream it leaves
    c->st = stream ? (hipStream_t)stream : c->own_st;
    return NFC_OK;
}

int nfc_get_counts(nfc_ctx *c, nfc_counts *out) {
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
    if (!c) return NFC_ERR_ARG;
    size_t n;
    int rc = edge_range(c, first, pos_out, cap, n_out, n);
    if (rc || !n) return rc;
    if (!code_out) return fail(c, NFC_ERR_ARG, "null output");
    if (c->edges_from_host) return fail(c, NFC_ERR_STATE, "the entries of nfc_push_edges carry the caller's indices: use nfc_read_edges");
    return fetch_entries(c, first, n, [&](const uint32_t *pos, const uint16_t *code, size_t cnt, size_t off) {
        memcpy(pos_out + off, pos, cnt * 4);
        memcpy(code_out + off, code, cnt * 2);
    });
}

int nfc_read_symbols(nfc_ctx *c, int type, size_t first, uint8_t *out, size_t cap, size_t *n_out) {
    if (!c || type < 0 || type > 1) return NFC_ERR_ARG;
    return read_range(c, c->d_sym[type].p, c->n_sym[type], 1, first, out, cap, n_out);
}

int nfc_read_packets(nfc_ctx *c, int type, nfc_packet *out, size_t cap, size_t *n_out) {
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
    if (!c || type < 0 || type > 1) return NFC_ERR_ARG;
    return read_range(c, c->d_bits[type].p, c->n_bits[type], 1, first, out, cap, n_out);
}

int nfc_read_val(nfc_ctx *c, size_t first, int8_t *out, size_t cap, size_t *n_out) {
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
    if (!c || level < 0 || level > 2) return NFC_ERR_ARG;
    c->timing = level;
    return NFC_OK;
}

int nfc_reset(nfc_ctx *c) {
    if (!c) return NFC_ERR_ARG;
    NOSUB(c);
    if (hipSetDevice(c->P.device) != hipSuccess) return fail(c, NFC_ERR_DEVICE, "hipSetDevice failed");
    init_carried(c);
    c->have_outputs = false;
    push_state(c, 0, true);   // carried values and a zeroed window in one launch
    return NFC_OK;
}

int nfc_prime(nfc_ctx *c, uint64_t start_index, float level) {
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
    c->h_dcarry.mil_state = h->miller_state;
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
    if (!c || !out) return NFC_ERR_ARG;
    *out = c->stats;
    out->redone_total = c->stats_redo_submitted;
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
    if (!p || !cur || !d || !n_out || type < 0 || type > 1) return NFC_ERR_ARG;
    if (!(p->samp_rate > 0) || p->max_len < 1) return NFC_ERR_ARG;
    const DecoderTables t = build_tables(p->samp_rate, p->max_len);
    const int nd = p->max_len + 1;
    int state = type ? 0 : ((0 + 1) << 1);
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (cur[i] < -1 || cur[i] > 2 || d[i] < 0 || d[i] > p->max_len) return NFC_ERR_ARG;
        const int li = (cur[i] + 1) * nd + d[i];
        uint8_t w;
        if (type) {
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

}  // extern "C"
