// host_context.h -- the context (nfc_ctx), the mirrored state block, launch helpers
// (part of nfc_amd.hip: included there, in this order, into one translation unit)
#pragma once


namespace {

std::string g_create_error;

// (the round-5 form of the tail: the certification shares the edge writer's launch -- kept beside k_tail for the A/B)
__global__ __launch_bounds__(256) void k_certify_and_write(CertLaunch C, EdgeArgs E, size_t nwords, const EdgeAgg *partials, const EdgeAgg *supers,
                                                          uint32_t *epos, uint16_t *ecode, uint32_t cap, bool own_prefix, uint32_t *total_out,
                                                          Last2 *last2_total, EdgeCarry *carry_out) {
    const uint32_t tiles = gridDim.x - C.blocks;
    if (blockIdx.x == 0) {
        certify_block(C.A, C.cert, nullptr, C.ring_next, C.carry, C.sum, C.blocks - 1, C.blocks);
        return;
    }
    if (blockIdx.x > tiles) {
        certify_block(C.A, C.cert, nullptr, C.ring_next, C.carry, C.sum, blockIdx.x - tiles - 1, C.blocks);
        return;
    }
    write_edges_tile(E, nwords, partials, supers, epos, ecode, cap, own_prefix, total_out, last2_total, carry_out, tiles - blockIdx.x, tiles);
}

// (re)allocations of device buffers by the calling thread: a batch's share is nfc_stats.device_allocs -- a stream in its steady state
// must show 0 (an allocation in the middle of a stream costs milliseconds: VERDICT r4, the hovering stream's second batch)
inline uint64_t &devbuf_allocs() {
    static thread_local uint64_t n = 0;
    return n;
}
inline bool &devbuf_trace() {   // (test build: NFC_TRACE_ALLOC names every (re)allocation on stderr)
    static bool on = false;
    return on;
}
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes, bool keep = false, hipStream_t st = nullptr) {
        if (bytes <= cap) return hipSuccess;
        devbuf_allocs()++;
        if (devbuf_trace()) fprintf(stderr, "[nfc] device buffer %p: %zu -> %zu bytes asked for\n", (void *)this, cap, bytes);
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
    TOT_SPEC = 40,      // u32: verdict of the speculative decode's check (decode.hip.h: dec_verify), 0 = every assumption that mattered held
    TOT_PKT0 = 72,      // PktCnt: per type, bits | closes << 32
    TOT_PKT1 = 80,
    TOT_LAST2 = 88,     // Last2 (8 bytes)
    TOT_NSYM = 96,      // u32[2]: symbols per packet type
    TOT_CERT = 104,     // CertSummary (24 bytes)
    TOT_FRAME = 128,    // FrameAgg (32 bytes): symbol counts, framing maps, bit / close counts of the batch
    TOT_BYTES = 160
};

// Everything the host mirrors after a batch, in one block so that one copy fetches it.
static_assert(sizeof(CertSummary) <= TOT_FRAME - TOT_CERT, "the verdict summary fits its slot of the totals");
struct DevState {
    Carry carry;
    EdgeCarry ecarry;
    DecCarry dcarry;
    uint8_t totals[TOT_BYTES];
    uint32_t seq[4];   // seq[0]: the batch the block belongs to, stamped by the batch's first kernel (k_fill): a mirror that
                       // does not carry the current number was not written by this batch's kernels
};

// what the threshold kernel of a batch submitted ahead may hold of a CU's 160 KB of LDS: the later stages of the batch before it run
// beside it (the writer stages 12 KB per workgroup, the speculative decode 25 KB)
constexpr size_t AHEAD_LDS_MAX = 96 * 1024;
constexpr int NRING = 4;   // window buffers: the carried one + one per batch that may be in flight (they rotate)
constexpr int NSUB = 3;    // batches that may be submitted and not yet waited for

struct nfc_ctx {
    nfc_params P;
    int L, mx, C, Lpad, wpb, twords;
    double factor;
    double hi_plus, lo_a, lo_b, hi_a, hi_b;
    int bands_ok, fast_ok, nfold, rows_per_step, C_min, wave_slots, lds_per_slot;
    uint32_t ring_carried = 0;   // nfc_stats.ring_slots_carried of the last batch adopted
    // the decode stage's speculative form (decode.hip.h: k_dec_spec): on unless NFC_DEC_SPEC=0; run-in edges per thread (2 / 4 / 8);
    // after a batch whose check failed the next batches take the three-launch form (spec_off_left counts them down)
    bool dec_spec = true, dec_spec_now = false;
    int dec_runin = 2, spec_off_left = 0, spec_fail_streak = 0;   // (streak: speculative attempts that failed in a row: the back-off doubles)
    // what the decode stage left for the readers: packet bits packed 32 to a word (the multi-launch stage) or a byte each (short
    // batches); symbol arrays not written yet (materialize_symbols: on the first nfc_read_symbols)
    bool bits_packed = false, sym_lazy = false, sym_own = false;
    FrameOut sym_P;
    uint32_t sym_n = 0, sym_tiles = 0;
    uint32_t decode_respeculated = 0;   // batches whose decode stage was repeated with the three-launch form (nfc_stats)
    int lean = 1, lean_k = 0, lean_rounds = 0, lean_slots = 0;   // pass 0 by k_threshold_lean (NFC_LEAN=0 turns it off), steps per superstep (NFC_LEAN_K)
    int n_cus = 1;                                             // compute units of the device
    // ... the rows' chunk lengths over the equal cut's, all but the last row's, by workgroups per CU (measured: host_threshold.h); [0]: NFC_WG_ROWBAL=a,b,c
    double rowbal_f[5][3] = {{1, 1, 1}, {1, 1, 1}, {1.02, 1, 1}, {1.045, 1.004, 1}, {1.036, 1.015, 0.990}};
    bool rowbal_set = false;
    bool wg_rowbal = true, rowbal_now = false;                 // chunks cut by dispatch row (host_threshold.h: thr_prepare; NFC_WG_ROWBAL=0: the equal cut); this batch is
    int wg = 1, wg_ok = 0, wg_nr = 4, wg_slots = 0, wg_slots_ahead = 0, wg_now = 0, wg_rounds = 0;
    int fine_left = 0, fine_adapt = 1, fine_mult = 4;   // batches still to be cut into fine_mult times as many chunks (after a batch that needed re-runs); NFC_CHUNK_ADAPT=0 turns it off   // pass 0 by k_threshold_wg (a chunk per workgroup; NFC_WG=0 turns it off), rounds
                                                                 // rows of 64 samples per step (NFC_WG_NR), resident workgroups, this batch uses it, rounds per superstep
    size_t wg_lds = 0, wg_lds_base = 0, wg_lds_bulk_max = 0;   // dynamic LDS of k_threshold_wg: with the staging ring / without any staging / the most a whole chunk's planes may bring it to
    bool wg_bulk = true, wg_bulk_now = false;                  // a chunk's plane words leave when the chunk is done (NFC_WG_BULK=0: always the ring); this launch may
    int wg_lone_max = 4, wg_lone_div = 64;   // ... how many failing chunks still count as lone: at most this many, and at most one in wg_lone_div
    int wg_rerun_lone = 1;   // a lone failing chunk of a clean batch is re-run by k_threshold_wg, gave up or not (host_threshold.h; NFC_WG_RERUN=0 in the test build: never)
    // re-runs by k_threshold_wg<KIND, 4, true> (failed rounds evaluated in place): where it applies, its LDS, up to how many failing chunks
    // of a round take it (more: k_threshold, whose one wave per chunk fills the machine where every chunk fails), the launch in progress
    int wg_ex_ok = 0, wg_ex_max = 0;
    size_t wg_ex_lds = 0;
    bool wg_ex_launch = false;
    bool wg_flags = false;   // test build: pass 0 by the instantiations with per-wave counters instead of a round's first barrier
    float lean_gfac = 1.3f, lean_gmin = 9.765625e-4f;   // drift allowance of the next superstep: max(gfac * B, gmin * ss)
    int gring = 0;   // this batch: the ring of a chunk in global memory instead of LDS
    int gring_ok = 0, gring_force = 0, wave_slots_g = 0;   // long windows qualify (NFC_RING=lds|global overrides the choice)
    DevBuf d_gring;
    uint32_t own_prefix_max = OWN_PREFIX_MAX_TILES;   // tile counts up to this need no prefix launches (NFC_OWN_PREFIX_MAX overrides)
    int use_small = 1;   // short batches take the one-launch edge / decode / framing kernel (NFC_NO_SMALL=1 turns it off)
    double expect_ms = 0.3;   // how long the stamp of a batch has taken to appear lately (wait_for_stamp)
    bool spin_wait = true;   // a batch's end is seen in the mirror's stamp word, not waited for on the stream (NFC_SPIN_WAIT=0; host_threshold.h)
    uint64_t selmask;
    float eps;               // certification margin of the speculative pass, relative to the window sum (1 %; host_threshold.h: eps_adapt says why it stays there)
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
    // debugging switches, read ONCE in nfc_create (an inherited environment must not reach the per-launch path)
    bool dbg_bad_launch = false, dbg_redo_submitted = false, dbg_no_submit_ahead = false, dbg_any = false, dbg_clk = false, dbg_trace = false;
    std::string dbg_clk_path;
    uint32_t batch_seq = 0;   // stamped into the state block by every batch's first kernel, checked in the mirror
    int timing = 0;   // 0: no events, 1: the threshold kernels' own start / stop events, 2: + batch total and stages as stream markers (nfc_set_timing)
    hipEvent_t kev[2 * 6] = {};  // start/stop pairs around the first k_threshold launches of a batch
    int n_kev = 0;
    std::string err;
    LaunchError launch_err;   // the first launch of the batch in work that the runtime rejected (launch_check.h)

    // tables
    DevBuf d_mil_map, d_man_map, d_mil_out, d_man_out, d_qmil_map, d_qmil_step;
    uint8_t mil_q_of[16], mil_canon[16];   // Miller state -> class of the quotient machine (0xFF: none) / -> canonical state (decoder_tables.h)
    DecTables T;

    // carried state
    DevBuf d_state, d_ring[NRING];   // the window: the carried one, the one the batch in work writes, and -- with batches submitted
                                     // ahead (nfc_submit_device) -- the ones THOSE write; they rotate
    DevState *hs = nullptr;        // pinned host mirror of d_state
    void *hs_dev = nullptr;        // the same memory as the device addresses it (kernels may fill the mirror themselves)
    uint8_t *h_stage = nullptr;    // pinned staging for nfc_get_state
    size_t h_stage_cap = 0;
    uint8_t *h_pk_stage = nullptr;     // pinned staging for the packet tables (build_packets)
    size_t h_pk_stage_cap = 0;
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
        uint32_t allocs = 0;   // buffers (re)allocated on its behalf so far (nfc_stats.device_allocs)
        bool fast = false, b_enqueued = false, timed = false, spec = false, tail = false;   // (spec: its decode stage ran in the speculative form)
    } sub[NSUB];
    int sub_count = 0;             // batches submitted and not yet waited for (sub[0] the oldest)
    uint32_t slot_next = 0;
    bool low_valid = false;        // Carry.low_nl / low_kl on the device describe the end of the last completed batch
    size_t lean_lds_per_cu = 0;
    size_t ahead_lds_per_cu = 0;   // LDS the threshold kernel of a batch submitted ahead holds per CU (the kernel that actually runs)
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
    // the fused tail (tail.hip.h: k_tail): status words of its look-backs, the ticket counter, the launch epoch; the packed bit arrays
    // alternate between two buffers -- a batch's launch clears the one the NEXT batch's tiles will or into
    DevBuf d_tail_st, d_tail_ticket, d_bits_alt[2];
    uint32_t tail_epoch = 0, tail_ticket_base = 0, tail_seq = 0xFFFFFFFFu;
    bool tail_on = false, tail_reset = true, tail_now = false;   // (tail_on: NFC_TAIL=1 in the test build)
    uint32_t tail_tw_used = 512, tail_dense_repeats = 0, tail_peak = 0, tail_peak_tw = 512;
    bool sym_from_tail = false;            // the last batch's out-bytes came from k_tail: the symbol reader builds its tile aggregates first
    uint32_t tail_tw = 512;         // words per tile: halves when a tile's entries did not fit the staging, grows back after calm batches
    int tail_tw_hold = 0;
    uint32_t bits_clean[2] = {0, 0}, alt_clean[2] = {0, 0};   // words of d_bits / d_bits_alt known to be zero
    int tail_occ[2] = {0, 0};                // resident workgroups per CU of k_tail<false / true>
    DevBuf d_spec;                           // per decode tile: its map, the state it assumed (decode.hip.h: DecSpec)
    DevBuf d_stage_bits[2], d_stage_cb[2], d_stage_ci[2], d_stage_q[2], d_stage_own;   // ... and what it stages for k_concat (TileStage)
    DevBuf d_pack;                           // nfc_get_state staging
    DevBuf d_gvtop;                          // per chunk: bound of the ring values (guard of the fp64 sums)
    DevBuf d_seqout;                         // sequential kernel: edge-timing state after its last sample
    uint32_t cap_edges = 0, cap_sym[2] = {0, 0};   // capacity estimates of the edge / symbol buffers
    // what the buffers behind those estimates are ALLOCATED for: at least the estimates, and room for a quarter of an entry per
    // sample (four times a clean capture's density; hovering load modulation doubles it) -- the estimates size the grids and follow
    // the stream, the allocations are made once per batch length (size_capacities)
    uint32_t alloc_edges = 0, alloc_sym[2] = {0, 0};
    uint64_t alloc_mark = 0;                       // devbuf_allocs() when the batch in work began
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
    if (c) c->err = buf, c->tail_reset = true;   // (after any failure the fused tail's ticket counter is re-armed: tail.hip.h)
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
// ... or, where it applies, with a chunk per workgroup (threshold_wg.hip.h)
template <int KIND>
void launch_wg(nfc_ctx *c, const ThrArgs &A, uint32_t nwork, hipEvent_t e0, hipEvent_t e1) {
    // the staging of the plane words (threshold_wg.hip.h): the whole chunk's when this launch may and the LDS has room, else the ring
    ThrArgs B = A;
    size_t lds = c->wg_lds;
    B.wg_stage_rounds = 2 * wg_flush_rounds(c->wg_nr);
    {
        const int rounds = A.C_max / wg_round_samples(c->wg_nr) + 2;   // (the longest chunk of the cut: thr_prepare)
        const size_t need = c->wg_lds_base + wg_stage_bytes(c->wg_nr, rounds);
        if (c->wg_bulk_now && c->wg_lds_bulk_max && need <= c->wg_lds_bulk_max) {
            B.wg_stage_rounds = rounds;
            lds = need;
        }
    }
    if (c->wg_ex_launch) {   // (re-runs that evaluate failed rounds in place: four rows per step whatever pass 0 ran with)
        B.wg_stage_rounds = 2 * wg_flush_rounds(4);
        lds = c->wg_ex_lds;
    }
    if (c->dbg_bad_launch) lds += (size_t)1 << 20;
    auto go = [&](auto kern) {
        if (e0) NFC_LAUNCH_EXT(kern, dim3(nwork), dim3(256), lds, c->st, e0, e1, 0, B);
        else NFC_LAUNCH(kern, dim3(nwork), dim3(256), lds, c->st, B);
    };
    if (c->wg_ex_launch) return go(k_threshold_wg<KIND, 4, true>);
#ifdef NFC_TEST_HOOKS
    if (c->wg_flags) {   // (NFC_WG_FLAGS=1: per-wave counters instead of a round's first barrier -- built and measured, threshold_wg.hip.h)
        if constexpr (KIND == IN_IQ_F32 || KIND == IN_ENV_F32) {
            if (c->wg_nr == 8) return go(k_threshold_wg<KIND, 8, false, true>);
        }
        return go(k_threshold_wg<KIND, 4, false, true>);
    }
#endif
    if constexpr (KIND == IN_IQ_F32 || KIND == IN_ENV_F32) {   // (the kinds eight rows per step are instantiated for: nfc_create chooses wg_nr)
        if (c->wg_nr == 8) return go(k_threshold_wg<KIND, 8>);
    }
    go(k_threshold_wg<KIND, 4>);
}
template <int KIND>
void launch_lean(nfc_ctx *c, const ThrArgs &A, uint32_t nwork, hipEvent_t e0, hipEvent_t e1) {
    if (c->wg_now) return launch_wg<KIND>(c, A, nwork, e0, e1);
    const uint32_t wpb = (uint32_t)c->wpb;
    const uint32_t blocks = (nwork + wpb - 1) / wpb;
    // (NFC_DEBUG_BAD_LAUNCH: a dynamic-LDS request the runtime must reject -- the test of the launch checks)
    const size_t lds = (size_t)wpb * c->Lpad * c->lds_per_slot + (c->dbg_bad_launch ? (size_t)1 << 20 : 0);
    auto go = [&](auto kern) {
        if (e0) NFC_LAUNCH_EXT(kern, dim3(blocks), dim3(64 * wpb), lds, c->st, e0, e1, 0, A);
        else NFC_LAUNCH(kern, dim3(blocks), dim3(64 * wpb), lds, c->st, A);
    };
    const bool b16 = (1 << c->nfold) == 16;
    if (b16) go(k_threshold_lean<KIND, 4, true>);
    else go(k_threshold_lean<KIND, 4, false>);
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
        case NFC_IN_ENV_F32: launch_lean<IN_ENV_F32>(c, A, nwork, e0, e1); break;
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
    int fresh = 0;
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
        // (a stream that starts here with a whole window in the batch: k_fill's short form)
        fresh = h.carry.stable == 0 && h.carry.filled == 0 && n >= (uint32_t)c->L && (void *)cr == c->d_state.p;
    }
    switch (c->P.input_kind) {
    case NFC_IN_IQ_F32: NFC_LAUNCH((k_fill<IN_IQ_F32>), dim3(1), dim3(FILL_BLOCK), (size_t)c->Lpad * 4, c->st, in, n, c->i16_scale, c->L, ring, cr, eci, ver, nchunks, sum, init, &((DevState *)c->d_state.p)->seq[0], c->batch_seq, fresh); break;
    case NFC_IN_ENV_F32: NFC_LAUNCH((k_fill<IN_ENV_F32>), dim3(1), dim3(FILL_BLOCK), (size_t)c->Lpad * 4, c->st, in, n, c->i16_scale, c->L, ring, cr, eci, ver, nchunks, sum, init, &((DevState *)c->d_state.p)->seq[0], c->batch_seq, fresh); break;
    case NFC_IN_REAL_F32_SQ: NFC_LAUNCH((k_fill<IN_REAL_F32_SQ>), dim3(1), dim3(FILL_BLOCK), (size_t)c->Lpad * 4, c->st, in, n, c->i16_scale, c->L, ring, cr, eci, ver, nchunks, sum, init, &((DevState *)c->d_state.p)->seq[0], c->batch_seq, fresh); break;
    default: NFC_LAUNCH((k_fill<IN_I16_SQ>), dim3(1), dim3(FILL_BLOCK), (size_t)c->Lpad * 4, c->st, in, n, c->i16_scale, c->L, ring, cr, eci, ver, nchunks, sum, init, &((DevState *)c->d_state.p)->seq[0], c->batch_seq, fresh); break;
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


}  // namespace
