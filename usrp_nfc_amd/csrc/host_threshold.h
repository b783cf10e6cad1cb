// host_threshold.h -- the threshold stage of a batch: plan, pass 0, certification rounds, sequential fallback
// (part of nfc_amd.hip: included there, in this order, into one translation unit)
#pragma once

namespace {

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
// The batch's last launch (k_pkt_finish) writes the state block into the host's mapped mirror and, last of all and behind a system
// fence, the batch's number into the mirror's stamp word.  Watching that word costs the host a few hundred nanoseconds after the
// write lands; hipStreamSynchronize comes back 8-10 us after the kernel has ended (the completion signal's way through the runtime) --
// and with one batch at a time the host's turn between two batches IS the step's idle time.  Bounded: after 2 ms without the stamp (a
// kernel that faulted never writes it) the stream is waited for in the ordinary way, which also reports the fault.  NFC_SPIN_WAIT=0.
// How much of the certification margin eps a batch's speculation used (CertSummary.worst: the largest L1 distance between a chunk's
// speculated and true incoming window, relative to its window sum) -- a diagnostic (test build: NFC_TRACE).  The margin itself stays
// at 1 %.  Measured, round 6, configs[2] (tag frames: loaded half bits 6 % above the HIGH threshold, less the noise), same call:
//   1 %    0.1542 ms per launch, roofline.frac 0.65   (the bench captures use 0.34 % of the window sum)
//   0.5 %  0.1432 ms, 0.70: the supersteps grow from one or two rounds to two to four
//   0.8 %  ONE chunk of the same capture gives up (a superstep grown on the head-room seen meets the next frame), and its re-run -- one
//          wave walking 98 304 samples -- makes the step 0.93 ms instead of 0.25
// How long supersteps grow is a heuristic; what a wrong guess costs is fifty times what a right one saves, and which margin happens to
// have none on a capture says nothing about the next capture: the margin is not tuned to the bench, and not adapted to the stream (a
// margin that followed what the stream used -- 1.6 times the worst seen, narrowed by a fifth per clean batch -- was built and walked
// straight through the 0.8 % above).  What would make a tighter margin safe is a cheap give-up, i.e. the in-place exact evaluation
// DESIGN.md 10 describes.  (NFC_EPS sets the margin in the test build.)
static inline void eps_adapt(nfc_ctx *c, const CertSummary &s) {
    if (!c->dbg_trace) return;
    float w;
    memcpy(&w, &s.worst, 4);
    fprintf(stderr, "[nfc] eps %.5f: the speculation used %.5f of the window sum at most, %u chunks failed their certification\n", (double)c->eps, (double)w, s.n_fail);
}

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#endif
}
static hipError_t wait_for_stamp(nfc_ctx *c) {
    // (a batch expected to take longer than the spin is worth -- a 1e9-sample batch runs 1.8 ms -- waits on the stream straight away)
    if (c->spin_wait && c->expect_ms < 1.2) {
        const volatile uint32_t *p = &c->hs->seq[1];
        const uint32_t want = c->stamp_b;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t k = 0;; k++) {
            if (*p == want) {
                std::atomic_thread_fence(std::memory_order_acquire);
                c->expect_ms = 0.75 * c->expect_ms + 0.25 * std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                return hipSuccess;
            }
            if ((k & 63u) == 63u) {
                const auto dt = std::chrono::steady_clock::now() - t0;
                if (dt > std::chrono::milliseconds(2)) break;
                if (dt > std::chrono::microseconds(400)) std::this_thread::yield();   // (past a clean batch's length: the core goes to whoever wants it between looks)
            }
            cpu_relax();
        }
        c->expect_ms = 4.0;   // (ran into the bound: the next batches of this stream wait on the stream; the figure decays below)
    } else if (c->spin_wait) {
        c->expect_ms *= 0.9;  // (... and the spin is tried again after a while: a stream's batches may have become short)
    }
    return hipStreamSynchronize(c->st);
}

struct ThrPlan {
    uint32_t nch;
    bool lean_applies;
    uint8_t *d_cert, *d_gflags, *d_gmin, *d_gmax;
    const uint8_t *h_cert, *h_gflags, *h_gmin, *h_gmax;
    // pinned staging behind the flag sections (one block, c->h_cflags): the chunks' sum bounds as the device left them, and what the
    // rounds of re-runs send down -- version bytes, the list of chunks to re-run, the list to certify (copies from pageable
    // memory are staged by the runtime, 20 us apiece: measured as the gaps between a round's launches)
    uint32_t *h_gvtop, *h_list_a, *h_list_b;
    uint8_t *h_ver;
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
    const bool lean_applies = c->lean && c->mx <= 500;   // (every input kind: a raw envelope that turns out negative makes its chunk give up)
    c->gring = c->gring_ok && (c->gring_force || (!lean_applies && (uint64_t)n >= (uint64_t)c->wave_slots_g * 8u * (uint64_t)c->L));
    // ... with a chunk per WORKGROUP where that kernel applies (threshold_wg.hip.h: max_len within one step, a ring that holds a round)
    c->wg_now = lean_applies && c->wg && c->wg_ok && !c->gring;
    if (!c->P.chunk_samples) {
        uint64_t slots = (uint64_t)(c->gring ? c->wave_slots_g : (c->wg_now ? (low_on_device ? c->wg_slots_ahead : c->wg_slots) : (lean_applies ? c->lean_slots : c->wave_slots)));
        // A stream whose last batches needed re-runs (a level that steps, load modulation at the threshold) is cut finer: a re-run
        // pass is ONE wave walking a chunk, so it costs what a chunk is long -- measured on the stress captures, 1e8 samples:
        // 4 x as many chunks take 2.14 -> 1.34 ms (level steps) and 7.3 -> 4.0 ms (hovering), 8 x is worse again (the rounds' fixed
        // costs); a clean stream pays 0.353 -> 0.406 ms for it, so the cut goes back after eight batches without a re-run.
        if (c->fine_left > 0 && c->wg_now) slots *= (uint64_t)c->fine_mult;
        // (every chunk owns three windows of summaries -- ring_in, ring_out[2] -- in device memory: however fine the cut, they stay
        // within 1 GiB, 4 x 1024 chunks of a 10 000-sample window take 0.5 GB)
        slots = std::min<uint64_t>(slots, std::max<uint64_t>(256, ((uint64_t)1 << 30) / ((uint64_t)12 * (uint64_t)c->L)));
        // (the lean kernel walks whole supersteps of lean_k steps: a chunk that is not a multiple of them ends on slow single steps;
        // the workgroup kernel whole rounds of four)
        const int stp = c->wg_now ? wg_round_samples(c->wg_nr) : 64 * c->rows_per_step * ((c->lean && !c->gring) ? c->lean_k : 1);
        uint64_t want = ((uint64_t)n + slots - 1) / slots;
        want = (want + stp - 1) / stp * stp;
        c->C = (int)std::max<uint64_t>(want, (uint64_t)c->C_min);
    }
    if (c->wg_now && c->C % wg_round_samples(c->wg_nr) != 0) c->wg_now = 0;   // (a configured chunk length that is not whole rounds: the one-wave kernel)
    const uint32_t off = 0u;   // (chunk c covers samples [c*C - off, (c+1)*C - off): the kernels here use off = 0)
    // The cut by dispatch row (threshold.hip.h: chunk_span).  The k-th workgroup the dispatcher hands a CU starts behind the k-1
    // before it and shares the CU with them for its whole life: measured with the cycle counter (round 5), the lives of the rows of
    // 256 workgroups are 0.965 / 0.985 / 1.010 / 1.042 of their mean with four per CU (1e8 samples, 1018 chunks of 98 304) and
    // 0.962 / 0.997 / 1.042 with three (av_window 10 000, 1e9 samples, 768 chunks), and the launch ends with the last row.  So the
    // rows get chunks whose lengths undo that -- the first rows longer, the last shorter, whole rounds each, the last row's length
    // whatever covers the batch -- when the batch is one full wave of resident workgroups and the chunk length is this file's own
    // choice.  NFC_WG_ROWBAL=0: the equal cut.
    const uint32_t cus = (uint32_t)std::max(1, c->n_cus);
    RowCut cut = equal_cut(n, (uint32_t)c->C, cus);
    const uint32_t slots_now = (uint32_t)(low_on_device ? c->wg_slots_ahead : c->wg_slots);   // (as the chunk length was chosen above)
    const uint32_t R = slots_now / cus;   // rows of a full wave: workgroups per CU
    // (the factors were measured on a 256-CU MI355X -- same-call A/Bs on three boxes, DESIGN.md 5.1c: on any other device the cut is equal)
    if (c->wg_rowbal && (c->n_cus == 256 || c->rowbal_set) && c->wg_now && !c->P.chunk_samples && !low_on_device && c->fine_left == 0 && slots_now == R * cus &&
        R >= 2 && R <= 4) {
        const uint32_t rs = (uint32_t)wg_round_samples(c->wg_nr);
        // (where the equal cut's chunks keep their planes in the LDS beside the ring, the longest chunk's must still fit: launch_wg)
        uint32_t max_len = 0;
        if (c->wg_lds_bulk_max && c->wg_lds_base + wg_stage_bytes(c->wg_nr, c->C / (int)rs + 2) <= c->wg_lds_bulk_max) {
            const size_t rounds = (c->wg_lds_bulk_max - c->wg_lds_base) / wg_stage_bytes(c->wg_nr, 1);
            max_len = rounds > 2 ? (uint32_t)(rounds - 2) * rs : rs;
        }
        cut = plan_row_cut(n, (uint32_t)c->C, rs, cus, R, c->rowbal_set ? c->rowbal_f[0] : c->rowbal_f[R], max_len);
    }
    c->rowbal_now = cut.by_row;
    const uint32_t nch = cut.nch;
    c->stats.n_chunks = nch;
    c->stats.chunk_samples = (uint32_t)c->C;
    const size_t nwords = ((size_t)n_all + 63) / 64 + 8;
    HIPCHK(c, planes_neg.ensure(nwords * 8));
    HIPCHK(c, planes_pos.ensure(nwords * 8));
    // The per-chunk buffers are sized for the FINE cut of this batch length straight away (a stream that turns out to need
    // re-runs is cut fine_mult times finer, host_threshold.h: fine_left -- from within the batch that finds out: an allocation in
    // the middle of a batch costs milliseconds).  Three windows of summaries per chunk: bounded by the cap on the chunk count.
    size_t nal = nch;
    if (c->fine_adapt && c->wg_now && !c->P.chunk_samples && c->fine_left == 0)
        nal = std::min<size_t>((size_t)nch * (size_t)c->fine_mult, std::max<size_t>(nch, ((size_t)1 << 30) / ((size_t)12 * (size_t)L)));
    for (int b = 0; b < 2; b++) {
        HIPCHK(c, c->d_ringout[b].ensure(nal * L * sizeof(float)));
        HIPCHK(c, c->d_touched[b].ensure(nal * c->twords * sizeof(uint32_t)));
        HIPCHK(c, c->d_info[b].ensure(nal * sizeof(ChunkInfo)));
    }
    HIPCHK(c, c->d_ringin.ensure(nal * L * sizeof(float)));
    HIPCHK(c, c->d_meta.ensure(nal * sizeof(RunMeta)));
    HIPCHK(c, c->d_ver.ensure(nal));
    HIPCHK(c, c->d_cflags.ensure((size_t)4 * nal));   // sections: cert | gflags | gmin | gmax
    if (c->h_cflags_cap < (size_t)20 * nal + 64) {
        devbuf_allocs()++;
        if (c->h_cflags) (void)hipHostFree(c->h_cflags);
        c->h_cflags_cap = (size_t)20 * nal + 4096;
        HIPCHK(c, hipHostMalloc((void **)&c->h_cflags, c->h_cflags_cap, hipHostMallocDefault));
    }
    uint8_t *d_cert = c->d_cflags.as<uint8_t>(), *d_gflags = d_cert + nch, *d_gmin = d_cert + 2 * (size_t)nch,
            *d_gmax = d_cert + 3 * (size_t)nch;
    const uint8_t *h_cert = c->h_cflags, *h_gflags = c->h_cflags + nch, *h_gmin = c->h_cflags + 2 * (size_t)nch,
                  *h_gmax = c->h_cflags + 3 * (size_t)nch;
    {
        uint8_t *q = c->h_cflags + (((size_t)4 * nch + 15) & ~(size_t)15);
        P.h_gvtop = (uint32_t *)q;
        P.h_list_a = (uint32_t *)(q + (size_t)4 * nch);
        P.h_list_b = (uint32_t *)(q + (size_t)8 * nch);
        P.h_ver = q + (size_t)12 * nch;
    }
    HIPCHK(c, c->d_list.ensure((size_t)nal * 8));   // two lists: the chunks a round re-runs, the chunks it certifies
    HIPCHK(c, c->d_gvtop.ensure((size_t)nal * 4));
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
    for (int r = 0; r < 4; r++) {
        A.row_len[r] = cut.row_len[r];
        A.row_start[r] = cut.row_start[r];
    }
    A.row_div = cut.row_div;
    A.C_max = (int)std::max(std::max(cut.row_len[0], cut.row_len[1]), std::max(cut.row_len[2], cut.row_len[3]));
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

// *recut_out (optional): pass 0's verdict says the stream is in the regime that needs re-runs and the batch is still on the coarse
// cut -- nothing of the attempt stands, the caller cuts finer and tries again.
static int threshold_span(nfc_ctx *c, const void *d_in_all, uint32_t n_all, uint32_t skip_all, uint32_t base, const EdgeCarry &ec,
                          const std::function<int()> *ahead, bool *clean, bool *need_seq_out, bool *recut_out = nullptr) {
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
        const bool lean = lean_applies && !c->gring && nch > 1;   // (max_len beyond 500 samples: not exercised, left to k_threshold)
        A.cert = d_cert;
        A.sum = (CertSummary *)(dT(c) + TOT_CERT);
        A.ksteps = c->wg_now ? c->wg_rounds : c->lean_rounds;
        A.gfac = c->lean_gfac;
        A.gfloor = c->lean_gmin;
        A.blk = 1 << c->nfold;
        const bool dbg_clk = lean && c->dbg_clk;
        if (dbg_clk) {
            HIPCHK(c, c->d_certinfo.ensure((size_t)nch * 32 * 5));   // (+ 4 waves x 4 counters per chunk in a profiling build)
            A.dbg_clk = c->d_certinfo.as<unsigned long long>();
        }
        c->wg_bulk_now = c->wg_bulk;   // (one batch at a time: nothing else wants the CU's LDS)
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
#ifdef NFC_WG_PROF
            if (c->wg_now) {
                std::vector<unsigned long long> pf((size_t)nch * 16);
                HIPCHK(c, hipMemcpy(pf.data(), (char *)c->d_certinfo.p + (size_t)nch * 32, (size_t)nch * 128, hipMemcpyDeviceToHost));
                double tk = 0, b1 = 0, b2 = 0, rr = 0;
                for (size_t k = 0; k < (size_t)nch * 4; k++) {
                    tk += (double)pf[4 * k];
                    b1 += (double)pf[4 * k + 1];
                    b2 += (double)pf[4 * k + 2];
                    rr += (double)pf[4 * k + 3];
                }
                const double nw = (double)nch * 4;
                fprintf(stderr, "[nfc] wg kernel, per wave: %.0f rounds; ticks waiting for samples %.0f, at the first barrier %.0f, at the closing barrier %.0f\n",
                        rr / nw, tk / nw, b1 / nw, b2 / nw);
            }
#endif
            fprintf(stderr, "[nfc] lean kernel, %u chunks, s_memtime ticks: first start .. last end %llu; per wave: incoming state %.0f, loop %.0f, "
                    "summary %.0f, longest wave %.0f, latest start %.0f\n", nch, t1 - t0, pro / nch, loop / nch, epi / nch, maxtot, maxstart);
            if (const char *path = c->dbg_clk_path.c_str()) {
                if (path[0] == '/' || path[0] == '.' || path[0] == 'g') {   // a file name: the raw stamps
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
        const bool dbg = c->dbg_any;
        bool first_round = true;
        c->h_list.clear();
        std::vector<uint8_t> tried_wg;   // 1: the workgroup kernel ran the chunk's latest re-run; 2: it has given up on it (or k_threshold has taken it since)
        bool ex_off = false;             // the in-place form is not used for the rest of this batch
        std::vector<uint8_t> ever_rerun; // chunks re-run at least once in this batch
        for (uint32_t k = 1; k < nch; k++) c->h_list.push_back(k);
        int rounds = 0;
        bool have_summary = false;
        CertSummary summary{};
        uint8_t *tot = dT(c);
        while (!c->h_list.empty()) {
            const uint32_t np = (uint32_t)c->h_list.size();
            if (first_round) {
                A.list = nullptr;   // k_certify: chunks 1 .. nch-1
            } else {   // (the list to certify: the second half of d_list, sent from pinned staging)
                memcpy(P.h_list_b, c->h_list.data(), (size_t)np * 4);
                HIPCHK(c, hipMemcpyAsync(c->d_list.as<uint32_t>() + nch, P.h_list_b, (size_t)np * 4, hipMemcpyHostToDevice, c->st));
                A.list = c->d_list.as<uint32_t>() + nch;
            }
            A.nlist = np;
            A.ver_zero = first_round ? 1 : 0;   // (pass 0 wrote every chunk's summary into buffer 0: k_fill zeroed the version bytes)
            if (dbg) HIPCHK(c, c->d_certinfo.ensure((size_t)nch * sizeof(CertInfo)));
            // the first round also resolves the end-of-batch state (last workgroup) and leaves its verdict as a
            // summary in the mirrored state block; later rounds (after re-runs) read the per-chunk flags
            CertSummary *d_sum = (CertSummary *)(tot + TOT_CERT);
            auto launch_certify = [&]() {
                NFC_LAUNCH(k_certify, dim3(cert_grid(np)), dim3(256), 0, c->st, A, d_cert,
                                   dbg ? c->d_certinfo.as<CertInfo>() : nullptr, c->d_ring[(c->ring_cur + 1) % NRING].as<float>(), dC(c),
                                   first_round ? d_sum : (CertSummary *)nullptr);
            };
            bool stamped_last = false;   // the round's last launch is the one that stamps the mirror (k_pkt_finish behind the writer)
            bool flags_with_verdict = false;
            if (first_round && ahead && !dbg) {
                // the stages that follow are enqueued now; their first full-width kernel takes the certification along
                c->cert = CertLaunch{A, d_cert, c->d_ring[(c->ring_cur + 1) % NRING].as<float>(), dC(c), d_sum, cert_grid(np)};
                c->cert_pending = true;
                const int rc = (*ahead)();
                if (c->cert_pending) {   // (a short batch's one-launch stage, or no edge stage at all: on its own then)
                    c->cert_pending = false;
                    launch_certify();
                    HIPCHK(c, mirror_async(c));
                } else {
                    stamped_last = c->timing < 2;   // (stream markers behind the stages: the stream is waited for)
                }
                if (rc) return rc;
                ran_ahead = true;
            } else {
                launch_certify();
                // (a stream whose batches have needed re-runs: the first round's flags travel with its verdict too -- one host turn less)
                flags_with_verdict = first_round && c->fine_left > 0;
                if (!first_round || flags_with_verdict) {   // (the flags, and the chunks' sum bounds for the exactness guard: all in the round's one host turn)
                    HIPCHK(c, hipMemcpyAsync(c->h_cflags, c->d_cflags.p, (size_t)4 * nch, hipMemcpyDeviceToHost, c->st));
                    HIPCHK(c, hipMemcpyAsync(P.h_gvtop, c->d_gvtop.p, (size_t)nch * 4, hipMemcpyDeviceToHost, c->st));
                }
                HIPCHK(c, mirror_async(c));
                if (first_round && ahead) {
                    const int rc = (*ahead)();
                    if (rc) return rc;
                    ran_ahead = true;
                }
            }
            if (stamped_last) HIPCHK(c, wait_for_stamp(c));
            else HIPCHK(c, hipStreamSynchronize(c->st));
            BATCHCHK(c, true);   // (the verdict summary and the totals are about to be read out of the mirror)
            std::vector<uint32_t> failing;
            if (first_round) {
                memcpy(&summary, c->hs->totals + TOT_CERT, sizeof summary);
                if (base == 0) eps_adapt(c, summary);
                if (summary.n_fail == 0 && !dbg) {
                    have_summary = true;
                    break;
                }
                // More than one chunk in 64 failed and the batch was cut for a clean stream: re-running those long chunks one wave
                // each is the expensive way round (a re-run pass costs what a chunk is long) -- pass 0 again on the fine cut costs
                // 0.2 ms and leaves short chunks to re-run.  (The next batches are cut fine from the start: fine_left.)
                if (recut_out && c->fine_adapt && c->wg_now && c->fine_left == 0 && !c->P.chunk_samples && c->fine_mult > 1 &&
                    (uint64_t)summary.n_fail * 64u > (uint64_t)nch) {
                    *recut_out = true;
                    A.ver_zero = 0;
                    return NFC_OK;
                }
                if (!flags_with_verdict) HIPCHK(c, hipMemcpy(c->h_cflags, c->d_cflags.p, (size_t)4 * nch, hipMemcpyDeviceToHost));
            }
            if (first_round && lean && !h_cert[0]) failing.push_back(0);   // chunk 0 gave up: re-run from the carried state
            for (uint32_t k : c->h_list)
                if (!h_cert[k]) failing.push_back(k);
            if (c->dbg_trace) {
                fprintf(stderr, "[nfc] round %d: n_fail %u, %zu failing of %u pending (cert[0] %d):", rounds, summary.n_fail, failing.size(), np, (int)h_cert[0]);
                for (size_t i = 0; i < failing.size() && i < 12; i++) fprintf(stderr, " %u(%x)", failing[i], (unsigned)h_gflags[failing[i]]);
                int why[8] = {0};
                for (uint32_t k : failing) why[(h_gflags[k] >> 4) & 7]++;
                fprintf(stderr, "; gave up by code:");
                for (int i = 0; i < 8; i++) fprintf(stderr, " %d", why[i]);
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
            A.ver_zero = 0;
            // A chunk right behind one that is re-run now and that did not give up itself is NOT re-run in this round: its own
            // evaluation may well be sound -- what failed is the comparison with a predecessor whose summary was worthless -- and
            // from the state resolved now it would only be evaluated against that worthless summary again.  It stays pending
            // (below: it can see a re-run chunk) and is certified against the predecessor's new summary in the next round.
            {
                std::vector<uint32_t> now;
                uint32_t last = 0xFFFFFFFFu;
                if (ever_rerun.empty()) ever_rerun.assign(nch, 0);
                for (uint32_t k : failing) {   // (ascending)
                    // (only a chunk still standing on its pass-0 evaluation: one that has been re-run already was evaluated from a
                    // resolved state that has moved since -- waiting would only cost it a round, measured: 3 general passes
                    // instead of 2 when every chunk fails)
                    const bool gave_up = (h_gflags[k] & 4) != 0;
                    if (!gave_up && !ever_rerun[k] && last != 0xFFFFFFFFu && k == last + 1) continue;
                    now.push_back(k);
                    last = k;
                }
                for (uint32_t k : now) ever_rerun[k] = 1;
                failing.swap(now);
            }
            // Re-runs from the exact state: by the workgroup kernel's in-place form where it applies and the round's failures fit the
            // machine (below), else by k_threshold, which decides everything in place and never gives up (so the rounds converge).
            // (The pass-0 form of the workgroup kernel re-running whatever did not give up was measured on the stress captures in rounds 4-6 --
            // the chunks behind a chunk that gave up mostly give up themselves, 6 passes / 4.7 ms against 3 / 2.1 -- and went in round 6.)
            // A LONE failure on an otherwise clean batch (round 6) -- a superstep that grew on the head-room it saw and met the next frame:
            // one chunk in a thousand -- takes the workgroup kernel too, gave up or not: from the exact state (no margin for a speculated
            // window) and with supersteps of at most two rounds it mostly gets through, and four waves walk a 98 304-sample chunk in 70 us
            // where the general kernel's one wave takes 670 (measured on configs[2] with the margin set to make a chunk give up: the step
            // 0.927 -> 0.526 ms, same call).  If it gives up again the general kernel takes it in the next round (tried_wg).
            const bool lone = c->wg_rerun_lone && failing.size() <= (size_t)c->wg_lone_max && (uint64_t)failing.size() * (uint64_t)c->wg_lone_div <= (uint64_t)nch;
            // Up to a machine-full of failing chunks (round 6): the workgroup kernel in the form that evaluates a failed round IN PLACE
            // (k_threshold_wg<KIND, 4, true>) -- it gives up only where a round is not four whole steps or a LOW run is out of sight.
            if (tried_wg.empty()) tried_wg.assign(nch, 0);
            {   // (a stream it keeps giving up on -- values out of range, a stream's first chunk: k_threshold alone for the rest of the batch)
                size_t ran = 0, gave = 0;
                for (uint32_t k : failing)
                    if (tried_wg[k] == 1) ran++, gave += (h_gflags[k] & 4) ? 1 : 0;
                if (gave * 4 > ran) ex_off = true;
            }
            const bool ex = c->wg_now && c->wg_ex_ok && !ex_off && failing.size() <= (size_t)c->wg_ex_max;
            std::vector<uint32_t> by_wg, by_general;
            for (uint32_t k : failing) {
                // (a chunk the in-place form has re-run and that did NOT give up may take it again -- what failed is its incoming state;
                // one it gave up on is k_threshold's for the rest of the batch: that kernel never gives up, so the rounds converge)
                if (tried_wg[k] == 1 && (h_gflags[k] & 4)) tried_wg[k] = 2;
                const bool ex_again = ex && tried_wg[k] == 1;
                if (c->wg_now && (!tried_wg[k] || ex_again) && (ex || lone)) {
                    by_wg.push_back(k);
                    tried_wg[k] = 1;
                } else {
                    by_general.push_back(k);
                    if (tried_wg[k]) tried_wg[k] = 2;
                }
            }
            if (c->dbg_trace) fprintf(stderr, "[nfc]   re-run: %zu by the workgroup kernel%s, %zu by k_threshold\n", by_wg.size(), ex ? " (in-place form)" : "", by_general.size());
            A.mode = 1;
            // One host turn per round (round 4: two before -- the re-runs were waited for before their certification was enqueued,
            // and every small copy came out of pageable memory): the lists travel from pinned staging, the version bytes of the
            // re-run chunks are flipped here and sent right behind the re-run launch, and the list to certify is made up before the
            // re-runs have run -- a re-run chunk is taken to leave slots untouched (which its new summary will say): a superset of
            // what has to be certified, and certifying a chunk nothing has changed for gives the verdict it had.
            {
                uint32_t *la = P.h_list_a;
                size_t nl_ = 0;
                for (uint32_t k : by_wg) la[nl_++] = k;
                for (uint32_t k : by_general) la[nl_++] = k;
                HIPCHK(c, hipMemcpyAsync(c->d_list.p, la, nl_ * 4, hipMemcpyHostToDevice, c->st));
            }
            if (!by_wg.empty()) {
                A.list = c->d_list.as<uint32_t>();
                A.nlist = (uint32_t)by_wg.size();
                A.ksteps = 2;
                c->wg_ex_launch = ex;
                launch_threshold_kind(c, A, A.nlist, true);
                c->wg_ex_launch = false;
                A.ksteps = c->wg_rounds;
            }
            if (!by_general.empty()) {
                A.list = c->d_list.as<uint32_t>() + by_wg.size();
                A.nlist = (uint32_t)by_general.size();
                launch_threshold_kind(c, A, A.nlist);
            }
            A.nlist = (uint32_t)failing.size();
            c->stats.threshold_passes++;
            passes++;
            c->stats.chunks_rerun += A.nlist;
            if (ex) c->stats.chunks_rerun_in_place += (uint32_t)by_wg.size();
            std::vector<uint8_t> ran(nch, 0);
            for (uint32_t k : failing) {
                ran[k] = 1;
                c->h_ver[k] ^= 1;
            }
            memcpy(P.h_ver, c->h_ver.data(), nch);
            HIPCHK(c, hipMemcpyAsync(c->d_ver.p, P.h_ver, nch, hipMemcpyHostToDevice, c->st));
            // pending: the re-run chunks (a predecessor may have been re-run beside them) and every
            // chunk that can see one of them through predecessors that left ring slots untouched
            c->h_list.clear();
            bool vis = false;
            for (uint32_t k = 0; k < nch; k++) {
                if (vis || ran[k]) c->h_list.push_back(k);
                const bool full = !ran[k] && !(h_gflags[k] & 2);   // (a re-run chunk: not known yet -- taken as not full)
                vis = ran[k] || (vis && !full);
            }
            if (++rounds > 2 * (int)nch + 2) return fail(c, NFC_ERR_INTERNAL, "threshold passes did not converge");
        }

        // can every fp64 sum of this batch be proven exact?  (otherwise the summation order matters)
        int emin = 255, emax = 0;
        bool flagged = false;
        uint32_t vtop = 0;
        if (have_summary && passes == 1) {
            emin = (int)(summary.eminmax & 0xFFFFu);
            emax = (int)(summary.eminmax >> 16);
            flagged = summary.flagged != 0;
            vtop = summary.vtop;
        } else {
            if (first_round) {   // (no round of re-runs has fetched them: the debugging path)
                HIPCHK(c, hipMemcpyAsync(c->h_cflags, c->d_cflags.p, (size_t)4 * nch, hipMemcpyDeviceToHost, c->st));
                HIPCHK(c, hipMemcpyAsync(P.h_gvtop, c->d_gvtop.p, (size_t)nch * 4, hipMemcpyDeviceToHost, c->st));
                HIPCHK(c, mirror_async(c));
                HIPCHK(c, hipStreamSynchronize(c->st));
                BATCHCHK(c, true);
            }
            // (otherwise the last round's certification brought flags, bounds and the mirror along: nothing ran since)
            for (uint32_t k = 0; k < nch; k++) {
                emin = std::min(emin, (int)h_gmin[k]);
                emax = std::max(emax, (int)h_gmax[k]);
                if (h_gflags[k] & 1) flagged = true;
                vtop = std::max(vtop, P.h_gvtop[k]);
            }
        }
        c->h_carry = c->hs->carry;
        carry_apply_fin(c->h_carry);
        const bool exact = sums_exact(c->h_carry, emin, emax, vtop);
        if (dbg) fprintf(stderr, "[nfc] guard: emin %d emax %d ss_emin %d ss_emax %d flagged %d exact %d\n", emin, emax,
                         c->h_carry.ss_emin, c->h_carry.ss_emax, (int)flagged, (int)exact);
        if (!exact || flagged) need_seq = true;
    }

    A.ver_zero = 0;
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
    c->stats.chunks_rerun_in_place = 0;
    c->stats.used_sequential = 0;
    const uint32_t span = (uint32_t)std::min<uint64_t>(((uint64_t)8 * c->L + STEP - 1) / STEP * STEP, 1u << 30);   // prefix per round
    EdgeCarry ec = c->h_ecarry;
    uint32_t base = 0;
    for (int round = 0;; round++) {
        bool need_seq = false, span_clean = false;
        // (while the stream is in the regime that needs re-runs, the later stages are not enqueued behind pass 0 on the chance that
        // it stands: they would run -- 0.17 ms of the machine -- before the host has seen the verdict, and run again after the re-runs)
        const bool optimistic = base == 0 && !(c->fine_adapt && c->fine_left > 0);
        bool recut = false;
        int rc = threshold_span(c, d_in, n, skip, base, ec, optimistic ? ahead : nullptr, &span_clean, &need_seq, &recut);
        if (rc) return rc;
        if (recut) {
            // the attempt's end-of-batch sum is void (k_fill's preparation would take it for the carried one), everything else it
            // wrote is written again
            c->fine_left = 8;
            c->cert_pending = false;
            HIPCHK(c, hipMemsetAsync((char *)dC(c) + offsetof(Carry, fin_valid), 0, sizeof(int32_t), c->st));
            span_clean = false;
            rc = threshold_span(c, d_in, n, skip, base, ec, nullptr, &span_clean, &need_seq, nullptr);
            if (rc) return rc;
        }
        if (!need_seq) {
            *clean = span_clean && base == 0;
            if (c->fine_adapt) {   // (a hint about the stream, not part of its state: it survives nfc_reset)
                if ((uint64_t)c->stats.chunks_rerun * 64u > (uint64_t)c->stats.n_chunks) c->fine_left = 8;
                else if (c->fine_left > 0) c->fine_left--;
            }
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


}  // namespace
