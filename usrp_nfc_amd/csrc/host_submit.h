// host_submit.h -- batches submitted ahead: nfc_submit_device / nfc_wait
// (part of nfc_amd.hip: included there, in this order, into one translation unit)
#pragma once

namespace {

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
           !(c->use_small && n <= SM_MAX_SAMPLES) && n > 0 && c->ahead_lds_per_cu <= AHEAD_LDS_MAX && !c->dbg_any && !c->dbg_clk && !c->dbg_no_submit_ahead;
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
    struct AllocNote {   // (whatever path leaves: the buffers (re)allocated here are this batch's)
        nfc_ctx::Submitted &b;
        uint64_t a0;
        ~AllocNote() { b.allocs += (uint32_t)(devbuf_allocs() - a0); }
    } alloc_note{b, devbuf_allocs()};
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
    A.ksteps = c->wg_now ? c->wg_rounds : c->lean_rounds;
    A.gfac = c->lean_gfac;
    A.gfloor = c->lean_gmin;
    A.blk = 1 << c->nfold;
    b.timed = b.timing >= 1;
    c->wg_bulk_now = false;   // (the other stages' workgroups run beside this kernel: its LDS stays small)
    launch_threshold_kind(c, A, P.nch, true, b.timed ? c->kev_sub[b.slot] : nullptr);
    const uint32_t np = P.nch - 1;
    A.nlist = np;
    A.ver_zero = 1;   // (pass 0 only: every summary is in buffer 0)
    NFC_LAUNCH(k_certify, dim3(cert_grid(np)), dim3(256), 0, c->st, A, P.d_cert, (CertInfo *)nullptr,
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
    struct AllocNote {
        nfc_ctx::Submitted &b;
        uint64_t a0;
        ~AllocNote() { b.allocs += (uint32_t)(devbuf_allocs() - a0); }
    } alloc_note{b, devbuf_allocs()};
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
    int rc;
    if (c->tail_on) {   // (tail.hip.h: one persistent launch; the edge stage alone where the decoders cannot run in it)
        const bool fused = tail_can_decode(c);
        rc = run_tail(c, b.n, 0u, b.g0);
        if (!rc && !fused) rc = run_decode(c, true);
    } else {
        rc = run_edges(c, b.n, 0u, b.g0);
        if (!rc) rc = run_decode(c);   // (its last launch mirrors the state block and stamps it)
    }
    if (rc) return rc;
    b.spec = c->dec_spec_now;
    b.tail = c->tail_now;
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
        nb.allocs = 0;   // (what the abandoned attempt allocated was charged to the batch that went through the synchronous path)
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
    uint32_t ne, ns[2], tail_v = 0;
    memcpy(&ne, c->hs->totals + TOT_EDGES, 4);
    memcpy(ns, c->hs->totals + TOT_NSYM, 8);
    eps_adapt(c, summary);
    if (summary.n_fail != 0) regular = false, why = "a chunk was not certified";
    else if (summary.flagged || !sums_exact(after, (int)(summary.eminmax & 0xFFFFu), (int)(summary.eminmax >> 16), summary.vtop)) regular = false, why = "sums not provably exact";
    else if (b.tail && (tail_v = ([&] { uint32_t v; memcpy(&v, c->hs->totals + TOT_SPEC, 4); return v & (TLV_DENSE | TLV_TIMEOUT); })())) {
        // (tail.hip.h: a tile's entries did not fit the staging -- the synchronous path repeats the batch with shorter tiles)
        regular = false, why = "a tile too dense for the fused tail";
        if (tail_v & TLV_DENSE) tail_note_dense(c);
    } else if (!(ne <= c->cap_edges && ns[0] + 2 <= c->cap_sym[0] && ns[1] + 2 <= c->cap_sym[1])) regular = false, why = "a capacity estimate was too small";
    else if (b.spec) {   // (decode.hip.h: dec_verify's verdict on the speculative decode of THIS batch, in the mirror its last launch wrote)
        uint32_t v;
        memcpy(&v, c->hs->totals + TOT_SPEC, 4);
        if (v) {
            // Only the decode stage was wrong: the threshold and edge stages stand, and so does everything submitted behind this
            // batch (their threshold stages start from this batch's threshold stage alone).  The decode stage is repeated in the
            // form that assumes nothing -- same carried values (the host mirrors are adopted below), same buffers -- on the main
            // stream, where the next batch's later stages are not enqueued before this one is popped; the batches behind it take
            // that form straight away (note_respeculation).
            note_respeculation(c);
            // (the context's per-batch fields -- cap_edges, last_g0, h_dcarry, pend_cur -- are still this batch's: the later stages of the
            // batches behind it are only enqueued once this one is popped)
            int rc = c->stamp_b == b.seq ? run_decode(c, true) : fail(c, NFC_ERR_INTERNAL, "decode repeat: the context has moved on to batch %u (this is %u)", c->stamp_b, b.seq);
            if (!rc && hipStreamSynchronize(c->st) != hipSuccess) rc = fail(c, NFC_ERR_DEVICE, "waiting for the repeated decode stage failed");
            if (!rc) rc = batch_ok(c, false);
            if (!rc && c->hs->seq[1] != b.seq) rc = fail(c, NFC_ERR_DEVICE, "state mirror is stale after the repeated decode stage (batch %u, mirror %u)", b.seq, c->hs->seq[1]);
            if (!rc) {
                memcpy(&v, c->hs->totals + TOT_SPEC, 4);
                if (v) rc = fail(c, NFC_ERR_INTERNAL, "the decode stage's three-launch form reported a speculation failure");
            }
            if (rc) {
                abandon();
                return rc;
            }
            memcpy(ns, c->hs->totals + TOT_NSYM, 8);
            b.spec = false;
            if (!(ns[0] + 2 <= c->cap_sym[0] && ns[1] + 2 <= c->cap_sym[1])) regular = false, why = "a capacity estimate was too small";
        }
    }
    if (c->dbg_redo_submitted && (c->dbg_fast_waits++ % 3u) == 2u) regular = false, why = "test hook";   // every third batch that ran ahead
    if (regular) {
        c->h_carry = after;
        c->h_ecarry = c->hs->ecarry;
        c->h_dcarry = c->hs->dcarry;
        c->n_edges = ne;
        c->n_sym[0] = ns[0];
        c->n_sym[1] = ns[1];
        update_estimates(c, b.n);
        c->dec_spec_now = b.spec;
        spec_batch_done(c);
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
        c->stats.device_allocs = b.allocs;
        c->ring_carried = summary.n_carried;   // (from this batch's own snapshot: the device's summary belongs to the next batch by now)
        c->ring_cur = (b.ring_in + 1) % NRING;
        c->nseen = b.g0 + b.n;
        c->last_in = b.d_in;
        c->have_outputs = true;
        c->low_valid = true;
        if (c->fine_adapt && c->fine_left > 0) c->fine_left--;   // (a batch without a re-run: run_threshold's bookkeeping, for this path)
        pop();
        return NFC_OK;
    }
    // The optimistic result does not stand: everything in flight is drained, the batch goes through the synchronous path
    // from the state before it (the host mirrors were last adopted there; its window buffer was not written since), and
    // the batches behind it start again from what that leaves.
    if (c->dbg_trace) fprintf(stderr, "[nfc] submitted batch %u processed again: %s\n", b.seq, why);
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
    const auto t_bp0 = std::chrono::steady_clock::now();
    if (nc) {
        // (through pinned staging: two copies of a few tens of KB into pageable vectors took 1.4 ms per batch -- the runtime stages
        // those itself, synchronously -- and were most of what `end_to_end` spent per piece: round 5)
        const size_t need = (size_t)nc * 12 + 64;
        if (c->h_pk_stage_cap < need) {
            devbuf_allocs()++;   // (a pinned regrow costs what a device one does: nfc_stats.device_allocs counts both)
            if (c->h_pk_stage) (void)hipHostFree(c->h_pk_stage);
            c->h_pk_stage = nullptr;
            c->h_pk_stage_cap = 0;
            const size_t cap = need + need / 2 + 65536;
            HIPCHK(c, hipHostMalloc((void **)&c->h_pk_stage, cap, hipHostMallocDefault));
            c->h_pk_stage_cap = cap;
        }
        uint64_t *idx = (uint64_t *)c->h_pk_stage;
        uint32_t *ends = (uint32_t *)(c->h_pk_stage + (size_t)nc * 8);
        HIPCHK(c, hipMemcpyAsync(idx, c->d_close_idx[t].p, (size_t)nc * 8, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(ends, c->d_close_end[t].p, (size_t)nc * 4, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        if (c->dbg_trace) fprintf(stderr, "[nfc] build_packets(%d): %u closes, copies %.3f ms\n", t, nc, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bp0).count());
        c->pk[t].reserve(nc);
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
    if (c->dbg_trace) fprintf(stderr, "[nfc] build_packets(%d): %zu packets, %.3f ms\n", t, c->pk[t].size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bp0).count());
    c->pk_ready[t] = true;
    return NFC_OK;
}

}  // namespace
