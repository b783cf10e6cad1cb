// host_stages.h -- edge stage, decode + framing stage, the one-launch form of short batches, process_batch
// (part of nfc_amd.hip: included there, in this order, into one translation unit)
#pragma once

namespace {

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
    E.sw = (uint32_t)(EW_WORDS * EW_SUPER);
    E.tps = (uint32_t)EW_SUPER;
    const size_t tiles = edge_num_tiles(nwords);    // EW_WORDS words per tile in both launches of the stage
    const size_t supers = edge_num_supers(nwords);  // the reduce pass: a workgroup per EW_SUPER tiles, both levels of aggregates
    HIPCHK(c, c->d_partials.ensure((tiles + supers + 2) * sizeof(EdgeAgg)));
    EdgeAgg *parts = c->d_partials.as<EdgeAgg>();
    EdgeAgg *sups = parts + tiles + 1;
    // launch 1: one aggregate per tile (first change, last two changes, entries it is sure of).  While the tiles are few,
    // each tile's workgroup of the writer folds its predecessors' aggregates itself (scan.hip.h: tile_prefix) and the
    // single-workgroup prefix launch is not needed.
    const bool own = tiles <= c->own_prefix_max;
    Last2 *last2_total = (Last2 *)(tot + TOT_LAST2);
    uint32_t *edges_total = (uint32_t *)(tot + TOT_EDGES);
    const EdgeAggOp op{E.mx, E.mx_magic};
    // (the first certification of the batch, when it is still to be launched, rides with the WRITER below: host_context.h)
    if (tiles) NFC_LAUNCH(k_edge_reduce, dim3((unsigned)supers), dim3(ER_BLOCK), 0, c->st, E, nwords, parts, sups);
    if (!own || !tiles)   // (long batches: the prefix launch over the SUPER-aggregates; the totals and the carry in its epilogue)
        scan_partials_with(c->st, op, supers, nullptr, (uint32_t)(EW_WORDS * EW_SUPER), sups, op.identity(), (EdgeAgg *)nullptr,
                           EdgeTotalEpilogue{E, edges_total, last2_total, dE(c)});
    const uint32_t cap = c->cap_edges;
    const size_t cap_al = std::max(cap, c->alloc_edges);
    HIPCHK(c, c->d_epos.ensure((cap_al + 8) * 4));
    HIPCHK(c, c->d_ecode.ensure((cap_al + 8) * 2));
    if (c->cert_pending && tiles) {
        c->cert_pending = false;
        NFC_LAUNCH(k_certify_and_write, dim3((unsigned)(c->cert.blocks + tiles)), dim3(SCAN_BLOCK), 0, c->st, c->cert, E, nwords, parts, sups,
                   c->d_epos.as<uint32_t>(), c->d_ecode.as<uint16_t>(), cap, own, edges_total, last2_total, dE(c));
    } else if (tiles) {
        NFC_LAUNCH(k_write_edges, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, E, nwords, parts, sups, c->d_epos.as<uint32_t>(),
                   c->d_ecode.as<uint16_t>(), cap, own, edges_total, last2_total, dE(c));
    }
    c->edges_from_host = false;
    c->tail_now = false;
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
        const size_t cs_al = c->edges_from_host ? cs : std::max(cs, c->alloc_sym[t]);   // (allocated with room: size_capacities)
        HIPCHK(c, c->d_sym[t].ensure(cs_al + 16));
        P.sym[t] = c->d_sym[t].as<uint8_t>();
        P.cap_sym[t] = cs;
        P.started_in[t] = (uint32_t)c->h_dcarry.pkt_started[t];
        if (!enabled[t]) continue;   // no symbols of this type (background.py:17-25); its carry stays
        const uint32_t pend = c->h_dcarry.pending[t];
        // (the open packet's carried bits come on top: room in steps of 64 K of them, so that a few bits more do not reallocate)
        const size_t pend_al = c->edges_from_host ? pend : ((((size_t)pend >> 16) + 1) << 16);
        HIPCHK(c, c->d_bits[t].ensure(pend_al + cs_al + 16));
        HIPCHK(c, c->d_pending[t][pn].ensure(pend_al + cs_al + 16));
        HIPCHK(c, c->d_pending[t][1 - pn].ensure(pend_al + cs_al + 16, true, c->st));   // (the other half with it: it is the next batch's, and holds this batch's carried bits)
        HIPCHK(c, c->d_close_end[t].ensure((cs_al + 4) * 4));
        HIPCHK(c, c->d_close_idx[t].ensure((cs_al + 4) * 8));
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

// After framing: the open packets' bits to the other half of the double buffer, the framing carry, the state block into the host's
// mapped mirror with the batch's stamp (decode.hip.h: k_pkt_finish) -- the last launch of a batch in both forms of the tail.
void launch_pkt_finish(nfc_ctx *c, const FrameOut &P, const bool (&enabled)[2], PktCnt *pk_total, FrameAgg *frame_total) {
    const int pn = 1 - c->pend_cur;
    PktFinish F;
    memset(&F, 0, sizeof F);
    F.packed = 1;
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
}

int run_decode(nfc_ctx *c, bool force_classic = false) {
    uint8_t *tot = dT(c);
    const uint32_t ce = c->cap_edges;                      // capacity; the count is on the device
    const uint32_t *ne_dev = (const uint32_t *)(tot + TOT_EDGES);
    const size_t tiles = dec_num_tiles(ce);
    // (allocated for alloc_edges: size_capacities -- the grids below follow the estimate `ce`)
    const size_t ce_al = c->edges_from_host ? ce : std::max(ce, c->alloc_edges);
    const size_t tiles_al = dec_num_tiles(ce_al);
    HIPCHK(c, c->d_states.ensure(ce_al + 16));   // one out-byte per edge
    HIPCHK(c, c->d_partials.ensure((tiles_al + 1) * sizeof(DecMaps)));
    HIPCHK(c, c->d_partials2.ensure((tiles_al + 1) * sizeof(FrameAgg)));
    // (the per-thread aggregates of the three-launch form: only where that form can run -- 24 + 16 bytes per 32 edges)
    HIPCHK(c, c->d_aggs.ensure((tiles_al * SCAN_BLOCK + 1) * sizeof(DecMaps)));
    HIPCHK(c, c->d_faggs.ensure((tiles_al * SCAN_BLOCK + 1) * sizeof(FramePk)));
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
    const bool own = tiles <= c->own_prefix_max;   // (scan.hip.h: tile_prefix -- no prefix launches while the tiles are few)
    DecMaps *map_total = (DecMaps *)(tot + TOT_DECMAP);
    // the speculative form (k_dec_spec: reduce + apply in one launch, every tile's incoming states from a run-in of its
    // predecessor's last edges) unless it is switched off, or the stream's last batches needed the three-launch form
    const uint32_t mil_class_in = (c->h_dcarry.mil_state >= 0 && c->h_dcarry.mil_state < 16) ? c->mil_q_of[c->h_dcarry.mil_state] : 0xFFu;
    const bool spec = c->dec_spec && c->spec_off_left == 0 && !force_classic && c->T.q_ok && mil_class_in != 0xFFu;   // (a state no edge sequence reaches: set from the host)
    c->dec_spec_now = spec;
    DecCarryEpilogue epi{spec ? (DecMaps *)nullptr : map_total, dec_state_in, dD(c), (uint32_t *)(tot + TOT_NSYM), pk_total,
                         {P.pend[0], P.pend[1]}, {P.started_in[0], P.started_in[1]}, {c->T.canon[0], c->T.canon[1], c->T.canon[2], c->T.canon[3]}};
    const uint32_t spec_state_in = mil_class_in | ((uint32_t)c->h_dcarry.man_state << 4);
    DecVerify V{nullptr, spec_state_in, {c->T.q_rep[0], c->T.q_rep[1]}, dD(c), (uint32_t *)(tot + TOT_SPEC)};
    // the packed bit arrays are or-ed into: the stage's first launch clears them
    ZeroJob Z;
    for (int t = 0; t < 2; t++) {
        Z.p[t] = (uint32_t *)P.bits[t];
        Z.n[t] = P.bits[t] ? (P.cap_bits[t] + 31u) / 32u + 1u : 0u;
        if (!tiles && Z.n[t]) HIPCHK(c, hipMemsetAsync(Z.p[t], 0, (size_t)Z.n[t] * 4, c->st));
    }
    TileStage S;
    memset(&S, 0, sizeof S);
    if (spec) {
        HIPCHK(c, c->d_spec.ensure((tiles_al + 1) * sizeof(DecSpec)));
        V.spec = c->d_spec.as<DecSpec>();
        // what the tiles stage for k_concat: bits and packet ends entered "started", per enabled packet type
        HIPCHK(c, c->d_stage_own.ensure((tiles_al + 1) * sizeof(FrameAgg)));
        S.own = c->d_stage_own.as<FrameAgg>();
        S.epos = P.epos;
        S.g0 = P.g0;
        S.idx64 = P.idx64;
        for (int t = 0; t < 2; t++) {
            if (!P.bits[t]) continue;
            HIPCHK(c, c->d_stage_bits[t].ensure((tiles_al + 1) * (size_t)FW_WORDS * 4));
            HIPCHK(c, c->d_stage_cb[t].ensure((tiles_al + 1) * (size_t)ST_CLOSES * 4));
            HIPCHK(c, c->d_stage_ci[t].ensure((tiles_al + 1) * (size_t)ST_CLOSES * 8));
            HIPCHK(c, c->d_stage_q[t].ensure((tiles_al + 1) * 4));
            S.bits[t] = c->d_stage_bits[t].as<uint32_t>();
            S.close_bit[t] = c->d_stage_cb[t].as<uint32_t>();
            S.close_idx[t] = c->d_stage_ci[t].as<uint64_t>();
            S.drop_bit[t] = c->d_stage_q[t].as<uint32_t>();
        }
        if (tiles) {
            if (lds_tables)
                NFC_LAUNCH(k_dec_spec<true>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, spec_state_in, c->dec_runin, outw,
                           fparts, c->d_spec.as<DecSpec>(), Z, S);
            else
                NFC_LAUNCH(k_dec_spec<false>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, spec_state_in, c->dec_runin, outw,
                           fparts, c->d_spec.as<DecSpec>(), Z, S);
        }
    } else {
        if (tiles) {
            if (lds_tables)
                NFC_LAUNCH(k_dec_reduce<true>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, dparts, daggs, Z);
            else
                NFC_LAUNCH(k_dec_reduce<false>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, dparts, daggs, Z);
        }
        if (!own) scan_partials<ComposeDec>(c->st, tiles, ne_dev, DEC_TILE, dparts, ComposeDec::identity_host(), map_total);
        if (tiles) {
            if (lds_tables)
                NFC_LAUNCH(k_dec_apply<true>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, dparts, daggs,
                                   dec_state_in, outw, fparts, c->d_faggs.as<FramePk>(), own, map_total);
            else
                NFC_LAUNCH(k_dec_apply<false>, dim3((unsigned)tiles), dim3(SCAN_BLOCK), 0, c->st, ecode, (size_t)ce, ne_dev, c->T, dparts, daggs,
                                   dec_state_in, outw, fparts, c->d_faggs.as<FramePk>(), own, map_total);
        }
    }
    if (!own) scan_partials<FrameAggOp>(c->st, tiles, ne_dev, DEC_TILE, fparts, FrameAggOp::identity(), frame_total, epi);
    if (spec)   // (the tiles' staged bits and packet ends to their places; its first workgroup checks the tiles' assumptions)
        NFC_LAUNCH(k_concat, dim3((unsigned)(std::max<size_t>(tiles, 1) + 1)), dim3(SCAN_BLOCK), 0, c->st, (size_t)ce, ne_dev, fparts, S, P, own, frame_total, epi, V);
    else
        NFC_LAUNCH(k_frame_write, dim3((unsigned)std::max<size_t>(tiles, 1)), dim3(SCAN_BLOCK), 0, c->st, outw, (size_t)ce, ne_dev, fparts,
                           c->d_faggs.as<FramePk>(), P, own, frame_total, epi, V);
    c->bits_packed = true;
    c->bits_clean[0] = c->bits_clean[1] = 0;   // (this stage cleared and filled them itself)
    c->sym_from_tail = false;
    c->sym_lazy = true;          // the symbol arrays are written when nfc_read_symbols asks for them (materialize_symbols)
    c->sym_P = P;
    c->sym_n = ce;
    c->sym_tiles = (uint32_t)tiles;
    c->sym_own = own;
    launch_pkt_finish(c, P, enabled, pk_total, frame_total);
    return NFC_OK;
}

#ifdef NFC_TEST_HOOKS
// ---------------------------------------------------------------------------
// the fused tail: edges, decoders and framing in one persistent launch (tail.hip.h)
// ---------------------------------------------------------------------------
// decode: the decoders run in the launch (the Miller decoder as its quotient machine, from a class the carried state belongs to);
// otherwise the launch is the edge stage alone and the three-launch decode follows.
bool tail_can_decode(const nfc_ctx *c) {
    const bool known = c->h_dcarry.mil_state >= 0 && c->h_dcarry.mil_state < 16 && c->mil_q_of[c->h_dcarry.mil_state] != 0xFFu;
    return c->T.q_ok && known && c->h_dcarry.man_state >= 0 && c->h_dcarry.man_state < 8;
}
int run_tail(nfc_ctx *c, uint32_t n, uint32_t skip, uint64_t g0) {
    uint8_t *tot = dT(c);
    const size_t nwords = ((size_t)n + 63) / 64;
    TailArgs A;
    memset(&A, 0, sizeof A);
    EdgeArgs &E = A.E;
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
    for (int b = 0; b < 64; b += c->mx) E.per_mask |= 1ull << b;
    const bool fresh = c->tail_seq != c->stamp_b;   // the batch's first attempt (a repeat finds its buffers used)
    c->tail_seq = c->stamp_b;
    // tiles: TL_WORDS_MAX words unless the stream's entries have been too dense for the staging lately (or look it by the estimate)
    if (fresh && c->tail_tw_hold > 0 && --c->tail_tw_hold == 0) c->tail_tw = std::min<uint32_t>(c->tail_tw * 2, TL_WORDS_MAX);
    // (and by the densest tile the last launch saw, scaled to the tile length: a stream that turns dense is met before it overflows)
    uint32_t tw = c->tail_tw;
    while (tw > 32 && (double)c->tail_peak * tw / (double)std::max(1u, c->tail_peak_tw) * 1.25 > (double)TL_CAP) tw /= 2;
    E.sw = tw;
    E.tps = 1;
    A.nwords = nwords;
    A.tw = tw;
    const size_t tiles = (nwords + tw - 1) / tw;
    A.ntiles = (uint32_t)tiles;
    const uint32_t cap = c->cap_edges;
    const size_t cap_al = std::max(cap, c->alloc_edges);
    HIPCHK(c, c->d_epos.ensure((cap_al + 8) * 4));
    HIPCHK(c, c->d_ecode.ensure((cap_al + 8) * 2));
    HIPCHK(c, c->d_states.ensure(cap_al + 32));
    A.epos = c->d_epos.as<uint32_t>();
    A.ecode = c->d_ecode.as<uint16_t>();
    A.outw = c->d_states.as<uint8_t>();
    A.cap = cap;
    c->edges_from_host = false;
    const bool decode = tail_can_decode(c);
    A.decode = decode ? 1 : 0;
    A.T = c->T;
    const bool enabled[2] = {c->T.tag != 0, c->T.reader != 0};
    FrameAgg *frame_total = (FrameAgg *)(tot + TOT_FRAME);
    PktCnt *pk_total = (PktCnt *)(tot + TOT_PKT0);
    FrameOut P;
    memset(&P, 0, sizeof P);
    if (decode) {
        // the packed bit arrays: tiles OR their boundary words in, so the words must be clear before ANY tile of the launch gets
        // there -- the launch of the batch before cleared this batch's buffer (two buffers alternate)
        if (fresh)
            for (int t = 0; t < 2; t++) {
                std::swap(c->d_bits[t], c->d_bits_alt[t]);
                std::swap(c->bits_clean[t], c->alt_clean[t]);
            }
        size_t cap_before[2] = {c->d_bits[0].cap, c->d_bits[1].cap};
        const int rf = frame_out(c, P, enabled);
        if (rf) return rf;
        for (int t = 0; t < 2; t++) {
            if (!P.bits[t]) continue;
            if (c->d_bits[t].cap != cap_before[t] || !fresh) c->bits_clean[t] = 0;
            const uint32_t need = (P.cap_bits[t] + 31u) / 32u + 1u;
            if (c->bits_clean[t] < need) HIPCHK(c, hipMemsetAsync(P.bits[t], 0, (size_t)need * 4, c->st));
            c->bits_clean[t] = 0;   // (used from here on)
            HIPCHK(c, c->d_bits_alt[t].ensure(c->d_bits[t].cap));
            A.Znext.p[t] = c->d_bits_alt[t].as<uint32_t>();
            A.Znext.n[t] = (uint32_t)std::min<size_t>(need + 4096, c->d_bits_alt[t].cap / 4);   // (room for a next batch that needs a little more)
            c->alt_clean[t] = A.Znext.n[t];
        }
    }
    A.P = P;
    const uint32_t mil_class_in = decode ? c->mil_q_of[c->h_dcarry.mil_state] : 0u;
    A.state0 = mil_class_in | ((uint32_t)(decode ? c->h_dcarry.man_state : 0) << 4);
    A.epi = DecCarryEpilogue{(DecMaps *)nullptr, 0u, dD(c), (uint32_t *)(tot + TOT_NSYM), pk_total,
                             {P.pend[0], P.pend[1]}, {P.started_in[0], P.started_in[1]}, {c->T.canon[0], c->T.canon[1], c->T.canon[2], c->T.canon[3]}};
    A.q_rep[0] = c->T.q_rep[0];
    A.q_rep[1] = c->T.q_rep[1];
    A.edges_total = (uint32_t *)(tot + TOT_EDGES);
    A.last2_total = (Last2 *)(tot + TOT_LAST2);
    A.carry_out = dE(c);
    A.frame_total = frame_total;
    A.verdict = (uint32_t *)(tot + TOT_SPEC);
    A.peak_out = (uint32_t *)(tot + TOT_RUNS);
    // the look-backs' status words: validated by the launch epoch, never cleared between launches -- but memory that comes fresh
    // from the allocator (or an epoch that wraps) is
    {
        const size_t need = tail_status_words(tiles) * 8;
        const size_t before = c->d_tail_st.cap;
        HIPCHK(c, c->d_tail_st.ensure(need));
        if (c->d_tail_st.cap != before || c->tail_epoch >= (1u << 30) - 2u) {
            HIPCHK(c, hipMemsetAsync(c->d_tail_st.p, 0, c->d_tail_st.cap, c->st));
            if (c->tail_epoch >= (1u << 30) - 2u) c->tail_epoch = 0;
        }
        HIPCHK(c, c->d_tail_ticket.ensure(256));
        if (c->tail_reset) {   // (the first launch, or a call that failed: the counter may not be where the host thinks)
            HIPCHK(c, hipMemsetAsync(c->d_tail_ticket.p, 0, 256, c->st));
            c->tail_ticket_base = 0;
            c->tail_reset = false;
        }
    }
    A.st = c->d_tail_st.as<uint64_t>();
    A.ticket = c->d_tail_ticket.as<uint32_t>();
    A.epoch = ++c->tail_epoch;
    A.ticket_base = c->tail_ticket_base;
    if (c->cert_pending) {
        c->cert_pending = false;
        A.C = c->cert;
    }
    const bool lds_tables = 4 * c->T.nd <= DEC_LDS_ROWS;
    if (!c->tail_occ[lds_tables]) {
        int occ = 0;
        const void *k = lds_tables ? (const void *)k_tail<true> : (const void *)k_tail<false>;
        HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, TL_BLOCK, lds_tables ? tail_table_bytes(c->T.nd) : 0));
        c->tail_occ[lds_tables] = std::max(1, occ);
    }
    const uint32_t njobs = A.ntiles + A.C.blocks;
    const uint32_t grid = std::min<uint32_t>(njobs, (uint32_t)(c->n_cus * c->tail_occ[lds_tables]));
    c->tail_ticket_base += njobs + grid;   // (every workgroup takes tickets until it draws one beyond the jobs)
    const size_t dyn = lds_tables && decode ? tail_table_bytes(c->T.nd) : 0;
    if (lds_tables) NFC_LAUNCH(k_tail<true>, dim3(grid), dim3(TL_BLOCK), dyn, c->st, A);
    else NFC_LAUNCH(k_tail<false>, dim3(grid), dim3(TL_BLOCK), dyn, c->st, A);
    c->tail_now = true;
    c->stats.tail_fused = 1;
    c->tail_tw_used = tw;
    if (!decode) return NFC_OK;   // (the caller goes on with run_decode)
    c->dec_spec_now = false;
    c->bits_packed = true;
    c->sym_lazy = true;
    c->sym_from_tail = true;
    c->sym_P = P;
    c->sym_n = cap;
    c->sym_tiles = (uint32_t)dec_num_tiles(cap);
    c->sym_own = c->sym_tiles <= c->own_prefix_max;
    launch_pkt_finish(c, P, enabled, pk_total, frame_total);
    return NFC_OK;
}
// the fused tail's verdict (host synchronised, mirror of this batch): 0, or TLV_DENSE / TLV_TIMEOUT
uint32_t tail_verdict(const nfc_ctx *c) {
    if (!c->tail_now) return 0;
    uint32_t v;
    memcpy(&v, c->hs->totals + TOT_SPEC, 4);
    return v & (TLV_DENSE | TLV_TIMEOUT);
}
// a tile's entries did not fit the staging: shorter tiles for this batch's repeat and for a while after
void tail_note_dense(nfc_ctx *c) {
    c->tail_tw = std::max<uint32_t>(32, c->tail_tw_used / 2);
    c->tail_tw_hold = 16;
    c->tail_dense_repeats++;
}
#else
constexpr uint32_t TLV_DENSE = 2u, TLV_TIMEOUT = 4u;
inline bool tail_can_decode(const nfc_ctx *) { return false; }
inline int run_tail(nfc_ctx *c, uint32_t, uint32_t, uint64_t) { return fail(c, NFC_ERR_INTERNAL, "the fused tail is not part of this build"); }
inline uint32_t tail_verdict(const nfc_ctx *) { return 0; }
inline void tail_note_dense(nfc_ctx *) {}
#endif

// The speculative decode's verdict (host synchronised, mirror of this batch): true = a tile's assumed state was wrong where it mattered
bool spec_failed(const nfc_ctx *c) {
    if (!c->dec_spec_now) return false;
    uint32_t v;
    memcpy(&v, c->hs->totals + TOT_SPEC, 4);
    return v != 0;
}
// ... then the stage is repeated in the three-launch form, and the next batches of the stream take that form straight away
// (frames longer than the run-in come in bursts: a long read, a firmware download); speculation is tried again after eight --
// after 16, 32 ... 256 while the attempts keep failing (a stream of nothing but long frames would otherwise pay a wasted
// speculative stage every ninth batch for ever); one speculative batch that stands takes the count back to eight
void note_respeculation(nfc_ctx *c) {
    c->decode_respeculated++;
    c->spec_off_left = 8 << std::min(c->spec_fail_streak, 5);
    c->spec_fail_streak++;
}
void spec_batch_done(nfc_ctx *c) {
    if (!c->dec_spec_now && c->spec_off_left > 0) c->spec_off_left--;
    else if (c->dec_spec_now) c->spec_fail_streak = 0;   // (only called for a batch whose speculative stage stood)
}

// the symbol arrays of the last batch, on demand (decode.hip.h: k_symbols_write)
int materialize_symbols(nfc_ctx *c) {
    if (!c->sym_lazy) return NFC_OK;
#ifdef NFC_TEST_HOOKS
    if (c->sym_tiles && c->sym_from_tail) {   // (k_tail leaves no aggregates per tile of DEC_TILE entries: made here, then scanned as the decode stage would)
        const uint32_t *ne_dev = (const uint32_t *)(dT(c) + TOT_EDGES);
        HIPCHK(c, c->d_partials2.ensure(((size_t)c->sym_tiles + 1) * sizeof(FrameAgg)));
        NFC_LAUNCH(k_sym_reduce, dim3(c->sym_tiles), dim3(SCAN_BLOCK), 0, c->st, c->d_states.as<uint8_t>(), (size_t)c->sym_n, ne_dev, c->d_partials2.as<FrameAgg>());
        if (!c->sym_own)
            scan_partials<FrameAggOp>(c->st, c->sym_tiles, ne_dev, DEC_TILE, c->d_partials2.as<FrameAgg>(), FrameAggOp::identity(), (FrameAgg *)nullptr);
        c->sym_from_tail = false;   // (the aggregates stand for further reads of this batch)
    }
#endif
    if (c->sym_tiles) {
        NFC_LAUNCH(k_symbols_write, dim3(c->sym_tiles), dim3(SCAN_BLOCK), 0, c->st, c->d_states.as<uint8_t>(), (size_t)c->sym_n,
                   (const uint32_t *)(dT(c) + TOT_EDGES), c->d_partials2.as<FrameAgg>(), c->sym_P, c->sym_own);
        HIPCHK(c, hipStreamSynchronize(c->st));
        BATCHCHK(c, false);
    }
    c->sym_lazy = false;   // (only now: a launch or a wait that failed leaves the arrays unwritten, and the next read tries again)
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
    A.E.sw = (uint32_t)(EW_WORDS * EW_SUPER);
    A.E.tps = (uint32_t)EW_SUPER;
    const uint32_t ce = c->cap_edges;
    const size_t ce_al = std::max(ce, c->alloc_edges);
    HIPCHK(c, c->d_epos.ensure((ce_al + 8) * 4));
    HIPCHK(c, c->d_ecode.ensure((ce_al + 8) * 2));
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
    c->dec_spec_now = false;
    c->bits_packed = false;   // (the one-launch stage keeps a byte per bit and writes the symbols itself)
    c->bits_clean[0] = c->bits_clean[1] = 0;
    c->tail_now = false;
    c->sym_lazy = false;
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
    // The ALLOCATIONS behind the estimates: a stream whose transition density changes (load modulation that starts hovering at the
    // threshold doubles it) must not grow its buffers in the middle -- VERDICT r4: the hovering stream's second batch took 4.4 ms
    // instead of 2.5, all of it hipMalloc.  Room for n / 4 entries (a clean capture has n / 15) and a symbol per entry; what does not
    // fit even that is grown by the repeat path of process_batch.  Monotonic: a short batch between long ones does not shrink them.
    // About 40 bytes hang off every entry allocated for (positions, codes, out-bytes, symbols, bits, pending x 2, packet ends x 2): a
    // 1e9-sample batch reserves 10 GB beside its 8 GB of input.  Where the device does not have that to spare the roomy figure
    // gives way to the estimate (the repeat path grows what a batch then overflows); INTEGRATION.md states the footprint.
    uint64_t ae = std::max<uint64_t>(std::min<uint64_t>((uint64_t)n / 4 + 65536, 0xFFFFFF00u), c->cap_edges);
    if (ae > c->alloc_edges) {   // (about to grow: once per batch length)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t fit = (uint64_t)free_b / 2 / 40;   // half of what is free, for everything that scales with the entries
            if (ae - c->alloc_edges > fit) ae = std::max<uint64_t>(c->cap_edges, c->alloc_edges + fit);
        }
    }
    c->alloc_edges = std::max(c->alloc_edges, (uint32_t)ae);
    for (int t = 0; t < 2; t++) c->alloc_sym[t] = std::max(c->alloc_sym[t], std::max(c->cap_sym[t], c->alloc_edges));
}
// densities for the next batch's estimates
void update_estimates(nfc_ctx *c, uint32_t n) {
    if (c->tail_now) {   // the densest tile of the batch just adopted (tail.hip.h), at the tile length it ran with
        uint32_t pk;
        memcpy(&pk, c->hs->totals + TOT_RUNS, 4);
        const double old = (double)c->tail_peak * c->tail_tw_used / (double)std::max(1u, c->tail_peak_tw) * 0.8;
        c->tail_peak = std::max(pk, (uint32_t)old);
        c->tail_peak_tw = c->tail_tw_used;
    }
    c->cap_edges_floor = 0;
    c->cap_sym_floor[0] = c->cap_sym_floor[1] = 0;
    c->edge_rate = std::max({(double)c->n_edges / (double)n, c->edge_rate * 0.9, 1.0 / 64});
    for (int t = 0; t < 2; t++)
        if (c->n_edges) c->sym_rate[t] = std::max((double)c->n_sym[t] / (double)c->n_edges, c->sym_rate[t] * 0.9);
}

// nfc_stats.ring_slots_carried of the batch just adopted: the host is synchronised with the stream and the mirror is this batch's
void note_carried(nfc_ctx *c) {
    CertSummary cs;
    memcpy(&cs, c->hs->totals + TOT_CERT, sizeof cs);
    c->ring_carried = cs.n_carried;
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
    c->alloc_mark = devbuf_allocs();
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
        note_carried(c);
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
        if (c->tail_on) {   // one persistent launch for the three stages (tail.hip.h); the edge stage alone where the decoders cannot run in it
            const bool fused = tail_can_decode(c);
            int r = run_tail(c, n, skip, g0);
            if (r) return r;
            if (!ev3_done && c->timing >= 2) HIPCHK(c, hipEventRecord(c->ev[3], c->st));
            ev3_done = true;
            return fused ? NFC_OK : run_decode(c, true);
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
        bool decode_only = false;   // the edges stand, the decode stage is repeated in the form that assumes nothing
        int respeculated = 0;       // (at most once per batch: the three-launch form assumes nothing -- its own bound, not the capacities')
        for (int attempt = 0;; attempt++) {
            if (attempt > 0 || !clean || decode_only) {
                if (decode_only) {   // (ms_decode then holds the failed speculative pass, the host's turn and the repeat: ev[3] stays where it was)
                    rc = run_decode(c, true);
                } else {
                    rc = edges_and_decode();
                }
                if (rc) return rc;
                HIPCHK(c, hipStreamSynchronize(c->st));
            }
            BATCHCHK(c, true);
            uint32_t ne, ns[2];
            memcpy(&ne, c->hs->totals + TOT_EDGES, 4);
            memcpy(ns, c->hs->totals + TOT_NSYM, 8);
            if (const uint32_t tv = tail_verdict(c)) {   // (tail.hip.h: a tile too dense for the staging -- or a look-back that gave up)
                if (tv & TLV_TIMEOUT) return fail(c, NFC_ERR_DEVICE, "the fused tail's look-back timed out");
                if (c->tail_tw_used <= 32) return fail(c, NFC_ERR_INTERNAL, "a tile of 2048 samples did not fit the fused tail's staging");
                tail_note_dense(c);
                decode_only = false;
                attempt--;   // (not a capacity attempt; the tile length at least halves every time)
                clean = false;
                continue;
            }
            const bool fit = ne <= c->cap_edges && ns[0] + 2 <= c->cap_sym[0] && ns[1] + 2 <= c->cap_sym[1];
            if (fit && spec_failed(c)) {   // (decode.hip.h: dec_verify -- a decode tile's assumed incoming state was wrong)
                if (respeculated++) return fail(c, NFC_ERR_INTERNAL, "the decode stage's three-launch form reported a speculation failure");
                note_respeculation(c);
                decode_only = true;
                attempt--;   // (not a capacity attempt)
                continue;
            }
            if (fit) {
                c->n_edges = ne;
                c->n_sym[0] = ns[0];
                c->n_sym[1] = ns[1];
                break;
            }
            decode_only = false;
            // a buffer was too small: the stages read carried values by value and wrote only write-only slots, so
            // they can simply run again with room for what was counted
            if (attempt >= 3) return fail(c, NFC_ERR_INTERNAL, "edge / symbol capacity did not settle");
            c->cap_edges_floor = (uint64_t)std::max(ne, c->cap_edges) * 5 / 4 + 65536;
            c->cap_sym_floor[0] = (uint64_t)ns[0] * 5 / 4 + 65536;
            c->cap_sym_floor[1] = (uint64_t)ns[1] * 5 / 4 + 65536;
        }
        update_estimates(c, n);
        spec_batch_done(c);
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
    note_carried(c);
    for (int i = 0; i < c->n_kev; i++) c->stats.ms_threshold_kernel[i] = elapsed_ms(c->kev[2 * i], c->kev[2 * i + 1]);
    c->stats.n_threshold_timed = (uint32_t)c->n_kev;
    c->nseen += n;
    c->last_in = d_in;
    c->have_outputs = true;
    c->stats.device_allocs = (uint32_t)(devbuf_allocs() - c->alloc_mark);
    // (the end-of-batch LOW bookkeeping on the device is what a batch submitted ahead may start from: only when the whole
    // batch went through one parallel attempt, whose last certification workgroup or k_finalize_state wrote it)
    c->low_valid = skip == 0 && c->stats.used_sequential == 0 && c->h_carry.stable;
    return NFC_OK;
}


}  // namespace
