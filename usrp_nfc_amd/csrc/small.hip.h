// small.hip.h -- edges, decoders and framing of a SHORT batch in one launch.
//
// The multi-launch stages (edges.hip.h, decode.hip.h) pay some twenty kernel boundaries per batch whatever
// its length; a batch of a few tens of thousands of samples (a shard's warm-up overlap, a GNU Radio work()
// call) is all boundary.  Here ONE workgroup walks the tiles of each stage in order, so every scan is a
// single pass with the running prefix in registers -- no tile aggregates, no partials pass, no second read.
// The per-item arithmetic is the same device code the large path runs (change_mask, event_mask,
// event_entry, the LUT walk, frame_agg_of / frame_write): only the orchestration differs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode.hip.h"
#include "edges.hip.h"
#include "scan.hip.h"

namespace nfc {

constexpr int SM_BLOCK = 256, SM_WAVES = SM_BLOCK / 64;   // four waves: room for registers (no scratch), short barriers
constexpr int SM_EVCAP = 4096;              // entries listed per round of the edge writer
constexpr uint32_t SM_MAX_SAMPLES = 1u << 18;   // batches up to this length take the one-launch path

struct SmallArgs {
    EdgeArgs E;
    size_t nwords;
    uint32_t *epos;         // entries as (batch-local sample position, code): edges.hip.h
    uint16_t *ecode;
    uint32_t cap_edges;
    DecTables T;
    uint32_t dec_state_in;
    FrameOut P;             // (epos: the array this launch writes)
    int32_t enabled[2];
    uint8_t *pending_next[2];
    uint32_t pending_cap[2];
    Last2 *tot_last2;
    uint32_t *tot_edges;
    DecMaps *tot_decmap;
    FrameAgg *tot_frame;
    PktCnt *tot_pk;
    uint32_t *tot_nsym;
    EdgeCarry *ecarry;
    DecCarry *dcarry;
    const uint32_t *mirror_src;   // the device state block -> the host's mapped mirror (decode.hip.h: PktFinish), or NULL
    uint32_t *mirror_dst;
    uint32_t mirror_words;
    uint32_t stamp_word, stamp;   // (decode.hip.h: PktFinish)
};

__global__ __launch_bounds__(SM_BLOCK) void k_small_stage(SmallArgs A) {
    // one LDS arena, carved per phase
    __shared__ __attribute__((aligned(16))) unsigned char smem[36864];
    __shared__ Last2 r_l2[SM_WAVES];
    __shared__ uint32_t r_u32[SM_WAVES];
    __shared__ FramePk r_fa[SM_WAVES];
    __shared__ DecMaps r_map[SM_WAVES];
    const int tid = threadIdx.x;

    // ================= edges: last-two-changes scan, event masks, entry offsets, entries =================
    uint32_t n_edges_all = 0;
    {
        uint64_t *s_ng = (uint64_t *)smem, *s_ps = s_ng + SM_BLOCK, *s_m = s_ps + SM_BLOCK;
        Last2 *s_ctx = (Last2 *)(s_m + SM_BLOCK);
        uint16_t *s_ev = (uint16_t *)(s_ctx + SM_BLOCK);   // 8 KB + 8 KB
        Last2 run_l2 = Last2Op::identity();
        for (size_t w0 = 0; w0 < A.nwords; w0 += SM_BLOCK) {
            const size_t w = w0 + tid;
            uint64_t ng = 0, ps = 0, m = 0;
            Last2 item = Last2Op::identity();
            if (w < A.nwords) {
                m = A.E.change_mask(w, ng, ps);
                if (m) {
                    const int b1 = 63 - __clzll((long long)m);
                    const uint64_t m2 = m & ~(1ull << b1);
                    item = Last2{(int32_t)(w * 64) + b1, m2 ? (int32_t)(w * 64) + (63 - __clzll((long long)m2)) : POS_NONE};
                }
            }
            Last2 tot_l2;
            const Last2 excl = block_exclusive<Last2Op, SM_WAVES>(item, r_l2, tot_l2);
            const Last2 ctxw = Last2Op::op(run_l2, excl);
            const uint64_t ev = (w < A.nwords) ? event_mask(A.E, w, ctxw, m) : 0ull;
            uint32_t total;
            const uint32_t off = block_exclusive<AddU32, SM_WAVES>((uint32_t)__popcll(ev), r_u32, total);
            s_ng[tid] = ng;
            s_ps[tid] = ps;
            s_m[tid] = m;
            s_ctx[tid] = ctxw;
            for (uint32_t rbase = 0; rbase < total; rbase += SM_EVCAP) {
                uint32_t k = off - rbase;   // wraps below the round: the unsigned compare drops those
                uint64_t e = ev;
                while (e) {
                    if (k < (uint32_t)SM_EVCAP) s_ev[k] = (uint16_t)((tid << 6) | (__ffsll((long long)e) - 1));
                    e &= e - 1;
                    k++;
                }
                __syncthreads();
                const uint32_t cnt = min((uint32_t)SM_EVCAP, total - rbase);
                for (uint32_t j = tid; j < cnt; j += SM_BLOCK) {
                    const uint32_t code = s_ev[j];
                    const int wl = (int)(code >> 6), b = (int)(code & 63u);
                    const int32_t p0 = (int32_t)((w0 + wl) * 64);
                    int v, d, t;
                    event_entry(A.E, p0, b, s_ng[wl], s_ps[wl], s_m[wl], s_ctx[wl], v, d, t);
                    const uint32_t g = n_edges_all + rbase + j;
                    if (g < A.cap_edges) {
                        A.epos[g] = (uint32_t)(p0 + b);
                        A.ecode[g] = edge_code(v, d, t, A.E.nd);
                    }
                }
                __syncthreads();
            }
            run_l2 = Last2Op::op(run_l2, tot_l2);
            n_edges_all += total;
        }
        if (tid == 0) {
            *A.tot_last2 = run_l2;
            *A.tot_edges = n_edges_all;
            if (A.E.skip < A.E.n) {   // (nothing but fill samples: unchanged)
                int lb, dur, st;
                A.E.state_before((int32_t)A.E.n, run_l2, lb, dur, st);
                A.ecarry->last_bit = lb;
                A.ecarry->dur = dur;
                A.ecarry->state = st;
            }
        }
    }
    __syncthreads();
    const size_t ne = min(n_edges_all, A.cap_edges);   // (an overflow makes the host repeat the batch with room)

    // ================= decoders: state maps, walk, symbols; framing in the same pass (decode.hip.h: edge domain) =================
    {
        uint4 *s_mil = (uint4 *)smem;                      // 8 KB   maps for composing
        uint2 *s_man = (uint2 *)(smem + 8192);             // 4 KB
        uint16_t *s_mstep = (uint16_t *)(smem + 12288);    // 16 KB  next state | out byte, for walking
        uint16_t *s_nstep = (uint16_t *)(smem + 28672);    // 8 KB
        const bool lds_tab = 4 * A.T.nd <= DEC_LDS_ROWS;
        if (lds_tab) {
            const int rows = 4 * A.T.nd;
            for (int i = tid; i < rows; i += SM_BLOCK) {
                if (A.T.reader) s_mil[i] = A.T.mil_map[i];
                if (A.T.tag) s_man[i] = A.T.man_map[i];
            }
            if (A.T.reader)
                for (int i = tid; i < rows * 2; i += SM_BLOCK) ((uint4 *)s_mstep)[i] = ((const uint4 *)A.T.mil_step)[i];
            if (A.T.tag)
                for (int i = tid; i < rows; i += SM_BLOCK) ((uint4 *)s_nstep)[i] = ((const uint4 *)A.T.man_step)[i];
            __syncthreads();
        }
        const uint4 *mil_map = lds_tab ? s_mil : A.T.mil_map;
        const uint2 *man_map = lds_tab ? s_man : A.T.man_map;
        const uint16_t *mil = lds_tab ? s_mstep : A.T.mil_step;
        const uint16_t *man = lds_tab ? s_nstep : A.T.man_step;
        DecMaps run_map = ComposeDec::identity();
        FrameAgg run_fa = FrameAggOp::identity();
        copy_pending(A.P, tid, SM_BLOCK);
        for (size_t e0 = 0; e0 < ne; e0 += (size_t)SM_BLOCK * DEC_ITEMS) {
            const size_t base = e0 + (size_t)tid * DEC_ITEMS;
            uint32_t c[8];
            load_codes(A.ecode, base, ne, c);
            DecMaps agg = ComposeDec::identity();
#pragma unroll
            for (int k = 0; k < DEC_ITEMS; k++) {
                const uint32_t code = (c[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                const uint32_t li = code & 0x3FFFu, route = code >> 14;
                if (route == 2u && A.T.reader) {
                    const uint4 v = mil_map[li];
#pragma unroll
                    for (int q = 0; q < 4; q++) agg.mil[q] = lookup16x4(v.x, v.y, v.z, v.w, agg.mil[q]);
                } else if (route == 1u && A.T.tag) {
                    const uint2 v = man_map[li];
                    agg.man[0] = __builtin_amdgcn_perm(v.y, v.x, agg.man[0]);
                    agg.man[1] = __builtin_amdgcn_perm(v.y, v.x, agg.man[1]);
                }
            }
            DecMaps tot_map;
            const DecMaps excl = block_exclusive<ComposeDec, SM_WAVES>(agg, r_map, tot_map);
            uint32_t st = ComposeDec::step(ComposeDec::op(run_map, excl), A.dec_state_in);
            uint32_t ow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < DEC_ITEMS; k++) {
                const uint32_t code = (c[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                const uint32_t li = code & 0x3FFFu, route = code >> 14;
                uint32_t w = 0;
                if (route == 2u && A.T.reader) {
                    const uint32_t e = mil[li * 16u + (st & 15u)];
                    w = e >> 8;
                    st = (st & ~15u) | (e & 15u);
                } else if (route == 1u && A.T.tag) {
                    const uint32_t e = man[li * 8u + ((st >> 4) & 7u)];
                    const uint32_t mo = e >> 8;
                    w = (mo & 3u) ? ((mo & 0xFCu) | 3u) : 0u;
                    st = (st & 15u) | ((e & 15u) << 4);
                }
                ow[k >> 2] |= w << (8 * (k & 3));
            }
            // symbols, packet bits and packet ends to their places (decode.hip.h: one aggregate carries all three offsets)
            FramePk tot_fa;
            const FramePk in_tile = block_exclusive<FramePkOp, SM_WAVES>(FramePkOp::pack(frame_agg_of(ow)), r_fa, tot_fa);
            if (ow[0] | ow[1] | ow[2] | ow[3]) frame_write(A.P, FrameAggOp::op(run_fa, FramePkOp::unpack(in_tile)), ow, base);
            run_map = ComposeDec::op(run_map, tot_map);
            run_fa = FrameAggOp::op(run_fa, FramePkOp::unpack(tot_fa));
        }
        __syncthreads();   // the bits and close offsets of every thread
        if (tid == 0) {
            *A.tot_decmap = run_map;
            *A.tot_frame = run_fa;
            const uint32_t st = ComposeDec::step(run_map, A.dec_state_in);
            A.dcarry->mil_state = (int32_t)byte_of(A.T.canon, st & 15u);   // (the canonical state of its class: DecTables)
            A.dcarry->man_state = (int32_t)(st >> 4);
            A.tot_nsym[1] = run_fa.cnt[1];   // Miller / reader
            A.tot_nsym[0] = run_fa.cnt[0];   // Manchester / tag
        }
        for (int t = 0; t < 2; t++) {
            const uint32_t nbits = A.P.pend[t] + fa_bits(run_fa, t, A.P.started_in[t]), ncl = fa_closes(run_fa, t, A.P.started_in[t]);
            if (tid == 0) A.tot_pk->v[t] = (uint64_t)nbits | ((uint64_t)ncl << 32);
            if (!A.enabled[t]) continue;
            if (nbits > A.P.cap_bits[t] || ncl > A.P.cap_close[t]) continue;   // the host repeats the stage with room
            const uint32_t from = ncl ? A.P.close_end[t][ncl - 1] : 0u;
            const uint32_t keep = nbits - from;
            for (uint32_t i = tid; i < keep && i < A.pending_cap[t]; i += SM_BLOCK) A.pending_next[t][i] = A.P.bits[t][from + i];
            if (tid == 0) {
                A.dcarry->pending[t] = keep;
                A.dcarry->pkt_started[t] = (int32_t)pm_apply(run_fa.fl[t], A.P.started_in[t]);
            }
        }
    }
    if (A.mirror_src) {   // the batch's last launch: the state block goes to the host's mirror from here
        __threadfence_block();
        __syncthreads();
        for (uint32_t i = tid; i < A.mirror_words; i += SM_BLOCK)
            if (i != A.stamp_word) A.mirror_dst[i] = A.mirror_src[i];
        __threadfence_system();   // (the stamp last: decode.hip.h, k_pkt_finish)
        __syncthreads();
        if (tid == 0) {
            ((volatile uint32_t *)A.mirror_dst)[A.stamp_word] = A.stamp;
        }
    }
}

}  // namespace nfc
