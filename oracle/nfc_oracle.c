/*
 * nfc_oracle.c -- CPU restatement (plain C) of the giech/usrp_nfc ISO-14443A
 * eavesdrop hot path: envelope -> gated running-mean threshold -> edge timing
 * -> Modified-Miller / Manchester symbol decode -> packet framing.
 *
 * TEST INFRASTRUCTURE ONLY.  The product (usrp_nfc_amd/, libnfc_amd.so) never
 * links, loads or calls this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED to the reference by tests/test_oracle_golden.py
 * (golden vectors produced by the unmodified reference modules, see
 * tests/golden/make_golden.py).  The GNU Radio envelope block is third-party
 * and absent; its fp32 formula is restated here (unpinned at that boundary).
 *
 * Citations are reference file:line, paths relative to /root/reference/code.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* utilities.py:7-14 */
enum { ERR_NONE = 0, ERR_TOO_SHORT = 2, ERR_TOO_LONG = 3, ERR_ENCODING = 4, ERR_INTERNAL = 5, ERR_WRONG_DUR = 6 };
/* utilities.py:17-23 -- same expressions, evaluated in IEEE double */
static const double T_FULL = 9.44;
static const double T_ZERO = 3.00;
#define T_HALF (T_FULL / 2)
#define T_ZERO_REM (T_FULL - T_ZERO)
#define T_ONE_REM (T_HALF - T_ZERO)
#define T_ONE_HALF (T_FULL + T_HALF)
/* packets.py:19-20 */
enum { TAG_TO_READER = 0, READER_TO_TAG = 1 };

typedef struct {
    double samp_rate, lo, hi;
    int32_t av_window, max_len, reader, tag;
} orc_params;

typedef struct {
    int64_t idx;  /* sample index (0-based, whole stream) at which the reference appended it */
    int32_t d;    /* duration in samples; the reference emits d * factor microseconds */
    int8_t v, t;
    int16_t pad;
} orc_edge;

typedef struct { void *p; size_t n, cap, esz; } vec;

static void vec_push(vec *v, const void *e) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 1024;
        v->p = realloc(v->p, v->cap * v->esz);
    }
    memcpy((char *)v->p + v->n * v->esz, e, v->esz);
    v->n++;
}

typedef struct {
    /* miller.py:19-29 */
    int prev, started, cur_type;
    double dur0, dur1, tol, lo, hi;
} miller_t;

typedef struct {
    /* manchester.py:15-25 */
    double lo, mid, hi;
    int prev_set, prev;
} manch_t;

typedef struct {
    /* packets.py:58-65 */
    int start_bit, started;
    vec cur; /* uint8 */
} pproc_t;

typedef struct orc {
    orc_params p;
    /* transition_sink.py:20-34 */
    double factor, total;
    double *ring;
    int64_t index, filled, dur, nseen;
    int last_bit, state, stable;
    miller_t mil;
    manch_t man;
    pproc_t pp[2];
    vec edges;      /* orc_edge */
    vec sym[2];     /* uint8 symbol streams per packet type */
    vec pk_type;    /* int8 */
    vec pk_len;     /* int32 */
    vec pk_bits;    /* uint8 */
    vec trace;      /* int8 val per stable sample (debug tap, not in the reference) */
    int want_trace;
} orc;

/* ---- packets.py:67-79 + 94-98 ---- */
static void pp_append(orc *o, int bit, int type) {
    pproc_t *pp = &o->pp[type];
    uint8_t b = (uint8_t)bit;
    vec_push(&o->sym[type], &b);
    if (bit != 0 && bit != 1) {
        if (pp->started) {
            if (pp->cur.n) { /* packets.py:97 `if ret` drops empty lists */
                int8_t t = (int8_t)type;
                int32_t n = (int32_t)pp->cur.n;
                vec_push(&o->pk_type, &t);
                vec_push(&o->pk_len, &n);
                for (size_t i = 0; i < pp->cur.n; i++) vec_push(&o->pk_bits, (uint8_t *)pp->cur.p + i);
            }
            pp->started = 0;
            pp->cur.n = 0;
        }
    } else if (!pp->started && bit == pp->start_bit) {
        pp->started = 1;
    } else {
        vec_push(&pp->cur, &b);
    }
}

/* ---- miller.py ---- */
enum { ST_BEGIN = 0, ST_ZS0 = 1, ST_OS0 = 2, ST_OS1 = 3 };

static int mil_stage(const miller_t *m) { /* :31-41 */
    if (m->dur0 == 0) return ST_BEGIN;
    if (m->cur_type == 0) return ST_ZS0;
    return m->dur1 == 0 ? ST_OS0 : ST_OS1;
}
static void mil_set_stage(miller_t *m, int st) { /* :43-59 */
    switch (st) {
    case ST_BEGIN: m->dur0 = 0; m->dur1 = 0; m->cur_type = 0; break;
    case ST_ZS0: m->dur0 = T_ZERO; m->cur_type = 0; break;
    case ST_OS0: m->dur0 = T_HALF; m->cur_type = 1; break;
    default: m->dur0 = T_HALF; m->dur1 = T_ZERO; m->cur_type = 1; break;
    }
}
static int mil_close(const miller_t *m, double dur, double av) { return fabs(dur - av) <= m->tol; } /* :62-63 */
static void mil_reset(miller_t *m) { m->started = 0; mil_set_stage(m, ST_BEGIN); }                 /* :65-67 */
static void mil_init(miller_t *m) {                                                                 /* :19-29 */
    m->prev = 0;
    m->tol = 1.5;
    m->lo = T_ZERO - m->tol;
    m->hi = 2 * T_FULL;
    mil_reset(m);
}

static int mil_begin(miller_t *m, int cur, double dur, int *r) { /* :73-96 */
    int n = 0;
    if (cur == 0) {
        if (mil_close(m, dur, T_ZERO)) { mil_set_stage(m, ST_ZS0); m->started = 1; }
        else r[n++] = ERR_TOO_LONG;
    } else if (m->started) {
        int bit = 0;
        if (m->prev == 0) bit = ERR_ENCODING;
        if (mil_close(m, dur, T_HALF)) mil_set_stage(m, ST_OS0);
        else if (mil_close(m, dur, T_FULL)) r[n++] = bit;
        else if (mil_close(m, dur, T_ONE_HALF)) { r[n++] = bit; mil_set_stage(m, ST_OS0); }
        else r[n++] = ERR_WRONG_DUR;
    }
    return n;
}
static int mil_zs0(miller_t *m, int cur, double dur, int *r) { /* :98-112 */
    int n = 0;
    if (cur == 0) r[n++] = ERR_ENCODING;
    else if (mil_close(m, dur, T_ZERO_REM)) { mil_set_stage(m, ST_BEGIN); r[n++] = 0; }
    else if (mil_close(m, dur, T_ZERO_REM + T_HALF)) { mil_set_stage(m, ST_OS0); r[n++] = 0; }
    else r[n++] = ERR_WRONG_DUR;
    return n;
}
static int mil_os0(miller_t *m, int cur, double dur, int *r) { /* :114-122 */
    int n = 0;
    if (cur != 0) r[n++] = ERR_ENCODING;
    else if (!mil_close(m, dur, T_ZERO)) r[n++] = ERR_WRONG_DUR;
    else mil_set_stage(m, ST_OS1);
    return n;
}
static int mil_os1(miller_t *m, int cur, double dur, int *r) { /* :124-148 */
    int n = 0;
    if (cur != 1) r[n++] = ERR_ENCODING;
    else if (mil_close(m, dur, T_ONE_REM)) { r[n++] = 1; mil_set_stage(m, ST_BEGIN); }
    else {
        r[n++] = 1;
        mil_set_stage(m, ST_BEGIN);
        dur -= T_ONE_REM;
        if (mil_close(m, dur, T_FULL)) r[n++] = 0;
        else if (mil_close(m, dur, T_HALF)) mil_set_stage(m, ST_OS0);
        else if (mil_close(m, dur, T_ONE_HALF)) { r[n++] = 0; mil_set_stage(m, ST_OS0); }
        else r[n++] = ERR_WRONG_DUR;
    }
    return n;
}
static void mil_step(orc *o, int cur, double dur) { /* body of :154-197 */
    miller_t *m = &o->mil;
    if (cur == 0 && fabs(dur - T_ZERO) < T_ZERO / 2) dur = T_ZERO; /* :157-158 */
    int err = ERR_NONE;
    int st = mil_stage(m);
    if ((dur < m->lo || dur > m->hi) && (st == ST_ZS0 || st == ST_OS1)) { /* :165-167 */
        pp_append(o, m->cur_type, READER_TO_TAG);
        err = ERR_TOO_LONG;
    } else if (dur < m->lo) err = ERR_TOO_SHORT;
    else if (dur > m->hi) err = ERR_TOO_LONG;
    if (err != ERR_NONE) { /* :173-176 */
        pp_append(o, err, READER_TO_TAG);
        mil_reset(m);
        return;
    }
    int r[4], n;
    if (st == ST_BEGIN) n = mil_begin(m, cur, dur, r);
    else if (st == ST_ZS0) n = mil_zs0(m, cur, dur, r);
    else if (st == ST_OS0) n = mil_os0(m, cur, dur, r);
    else n = mil_os1(m, cur, dur, r);
    for (int i = 0; i < n; i++) { /* :191-197 */
        pp_append(o, r[i], READER_TO_TAG);
        if (r[i] > 1) { mil_reset(m); m->prev = 0; }
        else m->prev = r[i];
    }
}

/* ---- manchester.py:30-61 ---- */
static void man_init(manch_t *m) {
    m->lo = T_HALF - 1;
    m->mid = T_HALF + 1;
    m->hi = 2 * T_HALF + 1;
    m->prev_set = 0;
    m->prev = 0;
}
static void man_step(orc *o, int cur, double dur) {
    manch_t *m = &o->man;
    int err = ERR_NONE;
    if (dur < m->lo) err = ERR_TOO_SHORT;
    else if (dur > m->hi) err = ERR_TOO_LONG;
    if (err != ERR_NONE) {
        m->prev_set = 0;
        m->prev = 0;
        pp_append(o, err, TAG_TO_READER);
        return;
    }
    int dual = dur > m->mid;
    int prev = m->prev;
    if (m->prev_set) {
        if (prev == cur || (prev != 0 && prev != 1)) { pp_append(o, ERR_INTERNAL, TAG_TO_READER); return; }
        pp_append(o, prev, TAG_TO_READER);
        m->prev_set = dual;
    } else {
        if (dual) { pp_append(o, ERR_ENCODING, TAG_TO_READER); return; }
        m->prev_set = 1;
    }
    m->prev = cur;
}

/* background.py:30-52 flattened: the grouping into same-type runs only batches
 * calls to stateful decoders, so routing each transition by its type is the
 * same computation. */
static void emit(orc *o, int v, int64_t d, int t) {
    orc_edge e;
    e.idx = o->nseen;
    e.d = (int32_t)d;
    e.v = (int8_t)v;
    e.t = (int8_t)t;
    e.pad = 0;
    vec_push(&o->edges, &e);
    double us = (double)d * o->factor; /* transition_sink.py:89,97: d*factor */
    if (t == TAG_TO_READER && o->p.tag) man_step(o, v, us);
    else if (t == READER_TO_TAG && o->p.reader) mil_step(o, v, us);
}

/* ---- transition_sink.py ---- */
orc *orc_create(const orc_params *p, int want_trace) {
    orc *o = (orc *)calloc(1, sizeof(orc));
    o->p = *p;
    o->factor = 1e6 / p->samp_rate; /* :21 */
    o->dur = 1;                     /* :22 */
    o->ring = (double *)calloc((size_t)(p->av_window > 0 ? p->av_window : 1), sizeof(double));
    o->edges.esz = sizeof(orc_edge);
    o->sym[0].esz = o->sym[1].esz = 1;
    o->pk_type.esz = 1;
    o->pk_len.esz = 4;
    o->pk_bits.esz = 1;
    o->trace.esz = 1;
    o->want_trace = want_trace;
    for (int t = 0; t < 2; t++) {
        o->pp[t].start_bit = (t == TAG_TO_READER) ? 1 : 0; /* packets.py:24-28 */
        o->pp[t].cur.esz = 1;
    }
    mil_init(&o->mil);
    man_init(&o->man);
    return o;
}

void orc_destroy(orc *o) {
    if (!o) return;
    free(o->ring); free(o->edges.p); free(o->sym[0].p); free(o->sym[1].p);
    free(o->pk_type.p); free(o->pk_len.p); free(o->pk_bits.p); free(o->trace.p);
    free(o->pp[0].cur.p); free(o->pp[1].cur.p);
    free(o);
}

static inline void one_sample(orc *o, float xf) {
    const int64_t L = o->p.av_window, mx = o->p.max_len;
    double bit = (double)xf; /* .tolist() of a float32 array, :39 */
    if (!o->stable) {        /* :109-125 */
        o->ring[o->filled++] = bit;
        if (o->filled == L) {
            double s = 0;
            for (int64_t i = 0; i < L; i++) s += o->ring[i]; /* :122 */
            o->total = s;
            o->dur = L % mx;                                  /* :123 */
            o->stable = 1;
        }
        o->nseen++;
        return;
    }
    double prev = o->ring[o->index];
    int prev_state = o->state;
    double ratio;
    if (o->total == 0) ratio = (bit == 0) ? 1 : o->p.hi + 0.1; /* :59-63 */
    else ratio = bit * (double)L / o->total;                   /* :65 */
    int val;
    double cur;
    if (o->p.lo > ratio) { val = -1; cur = prev; o->state = 2; }                        /* :67-70 */
    else if (o->state != 2 && ratio > o->p.hi) { val = 1; cur = prev; o->state = 1; }   /* :71-74 */
    else { val = 0; cur = bit; }                                                        /* :75-77 */
    o->ring[o->index] = cur;                 /* :80 */
    o->index = (o->index + 1) % L;           /* :81 */
    o->total += (cur - prev);                /* :82 */
    if (o->want_trace) { int8_t v8 = (int8_t)val; vec_push(&o->trace, &v8); }
    if (val == o->last_bit) o->dur += 1;     /* :84-85 */
    else {                                   /* :86-92 */
        int64_t d = prev_state == 0 ? mx : o->dur;
        int v = o->state == 2 ? o->last_bit + 1 : o->last_bit;
        emit(o, v, d, o->state - 1);
        o->dur = 1;
        o->last_bit = val;
    }
    if (o->dur > mx) {                       /* :95-99 */
        int v = o->state == 2 ? o->last_bit + 1 : o->last_bit;
        emit(o, v, mx, o->state - 1);
        o->dur = 1;
        o->state = 0;
    }
    o->nseen++;
}

void orc_push_env(orc *o, const float *x, size_t n) {
    if (o->p.av_window <= 0) return;
    for (size_t i = 0; i < n; i++) one_sample(o, x[i]);
}

/* gnuradio complex_to_mag_squared (decoder.py:27, usrp_src.py:31): fp32,
 * one rounding per product and per sum; -ffp-contract=off keeps it unfused. */
void orc_push_iq(orc *o, const float *iq, size_t n) {
    if (o->p.av_window <= 0) return;
    for (size_t i = 0; i < n; i++) {
        volatile float a = iq[2 * i] * iq[2 * i];
        volatile float b = iq[2 * i + 1] * iq[2 * i + 1];
        one_sample(o, a + b);
    }
}

/* WAV branch (decoder.py:25-28): real sample, Q = 0 -> fl(x*x) */
void orc_push_real_sq(orc *o, const float *x, size_t n) {
    if (o->p.av_window <= 0) return;
    for (size_t i = 0; i < n; i++) {
        volatile float a = x[i] * x[i];
        one_sample(o, a);
    }
}

size_t orc_n_edges(const orc *o) { return o->edges.n; }
const orc_edge *orc_edges(const orc *o) { return (const orc_edge *)o->edges.p; }
size_t orc_n_symbols(const orc *o, int type) { return o->sym[type].n; }
const uint8_t *orc_symbols(const orc *o, int type) { return (const uint8_t *)o->sym[type].p; }
size_t orc_n_packets(const orc *o) { return o->pk_type.n; }
const int8_t *orc_packet_types(const orc *o) { return (const int8_t *)o->pk_type.p; }
const int32_t *orc_packet_lens(const orc *o) { return (const int32_t *)o->pk_len.p; }
size_t orc_n_packet_bits(const orc *o) { return o->pk_bits.n; }
const uint8_t *orc_packet_bits(const orc *o) { return (const uint8_t *)o->pk_bits.p; }
size_t orc_n_trace(const orc *o) { return o->trace.n; }
const int8_t *orc_trace(const orc *o) { return (const int8_t *)o->trace.p; }
double orc_total(const orc *o) { return o->total; }
int64_t orc_nseen(const orc *o) { return o->nseen; }
void orc_clear_outputs(orc *o) {
    o->edges.n = o->sym[0].n = o->sym[1].n = 0;
    o->pk_type.n = o->pk_len.n = o->pk_bits.n = o->trace.n = 0;
}
