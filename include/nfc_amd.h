/*
 * nfc_amd.h -- C-ABI of the MI355X-native ISO-14443A IQ -> bit eavesdrop path.
 *
 * This is the drop-in boundary for the one hot path of giech/usrp_nfc:
 *
 *   envelope (gnuradio complex_to_mag_squared, decoder.py:27 / usrp_src.py:31)
 *   -> transition_sink   (transition_sink.py:10-125: gated running mean, lo/hi
 *                         ratio threshold with hysteresis, run-length timing)
 *   -> background router (background.py:30-52)
 *   -> miller_decoder / manchester_decoder (miller.py:13-197, manchester.py:13-61)
 *   -> PacketProcessor   (packets.py:57-98)
 *
 * The reference has no FFI of its own (it is pure Python on GNU Radio); the
 * functions below are what a ctypes binding placed inside the reference's
 * transition_sink.work()/background.append() would call.  INTEGRATION.md shows
 * that binding.  Plain C types only; every buffer is caller-allocated; every
 * function returns 0 on success or a negative nfc_status and leaves a message
 * for nfc_last_error().  A context is one stream; it is not thread-safe.  All
 * compute runs in hand-written HIP kernels on the selected device: there is no
 * CPU fallback, and nfc_create fails if no GPU is usable.
 */
#ifndef NFC_AMD_H
#define NFC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: nfc_stats grew (ran_ahead, redone_total, ring_slots_carried); i16_scale == 0 means sample / 32767 (GNU Radio's wavfile_source), not / 32768 */
/* 3: nfc_stats.reserved0 became decode_respeculated (same layout); the raw float32 envelope takes the fast threshold kernels */
/* 4: nfc_stats grew (device_allocs, tail_fused, chunks_rerun_in_place) */
#define NFC_AMD_ABI_VERSION 4

typedef enum {
    NFC_OK = 0,
    NFC_ERR_ARG = -1,      /* bad parameter */
    NFC_ERR_DEVICE = -2,   /* HIP runtime error / no device */
    NFC_ERR_NOMEM = -3,
    NFC_ERR_STATE = -4,    /* call sequence error */
    NFC_ERR_INTERNAL = -5  /* a kernel reported an impossible condition */
} nfc_status;

/* What one input sample is (replaces the source side of decoder.py:21-29). */
typedef enum {
    NFC_IN_IQ_F32 = 0,      /* interleaved float32 I,Q; x = fl(fl(I*I)+fl(Q*Q))  (uhd branch, usrp_src.py:31) */
    NFC_IN_ENV_F32 = 1,     /* float32 envelope already computed; x = sample                                    */
    NFC_IN_REAL_F32_SQ = 2, /* float32 real sample, Q = 0; x = fl(s*s)              (wav branch, decoder.py:25-28) */
    NFC_IN_I16_SQ = 3       /* int16 PCM; s = fl(pcm / 32767) (or fl(pcm * i16_scale)); x = fl(s*s)  (wavfile_source + wav branch) */
} nfc_input_kind;

/* nfc_params.flags */
#define NFC_FLAG_FORCE_SEQUENTIAL 1u /* run the exact one-lane sequential kernel for the threshold stage (slow; testing) */
#define NFC_FLAG_NO_EDGES 2u         /* stop after the threshold stage (val codes only; profiling)                      */

/* Constructor arguments of transition_sink (transition_sink.py:12) and background (background.py:17). */
typedef struct {
    double samp_rate;      /* samples per second; durations are reported as d * 1e6 / samp_rate microseconds */
    double lo_val;         /* 0.1  */
    double hi_val;         /* 1.1 for IQ input (decoder.py:23), 1.09 for the WAV branch (decoder.py:29) */
    int32_t av_window;     /* 2000 */
    int32_t max_len;       /* 50   */
    int32_t enable_reader; /* Modified-Miller decoder present (background.py:20) */
    int32_t enable_tag;    /* Manchester decoder present      (background.py:21) */
    int32_t input_kind;    /* nfc_input_kind */
    int32_t device;        /* HIP device ordinal */
    float i16_scale;       /* NFC_IN_I16_SQ only.  0: GNU Radio's wavfile_source normalisation, s = fl((float)pcm / 32767.0f) -- what
                            * decoder.py:25 feeds the path (gr-blocks wavfile_source_impl.cc divides 16-bit samples by 0x7FFF; GNU Radio
                            * is third party and absent from the reference tree, so this boundary is UNPINNED: SURVEY.md 8c).  > 0: s =
                            * fl((float)pcm * i16_scale) for a source normalised differently (1/32768 is a power of two; 1/32767 is not:
                            * fl(s*s) then rounds differently near the thresholds) */
    uint32_t flags;
    int32_t chunk_samples; /* samples per time chunk of the threshold kernel; 0 -> default */
    int32_t reserved;
} nfc_params;

/* One entry of the list transition_sink hands to its callback
 * (transition_sink.py:89-90,97): ((v, d*factor), t) plus the sample index at
 * which the reference appended it. 16 bytes. */
typedef struct {
    uint64_t idx; /* 0-based sample index in the whole stream */
    int32_t d;    /* duration in samples, 1..max_len */
    int8_t v;     /* -1, 0, 1, 2 */
    int8_t t;     /* cur_state - 1: -1 idle, 0 tag->reader, 1 reader->tag */
    int16_t pad;
} nfc_edge;

/* A packet as PacketProcessor.append_bit returns it (packets.py:67-79). */
typedef struct {
    uint64_t idx;     /* sample index of the edge whose symbol closed the packet */
    uint64_t bit_off; /* offset of its first bit in the per-type packet bit array of this batch */
    uint32_t n_bits;
    int32_t type;     /* 0 TAG_TO_READER, 1 READER_TO_TAG (packets.py:19-20) */
} nfc_packet;

typedef struct {
    uint64_t n_samples;    /* samples in the last batch */
    uint64_t n_edges;      /* transitions produced by the last batch */
    uint64_t n_symbols[2]; /* symbols handed to append_bit per packet type (0 tag, 1 reader) */
    uint64_t n_packets[2]; /* closed, non-empty packets per type */
    uint64_t n_packet_bits[2];
} nfc_counts;

typedef struct {
    double ms_total;          /* device time of the last batch, all kernels (hipEvent) */
    double ms_threshold;      /* envelope + threshold kernel launches of the last batch */
    double ms_edges;          /* run-length / edge extraction */
    double ms_decode;         /* Miller / Manchester / framing */
    uint32_t threshold_passes; /* launches of the threshold kernel (2 = speculate + verify) */
    uint32_t chunks_rerun;    /* chunks re-evaluated after the verify pass */
    uint32_t used_sequential; /* 1 if the exact sequential kernel ran */
    uint32_t n_chunks;
    uint64_t bytes_in;        /* input bytes of the last batch */
    double ms_threshold_kernel[6]; /* hipEvent duration of each k_threshold launch of the last batch (first 6) */
    uint32_t n_threshold_timed;
    uint32_t chunk_samples;   /* time-chunk length the threshold kernel used for the last batch (the nominal one: where the chunks
                               * are cut by dispatch row -- nfc_plan_row_cut -- they are up to 4 % longer or shorter, and n_chunks counts them) */
    uint32_t ran_ahead;       /* 1: the last batch's threshold stage ran ahead of the batch before it (nfc_submit_device) */
    uint32_t redone_total;    /* submitted batches of this context that had to be processed again synchronously */
    uint32_t ring_slots_carried; /* window slots whose value at the end of the last batch is still the one the batch started from
                                  * (no sample landing on them was accepted): 0 after a warm-up (nfc_prime + overlap) means the
                                  * window no longer depends on the level it was primed with */
    uint32_t decode_respeculated; /* batches of this context whose decode stage was repeated in the three-launch form because a tile of the
                                   * speculative form (a run-in of the predecessor tile's last edges instead of a scan over all tiles)
                                   * had assumed a decoder state that the check found wrong: a frame longer than the run-in */
    uint32_t device_allocs;   /* device / pinned buffers (re)allocated while the last batch was submitted and processed: 0 in the steady state
                               * of a stream -- the buffers are sized when a stream's first batch of a length is seen, with room for four
                               * times the transition density of a clean capture, so a stream that turns dense does not pay hipMalloc */
    uint32_t tail_fused;      /* 1: the last batch's edge, decode and framing stages ran as ONE launch (test build, NFC_TAIL=1); the product build: 0 */
    uint32_t chunks_rerun_in_place; /* ... of chunks_rerun: re-runs by the workgroup kernel in the form that evaluates a failed round in place
                                     * (up to a machine-full of failing chunks per round), the rest by the one-wave exact kernel */
} nfc_stats;

/* Everything a successor time chunk needs from its predecessor:
 * fixed header; the ring (av_window float32) follows in the caller's buffer. */
typedef struct {
    uint64_t n_seen;       /* samples consumed so far */
    double ss;             /* transition_sink._sum */
    int64_t last_low;      /* index of the last LOW sample, or -1 */
    int32_t filled;        /* transition_sink._filled */
    int32_t stable;        /* work has been rebound to work_stable */
    int32_t cur_state;     /* transition_sink._current_state */
    int32_t last_bit;      /* transition_sink._last_bit */
    int32_t dur;           /* transition_sink._dur */
    int32_t miller_state;  /* (stage, has_started, prev) packed: stage | started<<2 | prev<<3 */
    int32_t manch_state;   /* prev_set | (prev+1)<<1 */
    int32_t pkt_started[2];
    uint32_t n_pending_bits[2]; /* PacketProcessor._cur lengths (bits stay on the device) */
    int32_t av_window;
    int32_t reserved;
} nfc_state_header;

int nfc_abi_version(void);
int nfc_device_count(void);

typedef struct nfc_ctx nfc_ctx; /* opaque; one per stream */

int nfc_create(const nfc_params *params, nfc_ctx **out);
void nfc_destroy(nfc_ctx *ctx);
const char *nfc_last_error(const nfc_ctx *ctx); /* ctx may be NULL: message of the last failed nfc_create */

/* transition_sink.work(): consume n samples (any n >= 0, any chunking gives the
 * same concatenated outputs).  Host buffer: staged to the device first.
 * One call takes at most 2^30 samples (positions inside a batch are 32-bit: NFC_ERR_ARG beyond; push a longer capture in pieces,
 * the stream index itself is 64-bit). */
int nfc_push(nfc_ctx *ctx, const void *host_samples, size_t n);
/* Same, input already in device memory (16-byte aligned). */
int nfc_push_device(nfc_ctx *ctx, const void *dev_samples, size_t n);
/* Wait for the device; outputs of the last push are complete after it. */
int nfc_sync(nfc_ctx *ctx);
/* The decode and framing stages alone, for a caller that already has transitions (what background.append receives,
 * background.py:27-28: entries as nfc_read_edges returns them -- v, d in samples, t the route -1 / 0 / 1; idx only labels
 * the packets): Modified-Miller / Manchester decoding and packet framing on the GPU with the context's decoder and framing
 * state carried on, results through nfc_read_symbols / nfc_read_packets.  The threshold state is not touched. */
int nfc_push_edges(nfc_ctx *ctx, const nfc_edge *host_edges, size_t n);
/* Batches submitted ahead.  nfc_submit_device enqueues a batch and returns; nfc_wait completes the OLDEST submitted batch,
 * after which its outputs are read as after nfc_push_device -- valid until the next nfc_submit_device / nfc_wait / nfc_push*.
 * At most three batches may be in flight, so the steady state of a stream is
 *     submit(0); submit(1);  loop k: submit(k + 2); wait(k); read the outputs of k
 * and the threshold stage of batch k + 1 runs on the GPU beside the edge and decode stages of batch k (which need nothing of
 * it, and leave it most of the machine's issue slots), with batch k + 2's queued right behind it.  The input buffer of a submitted batch must stay untouched until
 * its nfc_wait returns.  Results are identical to nfc_push_device's: what runs ahead is checked in nfc_wait (certification
 * verdict, exactness guard, buffer capacities) and a batch that fails a check is processed again synchronously from the
 * state before it.  A batch that does not qualify (window not full yet, a short batch, state just set from the host, ...)
 * is simply processed inside its nfc_wait.  No other call that touches the stream state is accepted while batches are in
 * flight (NFC_ERR_STATE).  No reference counterpart: the reference is one synchronous work() call after another. */
int nfc_submit_device(nfc_ctx *ctx, const void *dev_samples, size_t n);
int nfc_wait(nfc_ctx *ctx);
int nfc_submitted(nfc_ctx *ctx); /* batches submitted and not yet waited for (0 .. 3) */
/* Enqueue this context's work on the caller's HIP stream (hipStream_t; NULL: back to the context's own), so that what the
 * caller enqueues there next -- a collective on the exported boundary states -- needs no host wait in between. */
int nfc_set_stream(nfc_ctx *ctx, void *stream);

/* Outputs of the LAST push (valid until the next push). */
int nfc_get_counts(nfc_ctx *ctx, nfc_counts *out);
/* The transitions of the batch (transition_sink.py:89-90,97).  On the device an entry is its batch-local sample position
 * (u32) and a 16-bit code; nfc_read_edges fetches those and builds the 16-byte records on the host.  A caller that can use
 * the compact form directly (6 bytes per entry over the link instead of 16) reads it with nfc_read_edges_compact:
 *   pos   sample index in the batch (stream index = n_seen before the push + pos)
 *   code  ((v + 1) * (max_len + 1) + d) | (t + 1) << 14     -- v, d, t as in nfc_edge
 * (not after nfc_push_edges, whose entries keep the caller's own indices). */
int nfc_read_edges(nfc_ctx *ctx, size_t first, nfc_edge *out, size_t cap, size_t *n_out);
int nfc_read_edges_compact(nfc_ctx *ctx, size_t first, uint32_t *pos_out, uint16_t *code_out, size_t cap, size_t *n_out);
/* symbols handed to CombinedPacketProcessor.append_bit(bit, type): 0/1 or an ErrorCode (utilities.py:7-14) */
int nfc_read_symbols(nfc_ctx *ctx, int type, size_t first, uint8_t *out, size_t cap, size_t *n_out);
/* closed packets of one type in stream order, and their bits (one byte per bit) */
int nfc_read_packets(nfc_ctx *ctx, int type, nfc_packet *out, size_t cap, size_t *n_out);
int nfc_read_packet_bits(nfc_ctx *ctx, int type, size_t first, uint8_t *out, size_t cap, size_t *n_out);
/* per-sample classification of the last batch (-1 LOW, 0 accepted, +1 HIGH); debugging tap */
int nfc_read_val(nfc_ctx *ctx, size_t first, int8_t *out, size_t cap, size_t *n_out);

/* Boundary state for multi-GPU time sharding / restart.  ring holds av_window floats; pending_bits holds the
 * open packets' bits (PacketProcessor._cur), type 0 first, hdr->n_pending_bits[0] + [1] bytes.  ring and
 * pending_bits may be NULL in nfc_get_state to query the header (and the sizes) only. */
int nfc_get_state(nfc_ctx *ctx, nfc_state_header *hdr, float *ring, size_t ring_cap, uint8_t *pending_bits,
                  size_t pending_cap);
int nfc_set_state(nfc_ctx *ctx, const nfc_state_header *hdr, const float *ring, size_t ring_len,
                  const uint8_t *pending_bits, size_t pending_len);
/* The same state written to DEVICE memory (16-byte aligned), asynchronously on the context's stream, for a
 * boundary exchange that goes GPU to GPU (RCCL all-gather straight from this buffer):
 *   [u32 len | 12 zero bytes | nfc_state_header | ring | pending bits]   len = bytes behind the 16-byte prefix.
 * When 16 + len exceeds cap only the prefix is written (the reader sees len and can ask again with room).
 * nfc_sync() before another stream or library reads the buffer. */
int nfc_export_state(nfc_ctx *ctx, void *device_dst, size_t cap, size_t *len_out);
/* Back to the state of a freshly created context (a new stream), keeping the device buffers. */
int nfc_reset(nfc_ctx *ctx);
/* Speculative start for a time shard that does not begin the stream (multi-GPU sharding): the
 * window full of `level` (the unloaded-carrier estimate), its exact sum, idle state machines, decoders reset,
 * n_seen = start_index.  Pushing an overlap that ends where the shard starts then converges to the true
 * boundary state.  No host-side ring: the window is filled on the device. */
int nfc_prime(nfc_ctx *ctx, uint64_t start_index, float level);

int nfc_get_stats(nfc_ctx *ctx, nfc_stats *out);
/* How much of nfc_stats' timing is collected.  Default 0: no events (the ms_* fields stay 0).  1: the threshold
 * kernels' own launch durations (ms_threshold_kernel), from start / stop events attached to the launches themselves.
 * 2: also ms_total and the per-stage split, from events recorded as markers between the kernels -- each marker costs
 * the stream a few microseconds. */
int nfc_set_timing(nfc_ctx *ctx, int level);

/* ---- "next" row f1: closed packets -> bytes -> commands, on the host -------------------------
 * and row f3: the CRYPTO1 sessions of MIFARE Classic (cipher.py, lfsr.py, fsm.py:133-154).
 * fsm.process_bits (fsm.py:218-238): frame-end repair (fsm.py:49-66), decryption while a session is up, odd-parity strip and
 * check (fsm.py:28-47), command lookup by protocol stage and leading bytes with CRC_A / BCC checks
 * (command.py:44-67,166-199; utilities.py:26-46), header / extra / CRC split (command.py:245-253), tag type and
 * UID tracking (fsm.py:165-216).  No device work. */
enum {
    NFC_CMD_UNKNOWN = -1,       /* bytes that match no command: CommandStructure("UNKNOWN", [], bytes) */
    NFC_CMD_PARITY_ERROR = -2   /* "PARITY ERROR" (fsm.py:225-227): no bytes, nothing dispatched */
};
enum {
    NFC_FRAME_EXTRA_ERROR = 1,      /* "EXTRA ERROR" (fsm.py:61) */
    NFC_FRAME_MANY_MORE_ERROR = 2,  /* "MANY MORE ERROR" (fsm.py:65) */
    NFC_FRAME_UID_MISMATCH = 4,     /* "MISMATCH BETWEEN READER-TAG UID" (fsm.py:186,195) */
    NFC_FRAME_ENCRYPTED = 8,        /* a CRYPTO1 session was up: the frame was decrypted first (fsm.py:133-154) */
    NFC_FRAME_AR_OK = 16, NFC_FRAME_AR_ERROR = 32,   /* reader answer vs suc64(nt): "AR OK" / "ERROR WITH AR" (fsm.py:203-208) */
    NFC_FRAME_AT_OK = 64, NFC_FRAME_AT_ERROR = 128   /* tag answer vs suc96(nt) (fsm.py:209-214) */
};
typedef struct nfc_frame {
    int32_t cmd;       /* index for nfc_command_info, or NFC_CMD_* */
    int32_t type;      /* 0 tag -> reader, 1 reader -> tag */
    uint32_t byte_off; /* nfc_fsm_process_packets: offset of the frame's bytes in the byte buffer */
    uint16_t n_bytes, n_header, n_extra, n_crc; /* bytes = header | extra | crc */
    uint32_t flags;    /* NFC_FRAME_* */
    uint16_t n_enc;    /* entries written to enc_out for this frame (what fsm._print_enc shows) */
    uint16_t pad;
} nfc_frame;
typedef struct nfc_command_info {
    char name[8];
    int32_t stage, type, crc, n_header, n_extra, xor_check;
    uint8_t header[2];
    uint8_t pad[2];
} nfc_command_info;
typedef struct nfc_fsm nfc_fsm; /* protocol state across packets: previous command, tag type, UID */

int nfc_fsm_create(nfc_fsm **out);
void nfc_fsm_destroy(nfc_fsm *f);
int nfc_fsm_reset(nfc_fsm *f);
/* one packet's bits (as nfc_read_packet_bits returns them); bytes_out needs n_bits / 9 + 1 bytes */
int nfc_fsm_process(nfc_fsm *f, const uint8_t *bits, size_t n_bits, int packet_type, nfc_frame *out, uint8_t *bytes_out,
                    size_t bytes_cap, uint16_t *enc_out /* NULL, or the same capacity: on-air byte | 0x100 if marked '!' */);
/* fsm.process_outgoing (fsm.py:68-112, wired at packets.py:88-90 as the emulator's encoder hook): a frame an emulator is about to
   send, bits with parity as the encoders take them; the machine follows it (tag type from an ATQA, the command in flight) and,
   while a MIFARE Classic session is up, encrypts it into bits_out (n_bits entries).  Returns 0, or 1 when the tag is an Ultralight:
   nothing was written, and the reference runs the frame through process_bits instead (the Python fsm does). */
int nfc_fsm_process_outgoing(nfc_fsm *f, const uint8_t *bits, size_t n_bits, int cmd, uint8_t *bits_out);
/* MIFARE Classic sector keys A / B (fsm.set_keys, fsm.py:157-160; both default to FF FF FF FF FF FF) */
int nfc_fsm_set_keys(nfc_fsm *f, const uint8_t key_a[6], const uint8_t key_b[6]);
/* a batch of packets in stream order: the rows of nfc_read_packets (both types merged by idx) over their bit arrays */
int nfc_fsm_process_packets(nfc_fsm *f, const nfc_packet *packets, size_t n_packets, const uint8_t *bits_type0,
                            const uint8_t *bits_type1, nfc_frame *frames_out, uint8_t *bytes_out, size_t bytes_cap,
                            size_t *bytes_used, uint16_t *enc_out /* NULL, or bytes_cap entries, indexed like bytes_out */);
int nfc_command_count(void);
int nfc_command_get(int cmd, nfc_command_info *out);
/* ISO 14443-3 type A CRC (utilities.py:30-41), low byte first */
int nfc_crc_a(const uint8_t *data, size_t n, uint8_t out[2]);

/* ---- "next" row f4: the transmit side as a device-side signal generator -------------------------
 * encode_bits of miller_encoder / manchester_encoder / binary_src.encoder (miller.py:200-233, manchester.py:64-79,
 * binary_src.py:17-20) on the host; binary_src.work (binary_src.py:64-103: int(dur * samp_rate / 1e6) samples per run,
 * complex64 level + 0j) and multiplier (multiplier.py:18-22: times A exp(j 2 pi f k / samp_rate)) as one kernel. */
typedef struct {
    int32_t level;  /* 0 / 1; 2 = binary_src's "temporary pause" marker (no samples) */
    int32_t pad;
    double dur_us;
} nfc_tx_run;
enum { NFC_TX_SAME = 0, NFC_TX_MANCHESTER = 1, NFC_TX_MILLER = 2 };
int nfc_tx_encode(int encoding, const uint8_t *bits, size_t n_bits, nfc_tx_run *out, size_t cap, size_t *n_out);
/* samples binary_src.work produces for the runs */
int nfc_tx_sample_count(const nfc_tx_run *runs, size_t n_runs, double samp_rate, uint64_t *n_samples);
/* renders the runs into dev_out (complex64, 32-byte aligned, cap_samples entries); carrier != 0 multiplies by the carrier,
 * sample k of this call having carrier index first_index + k.  kernel_ms (may be NULL): the kernel's duration by HIP events.
 * The carrier's phase is kept in 64-bit fixed point and its top 24 bits go into sincospif: a phase error of at most 3.7e-7 rad,
 * 1.5e-7 of the amplitude.  The reference's carrier is GNU Radio's sig_source_c -- third party, not under the reference tree:
 * parity at that boundary is unpinned (tx.hip.h states the arithmetic; tests/test_tx.py checks it against a float64
 * restatement within that tolerance). */
int nfc_tx_render_device(int device, const nfc_tx_run *runs, size_t n_runs, double samp_rate, int carrier, double freq,
                         float amp, uint64_t first_index, void *dev_out, size_t cap_samples, size_t *n_samples,
                         float *kernel_ms);

/* Device memory helpers so that a caller without HIP bindings (ctypes) can keep its input
 * resident in HBM and use nfc_push_device. */
int nfc_device_alloc(int device, size_t bytes, void **out);
int nfc_device_free(int device, void *p);
int nfc_device_upload(int device, void *dst, const void *src_host, size_t bytes);
int nfc_device_download(int device, void *dst_host, const void *src, size_t bytes);
/* The same for a caller that enqueues other libraries' work (an RCCL collective) next to a context's: a stream to share with
 * nfc_set_stream, a wait for it, a device-to-host copy on it, pinned host memory to copy into, a fill. */
int nfc_stream_create(int device, void **stream_out);
int nfc_stream_destroy(int device, void *stream);
int nfc_stream_sync(int device, void *stream);
int nfc_device_download_async(int device, void *dst_host, const void *src, size_t bytes, void *stream);
int nfc_device_fill(int device, void *dst, int byte_value, size_t bytes);
int nfc_host_alloc_pinned(size_t bytes, void **out);
int nfc_host_free_pinned(void *p);

/* Host-only helpers (no GPU needed): the duration LUTs the decode kernels use,
 * driven sequentially.  Used by the CPU test-suite to pin the tables to the
 * reference decoders' golden vectors.  type: 0 Manchester, 1 Miller. */
int nfc_host_decode_lut(const nfc_params *params, int type, const int8_t *cur, const int32_t *d, size_t n,
                        uint8_t *sym_out, size_t cap, size_t *n_out);
/* (type 2: Modified Miller through the decoder's QUOTIENT machine -- states that no sequence of transitions can tell apart are
 * one class, csrc/decoder_tables.h: miller_quotient -- which is what the speculative decode kernel walks: its state maps are
 * then 8 bytes.  Same symbols as type 1 by construction; the CPU suite checks it on the reference's decoder vectors.)
 * nfc_host_miller_classes: the class (0 .. n_classes - 1; 0xFF: a state no transition sequence reaches from the initial one)
 * and the canonical state of every one of the 16 Miller states (stage | has_started << 2 | prev << 3, miller.py:14-29).
 * The carried Miller state -- nfc_state_header.miller_state, the exported boundary state -- is always the canonical one:
 * `prev` reads 0 wherever the decoder cannot read it before writing it (miller.py:81 is its only read). */
int nfc_host_miller_classes(const nfc_params *params, uint8_t class_of[16], uint8_t canonical[16], int *n_classes);

/* The decoders themselves on the host, one transition at a time with any duration in microseconds (the walk the LUTs are
 * built from: csrc/decoder_tables.h), decoder state carried across calls in *state (0 before the first call; type 0
 * Manchester: manchester.py:30-61, type 1 Modified Miller: miller.py:153-197).  Behind usrp_nfc_amd.miller.miller_decoder /
 * manchester.manchester_decoder, the reference-named classes that keep `process_transition(list)`. */
int nfc_host_decode_steps(int type, const int8_t *cur, const double *dur_us, size_t n, int32_t *state, uint8_t *sym_out,
                          size_t cap, size_t *n_out);
/* the int16 -> float conversion of the NFC_IN_I16_SQ kernels, on the host (i16_scale as in nfc_params) */
float nfc_host_i16_to_float(int16_t pcm, float i16_scale);
/* How a batch of n samples is cut into the threshold stage's time chunks when the cut goes by dispatch row (csrc/chunk_cut.h; the
 * reference has no counterpart: its loop is one chunk, transition_sink.py:37-107 -- this is the parallel restatement's own geometry,
 * exposed so that the CPU suite can check it covers every sample exactly once).  C: the equal cut's chunk length, rs: samples per
 * round of the workgroup kernel (C a multiple of it), cus: compute units, rows: workgroups per CU (2 .. 4), factors: rows - 1 length
 * factors, max_len: longest chunk allowed (0: any).  out[0..3]: chunk length per row, out[4..7]: first sample per row, out[8]:
 * chunks per row, out[9]: chunks that begin inside the batch.  Returns 1 when the cut goes by row, 0 for the equal cut (the same
 * table with equal entries), negative on bad arguments.  Chunk c of row r = min(c / out[8], 3) covers
 * [out[4 + r] + (c - r * out[8]) * out[r], ... + out[r]) cut at n. */
int nfc_plan_row_cut(uint32_t n, uint32_t C, uint32_t rs, uint32_t cus, uint32_t rows, const double *factors, uint32_t max_len,
                     uint32_t out[10]);

#ifdef __cplusplus
}
#endif
#endif /* NFC_AMD_H */
