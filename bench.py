#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s -> decoded bits on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--samples S] [--workload miller|manchester|all|classic1k]

One "step" = one pass of the whole hot path (envelope -> threshold -> edges -> Miller/Manchester -> framing) over the
rank's batch of synthetic IQ, input already resident in HBM.  At N=1 the workload is BASELINE.json configs[1]: Miller-only
decode of 1e8 synthetic IQ samples @ 2 Msps.

N > 1: one process per GPU.  Launched by ``python -m torch.distributed.run ... bench.py --gpus N`` the ranks read RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; launched as plain ``python bench.py --gpus N`` this process spawns
the N ranks itself (before anything touches HIP) and relays rank 0's line.  Every rank holds one contiguous time chunk of a
single N x S-sample capture plus an overlap prefix; it decodes its chunk from a speculated boundary state, the boundary
states travel in ONE ncclAllGather per round (RCCL over xGMI through ctypes: usrp_nfc_amd/comm.py -- no PyTorch anywhere in
the path), and a rank whose speculation was wrong re-decodes (weak scaling; usrp_nfc_amd/sharding.py).
NFC_BENCH_BACKEND=host carries the states over TCP instead (ranks that share one GPU: plumbing tests).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
STREAM_CEILING_GBS = 7030.0   # what a read-only kernel with k_threshold_wg's access pattern (a 98 304-sample chunk per 256-thread workgroup,
                              # four per CU; a wave's 2 KB step of every 1 024-sample round asked for one round ahead, a barrier per round)
                              # reaches on this machine with non-temporal loads, as the kernel's are (5 970 with plain ones):
                              # tools/ubench/stream_chunks.hip, profiles/r04_stream_ceiling.txt


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)   # 0.4 ms each: the device reaches its steady clocks within the first few dozen
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--samples', type=float, default=0, help='samples per GPU (default: 1e8; 1e9 for classic1k)')
    ap.add_argument('--workload', default=None, choices=['miller', 'manchester', 'all', 'classic1k'],
                    help="BASELINE.json configs[1] / [2] / both decoders at 2 Msps, or configs[3] / [4]: the MIFARE Classic 1K "
                         "transaction of outputs/1k_with_enc.out at 10 Msps, 1e9 samples per GPU (a 1e8-sample capture tiled).  Default: miller; "
                         "with --gpus N > 1 and no --workload the line also carries configs[4] (classic1k, 1e9 samples per GPU) under other_configs")
    ap.add_argument('--input-kind', default='iq', choices=['iq', 'env', 'i16'],
                    help="what the caller hands over: fc32 IQ (8 B/sample, BASELINE's metric), the float32 envelope that the reference's own "
                         "transition_sink.work receives (4 B/sample, transition_sink.py:13-18), or 16-bit PCM as the WAV branch reads it (2 B/sample)")
    ap.add_argument('--chunk', type=int, default=0, help='time-chunk samples of the threshold kernel (0: library default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the end-to-end figure and the other configurations')
    ap.add_argument('--in-flight', type=int, default=3, choices=[1, 2, 3], help='one GPU: batches submitted and not yet waited for')
    ap.add_argument('--primary', default=None, choices=['sync', 'ahead'], help='what the headline steps are (default: sync on one GPU, ahead with several -- a rank\'s time shard is a stream, '
                    'its boundary protocol is paid once per shard; the per-step protocol is then the figure beside it) -- sync: every step a fresh stream pushed synchronously '
                    '(the threshold kernel with the machine to itself); ahead: consecutive batches of one stream, submitted ahead (nfc_submit_device / nfc_wait). '
                    'The other one is measured beside it unless --no-extras')
    ap.add_argument('--sync-steps', action='store_true', help='(same as --primary sync --no-extras for the stepping: kept for the profiling scripts)')
    a = ap.parse_args()
    a.workload_given = a.workload is not None
    a.workload = a.workload or 'miller'
    return a


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks (nothing HIP-related is imported before this point)
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(a):
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write('bench: ranks failed: %s\n' % bad)
        return 1
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------
def decoder_flags(workload):
    return dict(reader=workload in ('miller', 'all', 'classic1k'), tag=workload in ('manchester', 'all', 'classic1k'))


def stream_params(workload):
    """Sample rate and the constructor arguments that go with it.  At 10 Msps the reference's fixed-sample defaults
    (av_window 2000, max_len 50 samples) chop every bit period (hard part 5): they are scaled
    with the rate, as its keyword arguments allow (transition_sink.py:12)."""
    if workload == 'classic1k':
        return dict(samp_rate=10e6, hi_val=1.1, av_window=10000, max_len=250)
    return dict(samp_rate=2e6, hi_val=1.1)


WORKLOAD_NAMES = {'miller': 'configs[1]: Miller-only decode, synthetic IQ @2 Msps',
                  'manchester': 'configs[2]: Manchester-only decode, synthetic IQ @2 Msps',
                  'all': 'both decoders (-t all), Ultralight transaction, synthetic IQ @2 Msps',
                  'classic1k': 'configs[3]/[4]: -t all @10 Msps, MIFARE Classic 1K transaction tiled (av_window 10000, max_len 250)'}
TILE = 100_000_000   # classic1k: samples of the synthetic capture that is tiled to the configuration's length


def default_samples(workload):
    return 1_000_000_000 if workload == 'classic1k' else 100_000_000


# input kinds: (bytes per sample, elements per sample, what it is)
INPUT_KINDS = {'iq': (8, 2, 'fc32 IQ, 8 B/sample (envelope computed in the kernel)'),
               'env': (4, 1, 'float32 envelope, 4 B/sample: what transition_sink.work receives (transition_sink.py:13-18, decoder.py:27-33)'),
               'i16': (2, 1, '16-bit PCM, 2 B/sample: scaled by 1/32767 and squared in the kernel (the WAV branch, decoder.py:25-27)')}
PCM_SCALE = 40000.0   # i16: PCM value of a unit amplitude (the synthetic carrier, amplitude 0.5, sits at 20 000)


def api_kind(kind):
    from usrp_nfc_amd import api
    return {'iq': api.NFC_IN_IQ_F32, 'env': api.NFC_IN_ENV_F32, 'i16': api.NFC_IN_I16_SQ}[kind]


def convert_input(iq, kind):
    """The capture as the caller of that input kind would hold it (from the synthetic IQ)."""
    import numpy as np
    from usrp_nfc_amd import synth
    if kind == 'iq' or not len(iq):
        return iq if kind == 'iq' else np.zeros(0, np.float32 if kind == 'env' else np.int16)
    env = synth.envelope_f32(iq)
    if kind == 'env':
        return env
    return np.clip(np.rint(np.sqrt(env) * np.float32(PCM_SCALE)), -32768, 32767).astype(np.int16)


def oracle_push(o, x, kind):
    import numpy as np
    if kind == 'iq':
        o.push_iq(x)
    elif kind == 'env':
        o.push_env(x)
    else:   # (fl(pcm / 32767) as GNU Radio's wavfile_source scales it -- tests/test_host_abi.py pins the kernel's form of it --, then squared)
        o.push_real_sq((x.astype(np.float32) / np.float32(32767.0)).astype(np.float32))


OVERLAP_WINDOWS = (16, 32, 64)   # warm-up lengths a rank > 0 tries, in averaging windows (sharding.overlap_schedule)


def overlap_steps(workload, n_per_rank=None):
    """The warm-up lengths (samples) a rank > 0 tries, shortest first; the slice of its predecessor it holds is the longest."""
    from usrp_nfc_amd import sharding
    sp = stream_params(workload)
    steps = [sharding.shard_overlap(sp['samp_rate'], sp.get('av_window', 2000), windows=w) for w in OVERLAP_WINDOWS]
    if n_per_rank is not None:
        steps = [s for s in steps if s <= n_per_rank] or [n_per_rank // 256 * 256]
    return steps


def capture_overlap(workload, n_per_rank=None):
    return overlap_steps(workload, n_per_rank)[-1]


def make_capture_slice(workload, n_per_rank, rank, world):
    """Rank's time chunk of ONE capture of world*n_per_rank samples (plus the overlap samples before it).

    The modulation profile is a pure function of the global sample index (a frame sequence tiled after an idle lead-in
    that covers the averaging window); the noise comes from a per-rank PCG64 stream, the overlap region from the
    predecessor's stream, so neighbouring ranks agree on the samples they share.  classic1k: every rank's chunk is a
    TILE-sample capture repeated (BASELINE.json: "recordings/classic1k.wav tiled"); the returned array is ONE tile."""
    import numpy as np
    from usrp_nfc_amd import synth
    if workload == 'classic1k':
        frames, _ = synth.frames_from_trace(os.path.join(ROOT, 'tests', 'golden', '1k_with_enc.out'))
        period = synth.modulation_profile(frames, rate_msps=10.0, lead_in=0, tail=0)
        lead = 15000   # covers the 10000-sample window
        n_gen = min(n_per_rank, TILE)
    else:
        picks = {'miller': (0, 2, 4, 10), 'manchester': (1, 3, 5, 11), 'all': tuple(range(19))}[workload]
        frames = [(d, synth.frame_bits(data, sb)) for d, _, data, sb in (synth.ULTRALIGHT_TXN[i] for i in picks)]
        period = synth.modulation_profile(frames, rate_msps=2.0, lead_in=0, tail=0)
        lead = 3000
        n_gen = n_per_rank
    overlap = capture_overlap(workload, min(n_per_rank, n_gen))

    def profile(g_lo, g_hi):
        g = np.arange(g_lo, g_hi, dtype=np.int64)
        m = np.ones(len(g), np.float32)
        body = g >= lead
        m[body] = period[(g[body] - lead) % len(period)]
        return m

    def noisy(g_lo, g_hi, owner):
        # owner's stream covers [owner*n_gen, (owner+1)*n_gen); take the sub-range
        rng = np.random.Generator(np.random.PCG64([synth.SEED, owner]))
        base = owner * n_gen
        iq = rng.standard_normal(2 * n_gen, dtype=np.float32)[2 * (g_lo - base):2 * (g_hi - base)]
        iq *= np.float32(0.002)
        m = profile(g_lo, g_hi)
        iq[0::2] += (np.float32(0.5 * np.cos(0.3)) * m).astype(np.float32)
        iq[1::2] += (np.float32(0.5 * np.sin(0.3)) * m).astype(np.float32)
        return iq

    if workload == 'classic1k' and n_per_rank > TILE:
        # a rank's chunk = its tile repeated: the capture every rank decodes is rank 0's tile, so that the tile seams of
        # all ranks look alike and the overlap a rank holds is the END of that same tile
        own = noisy(0, n_gen, 0)
        ov = own[2 * (n_gen - overlap):].copy() if rank else np.zeros(0, np.float32)
        return ov, own
    lo = rank * n_per_rank
    own = noisy(lo, lo + n_per_rank, rank)
    if rank == 0:
        return np.zeros(0, np.float32), own
    ov = noisy(lo - overlap, lo, rank - 1)
    return ov, own


class Resident(object):
    """A rank's input in HBM: the host capture uploaded once -- a tile repeated `reps` times for classic1k."""

    def __init__(self, api, own, n, dev, kind='iq'):
        import numpy as np
        self.n = n
        bps, per, _ = INPUT_KINDS[kind]
        tile = len(own) // per
        self.reps = (n + tile - 1) // tile
        if self.reps == 1:
            self.buf = api.DeviceBuffer(own, dev)
        else:
            self.buf = api.DeviceBuffer(np.zeros(0, np.float32), dev, nbytes=bps * tile * self.reps)
            L = self.buf.L
            for k in range(self.reps):
                assert L.nfc_device_upload(dev, self.buf.ptr.value + bps * tile * k, own.ctypes.data, own.nbytes) == 0


# ---------------------------------------------------------------------------------------------------------------------
# one configuration on this rank
# ---------------------------------------------------------------------------------------------------------------------
def run_config(a, workload, n, steps, warmup, rank, world, local, comm, backend, want_parity, chunk=0, kind='iq'):
    import numpy as np
    from usrp_nfc_amd import api, sharding, synth
    ndev = max(1, api.device_count())
    dev = (local % ndev) if world > 1 else 0
    bps = INPUT_KINDS[kind][0]
    ov, own_iq = make_capture_slice(workload, n, rank, world)
    own = convert_input(own_iq, kind)      # (what is resident in HBM and what the oracle is given: the caller's form of the capture)
    if kind != 'iq':
        del own_iq
    flags = decoder_flags(workload)
    res = Resident(api, own, n, dev, kind)
    d_ov = api.DeviceBuffer(convert_input(ov, kind), dev) if len(ov) else None
    ctx = api.NfcContext(input_kind=api_kind(kind), device=dev, chunk_samples=chunk, **stream_params(workload), **flags)
    level = sharding.carrier_level(synth.envelope_f32(ov[:2 * 4096])) if len(ov) else 0.0
    g_lo = rank * n
    n_ov = len(ov) // 2
    redo = [0]
    force_exchange = bool(os.environ.get('NFC_BENCH_FORCE_EXCHANGE'))   # 1-rank smoke of the collective path

    ov_steps = [s for s in overlap_steps(workload, n) if s <= n_ov] if n_ov else None
    if ov_steps:   # the carrier level where each try starts
        level = [sharding.carrier_level(synth.envelope_f32(ov[2 * (n_ov - s):2 * (n_ov - s) + 2 * 4096])) for s in ov_steps]

    def push_overlap(nov):   # the LAST nov samples before the shard (lengths are multiples of 256 samples: 16-byte aligned)
        ctx.push_device(d_ov.ptr.value + bps * (n_ov - nov), nov)

    def one_step():
        redo[0] += sharding.decode_shard(ctx, comm, push_overlap, lambda: ctx.push_device(res.buf, n),
                                         g_lo - n_ov, level, force_exchange=force_exchange, overlap_steps=ov_steps, shard_start=g_lo)

    def barrier():
        if hasattr(comm, 'barrier'):
            comm.barrier()
        ctx.sync()

    # Two ways to step, both measured (the other one lands beside the headline unless --no-extras):
    #  sync   every step a fresh stream pushed synchronously (nfc_push_device) -- with several ranks the whole sharding
    #         protocol per step (reset / prime, overlap, own chunk, boundary exchange).  Nothing runs beside the threshold
    #         kernel: its launches are the kernel with the machine to itself.  The headline unless --primary ahead.
    #  ahead  the steps are consecutive batches of ONE stream per rank -- the rank's slice of the capture again and again,
    #         the stream state carried on -- and batch k + 1 is submitted before batch k is waited for (nfc_submit_device /
    #         nfc_wait): its threshold stage then runs beside the edge and decode stages of batch k.  With several ranks
    #         the rank's time shard IS that stream: `steps` batches long, primed and warmed up on the overlap before it,
    #         its speculated start state checked against the predecessor's true end state in ONE exchange after the last
    #         batch (a mismatch re-decodes the shard).
    sharded = world > 1 or force_exchange

    def stream_steps(count, timed_every, acc):
        """count consecutive batches of the stream, a.in_flight in flight; timed_every: every that-many-th threshold launch
        carries its own start / stop HIP events (nfc_set_timing; a timed launch costs the step a few us, so not all of them are)"""
        if not count:
            return
        want = lambda j: bool(timed_every) and j % timed_every == 3   # (not the region's very first launches)
        cur = [None]

        def submit(j):
            t = 1 if want(j) else 0
            if t != cur[0]:
                ctx.set_timing(t)
                cur[0] = t
            ctx.submit_device(res.buf, n)
        nxt = 0
        for k in range(count):
            while nxt < count and nxt < k + a.in_flight:   # (three in flight: the threshold stage of batch k + 2 queued behind k + 1's)
                submit(nxt)
                nxt += 1
            ctx.wait()
            if timed_every:
                st = ctx.stats()
                acc['n_ahead'] += int(st.ran_ahead)
                if st.n_threshold_timed:
                    acc['kernel_ms'].extend(st.ms_threshold_kernel[i] for i in range(st.n_threshold_timed))
                    acc['n_pass'].append(st.threshold_passes)
        ctx.set_timing(0)

    def region(mode):
        """warm-up + exactly `steps` timed steps in the given mode, bracketed by barrier + device sync; max over ranks"""
        acc = {'kernel_ms': [], 'n_pass': [], 'n_ahead': 0, 'mode': mode}
        redo[0] = 0
        if mode == 'ahead' and sharded:
            def shard(count, timed_every):
                redo[0] += sharding.decode_shard(ctx, comm, push_overlap, lambda: stream_steps(count, timed_every, acc),
                                                 g_lo - n_ov, level, force_exchange=force_exchange, overlap_steps=ov_steps, shard_start=g_lo)
            if warmup:
                shard(warmup, 0)
            redo[0] = 0
            barrier()
            t0 = time.perf_counter()
            shard(steps, 8)
        elif mode == 'ahead':
            ctx.reset()
            stream_steps(warmup, 0, acc)
            barrier()
            t0 = time.perf_counter()
            stream_steps(steps, 8, acc)
        else:
            for _ in range(warmup):
                one_step()
            redo[0] = 0
            barrier()
            t0 = time.perf_counter()
            for k in range(steps):
                # every 8th k_threshold launch of a long region carries its own start / stop HIP events, every 2nd of a short one
                # (<= 32 steps: 10 of the driver's 20).  Not every one: a timed launch costs its step 14 us -- measured, round 6, 20 steps
                # on one box: 0.2545 / 0.2464 / 0.2432 ms per step with every / every 2nd / every 4th launch timed, the launches themselves
                # 0.139 ms whichever (NFC_BENCH_TIMED_STRIDE=1 times them all)
                stride = int(os.environ.get('NFC_BENCH_TIMED_STRIDE', '0')) or (8 if steps > 32 else 2)
                timed = (k % 8 == 3) if stride == 8 else (k % stride == stride - 1)   # (stride 8: not the region's very first launch)
                if stride > 1 and (timed or (k and ((k - 1) % 8 == 3 if stride == 8 else (k - 1) % stride == stride - 1))):
                    ctx.set_timing(1 if timed else 0)
                elif stride == 1 and k == 0:
                    ctx.set_timing(1)
                one_step()
                if timed:
                    st = ctx.stats()
                    acc['kernel_ms'] += [st.ms_threshold_kernel[i] for i in range(st.n_threshold_timed)]
                    acc['n_pass'].append(st.threshold_passes)
            ctx.set_timing(0)
        barrier()
        dt = time.perf_counter() - t0
        if hasattr(comm, 'max_over_ranks'):
            dt = comm.max_over_ranks(dt)
        acc['dt'] = dt
        acc['redo'] = redo[0]
        return acc

    def summary(acc):
        ka = float(np.mean(acc['kernel_ms'])) if acc['kernel_ms'] else float('nan')
        ach = float(bps) * n / (ka * 1e-3) / 1e9
        # stepping: 'sync' = a fresh stream per step, pushed synchronously (nfc_push_device): nothing runs beside the threshold kernel;
        # 'ahead' = consecutive batches of one stream per rank, in_flight of them submitted before the oldest is waited for (with several
        # ranks the rank's time shard is that stream: one boundary exchange after its last batch)
        what = acc['mode']
        return {'stepping': what, 'in_flight': a.in_flight if what == 'ahead' else 1, 'ran_ahead': acc['n_ahead'] if what == 'ahead' else 0,
                'ms_per_step': acc['dt'] / steps * 1e3, 'value': world * n * steps / acc['dt'] / 1e6, 'unit': 'Msamples/s',
                'steps': steps, 'boundary_redos': acc['redo'],
                'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                             'avg_launch_ms': ka, 'launches_timed': len(acc['kernel_ms']),
                             'launches_per_step': float(np.mean(acc['n_pass'])) if acc['n_pass'] else None}}

    primary = a.primary or ('ahead' if world > 1 else 'sync')
    ahead = primary == 'ahead' and not a.sync_steps
    # (the region that lands beside the headline goes first: the headline's steps then run on a machine that is already at its clocks)
    other = None if (a.no_extras or a.sync_steps) else region('sync' if ahead else 'ahead')
    prim = region('ahead' if ahead else 'sync')
    dt, kernel_ms, n_pass = prim['dt'], prim['kernel_ms'], prim['n_pass']
    redo[0] = prim['redo']
    ctx.set_timing(2)   # one more, untimed, step for the per-stage split reported beside the headline (synchronous: stream markers)
    if ahead and not sharded:
        ctx.push_device(res.buf, n)
    else:
        one_step()   # (with several ranks: the whole protocol once more on a fresh stream -- every rank's context then holds the decode of ITS shard)
    st = ctx.stats()
    cnt = ctx.counts()
    n_edges = int(cnt.n_edges)
    gathered = None
    if want_parity and world > 1:
        # that step was the whole sharding protocol once more: every rank's context holds the decode of ITS shard -- digests of
        # it travel to rank 0 (JSON over the communicator), which checks every one against the oracle (sharded_parity)
        gathered = comm.gather_objects(result_digest(ctx.edges(), ctx.symbols(0), ctx.symbols(1), ctx.packets()))
    ov_used = None
    if world > 1:   # the warm-up length every rank ended on (rank 0 has none)
        ov_used = comm.gather_objects(int(getattr(ctx, 'overlap_used', 0) or 0))
    out = None
    if rank == 0:
        k_avg = float(np.mean(kernel_ms)) if kernel_ms else float('nan')
        thr_bytes = float(bps) * n                       # 8 B per sample (fc32 IQ; 4 / 2 for the other input kinds) read by the envelope + threshold kernel ...
        edge_bytes = 16.0 * n_edges                      # ... 16 B per emitted edge (the nfc_edge record) in SURVEY's count; the edge
        stored_bytes = 6.0 * n_edges                     # stage now keeps 6 B per entry (u32 position + u16 code), records built on read
        achieved = thr_bytes / (k_avg * 1e-3) / 1e9
        traffic, tsrc = hbm_traffic(workload, n, kind)
        out = {
            'ms_per_step': dt / steps * 1e3,
            'value': world * n * steps / dt / 1e6,
            'config': {'workload': WORKLOAD_NAMES[workload], 'input_kind': INPUT_KINDS[kind][2], 'samples_per_gpu': n, 'time_chunk_samples': int(st.chunk_samples),
                       'time_chunks': int(st.n_chunks), 'parallelism': 'time-chunk x%d' % world, 'edges_per_gpu': n_edges,
                       'symbols_reader': int(cnt.n_symbols[1]), 'symbols_tag': int(cnt.n_symbols[0]),
                       'packets': int(cnt.n_packets[0] + cnt.n_packets[1]), 'boundary_redos': redo[0],
                       'shard_overlap_samples': overlap_steps(workload, n) if world > 1 else [], 'overlap_used': ov_used,
                       'exchange': backend if world > 1 else 'none',
                       'rccl_ranks_seen': getattr(comm, 'ranks_seen', None),
                       'stepping': prim['mode'], 'in_flight': a.in_flight if prim['mode'] == 'ahead' else 1,
                       # several ranks: what the boundary protocol adds to a step when it is run PER STEP (prime, warm-up on the overlap,
                       # state export, all-gather, verify) -- the per-step region's step less the device time of the own chunk alone
                       'protocol_us_per_step': ((((other if ahead else prim)['dt'] / steps * 1e3) - st.ms_total) * 1e3
                                                if sharded and (other or not ahead) and st.ms_total > 0 else None)},
            # (kernel: the fused envelope + gated-mean threshold kernel, a time chunk per workgroup; k_threshold_lean / k_threshold where it does not apply)
            'roofline': {'bound': 'hbm', 'kernel': 'k_threshold_wg',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': tsrc, 'avg_launch_ms': k_avg, 'launches_timed': len(kernel_ms),
                         'launches_per_step': float(np.mean(n_pass)) if n_pass else None,
                         'algorithmic_bytes_per_launch': thr_bytes,
                         'streaming_read_ceiling': STREAM_CEILING_GBS, 'frac_of_streaming_ceiling': achieved / STREAM_CEILING_GBS,
                         # (bytes_per_sample only: the 16 B/edge of SURVEY 8(d) are written by the edge stage, listed under edge_stage and
                         # counted in whole_path; with --primary ahead the timed launches run BESIDE the previous batch's later stages)
                         'bytes_per_sample': bps,
                         # what README / DESIGN quote beside frac, inside the object the driver's record keeps:
                         'whole_path_frac': (thr_bytes + edge_bytes) / (dt / steps) / 1e9 / HBM_PEAK_GBS,
                         # ... and with the 6 B per entry the edge stage actually stores instead of SURVEY 8(d)'s 16 B per edge
                         'whole_path_frac_stored': (thr_bytes + stored_bytes) / (dt / steps) / 1e9 / HBM_PEAK_GBS,
                         'tail_us_per_step': (dt / steps * 1e3 - k_avg) * 1e3,   # everything of a step but the threshold kernel: later stages, k_fill, launch gaps, the host's turn
                         'stages_us_extra_step': {'threshold': st.ms_threshold * 1e3, 'edges': st.ms_edges * 1e3, 'decode': st.ms_decode * 1e3},
                         ('one_batch_at_a_time_ms_per_step' if ahead else 'submitted_ahead_ms_per_step'): (other['dt'] / steps * 1e3 if other else None),
                         ('one_batch_at_a_time_avg_launch_ms' if ahead else 'submitted_ahead_avg_launch_ms'):
                             (float(np.mean(other['kernel_ms'])) if other and other['kernel_ms'] else None),
                         'submitted_ahead_ran_ahead': ((prim if ahead else other)['n_ahead'] if (ahead or other) else None)},
            'edge_stage': {'algorithmic_bytes': edge_bytes, 'bytes_stored': stored_bytes, 'stage_ms': st.ms_edges,
                           # (all kernels of the edge stage of the extra, marker-timed step; algorithmic_bytes is SURVEY 8(d)'s 16 B per edge,
                           # the stage stores 6 B per entry and nfc_read_edges builds the 16-byte records)
                           'achieved_GBs': (edge_bytes / (st.ms_edges * 1e-3) / 1e9) if st.ms_edges > 0 else None},
            'whole_path': {'algorithmic_bytes': thr_bytes + edge_bytes,
                           'achieved_GBs': (thr_bytes + edge_bytes) / (dt / steps) / 1e9, 'frac': (thr_bytes + edge_bytes) / (dt / steps) / 1e9 / HBM_PEAK_GBS},
            ('one_batch_at_a_time' if ahead else 'submitted_ahead'): (summary(other) if other else None),
            'stage_ms_extra_step': {'total_device': st.ms_total, 'threshold': st.ms_threshold, 'edges': st.ms_edges,
                                    'decode': st.ms_decode, 'used_sequential': int(st.used_sequential)},
        }
        if want_parity:
            out['parity'] = parity_check(workload, own, flags, n, True, kind)
            if gathered is not None:
                out['parity']['sharded'] = sharded_parity(workload, n, world, flags, gathered)
                if out['parity']['sharded'].get('all_equal') is False:
                    raise SystemExit('bench: a rank\'s shard differs from the oracle\'s cut of the whole capture: %s' % json.dumps(out['parity']['sharded']))
            # the context the timed loop ran in must have produced the same decode: a fresh stream per step / rank 0's shard
            # starts the stream; or, with consecutive batches of one stream, its steady state (when the rounds repeat)
            want_counts = out['parity'] if (ahead and not sharded) else out['parity']['first_round']
            if ahead and not sharded and not out['parity'].get('stationary'):
                out['parity']['timed_loop_counts_equal'] = None
            else:
                same = bool(n_edges == want_counts['n_edges'] and int(cnt.n_packets[0] + cnt.n_packets[1]) == want_counts['n_packets'])
                out['parity']['timed_loop_counts_equal'] = same
                if not same:
                    raise SystemExit('bench: the timed loop decoded %d edges / %d packets, the oracle %d / %d' % (
                        n_edges, int(cnt.n_packets[0] + cnt.n_packets[1]), want_counts['n_edges'], want_counts['n_packets']))
    ctx.set_stream(None)   # back on its own stream before the communicator's goes away
    ctx.close()
    return out, own, flags


def hbm_traffic(workload, n, kind='iq'):
    """HBM bytes per threshold launch from the rocprofv3 PMC passes recorded under profiles/ (separate --pmc FETCH_SIZE /
    WRITE_SIZE runs of this same command; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md): a record of an
    earlier run, not a measurement of this one -- `traffic_source` says so.  profiles/hbm_traffic.json is a list of records
    keyed by workload and sample count (tools/profiles.sh writes it); None unless one matches."""
    try:
        recs = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
        if isinstance(recs, dict):
            recs = [recs]
        for rec in recs:
            if rec.get('workload') == workload and int(rec.get('samples', 0)) == n and rec.get('input_kind', 'iq') == kind:
                return rec['bytes_per_launch'], 'profiles/hbm_traffic.json (%s)' % rec.get('kernel', 'k_threshold')
    except Exception:
        pass
    return None, None


def stress_config(a, name, n, steps=6):
    """The unhappy path, one batch at a time on fresh streams (usrp_nfc_amd/synth.py: stress_workload): what a step costs when
    chunks of the threshold stage give up or cannot be certified and are evaluated again -- threshold passes, chunks re-run,
    parity against the C oracle on the whole capture."""
    import numpy as np
    from oracle import c_oracle as co
    from usrp_nfc_amd import api, synth
    kw = {'stress_dropouts': dict(depth=0.08, sigma=0.002, step=1.0),
          'stress_dropouts_steps': dict(depth=0.08, sigma=0.002),
          'stress_hover': dict()}[name]
    iq = synth.stress_workload(n, **kw)
    buf = api.DeviceBuffer(iq)
    out = {'workload': name, 'samples': n, 'generator': 'synth.stress_workload(%s)' % ', '.join('%s=%s' % kv for kv in sorted(kw.items())),
           'what': {'stress_dropouts': 'the -t all workload with a 400-sample loss of signal every 1e6 samples',
                    'stress_dropouts_steps': 'the same with a +-15 % level step behind every loss of signal',
                    'stress_hover': 'tag load modulation at mag^2 x 1.10 = hi_val exactly, five times the noise, drop-outs and level steps: '
                                    'every loaded half bit hovers at the HIGH threshold, no chunk of the speculative pass can be certified'}[name]}
    with api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **stream_params('all'), **decoder_flags('all')) as ctx:
        # (one batch of the CLEAN -t all capture first: the context's buffers exist and it cuts its batches for a clean stream --
        # first_step_ms below is then what finding out about the regime costs, not what hipMalloc costs)
        clean = api.DeviceBuffer(synth.workload('all', n))
        ctx.push_device(clean, n)
        ctx.sync()
        clean.free()
        ts, allocs = [], []
        for k in range(steps):
            ctx.reset()
            ctx.sync()
            t0 = time.perf_counter()
            ctx.push_device(buf, n)
            ctx.sync()
            ts.append(time.perf_counter() - t0)
            allocs.append(int(ctx.stats().device_allocs))   # (buffers (re)allocated inside the batch: 0 is what a stream may expect)
        st = ctx.stats()
        # (the first step is the stream's first batch in this regime: pass 0 runs on the clean stream's chunking, its verdict says the
        # stream needs re-runs, and the batch is cut four times finer there and then -- host_threshold.h: recut; the context keeps the
        # fine cut for as long as batches need re-runs, fine_left -- which is what the median shows)
        out.update({'ms_per_step': float(np.median(ts[1:])) * 1e3, 'steps': steps - 1, 'first_step_ms': ts[0] * 1e3, 'steps_ms': [round(t * 1e3, 4) for t in ts],
                    'device_allocs': allocs, 'worst_over_median': float(max(ts[1:]) / np.median(ts[1:])),
                    'threshold_passes': int(st.threshold_passes), 'chunks_rerun': int(st.chunks_rerun), 'chunks_rerun_in_place': int(st.chunks_rerun_in_place), 'n_chunks': int(st.n_chunks),
                    'chunk_samples': int(st.chunk_samples), 'used_sequential': int(st.used_sequential)})
        if not a.no_parity:
            o = co.COracle(**stream_params('all'), **decoder_flags('all'))
            o.push_iq(iq)
            ge, oe = ctx.edges(), o.edges()
            out['parity'] = {'edges_equal': bool(len(ge) == len(oe) and np.array_equal(ge['idx'].astype(np.int64), oe['idx']) and np.array_equal(ge['d'], oe['d'])
                                                 and np.array_equal(ge['v'], oe['v']) and np.array_equal(ge['t'], oe['t'])),
                             'symbols_equal': bool(all(np.array_equal(ctx.symbols(t), o.symbols(t)) for t in (0, 1))),
                             'packets_equal': bool(ctx.packets() == o.packets()), 'n_edges': int(len(oe))}
    return out


def parity_check(workload, own, flags, n, ahead=False, kind='iq'):
    """Rank 0's chunk, GPU vs the pinned C oracle, full size, every pass compared in full (edges, symbols, packets).
    A tiled capture goes tile by tile.  ahead: as the timed loop does it -- ONE stream, the capture again and again, batch
    k + 2 submitted before batch k is waited for; four rounds (two for a tiled capture), so that a batch that follows a
    synchronous one and a batch that follows one that ran ahead are both compared; the counts reported are the last round's."""
    import numpy as np
    from oracle import c_oracle as co
    from usrp_nfc_amd import api
    tile = len(own) // INPUT_KINDS[kind][1]
    reps = (n + tile - 1) // tile
    rounds = (4 if reps == 1 else 2) if ahead else 1
    ctx = api.NfcContext(input_kind=api_kind(kind), **stream_params(workload), **flags)
    o = co.COracle(**stream_params(workload), **flags)
    ok_edges = ok_sym = ok_pk = True
    per_round = []
    n_ahead = 0
    buf = api.DeviceBuffer(own) if ahead else None
    total = rounds * reps
    nxt = 0
    for k in range(total):   # the stream carries over from pass to pass on both sides; outputs are compared per pass
        if ahead:
            while nxt < total and nxt < k + 3:   # (three in flight, as the timed loop)
                ctx.submit_device(buf, tile)
                nxt += 1
            ctx.wait()
            n_ahead += int(ctx.stats().ran_ahead)
        else:
            ctx.push(own)
        o.clear_outputs()
        oracle_push(o, own, kind)
        ge, oe = ctx.edges(), o.edges()
        ok_edges &= bool(len(ge) == len(oe) and np.array_equal(ge['idx'].astype(np.int64), oe['idx']) and np.array_equal(ge['d'], oe['d'])
                         and np.array_equal(ge['v'], oe['v']) and np.array_equal(ge['t'], oe['t']))
        ok_sym &= all(np.array_equal(ctx.symbols(t), o.symbols(t)) for t in (0, 1))
        gp = ctx.packets()
        ok_pk &= gp == o.packets()
        if k % reps == 0:
            per_round.append([0, 0])
        per_round[-1][0] += len(oe)
        per_round[-1][1] += len(gp)
    ctx.close()
    out = {'vs': 'oracle/nfc_oracle.c on the same %d samples' % n, 'edges_equal': bool(ok_edges), 'symbols_equal': bool(ok_sym),
           'packets_equal': bool(ok_pk), 'n_edges': int(per_round[-1][0]), 'n_packets': int(per_round[-1][1]), 'tiles': reps}
    if ahead:
        out.update({'passes': 'one stream, %d rounds over the capture, batches submitted ahead (%d of %d ran ahead)' % (rounds, n_ahead, total),
                    'first_round': {'n_edges': int(per_round[0][0]), 'n_packets': int(per_round[0][1])},
                    'stationary': bool(per_round[-1] == per_round[-2])})
    return out


def result_digest(edges, sym0, sym1, packets):
    """64-bit digests (and counts) of one rank's decode of its shard: the edge list with GLOBAL sample indices, the two symbol
    streams, the packets.  Same function on the GPU side and on the oracle side of sharded_parity."""
    import hashlib
    import numpy as np

    def h(*arrs):
        m = hashlib.sha256()
        for a_ in arrs:
            m.update(np.ascontiguousarray(a_).tobytes())
        return m.hexdigest()[:16]
    return {'n_edges': int(len(edges)), 'n_packets': int(len(packets)),
            'edges': h(np.asarray(edges['idx'], '<i8'), np.asarray(edges['d'], '<i4'), np.asarray(edges['v'], 'i1'), np.asarray(edges['t'], 'i1')),
            'symbols': h(np.asarray(sym0, np.uint8)) + h(np.asarray(sym1, np.uint8)),
            'packets': h(np.array([t for t, b in packets], 'i1'), np.array([len(b) for t, b in packets], '<i4'),
                         np.array([bit for t, b in packets for bit in b], np.uint8))}


SHARDED_PARITY_CAP = 800_000_000   # samples of the whole capture the stitched parity leg regenerates and decodes on rank 0's host


def sharded_parity(workload, n, world, flags, gathered):
    """EVERY rank against the oracle: rank 0 regenerates the whole world * n sample capture shard by shard, runs the pinned C
    oracle ONCE over it (one stream, from sample 0), cuts its outputs at the shard boundaries and compares each cut with what
    the rank that decoded that shard reported (gathered: every rank's result_digest of its last protocol step).
    A TILED capture (classic1k beyond one tile: BASELINE.json configs[3] / [4]) needs no regeneration -- every rank's chunk is the
    same tile again and again -- so it is checked at full size whatever that is (8 x 1e9 samples: half a minute of the C oracle);
    a generated capture is regenerated shard by shard up to NFC_BENCH_SHARDED_PARITY_CAP samples."""
    from oracle import c_oracle as co
    total = world * n
    tiled = workload == 'classic1k' and n > TILE
    cap = int(float(os.environ.get('NFC_BENCH_SHARDED_PARITY_CAP', SHARDED_PARITY_CAP)))
    if total > cap and not tiled:
        return {'skipped': 'the capture has %d samples, the stitched leg regenerates at most %d (NFC_BENCH_SHARDED_PARITY_CAP raises it)' % (total, cap),
                'ranks_reported': gathered}
    o = co.COracle(**stream_params(workload), **flags)
    equal, want = [], []
    own_r = None
    t0 = time.perf_counter()
    for r in range(world):
        if own_r is None or not tiled:
            _, own_r = make_capture_slice(workload, n, r, world)
        tile = len(own_r) // 2
        o.clear_outputs()
        for _ in range((n + tile - 1) // tile):   # (a tiled capture: the rank's chunk is the tile again and again)
            o.push_iq(own_r)
        d = result_digest(o.edges(), o.symbols(0), o.symbols(1), o.packets())
        want.append(d)
        equal.append(bool(d == gathered[r]))
    del own_r
    return {'vs': 'oracle/nfc_oracle.c, ONE stream over the whole %d-sample capture, cut at the shard boundaries' % total,
            'ranks_equal': equal, 'all_equal': bool(all(equal)), 'n_edges': [d['n_edges'] for d in want], 'n_packets': [d['n_packets'] for d in want],
            'oracle_seconds': time.perf_counter() - t0}


def cpu_baseline(own, flags, params, kind='iq'):
    """The reference's CPU path timed on this host: the pinned C port of the per-sample loop (1 core),
    and -- for the reference's own language -- the line-for-line Python restatement on a prefix."""
    from oracle import c_oracle as co, py_oracle as po
    from usrp_nfc_amd import synth
    import numpy as np
    o = co.COracle(**params, **flags)
    n = len(own) // INPUT_KINDS[kind][1]
    t0 = time.perf_counter()
    oracle_push(o, own, kind)
    tc = time.perf_counter() - t0
    npy = min(n, 4_000_000)
    if kind == 'iq':
        x = synth.envelope_f32(own[:2 * npy])
    elif kind == 'env':
        x = own[:npy]
    else:
        x = (own[:npy].astype(np.float32) / np.float32(32767.0)).astype(np.float32)
        x = x * x
    t0 = time.perf_counter()
    po.run_path(x, chunk=8192, **params, **flags)
    tp = time.perf_counter() - t0
    return {'value': n / tc / 1e6, 'unit': 'Msamples/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle/nfc_oracle.c, whole %d-sample workload, 1 thread (the reference is a single-threaded Python loop)' % n,
            'python_restatement_msamples_s': npy / tp / 1e6,
            'python_sample': 'oracle/py_oracle.py, first %d samples, 8192-sample work() calls (no GNU Radio: envelope by numpy)' % npy,
            'host_cpus': os.cpu_count()}


def _end_to_end_pass(workload, own, flags):
    """Host IQ in, decoded commands out (SURVEY.md 8(d) "separately end-to-end"): pinned host samples -> H2D in pieces -> GPU path ->
    transitions (6 B each), packet tables and packet bits D2H -> packets to bytes / commands on the host (fsm.process_packets, C).
    Three threads, one per resource: the uploader keeps the H2D direction of the link busy (a ring of three device buffers), the main
    thread pushes a piece and reads its outputs back into pinned arrays (the D2H direction: the copy engine writes them in place),
    a worker turns packets into commands.  What bounds it: the link, 8 B per sample in."""
    import ctypes as C
    import queue
    import numpy as np
    from usrp_nfc_amd import api, fsm, _lib
    L = _lib.load()
    n = len(own) // 2
    piece = 1 << 22
    pin = C.c_void_p()
    assert L.nfc_host_alloc_pinned(own.nbytes, C.byref(pin)) == 0
    C.memmove(pin, own.ctypes.data, own.nbytes)
    NB = 3
    bufs = [api.DeviceBuffer(np.zeros(0, np.float32), 0, nbytes=8 * piece) for _ in range(NB)]
    ctx = api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **stream_params(workload), **flags)
    machine = fsm.fsm(callback=lambda cmd, st: None)
    pieces = [(o, min(piece, n - o)) for o in range(0, n, piece)]
    outs = [(api.PinnedArray(piece // 4 + 65536, np.uint32), api.PinnedArray(piece // 4 + 65536, np.uint16)) for _ in range(2)]
    uploaded = [threading.Event() for _ in pieces]
    consumed = [threading.Event() for _ in pieces]   # piece k's device buffer may be overwritten (its push has returned)
    err = []

    def uploader():
        try:
            for k, (o, m) in enumerate(pieces):
                if k >= NB:
                    consumed[k - NB].wait()
                assert L.nfc_device_upload(0, bufs[k % NB].ptr, pin.value + 8 * o, 8 * m) == 0
                uploaded[k].set()
        except Exception as e:   # (the main thread must not wait for ever)
            err.append(e)
            for ev in uploaded:
                ev.set()

    work = queue.Queue()
    n_frames = [0]
    t_fsm = [0.0]

    def protocol():
        while True:
            item = work.get()
            if item is None:
                return
            tabs, bits = item
            t1 = time.perf_counter()
            table = np.concatenate(tabs)
            table = table[np.argsort(table['idx'], kind='stable')]
            table = table[table['n_bits'] > 0]
            if len(table):
                frames, _ = machine.process_packets(table, bits[0], bits[1], dispatch=False)
                n_frames[0] += len(frames)
            t_fsm[0] += time.perf_counter() - t1

    n_edges = 0
    t_push = t_read = t_join = 0.0   # where the main thread's time goes: GPU path, read-back, waiting for the upload
    th_u, th_p = threading.Thread(target=uploader, daemon=True), threading.Thread(target=protocol, daemon=True)   # (daemon: a failure on the main thread must not leave the process waiting for them)
    t0 = time.perf_counter()
    th_u.start()
    th_p.start()
    for k, (o, m) in enumerate(pieces):
        ta = time.perf_counter()
        uploaded[k].wait()
        if err:
            raise err[0]
        tb = time.perf_counter()
        t_join += tb - ta
        ctx.push_device(bufs[k % NB], m)
        consumed[k].set()
        tc = time.perf_counter()
        t_push += tc - tb
        pos, code = ctx.edges_compact(out=(outs[k & 1][0].array, outs[k & 1][1].array))
        n_edges += len(pos)
        tabs = [ctx.packet_table(t) for t in (0, 1)]
        bits = [ctx.packet_bits(t) for t in (0, 1)]
        work.put((tabs, bits))
        t_read += time.perf_counter() - tc
    work.put(None)
    th_p.join()
    th_u.join()
    dt = time.perf_counter() - t0
    ctx.close()
    for a, b in outs:
        a.free()
        b.free()
    L.nfc_host_free_pinned(pin)
    return {'value': n / dt / 1e6, 'unit': 'Msamples/s', 'ms_total': dt * 1e3, 'samples': n, 'edges_to_host': n_edges,
            'commands': n_frames[0], 'piece_samples': piece,
            'what': 'pinned host IQ -> H2D (uploader thread, three device buffers) -> GPU path -> transitions (compact: 6 B each, into pinned arrays) + packets D2H -> fsm (C, worker thread)',
            # (the link itself moves 57 GB/s either way from pinned memory, measured with a bare hipMemcpy on this pool: link_GBs_in is
            # what the pass keeps of the H2D direction -- main_thread_ms says where the rest went)
            'link_GBs': (8 * n + 6 * n_edges) / dt / 1e9, 'link_GBs_in': 8 * n / dt / 1e9, 'mb_in': 8 * n / 1e6, 'mb_out': 6 * n_edges / 1e6,
            'main_thread_ms': {'gpu_path': t_push * 1e3, 'read_back': t_read * 1e3, 'waiting_for_upload': t_join * 1e3},
            'worker_ms': {'fsm': t_fsm[0] * 1e3}}


def end_to_end(workload, own, flags):
    """Two passes over the capture, the second reported: the first one sizes the context's buffers and the pinned staging areas
    (allocations a stream makes once) -- its own figure rides along as first_pass_msamples_s."""
    first = _end_to_end_pass(workload, own, flags)
    second = _end_to_end_pass(workload, own, flags)
    second['first_pass_msamples_s'] = first['value']
    return second


def rank_main(a):
    import numpy as np
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if a.gpus != world and not (a.gpus == 1 and world == 1):
        raise SystemExit('bench: --gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    from usrp_nfc_amd import api, comm as cm, sharding
    backend = os.environ.get('NFC_BENCH_BACKEND', 'rccl')
    backend = 'host' if backend in ('host', 'gloo', 'tcp') else 'rccl'
    force_x = bool(os.environ.get('NFC_BENCH_FORCE_EXCHANGE'))
    ndev = max(1, api.device_count())
    if world > 1 or force_x:
        if backend == 'rccl':
            if world > ndev:
                raise SystemExit('bench: %d ranks but %d GPU(s): RCCL wants one rank per GPU (NFC_BENCH_BACKEND=host shares a GPU)' % (world, ndev))
            comm = cm.RcclComm(local % ndev)
        else:
            comm = cm.HostComm()
    else:
        comm = sharding.LocalComm()
    assert comm.world == world
    n = int(a.samples) if a.samples else default_samples(a.workload)
    kind = a.input_kind
    if kind != 'iq' and world > 1:
        raise SystemExit('bench: --input-kind %s is a one-GPU figure (the sharded runs take fc32 IQ, BASELINE.json\'s metric)' % kind)
    out, own, flags = run_config(a, a.workload, n, a.steps, a.warmup, rank, world, local, comm, backend, not a.no_parity, a.chunk, kind)
    if rank == 0:
        line = {'metric': 'IQ Msamples/s -> decoded bits (2 Msps stream)' if a.workload != 'classic1k' else 'IQ Msamples/s -> decoded bits (10 Msps stream)',
                'value': out['value'], 'unit': 'Msamples/s', 'n_gpus': comm.world, 'steps': a.steps, 'warmup': a.warmup,
                'ms_per_step': out['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': 'f32 envelope / f64 window sums / u8 symbols', 'data': 'synthetic'}
        line.update({k: v for k, v in out.items() if k not in ('value', 'ms_per_step')})
        if not a.no_cpu_baseline and world == 1:   # the CPU baseline is timed on rank 0 of the 1-GPU run only
            line['cpu_baseline'] = cpu_baseline(own, flags, stream_params(a.workload), kind)
        if world == 1 and not a.no_extras and a.workload == 'miller' and n == 100_000_000 and kind == 'iq':
            # the figures SURVEY.md 8(d) asks for beside the headline, measured in this same invocation
            line['end_to_end'] = end_to_end(a.workload, own, flags)
            # (inside the object the driver's record keeps whole: VERDICT r05 item 6)
            line['roofline']['end_to_end_msamples_s'] = line['end_to_end']['value']
            line['roofline']['end_to_end_link_GBs_in'] = line['end_to_end']['link_GBs_in']
            del own
            others = []
            for wl, nn, st in (('manchester', 100_000_000, 20), ('classic1k', 1_000_000_000, 10)):
                o2, _, _ = run_config(a, wl, nn, st, 3, 0, 1, 0, sharding.LocalComm(), 'none', not a.no_parity)
                sa = o2.get('submitted_ahead') or {}
                others.append({'workload': WORKLOAD_NAMES[wl], 'samples': nn, 'steps': st, 'ms_per_step': o2['ms_per_step'], 'value': o2['value'],
                               'unit': 'Msamples/s', 'roofline': o2['roofline'], 'time_chunks': o2['config']['time_chunks'],
                               'submitted_ahead': {k: sa.get(k) for k in ('ms_per_step', 'value', 'ran_ahead', 'steps')} if sa else None,
                               'parity': o2.get('parity')})
            # the other input kinds of the boundary on configs[1]'s capture: the float32 envelope the reference's own sink is handed
            # (4 B/sample) and 16-bit PCM (2 B/sample).  Same kernel, same ~0.19 ms launch -- it is bound by instruction issue,
            # not by bytes -- so the fraction of the HBM roofline falls with the bytes per sample; each entry says its own.
            for kd in ('env', 'i16'):
                o2, _, _ = run_config(a, 'miller', 100_000_000, 20, 3, 0, 1, 0, sharding.LocalComm(), 'none', not a.no_parity, 0, kd)
                others.append({'workload': WORKLOAD_NAMES['miller'], 'input_kind': INPUT_KINDS[kd][2], 'samples': 100_000_000, 'steps': 20,
                               'ms_per_step': o2['ms_per_step'], 'value': o2['value'], 'unit': 'Msamples/s', 'roofline': o2['roofline'],
                               'time_chunks': o2['config']['time_chunks'], 'parity': o2.get('parity')})
            line['other_configs'] = others
            # the unhappy path: chunks that give up / cannot be certified
            line['stress'] = [stress_config(a, nm, 100_000_000) for nm in ('stress_dropouts', 'stress_dropouts_steps', 'stress_hover')]
            clean = out['ms_per_step']
            for e in line['stress']:
                e['vs_clean_step'] = e['ms_per_step'] / clean
    own = None   # (the capture's host copy is not needed any more)
    if world > 1 and not a.workload_given and not a.samples and not a.no_extras:
        # BASELINE.json configs[4]: the -t all decode of the MIFARE Classic 1K capture at 10 Msps, 1e9 samples per GPU, time-sharded
        # over the same ranks (every rank in step: the protocol's collectives run inside)
        st4 = min(a.steps, 10)
        o4, _, _ = run_config(a, 'classic1k', default_samples('classic1k'), st4, min(a.warmup, 3), rank, world, local, comm, backend, not a.no_parity)
        if rank == 0:
            line['other_configs'] = [{'workload': WORKLOAD_NAMES['classic1k'], 'baseline_config': 'configs[4]', 'samples_per_gpu': default_samples('classic1k'),
                                      'n_gpus': world, 'steps': st4, 'ms_per_step': o4['ms_per_step'], 'value': o4['value'], 'unit': 'Msamples/s',
                                      'scaling': 'weak', 'roofline': o4['roofline'], 'config': o4['config'], 'submitted_ahead': o4.get('submitted_ahead'),
                                      'parity': o4.get('parity')}]
    if rank == 0:
        print(json.dumps(line))
        sys.stdout.flush()
    if hasattr(comm, 'barrier'):
        comm.barrier()
    if hasattr(comm, 'close'):
        comm.close()


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(a))   # (this process never touches HIP)
    rank_main(a)


if __name__ == '__main__':
    main()
