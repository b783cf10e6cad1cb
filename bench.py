#!/usr/bin/env python3
"""bench.py -- IQ Msamples/s -> decoded bits on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--samples S] [--workload miller|manchester|all]

One "step" = one pass of the whole hot path (envelope -> threshold -> edges -> Miller/Manchester ->
framing) over the rank's batch of synthetic IQ, input already resident in HBM.  At N=1 the workload is
BASELINE.json configs[1]: Miller-only decode of 1e8 synthetic IQ samples @ 2 Msps.  For N>1 (launched by
torch.distributed.run, one rank per GPU) every rank holds one contiguous time chunk of a single N*1e8-sample
capture plus an overlap prefix; it decodes its chunk from a speculated boundary state, the true boundary
states travel over RCCL (all_gather), and a rank whose speculation was wrong re-decodes (weak scaling).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from usrp_nfc_amd import api, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
OVERLAP = 8192          # samples of the predecessor's chunk each rank > 0 also holds: the 2000-sample window plus twice the longest
                        # ISO 14443A frame at 2 Msps (163 bits ~ 3.1 k samples); a wrong guess costs a re-decode, never exactness


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)   # 0.45 ms each: the device reaches its steady clocks within the first few dozen
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--samples', type=float, default=1e8, help='samples per GPU')
    ap.add_argument('--workload', default='miller', choices=['miller', 'manchester', 'all', 'classic1k'],
                    help="BASELINE.json configs[1] / [2] / both decoders at 2 Msps, or configs[3] / [4]: the MIFARE Classic 1K "
                         "transaction of outputs/1k_with_enc.out at 10 Msps (pass --samples 1e9)")
    ap.add_argument('--chunk', type=int, default=0, help='time-chunk samples of the threshold kernel (0: library default)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    return ap.parse_args()


def decoder_flags(workload):
    return dict(reader=workload in ('miller', 'all', 'classic1k'), tag=workload in ('manchester', 'all', 'classic1k'))


def stream_params(workload):
    """Sample rate and the constructor arguments that go with it.  At 10 Msps the reference's fixed-sample defaults
    (av_window 2000, max_len 50 samples) chop every bit period (hard part 5): they are scaled
    with the rate, as its keyword arguments allow (transition_sink.py:12)."""
    if workload == 'classic1k':
        return dict(samp_rate=10e6, hi_val=1.1, av_window=10000, max_len=250)
    return dict(samp_rate=2e6, hi_val=1.1)


def shard_overlap(workload):
    return OVERLAP * 5 if workload == 'classic1k' else OVERLAP   # the same time span at 10 Msps


def make_capture_slice(workload, n_per_rank, rank, world):
    """Rank's time chunk of ONE capture of world*n_per_rank samples (plus OVERLAP samples before it).

    The modulation profile is a pure function of the global sample index (a frame sequence tiled after a
    3000-sample idle lead-in); the noise comes from a per-rank PCG64 stream, the overlap region from the
    predecessor's stream, so neighbouring ranks agree on the samples they share."""
    if workload == 'classic1k':
        frames, _ = synth.frames_from_trace(os.path.join(ROOT, 'tests', 'golden', '1k_with_enc.out'))
        period = synth.modulation_profile(frames, rate_msps=10.0, lead_in=0, tail=0)
        lead = 15000   # covers the 10000-sample window
    else:
        picks = {'miller': (0, 2, 4, 10), 'manchester': (1, 3, 5, 11), 'all': tuple(range(19))}[workload]
        frames = [(d, synth.frame_bits(data, sb)) for d, _, data, sb in (synth.ULTRALIGHT_TXN[i] for i in picks)]
        period = synth.modulation_profile(frames, rate_msps=2.0, lead_in=0, tail=0)
        lead = 3000
    overlap = shard_overlap(workload)

    def profile(g_lo, g_hi):
        g = np.arange(g_lo, g_hi, dtype=np.int64)
        m = np.ones(len(g), np.float32)
        body = g >= lead
        m[body] = period[(g[body] - lead) % len(period)]
        return m

    def noisy(g_lo, g_hi, owner):
        # owner's stream covers [owner*n, (owner+1)*n); take the sub-range
        rng = np.random.Generator(np.random.PCG64([synth.SEED, owner]))
        base = owner * n_per_rank
        iq = rng.standard_normal(2 * n_per_rank, dtype=np.float32)[2 * (g_lo - base):2 * (g_hi - base)]
        iq *= np.float32(0.002)
        m = profile(g_lo, g_hi)
        iq[0::2] += (np.float32(0.5 * np.cos(0.3)) * m).astype(np.float32)
        iq[1::2] += (np.float32(0.5 * np.sin(0.3)) * m).astype(np.float32)
        return iq

    lo = rank * n_per_rank
    own = noisy(lo, lo + n_per_rank, rank)
    if rank == 0:
        return np.zeros(0, np.float32), own
    ov = noisy(lo - overlap, lo, rank - 1)
    return ov, own


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    n = int(a.samples)
    dist = None
    backend = os.environ.get('NFC_BENCH_BACKEND', 'nccl')   # 'gloo': plumbing test of the N>1 path on one GPU
    force_x = bool(os.environ.get('NFC_BENCH_FORCE_EXCHANGE')) and 'MASTER_ADDR' in os.environ   # 1-rank smoke of the RCCL path
    if world > 1 or force_x:
        import torch
        import torch.distributed as dist
        ndev = max(1, torch.cuda.device_count())
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')
    dev = (local % ndev) if world > 1 else 0

    ov, own = make_capture_slice(a.workload, n, rank, world)
    flags = decoder_flags(a.workload)
    d_own = api.DeviceBuffer(own, dev)
    d_ov = api.DeviceBuffer(ov, dev) if len(ov) else None
    ctx = api.NfcContext(input_kind=api.NFC_IN_IQ_F32, device=dev, chunk_samples=a.chunk, **stream_params(a.workload), **flags)

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            if backend == 'nccl':
                torch.cuda.synchronize()

    from usrp_nfc_amd import sharding
    if dist is not None:
        import torch
        comm = sharding.TorchDistComm(dist, torch.device('cuda', local) if backend == 'nccl' else torch.device('cpu'))   # nccl == RCCL over xGMI
        if backend == 'nccl':
            # a stream of its own as torch's current one: the decode context joins it (nfc_set_stream), so the exported
            # boundary states are ordered before the all-gather on the device, without a host wait
            torch.cuda.set_stream(torch.cuda.Stream(device=local))
    else:
        comm = sharding.LocalComm()
    level = sharding.carrier_level(synth.envelope_f32(ov[:2 * 4096])) if len(ov) else 0.0
    g_lo = rank * n
    redo_count = 0

    force_exchange = bool(os.environ.get('NFC_BENCH_FORCE_EXCHANGE'))
    n_ov = len(ov) // 2

    def push_overlap():
        ctx.push_device(d_ov, n_ov)

    def push_own():
        ctx.push_device(d_own, n)

    def one_step():
        nonlocal redo_count
        redo_count += sharding.decode_shard(ctx, comm, push_overlap, push_own, g_lo - n_ov, level, force_exchange=force_exchange)

    for _ in range(a.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    n_pass = []
    for k in range(a.steps):
        # every 8th k_threshold launch of the timed region carries its own start / stop HIP events (nfc_amd.h:
        # nfc_set_timing; a timed launch costs the step ~10 us, so not all of them are); the host reads the statistics
        # of those steps only (the device idles while the host is between two pushes)
        timed = k % 8 == 0
        if timed or k % 8 == 1:
            ctx.set_timing(1 if timed else 0)
        one_step()
        if timed:
            st = ctx.stats()
            kernel_ms += [st.ms_threshold_kernel[i] for i in range(st.n_threshold_timed)]
            n_pass.append(st.threshold_passes)
    barrier()
    dt = time.perf_counter() - t0
    ctx.set_timing(2)   # one more, untimed, step for the per-stage split reported beside the headline
    one_step()
    st = ctx.stats()
    if dist is not None:
        import torch
        tmax = torch.tensor([dt], device='cuda' if backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    cnt = ctx.counts()
    st = ctx.stats()
    n_edges = int(cnt.n_edges)
    out = None
    if rank == 0:
        ms_step = dt / a.steps * 1e3
        k_avg = float(np.mean(kernel_ms)) if kernel_ms else float('nan')
        alg_bytes = 8.0 * n + 16.0 * n_edges                       # 8 B/sample + 16 B/edge
        achieved = alg_bytes / (k_avg * 1e-3) / 1e9
        out = {
            'metric': 'IQ Msamples/s -> decoded bits (2 Msps stream)',
            'value': world * n * a.steps / dt / 1e6,
            'unit': 'Msamples/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 envelope / f64 window sums / u8 symbols', 'data': 'synthetic',
            'config': {'workload': {'miller': 'configs[1]: Miller-only decode, synthetic IQ @2 Msps',
                                    'manchester': 'configs[2]: Manchester-only decode, synthetic IQ @2 Msps',
                                    'all': 'both decoders (-t all), Ultralight transaction, synthetic IQ @2 Msps',
                                    'classic1k': 'configs[3]/[4]: -t all @10 Msps, MIFARE Classic 1K transaction tiled '
                                                 '(av_window 10000, max_len 250)'}[a.workload],
                       'samples_per_gpu': n, 'time_chunk_samples': int(st.chunk_samples), 'time_chunks': int(st.n_chunks), 'parallelism': 'time-chunk x%d' % world,
                       'edges_per_gpu': n_edges, 'symbols_reader': int(cnt.n_symbols[1]),
                       'symbols_tag': int(cnt.n_symbols[0]), 'packets': int(cnt.n_packets[0] + cnt.n_packets[1]),
                       'boundary_redos': redo_count},
            'roofline': {'bound': 'hbm', 'kernel': 'k_threshold (fused envelope + gated-mean threshold)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': hbm_traffic(a, n), 'avg_launch_ms': k_avg, 'launches_timed': len(kernel_ms),
                         'launches_per_step': float(np.mean(n_pass)) if n_pass else None,
                         'algorithmic_bytes_per_launch': alg_bytes},
            'stage_ms_extra_step': {'total_device': st.ms_total, 'threshold': st.ms_threshold, 'edges': st.ms_edges,
                                   'decode': st.ms_decode, 'used_sequential': int(st.used_sequential)},
        }
        if not a.no_parity:
            out['parity'] = parity_check(a, own, flags, n)
            # the context the timed loop ran in must have produced the same decode (rank 0's shard starts the stream, so
            # its outputs are those of the fresh decode the oracle was compared with): not only the side context is checked
            out['parity']['timed_loop_counts_equal'] = bool(
                n_edges == out['parity']['n_edges'] and int(cnt.n_packets[0] + cnt.n_packets[1]) == out['parity']['n_packets'])
            if not out['parity']['timed_loop_counts_equal']:
                raise SystemExit('bench: the timed loop decoded %d edges / %d packets, the oracle %d / %d' % (
                    n_edges, int(cnt.n_packets[0] + cnt.n_packets[1]), out['parity']['n_edges'], out['parity']['n_packets']))
        if not a.no_cpu_baseline and world == 1:   # the CPU baseline is timed on rank 0 of the 1-GPU run only
            out['cpu_baseline'] = cpu_baseline(own, flags, stream_params(a.workload))
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        ctx.set_stream(None)   # back on its own stream before torch's streams go away
        dist.barrier()
        dist.destroy_process_group()


def hbm_traffic(a, n):
    """HBM bytes per k_threshold launch from the rocprofv3 PMC passes recorded under profiles/ (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md).  None unless the recorded run matches this workload."""
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
        if rec.get('workload') == a.workload and int(rec.get('samples', 0)) == n:
            return rec['bytes_per_launch']
    except Exception:
        pass
    return None


def parity_check(a, own, flags, n):
    """Rank 0's chunk decoded from a fresh stream, GPU vs the pinned C oracle, full size."""
    from oracle import c_oracle as co
    ctx = api.NfcContext(input_kind=api.NFC_IN_IQ_F32, **stream_params(a.workload), **flags)
    ctx.push(own)
    o = co.COracle(**stream_params(a.workload), **flags)
    o.push_iq(own)
    ge, oe = ctx.edges(), o.edges()
    ok_edges = len(ge) == len(oe) and np.array_equal(ge['idx'].astype(np.int64), oe['idx']) and \
        np.array_equal(ge['d'], oe['d']) and np.array_equal(ge['v'], oe['v']) and np.array_equal(ge['t'], oe['t'])
    ok_sym = all(np.array_equal(ctx.symbols(t), o.symbols(t)) for t in (0, 1))
    gp = ctx.packets()
    ok_pk = gp == o.packets()
    ctx.close()
    return {'vs': 'oracle/nfc_oracle.c on the same %d samples' % n, 'edges_equal': bool(ok_edges),
            'symbols_equal': bool(ok_sym), 'packets_equal': bool(ok_pk), 'n_edges': int(len(oe)), 'n_packets': len(gp)}


def cpu_baseline(own, flags, params):
    """The reference's CPU path timed on this host: the pinned C port of the per-sample loop (1 core),
    and -- for the reference's own language -- the line-for-line Python restatement on a prefix."""
    from oracle import c_oracle as co, py_oracle as po
    o = co.COracle(**params, **flags)
    n = len(own) // 2
    t0 = time.perf_counter()
    o.push_iq(own)
    tc = time.perf_counter() - t0
    npy = min(n, 4_000_000)
    x = synth.envelope_f32(own[:2 * npy])
    t0 = time.perf_counter()
    po.run_path(x, chunk=8192, **params, **flags)
    tp = time.perf_counter() - t0
    return {'value': n / tc / 1e6, 'unit': 'Msamples/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle/nfc_oracle.c (C restatement of transition_sink+decoders), whole %d-sample workload, '
                      '1 thread; the reference itself is a single-threaded Python loop' % n,
            'python_restatement_msamples_s': npy / tp / 1e6,
            'python_sample': 'oracle/py_oracle.py on the first %d samples in 8192-sample work() calls '
                             '(GNU Radio is not installed: envelope by numpy)' % npy,
            'host_cpus': os.cpu_count()}


if __name__ == '__main__':
    main()
