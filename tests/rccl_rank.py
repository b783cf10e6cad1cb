"""One rank of the two-process RCCL test (tests/test_sharding.py: test_two_processes_rccl): started as a fresh child process per
GPU with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT in its environment.  Decodes its time shard of the 10 Msps MIFARE Classic
capture with sharding.decode_shard over comm.RcclComm (ncclAllGather of the boundary states between two devices) and prints one
JSON line: re-decodes, what RCCL says the communicator spans, digests of its decode and of the oracle's cut of the whole capture."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from oracle import c_oracle as co
    from usrp_nfc_amd import api, comm as cm, sharding, synth
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    sabotage = int(os.environ.get('NFC_TEST_SABOTAGE', '-1'))
    n_per = 700_000
    params = dict(samp_rate=10e6, hi_val=1.1, av_window=10000, max_len=250)
    frames, _ = synth.frames_from_trace(os.path.join(ROOT, 'tests', 'golden', '1k_with_enc.out'))
    m = synth.tiled_profile(synth.modulation_profile(frames, rate_msps=10.0, lead_in=0, tail=0), world * n_per)
    m[:15000] = 1.0   # idle lead-in that covers the window
    iq = synth.iq_from_profile(m, seed=11)
    overlap = sharding.shard_overlap(10e6, 10000)
    lo = rank * n_per
    nov = (512 if sabotage == rank else overlap) if rank else 0
    own, ov = iq[2 * lo:2 * (lo + n_per)], iq[2 * (lo - nov):2 * lo]
    dev = int(os.environ.get('LOCAL_RANK', rank)) % max(1, api.device_count())
    comm = cm.RcclComm(dev)
    try:
        ctx = api.NfcContext(input_kind=api.NFC_IN_IQ_F32, device=dev, **params)
        level = sharding.carrier_level(synth.envelope_f32(ov[:2 * 4096])) if rank else 0.0
        redos = sharding.decode_shard(ctx, comm, lambda: ctx.push(ov), lambda: ctx.push(own), lo - nov, level)
        got = bench.result_digest(ctx.edges(), ctx.symbols(0), ctx.symbols(1), ctx.packets())
        # the oracle over the capture up to the end of this rank's shard, its outputs cut at the shard's start
        o = co.COracle(**params)
        if lo:
            o.push_iq(iq[:2 * lo])
            o.clear_outputs()
        o.push_iq(own)
        want = bench.result_digest(o.edges(), o.symbols(0), o.symbols(1), o.packets())
        print(json.dumps({'rank': rank, 'redos': redos, 'ranks_seen': comm.ranks_seen, 'device': dev, 'got': got, 'want': want}))
        sys.stdout.flush()
        comm.barrier()
        ctx.set_stream(None)
        ctx.close()
    finally:
        comm.close()


if __name__ == '__main__':
    main()
