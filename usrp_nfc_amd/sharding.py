"""Time sharding of one long capture over the GPUs of a node.

Rank r decodes samples [r*n, (r+1)*n).  The decode of a time chunk needs the exact stream state at
its first sample (ring of accepted samples, running sum, edge-timing variables, decoder and framing
states, bits of a packet that straddles the boundary).  Protocol:

1. speculate: rank r > 0 also holds ``overlap`` samples of its predecessor's chunk.  It starts there
   from a primed state (every ring slot at the estimated carrier level, idle state machines) and runs
   the overlap; the threshold stage forgets its initial ring as slots are overwritten by accepted
   samples, the edge / decoder / framing machines re-synchronise at the next frame gap, so at the
   chunk boundary the state is -- almost always -- exactly the true one;
2. decode the own chunk from that state;
3. exchange: ONE all-gather per round of every rank's (speculated START state, true END state) pair
   (about 8.2 KB each: header + ring + pending packet bits; comm.RcclComm: ncclAllGather over xGMI on a GPU
   node; comm.HostComm: TCP through rank 0 for ranks that share a GPU and for the CPU tests), so that every
   rank can evaluate every comparison itself -- no second collective;
4. verify: rank r's speculated start state must equal rank r-1's true end state, bit for bit.
   By induction from rank 0 (whose start is the stream start) all chunks are exact when every
   comparison holds.  Otherwise the first mismatching rank re-decodes its chunk from the true state and
   the exchange repeats (at most world-1 extra rounds).

No data-path collective: samples never move between ranks.
"""
import numpy as np


def carrier_level(env_head):
    """Robust unloaded-carrier level of an envelope excerpt (the estimate the threshold kernel
    itself speculates with: mean of the lower part of the upper cluster)."""
    x = np.asarray(env_head, np.float32)
    half = 0.5 * float(x.max())
    upper = x[x >= half]
    ca = float(upper.mean())
    sel = x[(x >= half) & (x <= ca)]
    return float(sel.mean() if len(sel) else ca)


_EMPTY = np.zeros(0, np.uint8)
PREFIX = 16          # [u32 length | 12 zero bytes] in front of every state in an exchange slot (nfc_export_state)
PENDING_ROOM = 4096  # bytes of open-packet bits a slot has room for (a frame is a few hundred bits)


def slot_bytes(av_window, header_bytes=96):
    n = PREFIX + header_bytes + 4 * int(av_window) + PENDING_ROOM
    return (n + 15) // 16 * 16


class LocalComm(object):
    """Single-process stand-in (world size 1): nothing is ever exchanged."""
    world, rank = 1, 0
    device_slots = False


def shard_overlap(samp_rate, av_window, longest_frame_bits=164, windows=16):
    """Samples of its predecessor's chunk a rank > 0 also decodes (its speculation warm-up).

    Two things have to converge before the boundary.  The state machines (edge timing, decoders, framing) re-synchronise
    at a frame gap: twice the longest frame of the capture at the stream's rate -- by default an 18-byte answer with parity,
    start and end bit, 164 bit periods of 128 / fc = 9.44 us -- always contains one.  The ring of
    accepted samples starts from a level estimate in every slot, and a slot only takes its true value when a sample that
    lands on it is ACCEPTED; inside frames a third to a half of the samples are rejected (pauses, loaded half bits), so a
    slot needs several passes of the window before the chance that it was rejected every time is negligible: `windows`
    averaging windows (16: measured -- the 10 Msps MIFARE Classic capture, frames back to back with 150 us gaps, still had
    stale slots after 4 windows on every boundary).  A wrong guess costs a re-decode, never exactness; the warm-up itself
    costs windows * av_window samples per rank (0.02 % of a 1e9-sample shard).  Rounded up to a multiple of 256."""
    frame = int(np.ceil(longest_frame_bits * 128.0 / 13.56e6 * samp_rate))
    n = int(windows) * int(av_window) + 2 * frame
    return (n + 255) // 256 * 256


def overlap_schedule(samp_rate, av_window, available, longest_frame_bits=164):
    """Warm-up lengths to try, shortest first: 16, 32, 64 ... averaging windows (+ two frames), as far as `available` samples
    before the shard exist.  decode_shard stops at the first one after which the engine says its window has converged."""
    out, w = [], 16
    while True:
        n = shard_overlap(samp_rate, av_window, longest_frame_bits, w)
        if n > available:
            break
        out.append(n)
        w *= 2
    return out or ([available] if available > 0 else [])


def decode_shard(engine, comm, push_overlap, push_own, start_index, level, force_exchange=False, overlap_steps=None, shard_start=None):
    """Run the protocol for this rank.

    engine: reset(), prime(start_index, level), state_blob(), set_state_blob(blob)
    push_overlap(): feeds the overlap samples (rank > 0) -- outputs are discarded
    push_own():     feeds the rank's own chunk; the engine then holds that chunk's outputs
    force_exchange: run the exchange even with a single rank (exercises the collective path on one GPU).
    overlap_steps:  (optional) warm-up lengths to try, shortest first (overlap_schedule); push_overlap is then called with the
                    length -- it feeds the LAST that-many samples before the shard -- and shard_start is the shard's first sample
                    index.  The warm-up stops at the first length after which ``engine.window_converged()`` holds: every window
                    slot has taken an accepted sample since the prime, i.e. the window no longer depends on the level it was
                    primed with (the fixed 16 windows are the first try, not a constant to tune per capture).
    Returns the number of re-decodes this rank had to do; ``engine.overlap_used`` is the warm-up length that was used.
    """
    rank, world = comm.rank, comm.world
    exchanging = world > 1 or (force_exchange and hasattr(comm, 'exchange'))
    on_device = exchanging and comm.device_slots and hasattr(engine, 'export_state')
    if exchanging:
        comm.bind(engine.av_window, getattr(engine, 'state_bytes', None))   # state_bytes: an engine with its own blob format
    # the engine works on the collective's stream: its exported states are ordered before the all-gather without a host wait
    # (needs a stream of its own as torch's current one: the default stream has no handle to share)
    same_stream = bool(on_device and hasattr(engine, 'set_stream') and comm.stream_handle())
    if same_stream:
        engine.set_stream(comm.stream_handle())

    def capture(slot):   # the engine's current state into an exchange slot
        if not exchanging:
            return
        if on_device:
            engine.export_state(comm.slot_ptr(slot), comm.half)   # GPU -> the device send buffer, asynchronously
        else:
            comm.put(slot, engine.state_blob())

    if rank == 0:
        engine.reset()
        if exchanging and not on_device:
            comm.put(0, _EMPTY)   # (a device send buffer starts zeroed and rank 0 never writes its slot 0)
    elif overlap_steps:
        used = 0
        for k, nov in enumerate(overlap_steps):
            engine.prime(shard_start - nov, level[k] if hasattr(level, '__len__') else level)   # (a level per try: estimated where it starts)
            push_overlap(nov)
            used = nov
            if not hasattr(engine, 'window_converged') or engine.window_converged():
                break
        try:
            engine.overlap_used = used
        except AttributeError:
            pass
        capture(0)
    else:
        engine.prime(start_index, level)
        push_overlap()
        capture(0)
    push_own()
    redos = 0
    if not exchanging:
        return redos
    for _ in range(world):
        # one collective per round: every rank sees every (speculated start, true end) pair and therefore
        # reaches the same verdict without a second exchange
        capture(1)
        if on_device and not same_stream:
            engine.sync()
        pairs = comm.exchange()
        bad = [r for r in range(1, world)
               if not (pairs[r][0].size == pairs[r - 1][1].size and np.array_equal(pairs[r][0], pairs[r - 1][1]))]
        if not bad:
            break
        if rank == bad[0]:
            # this rank's predecessors are exact, so its predecessor's end state is the truth
            engine.set_state_blob(pairs[rank - 1][1].copy())
            capture(0)
            push_own()
            redos += 1
    return redos
