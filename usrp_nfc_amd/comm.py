"""Rank-to-rank exchange for the time sharding of sharding.py, without PyTorch.

Two carriers of the boundary states, same interface (world, rank, device_slots, bind, slot_ptr / put, stream_handle,
exchange, barrier, max_over_ranks, close):

* RcclComm -- ctypes binding of librccl.so: ncclAllGather of the (speculated start | true end) frames straight out of the
  device buffer nfc_export_state writes, on the stream the decode context works on; one process per GPU, one HIP runtime
  per process (librccl and libnfc_amd share libamdhip64).  The 128-byte ncclUniqueId travels through a rendezvous file.
* HostComm -- plain TCP through rank 0, states staged through host memory (nfc_get_state): for ranks that share one GPU
  (RCCL refuses two ranks on a device) and for CPU engines in the tests.

Rendezvous: the ranks of one job share a parent (torch.distributed.run's agent, or bench.py's own spawner) and
MASTER_PORT, so ``$NFC_RDZV_DIR or /tmp / nfc_rdzv_<MASTER_PORT>_<ppid>_<tag>`` names a file only they agree on; rank 0
writes it atomically (temporary name + rename) and removes it when the communicator closes.  MASTER_PORT itself is never
bound here: under torch.distributed.run it belongs to the launcher's own store.
"""
import ctypes as C
import os
import json
import secrets
import socket
import stat
import struct
import time

import numpy as np

from .sharding import PREFIX, slot_bytes


def env_rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def _rdzv_dir():
    """A directory only this user can write: $NFC_RDZV_DIR, or <tmp>/nfc_rdzv_<uid> created 0700 (an existing one must be
    ours and closed to others -- a world-writable /tmp name could be planted by another local user)."""
    d = os.environ.get('NFC_RDZV_DIR')
    if d:
        return d
    d = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'nfc_rdzv_%d' % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        st = os.lstat(d)
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise RuntimeError('rendezvous directory %s is not a private directory of this user' % d)
    return d


def _rdzv_path(tag):
    return os.path.join(_rdzv_dir(), 'nfc_rdzv_%s_%d_%s' % (os.environ.get('MASTER_PORT', '0'), os.getppid(), tag))


def rdzv_publish(tag, payload):
    path = _rdzv_path(tag)
    tmp = '%s.%d.tmp' % (path, os.getpid())
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, 'wb') as f:
        f.write(payload)
    os.replace(tmp, path)
    return path


def rdzv_fetch(tag, timeout=120.0, min_mtime=0.0):
    """The payload rank 0 published under `tag` (a file older than min_mtime is a previous job's leftover)."""
    path = _rdzv_path(tag)
    t_end = time.time() + timeout
    while True:
        try:
            if os.path.getmtime(path) >= min_mtime:
                with open(path, 'rb') as f:
                    return f.read()
        except OSError:
            pass
        if time.time() > t_end:
            raise RuntimeError('rendezvous: %s did not appear within %.0f s' % (path, timeout))
        time.sleep(0.01)


def _frames_to_pairs(got, world, half):
    pairs = []
    for r in range(world):
        pair = []
        for slot in range(2):
            ln = int(got[r, slot, :4].view('<u4')[0])
            if PREFIX + ln > half:   # every rank sees this and fails alike (no rank is left waiting)
                raise RuntimeError('rank %d: boundary state of %d bytes exceeds the %d-byte exchange slot' % (r, ln, half))
            pair.append(got[r, slot, PREFIX:PREFIX + ln].copy())
        pairs.append(tuple(pair))
    return pairs


# ---------------------------------------------------------------------------------------------------------------------
# TCP through rank 0
# ---------------------------------------------------------------------------------------------------------------------
def _send_msg(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise RuntimeError('peer closed the connection')
        buf += part
    return bytes(buf)


def _recv_msg(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class HostComm(object):
    """All-gather of byte frames over TCP: every rank sends its frame to rank 0, rank 0 answers with all of them."""
    device_slots = False

    def __init__(self, rank=None, world=None, tag='host', timeout=120.0):
        r, w = env_rank_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.half = 0
        self._t0 = time.time() - 1.0
        self._peers = []
        self._sock = None
        self._tag = tag
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(('127.0.0.1', 0))   # an ephemeral port: its number travels through the rendezvous file
            srv.listen(self.world)
            token = secrets.token_bytes(16)   # a peer must present it: the file is readable by this user only
            self._path = rdzv_publish(tag, struct.pack('<I', srv.getsockname()[1]) + token)
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                try:
                    hello = _recv_exact(conn, 20)
                except (RuntimeError, OSError):
                    conn.close()
                    continue
                (pr,) = struct.unpack('<I', hello[:4])
                if hello[4:] != token or not (1 <= pr < self.world) or pr in peers:
                    conn.close()   # not a rank of this job (or a rank twice): dropped, the wait goes on
                    continue
                conn.settimeout(None)
                peers[pr] = conn
            srv.close()
            self._peers = [peers[k] for k in range(1, self.world)]
        else:
            t_end = time.time() + timeout
            while True:
                # (fetched again on every attempt: a leftover of a crashed job with the same name is replaced by rank 0's)
                raw = rdzv_fetch(tag, max(1.0, t_end - time.time()), self._t0 - 600.0)
                (port,) = struct.unpack('<I', raw[:4])
                try:
                    s = socket.create_connection(('127.0.0.1', port), timeout=timeout)
                    break
                except OSError:
                    if time.time() > t_end:
                        raise
                    time.sleep(0.02)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(struct.pack('<I', self.rank) + raw[4:20])
            self._sock = s

    # -- collectives on small host payloads --
    def allgather_bytes(self, payload):
        if self.world == 1:
            return [bytes(payload)]
        if self.rank == 0:
            parts = [bytes(payload)] + [_recv_msg(c) for c in self._peers]
            blob = b''.join(struct.pack('<Q', len(p)) + p for p in parts)
            for c in self._peers:
                _send_msg(c, blob)
            return parts
        _send_msg(self._sock, bytes(payload))
        blob = _recv_msg(self._sock)
        parts, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from('<Q', blob, off)
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def barrier(self):
        self.allgather_bytes(b'')

    def max_over_ranks(self, value):
        return max(struct.unpack('<d', p)[0] for p in self.allgather_bytes(struct.pack('<d', float(value))))

    def gather_objects(self, obj):
        """JSON-gather of plain data (numbers, strings, lists, dicts) for the bench's and the tests' bookkeeping: every
        rank gets every object (tuples come back as lists).  Nothing a peer sends is ever executed."""
        return [json.loads(p.decode()) for p in self.allgather_bytes(json.dumps(obj).encode())]

    # -- the boundary exchange --
    def bind(self, av_window, state_bytes=None):
        half = slot_bytes(av_window) if state_bytes is None else (PREFIX + int(state_bytes) + 15) // 16 * 16
        if half != self.half:
            self.half = half
            self._send = np.zeros(2 * half, np.uint8)

    def stream_handle(self):
        return None

    def put(self, slot, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        frame = self._send[slot * self.half:(slot + 1) * self.half]
        frame[:] = 0
        frame[:4] = np.array([blob.size], '<u4').view(np.uint8)
        if PREFIX + blob.size <= self.half:
            frame[PREFIX:PREFIX + blob.size] = blob

    def exchange(self):
        parts = self.allgather_bytes(self._send.tobytes())
        got = np.frombuffer(b''.join(parts), np.uint8).reshape(self.world, 2, self.half)
        return _frames_to_pairs(got, self.world, self.half)

    def close(self):
        for c in self._peers:
            c.close()
        if self._sock is not None:
            self._sock.close()
        self._peers, self._sock = [], None
        if self.rank == 0 and self.world > 1:
            try:
                os.remove(self._path)
            except OSError:
                pass


# ---------------------------------------------------------------------------------------------------------------------
# RCCL
# ---------------------------------------------------------------------------------------------------------------------
class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]   # rccl.h: NCCL_UNIQUE_ID_BYTES


NCCL_UINT8, NCCL_FLOAT64 = 1, 8      # ncclDataType_t
NCCL_MAX = 2                         # ncclRedOp_t

_rccl = None


def load_rccl():
    global _rccl
    if _rccl is None:
        last = None
        for name in (os.environ.get('NFC_RCCL_LIB'), 'librccl.so', '/opt/rocm/lib/librccl.so', 'librccl.so.1'):
            if not name:
                continue
            try:
                _rccl = C.CDLL(name)
                break
            except OSError as e:
                last = e
        if _rccl is None:
            raise RuntimeError('librccl.so not found: %s' % last)
        vp = C.c_void_p
        _rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        _rccl.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, _UniqueId, C.c_int]
        _rccl.ncclAllGather.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
        _rccl.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
        _rccl.ncclCommDestroy.argtypes = [vp]
        _rccl.ncclCommCount.argtypes = [vp, C.POINTER(C.c_int)]
        _rccl.ncclGetErrorString.argtypes = [C.c_int]
        _rccl.ncclGetErrorString.restype = C.c_char_p
    return _rccl


class RcclComm(object):
    """ncclAllGather of the boundary frames over xGMI; one rank per GPU."""
    device_slots = True

    def __init__(self, device, rank=None, world=None, tag='rccl', timeout=120.0):
        from . import _lib
        r, w = env_rank_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.device = int(device)
        self.L = _lib.load()
        self.N = load_rccl()
        self.half = 0
        self._send = self._recv = self._host = None
        t0 = time.time() - 1.0
        uid = _UniqueId()
        if self.rank == 0:
            self._ck(self.N.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
            self._path = rdzv_publish(tag, C.string_at(C.addressof(uid), 128))
        else:
            raw = rdzv_fetch(tag, timeout, t0 - 600.0)
            C.memmove(C.addressof(uid), raw, 128)
        self.stream = C.c_void_p()
        self._ckl(self.L.nfc_stream_create(self.device, C.byref(self.stream)), 'nfc_stream_create')
        self.comm = C.c_void_p()
        self._ck(self.N.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), 'ncclCommInitRank')
        cnt = C.c_int(0)
        self._ck(self.N.ncclCommCount(self.comm, C.byref(cnt)), 'ncclCommCount')
        self.ranks_seen = int(cnt.value)   # what RCCL itself says the communicator spans
        if self.ranks_seen != self.world:
            raise RuntimeError('RCCL communicator spans %d ranks, %d expected' % (self.ranks_seen, self.world))
        # scratch for the scalar collectives (barrier, max of a double)
        self._scal = C.c_void_p()
        self._scal_host = C.c_void_p()
        self._ckl(self.L.nfc_device_alloc(self.device, 64, C.byref(self._scal)), 'nfc_device_alloc')
        self._ckl(self.L.nfc_host_alloc_pinned(64, C.byref(self._scal_host)), 'nfc_host_alloc_pinned')

    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s: %s' % (what, self.N.ncclGetErrorString(rc).decode()))

    def _ckl(self, rc, what):
        """A call into libnfc_amd (made unconditionally -- never inside an assert, which python -O strips)."""
        if rc != 0:
            raise RuntimeError('%s failed (%d): %s' % (what, rc, (self.L.nfc_last_error(None) or b'').decode()))

    def bind(self, av_window, state_bytes=None):
        half = slot_bytes(av_window) if state_bytes is None else (PREFIX + int(state_bytes) + 15) // 16 * 16
        if half == self.half:
            return
        self._free_buffers()
        self.half = half
        self._send, self._recv, self._host = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ckl(self.L.nfc_device_alloc(self.device, 2 * half, C.byref(self._send)), 'nfc_device_alloc')
        self._ckl(self.L.nfc_device_alloc(self.device, self.world * 2 * half, C.byref(self._recv)), 'nfc_device_alloc')
        self._ckl(self.L.nfc_host_alloc_pinned(self.world * 2 * half, C.byref(self._host)), 'nfc_host_alloc_pinned')
        self._ckl(self.L.nfc_device_fill(self.device, self._send, 0, 2 * half), 'nfc_device_fill')   # (rank 0 never writes its slot 0)

    def slot_ptr(self, slot):
        return self._send.value + slot * self.half

    def stream_handle(self):
        return self.stream.value

    def exchange(self):
        n = 2 * self.half
        self._ck(self.N.ncclAllGather(self._send, self._recv, n, NCCL_UINT8, self.comm, self.stream), 'ncclAllGather')
        self._ckl(self.L.nfc_device_download_async(self.device, self._host, self._recv, self.world * n, self.stream), 'nfc_device_download_async')
        self._ckl(self.L.nfc_stream_sync(self.device, self.stream), 'nfc_stream_sync')
        got = np.frombuffer((C.c_uint8 * (self.world * n)).from_address(self._host.value), np.uint8).reshape(self.world, 2, self.half)
        return _frames_to_pairs(got, self.world, self.half)

    def max_over_ranks(self, value):
        host = (C.c_double * 1).from_address(self._scal_host.value)
        host[0] = float(value)
        self._ckl(self.L.nfc_device_upload(self.device, self._scal, self._scal_host, 8), 'nfc_device_upload')
        self._ck(self.N.ncclAllReduce(self._scal, self._scal, 1, NCCL_FLOAT64, NCCL_MAX, self.comm, self.stream), 'ncclAllReduce')
        self._ckl(self.L.nfc_device_download_async(self.device, self._scal_host, self._scal, 8, self.stream), 'nfc_device_download_async')
        self._ckl(self.L.nfc_stream_sync(self.device, self.stream), 'nfc_stream_sync')
        return float(host[0])

    def barrier(self):
        self.max_over_ranks(0.0)

    def allgather_bytes(self, payload, cap=4096):
        """Small host payloads (<= cap - 4 bytes each) through ncclAllGather: the bench's per-rank digests and counts."""
        payload = bytes(payload)
        if len(payload) + 4 > cap:
            raise RuntimeError('payload of %d bytes exceeds the %d-byte gather slot' % (len(payload), cap))
        if getattr(self, '_gcap', 0) != cap:
            self._free_gather()
            self._gsend, self._grecv, self._ghost = C.c_void_p(), C.c_void_p(), C.c_void_p()
            self._ckl(self.L.nfc_device_alloc(self.device, cap, C.byref(self._gsend)), 'nfc_device_alloc')
            self._ckl(self.L.nfc_device_alloc(self.device, self.world * cap, C.byref(self._grecv)), 'nfc_device_alloc')
            self._ckl(self.L.nfc_host_alloc_pinned(self.world * cap, C.byref(self._ghost)), 'nfc_host_alloc_pinned')
            self._gcap = cap
        host = (C.c_uint8 * (self.world * cap)).from_address(self._ghost.value)
        frame = struct.pack('<I', len(payload)) + payload
        C.memmove(self._ghost.value, frame, len(frame))
        self._ckl(self.L.nfc_device_upload(self.device, self._gsend, self._ghost, cap), 'nfc_device_upload')
        self._ck(self.N.ncclAllGather(self._gsend, self._grecv, cap, NCCL_UINT8, self.comm, self.stream), 'ncclAllGather')
        self._ckl(self.L.nfc_device_download_async(self.device, self._ghost, self._grecv, self.world * cap, self.stream), 'nfc_device_download_async')
        self._ckl(self.L.nfc_stream_sync(self.device, self.stream), 'nfc_stream_sync')
        raw = bytes(host)
        out = []
        for r in range(self.world):
            (n,) = struct.unpack_from('<I', raw, r * cap)
            out.append(raw[r * cap + 4:r * cap + 4 + n])
        return out

    def gather_objects(self, obj):
        return [json.loads(p.decode()) for p in self.allgather_bytes(json.dumps(obj).encode())]

    def _free_gather(self):
        if getattr(self, '_gcap', 0):
            self.L.nfc_device_free(self.device, self._gsend)
            self.L.nfc_device_free(self.device, self._grecv)
            self.L.nfc_host_free_pinned(self._ghost)
            self._gcap = 0

    def _free_buffers(self):
        if self._send is not None:
            self.L.nfc_device_free(self.device, self._send)
            self.L.nfc_device_free(self.device, self._recv)
            self.L.nfc_host_free_pinned(self._host)
            self._send = self._recv = self._host = None

    def close(self):
        if getattr(self, 'comm', None):
            self.L.nfc_stream_sync(self.device, self.stream)
            self.N.ncclCommDestroy(self.comm)
            self.comm = None
            self._free_buffers()
            self._free_gather()
            self.L.nfc_device_free(self.device, self._scal)
            self.L.nfc_host_free_pinned(self._scal_host)
            self.L.nfc_stream_destroy(self.device, self.stream)
            if self.rank == 0:
                try:
                    os.remove(self._path)
                except OSError:
                    pass
