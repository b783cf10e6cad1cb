"""Test infrastructure: the boundary exchange of usrp_nfc_amd.sharding carried by torch.distributed's gloo backend (CPU).
The product's own carriers (usrp_nfc_amd/comm.py: RCCL, TCP) need no PyTorch; this one exists so that the N > 1 protocol is
also exercised over a stock collective library in the CPU suite."""
import numpy as np

from usrp_nfc_amd.sharding import PREFIX, slot_bytes


class GlooComm(object):
    device_slots = False

    def __init__(self, dist):
        import torch
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self._torch = torch
        self.half = 0

    def bind(self, av_window, state_bytes=None):
        half = slot_bytes(av_window) if state_bytes is None else (PREFIX + int(state_bytes) + 15) // 16 * 16
        if half == self.half:
            return
        torch = self._torch
        self.half = half
        self._send = torch.zeros(2 * half, dtype=torch.uint8)
        self._recv = torch.zeros(self.world * 2 * half, dtype=torch.uint8)
        self._recv_parts = list(self._recv.chunk(self.world))

    def stream_handle(self):
        return None

    def put(self, slot, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        frame = np.zeros(self.half, np.uint8)
        frame[:4] = np.array([blob.size], '<u4').view(np.uint8)
        if PREFIX + blob.size <= self.half:
            frame[PREFIX:PREFIX + blob.size] = blob
        self._send[slot * self.half:(slot + 1) * self.half] = self._torch.from_numpy(frame)

    def exchange(self):
        self.dist.all_gather(self._recv_parts, self._send)
        got = self._recv.numpy().reshape(self.world, 2, self.half)
        pairs = []
        for r in range(self.world):
            pair = []
            for slot in range(2):
                ln = int(got[r, slot, :4].view('<u4')[0])
                if PREFIX + ln > self.half:
                    raise RuntimeError('rank %d: boundary state of %d bytes exceeds the %d-byte exchange slot' % (r, ln, self.half))
                pair.append(got[r, slot, PREFIX:PREFIX + ln].copy())
            pairs.append(tuple(pair))
        return pairs
