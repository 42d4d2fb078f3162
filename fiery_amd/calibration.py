"""Calibration table: host-inverted camera matrices served to a graph-capturable device lookup.

`get_geometry` (reference: fiery/models/fiery.py:193-208) needs `R . K^-1` per camera; the reference's CPU path gets
`K^-1` from LAPACK (`torch.inverse`).  The device closed form equals LAPACK bit for bit only for zero-skew pinhole
intrinsics; a general 3x3 through the adjugate lands a few ulp away, and a few ulp in a matrix entry moves points that
sit on a voxel edge into the neighbouring cell.  Computing the matrices on the host per call (`camera_matrix_mode =
'host'`) is exact for any K but reads the calibration back every step - a synchronisation, and not capturable.

A rig's calibration is a handful of matrices that repeat for the whole drive, so this table keys the host result on the
calibration's CONTENT instead:

* an entry = the 21 words that enter the computation (K's nine, the three rows of [R | t]) -> the twelve numbers
  `host_camera_matrices` returns for them (the reference's own operators: `torch.inverse`, `matmul`);
* `fiery_camera_matrices_cached` (csrc/lift_splat.hip) hashes each camera's 21 words, probes the table in HBM, copies the
  twelve numbers on a hit; on a miss it evaluates the device form and appends the camera's words to a miss list;
* the miss list is copied to pinned host memory behind the kernel (a memcpy node when captured); `lookup` looks at the
  PREVIOUS calls' lists before it launches - no wait: a list still in flight is picked up one call later - and `prime`s
  the table with what it finds.  Whatever words the host reads, the entry it makes is right for exactly those words (the
  value is computed from the key), so a torn read can only add an entry nobody asks for.

Consequences: with the rig's calibrations primed (`Fiery.prime_calibrations`, or the capture call of
`bev_forward_graph` / `forward_graph`, which primes from its own arguments) every replay is bit-exact against the
reference's CPU path for ANY intrinsics; a calibration never seen before is served by the device form for the calls until
its miss list has landed (counted in `stats`), then exactly.
"""
import numpy as np
import torch

ENTRY_WORDS = 36        # FIERY_CALIB_ENTRY_WORDS: [0] occupied, [1..21] key, [22..33] value, [34..35] unused
KEY_WORDS = 21          # FIERY_CALIB_KEY_WORDS
MISS_WORDS = 24         # FIERY_CALIB_MISS_WORDS: key, camera index, two unused
MISS_HEADER = 4         # FIERY_CALIB_MISS_HEADER: [0] entries written, [1] misses in the launch, [2] launch number, [3] unused
PROBES = 16             # FIERY_CALIB_PROBES


def key_words(intrinsics, extrinsics):
    """(..., 3, 3), (..., 4, 4) float32 on the host -> (N, 21) uint32: the bit patterns the camera matrices depend on."""
    K = np.ascontiguousarray(intrinsics, dtype=np.float32).reshape(-1, 9)
    E = np.ascontiguousarray(extrinsics, dtype=np.float32).reshape(-1, 16)[:, :12]
    return np.concatenate([K, E], axis=1).view(np.uint32)


def hash_words(words):
    """FNV-1a over the 21 words with a final fold - the device kernel's `calibration_hash`, word for word."""
    h = np.full(words.shape[0], 2166136261, dtype=np.uint64)
    for i in range(words.shape[1]):
        h = ((h ^ words[:, i].astype(np.uint64)) * np.uint64(16777619)) & np.uint64(0xFFFFFFFF)
    return (h ^ (h >> np.uint64(15))).astype(np.uint32)


class CalibrationTable:
    def __init__(self, lib, device, slots=4096, miss_capacity=64):
        assert slots & (slots - 1) == 0 and slots >= 64
        self.lib, self.device, self.slots, self.miss_capacity = lib, torch.device(device), slots, miss_capacity
        self.host = np.zeros((slots, ENTRY_WORDS), dtype=np.uint32)
        self.known = {}                                             # key bytes -> slot
        self.table = torch.zeros(slots * ENTRY_WORDS, dtype=torch.int32, device=self.device)
        n_miss = MISS_HEADER + miss_capacity * MISS_WORDS + 1       # (+ the launch number again, behind the entries)
        self.misses = torch.zeros(n_miss, dtype=torch.int32, device=self.device)
        pin = self.device.type == 'cuda'
        self.landed = torch.zeros(n_miss, dtype=torch.int32, pin_memory=pin)
        self.seen_launch = 0
        # keys a miss list has shown ONCE (bounded; see absorb_miss_lists): a key is filed when it comes back
        self.candidates = set()
        self.max_candidates = 4 * slots
        # changed rows travel through a pinned staging buffer, `upload_rows` rows at a time, without blocking the host
        self.upload_rows = 256
        self.staging = torch.zeros(self.upload_rows * ENTRY_WORDS, dtype=torch.int32, pin_memory=pin)
        self.staging_event = None
        self.stats = dict(entries=0, primed=0, from_miss_lists=0, rejected=0, miss_lists=0, seen_once=0)

    # -- host side --------------------------------------------------------------------------------------
    def prime(self, intrinsics, extrinsics):
        """Add the calibrations in (..., 3, 3), (..., 4, 4) (any device; read back if they live on the GPU - this is the
        synchronous way in, for set-up time).  Returns the number of new entries."""
        from .model import host_camera_matrices
        K = intrinsics.detach().to(device='cpu', dtype=torch.float32).reshape(-1, 3, 3)
        E = extrinsics.detach().to(device='cpu', dtype=torch.float32).reshape(-1, 4, 4)
        words = key_words(K.numpy(), E.numpy())
        fresh, first = [], set()
        for i in range(words.shape[0]):
            kb = words[i].tobytes()
            if kb not in self.known and kb not in first:
                first.add(kb)
                fresh.append(i)
        if not fresh:
            return 0
        room = self.slots // 2 - len(self.known)
        if len(fresh) > room:                                       # (no host work for entries that cannot be filed)
            self.stats['rejected'] += len(fresh) - max(room, 0)
            fresh = fresh[:max(room, 0)]
            if not fresh:
                return 0
        try:
            values = host_camera_matrices(K[fresh], E[fresh]).numpy().view(np.uint32)
            usable = [True] * len(fresh)
        except RuntimeError:                                      # a singular K in the batch: one by one
            values, usable = np.zeros((len(fresh), 12), dtype=np.uint32), []
            for j, i in enumerate(fresh):
                try:
                    values[j] = host_camera_matrices(K[i:i + 1], E[i:i + 1]).numpy().view(np.uint32)[0]
                    usable.append(True)
                except RuntimeError:
                    usable.append(False)                          # (the reference raises on it too)
        hashes = hash_words(words[fresh])
        added = 0
        changed = []
        for j, i in enumerate(fresh):
            slot = self._free_slot(int(hashes[j])) if usable[j] else -1
            if slot < 0 or len(self.known) >= self.slots // 2:
                self.stats['rejected'] += 1                       # stays on the device form
                continue
            self.host[slot, 0] = 1
            self.host[slot, 1:1 + KEY_WORDS] = words[i]
            self.host[slot, 1 + KEY_WORDS:1 + KEY_WORDS + 12] = values[j]
            self.known[words[i].tobytes()] = slot
            changed.append(slot)
            added += 1
        if added:
            self._upload(sorted(changed))
            self.stats['entries'] = len(self.known)
        return added

    def _upload(self, slots):
        """The changed rows only (144 B each), ordered on the current stream in front of the next lookup.  On the GPU they
        go through the pinned staging buffer with a non-blocking copy per run of consecutive slots; the host waits only if
        the previous batch of rows has not left the staging buffer yet."""
        table = self.table.view(self.slots, ENTRY_WORDS)
        host = self.host.view(np.int32)
        if self.device.type != 'cuda':
            idx = torch.as_tensor(slots, dtype=torch.long)
            table[idx] = torch.from_numpy(host[slots])
            return
        staging = self.staging.view(self.upload_rows, ENTRY_WORDS)
        for lo in range(0, len(slots), self.upload_rows):
            part = slots[lo:lo + self.upload_rows]
            if self.staging_event is not None:
                self.staging_event.synchronize()                    # (the previous batch has long left: set-up time, or a step ago)
            staging[:len(part)] = torch.from_numpy(host[part])
            idx = torch.as_tensor(part, dtype=torch.long).pin_memory().to(self.device, non_blocking=True)
            table.index_copy_(0, idx, staging[:len(part)].to(self.device, non_blocking=True))
            self.staging_event = torch.cuda.Event()
            self.staging_event.record()

    def _free_slot(self, h):
        for probe in range(PROBES):
            slot = (h + probe) & (self.slots - 1)
            if self.host[slot, 0] == 0:
                return slot
        return -1

    def absorb_miss_lists(self):
        """Entries for the calibrations earlier lookups missed, from whatever miss list has landed in pinned memory."""
        if len(self.known) >= self.slots // 2:                      # full (calibrations that never repeat, e.g. a randomised
            return 0                                                # K): everything new stays on the device form, at no host cost
        landed = self.landed.numpy().view(np.uint32)
        launch, count = int(landed[2]), int(landed[0])
        if launch == self.seen_launch or launch != int(landed[-1]):   # nothing new, or a copy in flight
            return 0
        self.seen_launch = launch
        if count == 0:
            return 0
        count = min(count, self.miss_capacity)
        rows = landed[MISS_HEADER:MISS_HEADER + count * MISS_WORDS].reshape(count, MISS_WORDS)[:, :KEY_WORDS].copy()
        f = rows.view(np.float32)
        K = torch.from_numpy(f[:, :9].reshape(count, 3, 3).copy())
        E = torch.zeros(count, 4, 4)
        E[:, :3, :] = torch.from_numpy(f[:, 9:21].reshape(count, 3, 4).copy())
        E[:, 3, 3] = 1.0
        self.stats['miss_lists'] += 1
        # A calibration is filed when a miss list shows it for the SECOND time: what repeats is a rig's calibration; the
        # reference's loader (fiery/data.py:172-209) builds sensor_to_lidar from each sample's own ego poses, so its
        # extrinsics never repeat - filing those would cost host LAPACK + an upload per step and fill the table with
        # one-off keys (and make a step's output depend on the calls before it).  One-off keys stay on the device form.
        again = []
        for i in range(count):
            kb = rows[i].tobytes()
            if kb in self.known:
                continue
            if kb in self.candidates:
                self.candidates.discard(kb)
                again.append(i)
            else:
                if len(self.candidates) >= self.max_candidates:
                    self.candidates.clear()                         # bounded memory; a repeating key comes back soon enough
                self.candidates.add(kb)
                self.stats['seen_once'] += 1
        if not again:
            return 0
        added = self.prime(K[again], E[again])
        self.stats['from_miss_lists'] += added
        return added

    # -- device side ------------------------------------------------------------------------------------
    def lookup(self, intrinsics, extrinsics):
        """(..., 3, 3), (..., 4, 4) on the device -> (N, 12) camera matrices.  Capturable; never waits for the device
        (outside a capture it first absorbs the miss lists that have landed)."""
        capturing = self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.absorb_miss_lists()
        K = intrinsics.reshape(-1, 3, 3).float().contiguous()
        E = extrinsics.reshape(-1, 4, 4).float().contiguous()
        cam = self.lib.camera_matrices_cached(K, E, self.table, self.slots, self.misses, self.miss_capacity)
        self.landed.copy_(self.misses, non_blocking=True)
        return cam
