"""Feature-file fast path: `.npy` feature files -> test-mode tiles resident in HBM.

Replaces the reference's per-frame Python loop (feature_dataset.py:359-367: 512*S tensor indexings + torch.cat per
video) by ONE vectorised gather into a pinned host buffer and an asynchronous host->device copy on a side HIP
stream, double-buffered so the copy of video i+1 overlaps the head's kernels on video i (the head is HBM-bound
and would otherwise be host-starved at ~10^6 features/s).  Index semantics are exactly the reference's
(feature_index.py, pinned to the reference's tables)."""
from __future__ import annotations

from typing import Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

from . import feature_index as FI


class FeatureStream:
    """Iterates over videos: yields (features [1, ncrops, 512*S, D] on `device`, num_frames, segment_size, path)."""

    def __init__(self, paths: Sequence[str], num_segments: int = 32, seg_length: int = 16, stride: int = 1,
                 ncrops: int = 1, device: Optional[torch.device] = None, max_tiles: int = 64, readers: int = 4):
        self.paths = list(paths)
        self.N, self.L, self.stride, self.ncrops = num_segments, seg_length, stride, ncrops
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned = [None, None]
        self._copied = [None, None]      # per slot: event recorded after the last H2D copy that READ the pinned buffer
        self.max_tiles = max_tiles
        self.readers = max(1, int(readers))       # threads that fill the videos of one batched() group concurrently

    @staticmethod
    def _npy_header(fh):
        """(shape, fortran_order, dtype) of an open .npy file, positioned at its data (numpy.lib.format, versions 1-3)."""
        from numpy.lib import format as NF
        major, minor = NF.read_magic(fh)
        if (major, minor) == (1, 0):
            return NF.read_array_header_1_0(fh)
        if (major, minor) == (2, 0):
            return NF.read_array_header_2_0(fh)
        return NF._read_array_header(fh, version=(major, minor))

    def _slot_buffer(self, slot: int, need: int, D: int) -> torch.Tensor:
        # The pinned buffer of this slot was the source of an asynchronous copy two videos ago: that copy must have
        # finished reading it before the host overwrites it (the device-side wait_event in __iter__ orders the CONSUMER
        # stream only, not the host).
        if self._copied[slot] is not None:
            self._copied[slot].synchronize()
            self._copied[slot] = None
        buf = self._pinned[slot]
        if buf is None or buf.numel() < need:
            # pinning is slow (~1 GB/s): start at 16 tiles and grow by half so that a run re-pins a handful of times
            grown = 0 if buf is None else buf.numel() * 3 // 2
            buf = torch.empty(max(need, grown, self.ncrops * 16 * 512 * D), dtype=torch.float32).pin_memory()
            self._pinned[slot] = buf
        return buf

    def _host_tile(self, path: str, slot: int):
        # Fast path (one crop, stride 1, C-ordered float32 file -- the reference's feature files): the tile is the file's
        # rows in order followed by its first rows again (indices (s + i) % T, feature_dataset.py:362), so the file is READ
        # STRAIGHT INTO the pinned buffer (one copy out of the page cache, no intermediate array, no index gather) and the
        # wrap-around rows are copied inside it.
        if self.ncrops == 1 and self.stride == 1:
            with open(path, "rb") as fh:
                shape, fortran, dtype = self._npy_header(fh)
                if len(shape) == 2 and not fortran and dtype == np.dtype("<f4"):
                    T, D = shape
                    starts, S = FI.test_start_indices(T, self.N, self.L, self.stride)
                    rows = len(starts) * self.L
                    view = self._slot_buffer(slot, rows * D, D)[: rows * D].view(1, rows, D)
                    dst = view.numpy()[0]
                    got = fh.readinto(memoryview(dst[:T]).cast("B"))
                    if got != T * D * 4:
                        raise IOError(f"{path}: short read ({got} of {T * D * 4} bytes)")
                    for r in range(T, rows, T):
                        n = min(T, rows - r)
                        dst[r:r + n] = dst[:n]
                    return view, T, S
        arr = np.load(path, mmap_mode="r", allow_pickle=False)               # [T*ncrops, D] float32
        D = arr.shape[-1]
        T = arr.shape[0] // self.ncrops
        starts, S = FI.test_start_indices(T, self.N, self.L, self.stride)
        if self.stride == 1 and arr.dtype == np.float32 and arr.ndim == 2 and arr.flags.c_contiguous:
            # several crops, stride 1: crop c of the tile is rows c, c + ncrops, ... of the file in order, then its first rows
            # again -- one strided copy per crop straight out of the mapping into the pinned slot, no index gather
            rows = len(starts) * self.L
            need = self.ncrops * rows * D
            view = self._slot_buffer(slot, need, D)[:need].view(self.ncrops, rows, D)
            dst = view.numpy()
            src = arr.reshape(T, self.ncrops, D)
            for c in range(self.ncrops):
                np.copyto(dst[c, :T], src[:, c, :])
            for r in range(T, rows, T):
                n = min(T, rows - r)
                dst[:, r:r + n] = dst[:, :n]
            return view, T, S
        idx = FI.frame_index_table(starts, self.L, self.stride, T)
        rows = idx.shape[0]
        need = self.ncrops * rows * D
        view = self._slot_buffer(slot, need, D)[:need].view(self.ncrops, rows, D)
        src = np.asarray(arr).reshape(T, self.ncrops, D)
        np.take(src, idx, axis=0, out=view.numpy().transpose(1, 0, 2)) if self.ncrops == 1 else \
            view.numpy().__setitem__(slice(None), src[idx].transpose(1, 0, 2))
        return view, T, S

    # ---- several videos per batch (AnomalyCLIP.forward_test_many / AnomalyCLIPModule.score_videos)
    def _geometry(self, path: str):
        """(T, S, rows per crop, D) of a feature file from its header alone"""
        with open(path, "rb") as fh:
            shape, fortran, dtype = self._npy_header(fh)
        T = shape[0] // self.ncrops
        starts, S = FI.test_start_indices(T, self.N, self.L, self.stride)
        return T, S, len(starts) * self.L, shape[-1]

    def _fill(self, path: str, dst: np.ndarray, T: int, rows: int):
        """test-mode tile of one video into dst [ncrops, rows, D] (pinned memory): same index semantics as _host_tile"""
        D = dst.shape[-1]
        if self.stride == 1:
            if self.ncrops == 1:
                with open(path, "rb") as fh:
                    shape, fortran, dtype = self._npy_header(fh)
                    if len(shape) == 2 and not fortran and dtype == np.dtype("<f4"):
                        got = fh.readinto(memoryview(dst[0, :T]).cast("B"))
                        if got != T * D * 4:
                            raise IOError(f"{path}: short read ({got} of {T * D * 4} bytes)")
                        for r in range(T, rows, T):
                            n = min(T, rows - r)
                            dst[0, r:r + n] = dst[0, :n]
                        return
            arr = np.load(path, mmap_mode="r", allow_pickle=False)
            if arr.dtype == np.float32 and arr.ndim == 2 and arr.flags.c_contiguous:
                src = arr.reshape(T, self.ncrops, D)
                for c in range(self.ncrops):
                    np.copyto(dst[c, :T], src[:, c, :])
                for r in range(T, rows, T):
                    n = min(T, rows - r)
                    dst[:, r:r + n] = dst[:, :n]
                return
        arr = np.load(path, mmap_mode="r", allow_pickle=False)
        starts, _ = FI.test_start_indices(T, self.N, self.L, self.stride)
        idx = FI.frame_index_table(starts, self.L, self.stride, T)
        src = np.asarray(arr).reshape(T, self.ncrops, D)
        dst[:] = src[idx].transpose(1, 0, 2)

    def _host_group(self, paths: Sequence[str], slot: int):
        geo = [self._geometry(p) for p in paths]
        D = geo[0][3]
        need = sum(self.ncrops * g[2] * D for g in geo)
        flat = self._slot_buffer(slot, need, D)[:need]
        jobs, off = [], 0
        for p, (T, S, rows, _) in zip(paths, geo):
            n = self.ncrops * rows * D
            jobs.append((p, flat[off:off + n].view(self.ncrops, rows, D).numpy(), T, rows))
            off += n
        # the videos of a group land in disjoint slices of the pinned slot: filled by `self.readers` threads at once (one
        # thread copies out of the page cache at ~8-10 GB/s; file reads and large numpy copies release the GIL)
        if self.readers > 1 and len(jobs) > 1:
            list(self._fill_pool().map(lambda j: self._fill(*j), jobs))
        else:
            for j in jobs:
                self._fill(*j)
        return flat.view(-1, D), [(T, S, rows, p) for p, (T, S, rows, _) in zip(paths, geo)]

    def _fill_pool(self):
        pool = self.__dict__.get("_pool")
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor
            pool = self.__dict__["_pool"] = ThreadPoolExecutor(max_workers=self.readers)
        return pool

    def batched(self, videos: int = 8, max_tiles: int = 96, first: int = 2):
        """Iterates over GROUPS of consecutive videos: yields (features [sum_v ncrops * rows_v, D] on the device -- video after
        video, each crop-major in frame order --, [(num_frames, segment_size, rows_per_crop, path), ...]).  A group holds up
        to `videos` videos and `max_tiles` 512-frame tiles (per crop); one pinned slot, ONE host-to-device copy per group, the
        next group read by the reader thread meanwhile.  The FIRST group holds at most `first` videos: nothing overlaps its
        read + copy, so it is kept short (the consumer starts after ~1 ms instead of a full group's ~5 ms)."""
        from concurrent.futures import ThreadPoolExecutor
        groups, cur, tiles = [], [], 0
        for p in self.paths:
            S = self._geometry(p)[1]
            if cur and (len(cur) >= (videos if groups else max(1, min(first, videos))) or tiles + S > max_tiles):
                groups.append(cur)
                cur, tiles = [], 0
            cur.append(p)
            tiles += S
        if cur:
            groups.append(cur)
        # group i is handed to the consumer as soon as its copy is ENQUEUED (the consumer's stream waits for the copy on the
        # device); the reader thread fills the other slot with group i + 1 meanwhile, and that group's copy overlaps
        # whatever the consumer queued for group i
        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(self._host_group, groups[0], 0) if groups else None
            for i in range(len(groups)):
                slot = i & 1
                view, meta = fut.result()
                fut = pool.submit(self._host_group, groups[i + 1], (i + 1) & 1) if i + 1 < len(groups) else None
                with torch.cuda.stream(self._copy_stream):
                    dev = view.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                self._copied[slot] = ev
                torch.cuda.current_stream().wait_event(ev)
                dev.record_stream(torch.cuda.current_stream())
                yield dev, meta

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, int, int, str]]:
        # One reader thread runs a video ahead: file i + 1 is read into its pinned slot (file I/O and large copies release
        # the GIL) while this thread issues the copy of video i and the consumer launches its kernels.  Slot s is refilled
        # for video i + 2 only after the copy of video i has finished reading it (_slot_buffer).
        from concurrent.futures import ThreadPoolExecutor
        pending = None
        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(self._host_tile, self.paths[0], 0) if self.paths else None
            for i, path in enumerate(self.paths):
                slot = i & 1
                view, T, S = fut.result()
                fut = pool.submit(self._host_tile, self.paths[i + 1], (i + 1) & 1) if i + 1 < len(self.paths) else None
                with torch.cuda.stream(self._copy_stream):
                    dev = view.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                self._copied[slot] = ev
                if pending is not None:
                    yield pending
                torch.cuda.current_stream().wait_event(ev)          # consumer stream waits for this copy only
                dev.record_stream(torch.cuda.current_stream())
                pending = (dev.unsqueeze(0), T, S, path)
            if pending is not None:
                yield pending
