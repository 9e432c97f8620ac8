"""Dataset decoding in worker PROCESSES with a shared-memory ring.

The reference feeds `save_depth` from a `DataLoader(..., num_workers=10)` (eval.py:305): worker
processes decode the PNGs, the samples come back pickled through pipes.  One cfg2 sample is 5 PNG
decodes + rescale + crop = ~130 ms of CPU work and 19.7 MB of float32 pixels; at the GPU's rate
(hundreds of maps per second) neither a thread pool (the numpy part holds the GIL: 50 maps/s) nor
pickled hand-over (one unpickle of 19.7 MB per sample in the parent) keeps up.  Here

  * `procs` worker processes are started with the `spawn` method -- fresh interpreters that never
    inherit a HIP context, so the pool can be created before or after the parent touched the GPU;
  * each worker runs `dataset[i]` and writes the image block straight into a slot of one
    `multiprocessing.shared_memory` segment; only the small rest of the sample (matrices, depth
    values, names) travels through a queue;
  * samples are dealt to the workers in contiguous chunks, so a worker's decoded-image cache
    (`EvalDataset(cache_images=...)`: every view is the reference view once and a source view of its
    neighbours several times) actually hits;
  * the parent hands samples out in dataset order; the `imgs` array of a sample is a view of its
    slot and stays valid until `release(sample)` (or the next-but-`slots` sample).
"""
from __future__ import annotations

import multiprocessing as mp
import os
import queue
import traceback
from multiprocessing import shared_memory

import numpy as np

_SENTINEL = None


def _worker(dataset, shm_name, slot_bytes, tasks, results):
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        while True:
            job = tasks.get()
            if job is _SENTINEL:
                return
            seq, idx, slot = job
            try:
                s = dict(dataset[idx])
                imgs = np.ascontiguousarray(s.pop("imgs"))
                if imgs.nbytes > slot_bytes:
                    raise ValueError(f"sample {idx}: {imgs.nbytes} bytes of images exceed the slot size {slot_bytes}")
                dst = np.ndarray(imgs.shape, imgs.dtype, buffer=shm.buf, offset=slot * slot_bytes)
                dst[...] = imgs
                results.put((seq, slot, imgs.shape, imgs.dtype.str, s, None))
            except BaseException:  # noqa: BLE001 - forwarded to the parent
                results.put((seq, slot, None, None, None, traceback.format_exc()))
    finally:
        shm.close()


class DecoderPool:
    """`for sample in pool.imap(indices): ...; pool.release(sample)` -- samples in order."""

    def __init__(self, dataset, procs: int = 8, chunk: int = 4, slots: int | None = None,
                 slot_bytes: int | None = None):
        if procs < 1:
            raise ValueError("procs must be >= 1")
        self.dataset, self.procs, self.chunk = dataset, procs, max(1, chunk)
        self.slots = slots or (2 * procs * self.chunk)
        if self.slots < procs * self.chunk:
            raise ValueError("need at least procs * chunk slots for in-order delivery")
        self._slot_bytes = slot_bytes
        self._ctx = mp.get_context("spawn")
        self._shm = None
        self._workers = []
        self._tasks = []
        self._results = None
        self._slot_of = {}

    # -- lifecycle ---------------------------------------------------------------------------
    def _start(self, first_index):
        if self._shm is not None:
            return
        if self._slot_bytes is None:   # learn the image block size from one item (decoded here, once)
            probe = np.asarray(self.dataset[first_index]["imgs"])
            self._slot_bytes = (probe.nbytes + 4095) // 4096 * 4096
        self._shm = shared_memory.SharedMemory(create=True, size=self.slots * self._slot_bytes)
        self._results = self._ctx.Queue()
        for _ in range(self.procs):
            q = self._ctx.Queue()
            p = self._ctx.Process(target=_worker, args=(self.dataset, self._shm.name, self._slot_bytes, q,
                                                        self._results), daemon=True)
            p.start()
            self._tasks.append(q)
            self._workers.append(p)

    def close(self):
        for q in self._tasks:
            try:
                q.put(_SENTINEL)
            except (OSError, ValueError):
                pass
        for p in self._workers:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self._workers, self._tasks = [], []
        if self._shm is not None:
            self._shm.close()
            try:
                self._shm.unlink()
            except FileNotFoundError:
                pass
            self._shm = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # best effort: never leave a segment behind in /dev/shm
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- iteration ---------------------------------------------------------------------------
    def release(self, sample):
        """The consumer is done with `sample["imgs"]` (e.g. after the copy to the device)."""
        slot = self._slot_of.pop(id(sample), None)
        if slot is not None:
            self._free.append(slot)

    def imap(self, indices):
        indices = list(indices)
        if not indices:
            return
        self._start(indices[0])
        self._free = list(range(self.slots))
        self._slot_of = {}
        n = len(indices)
        submitted = 0          # next position to hand to a worker
        pending = {}           # seq -> result
        handed = []            # samples yielded and not yet released, oldest first
        for pos in range(n):
            # keep the workers fed: chunk c of positions [c*chunk, (c+1)*chunk) goes to worker c % procs
            while submitted < n and submitted < pos + self.slots:
                if not self._free:
                    if handed:   # the consumer did not release explicitly: recycle the oldest hand-out
                        self.release(handed.pop(0))
                        continue
                    break
                slot = self._free.pop()
                self._tasks[(submitted // self.chunk) % self.procs].put((submitted, indices[submitted], slot))
                submitted += 1
            while pos not in pending:
                try:
                    seq, slot, shape, dtype, rest, err = self._results.get(timeout=1.0)
                except queue.Empty:
                    dead = [p.exitcode for p in self._workers if not p.is_alive()]
                    if dead:
                        raise RuntimeError(f"decoder worker died (exit codes {dead})") from None
                    continue
                if err is not None:
                    raise RuntimeError(f"dataset item {indices[seq]} failed in a decoder process:\n{err}")
                pending[seq] = (slot, shape, dtype, rest)
            slot, shape, dtype, rest = pending.pop(pos)
            sample = dict(rest)
            sample["imgs"] = np.ndarray(shape, np.dtype(dtype), buffer=self._shm.buf, offset=slot * self._slot_bytes)
            self._slot_of[id(sample)] = slot
            handed.append(sample)
            handed = [s for s in handed if id(s) in self._slot_of]
            yield sample


def default_procs() -> int:
    """Decoder processes for one GPU's share of the host (a 1-GPU box of this pool gets 16 cores)."""
    return max(1, min(16, (os.cpu_count() or 2) - 2))


# ---------------------------------------------------------------------------------------------------
# View-level pool: every image is decoded ONCE per run and shared between the samples that use it.
# ---------------------------------------------------------------------------------------------------
def _view_worker(dataset, shm_name, slot_bytes, tasks, results):
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        while True:
            job = tasks.get()
            if job is _SENTINEL:
                return
            path, slot = job
            try:
                arr, adjust = dataset.decode_view(path)
                if arr.nbytes > slot_bytes:
                    raise ValueError(f"{path}: {arr.nbytes} bytes exceed the slot size {slot_bytes}")
                np.ndarray(arr.shape, arr.dtype, buffer=shm.buf, offset=slot * slot_bytes)[...] = arr
                results.put((path, slot, arr.shape, arr.dtype.str, adjust, None))
            except BaseException:  # noqa: BLE001 - forwarded to the parent
                results.put((path, slot, None, None, None, traceback.format_exc()))
    finally:
        shm.close()


class ViewDecoderPool:
    """Samples in dataset order with `imgs` = LIST of per-view arrays [3,H,W] living in shared memory.

    In an eval run every view is the reference view of one sample and a source view of several
    neighbours (nviews - 1 on average), so decoding per SAMPLE repeats each PNG decode nviews times
    (DecoderPool above; a per-worker cache only recovers part of it).  Here the unit of work is the
    VIEW: the parent walks the samples `lookahead` ahead, asks the workers for every image that is not
    already in the shared-memory cache (`slots` entries, least recently used first out, entries of
    samples still ahead or handed out are pinned) and assembles a sample from the cached views and the
    (cheap, in-parent) camera parsing.  Needs a dataset with `view_plan`, `decode_view`, `assemble`
    (dataset_eval.EvalDataset); results are bit-identical to `dataset[i]`.
    """

    def __init__(self, dataset, procs: int = 8, slots: int = 96, lookahead: int = 8, slot_bytes: int | None = None):
        self.dataset, self.procs, self.slots, self.lookahead = dataset, max(1, procs), slots, max(1, lookahead)
        self._slot_bytes = slot_bytes
        self._ctx = mp.get_context("spawn")
        self._shm = None
        self._workers, self._tasks, self._results = [], None, None

    def _start(self, first_index):
        if self._shm is not None:
            return
        if self._slot_bytes is None:
            arr, _ = self.dataset.decode_view(self.dataset.view_plan(first_index)[1][0][0])
            self._slot_bytes = (arr.nbytes + 4095) // 4096 * 4096
        self._shm = shared_memory.SharedMemory(create=True, size=self.slots * self._slot_bytes)
        self._tasks, self._results = self._ctx.Queue(), self._ctx.Queue()
        for _ in range(self.procs):
            p = self._ctx.Process(target=_view_worker, args=(self.dataset, self._shm.name, self._slot_bytes,
                                                             self._tasks, self._results), daemon=True)
            p.start()
            self._workers.append(p)
        self._cache = {}        # path -> [slot, state ('flying' | 'ready'), shape, dtype, adjust, pins]
        self._free = list(range(self.slots))
        self._pins_of = {}      # id(sample) -> [paths]

    def close(self):
        for _ in self._workers:
            try:
                self._tasks.put(_SENTINEL)
            except (OSError, ValueError):
                pass
        for p in self._workers:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self._workers = []
        if self._shm is not None:
            try:
                self._shm.close()
            except BufferError:      # the consumer still holds views of the cache (or page-locked it)
                pass
            try:
                self._shm.unlink()
            except FileNotFoundError:
                pass
            self._shm = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def release(self, sample):
        for path in self._pins_of.pop(id(sample), ()):
            self._cache[path][5] -= 1

    def _request(self, path) -> bool:
        """Pin `path` for a sample ahead; start its decode if it is not cached.  False = no slot free."""
        ent = self._cache.get(path)
        if ent is not None:
            ent[5] += 1
            self._cache[path] = self._cache.pop(path)      # most recently used last
            return True
        if not self._free:
            victim = next((k for k, e in self._cache.items() if e[5] == 0 and e[1] == "ready"), None)
            if victim is None:
                return False
            self._free.append(self._cache.pop(victim)[0])
        slot = self._free.pop()
        self._cache[path] = [slot, "flying", None, None, None, 1]
        self._tasks.put((path, slot))
        return True

    def imap(self, indices):
        indices = list(indices)
        if not indices:
            return
        self._start(indices[0])
        plans = {}
        planned = 0            # samples [0, planned) have their views pinned / requested
        handed = []
        for pos in range(len(indices)):
            while planned < len(indices) and planned < pos + self.lookahead:
                plan = plans.get(planned) or self.dataset.view_plan(indices[planned])
                plans[planned] = plan
                paths = [p for p, _ in plan[1]]
                got = []
                for p in paths:
                    if not self._request(p):
                        break
                    got.append(p)
                if len(got) < len(paths):          # cache full of pinned views: undo and retry later
                    for p in got:
                        self._cache[p][5] -= 1
                    if planned == pos:
                        if handed:                 # recycle the oldest hand-out the consumer kept
                            self.release(handed.pop(0))
                            continue
                        raise RuntimeError("ViewDecoderPool: `slots` is smaller than one sample's views")
                    break
                planned += 1
            filename, views = plans.pop(pos)
            paths = [p for p, _ in views]
            while any(self._cache[p][1] != "ready" for p in paths):
                try:
                    path, slot, shape, dtype, adjust, err = self._results.get(timeout=1.0)
                except queue.Empty:
                    dead = [p.exitcode for p in self._workers if not p.is_alive()]
                    if dead:
                        raise RuntimeError(f"decoder worker died (exit codes {dead})") from None
                    continue
                if err is not None:
                    raise RuntimeError(f"decoding {path} failed in a decoder process:\n{err}")
                ent = self._cache[path]
                ent[1:5] = ["ready", shape, dtype, adjust]
            ents = [self._cache[p] for p in paths]
            sample = self.dataset.assemble(indices[pos], [e[4] for e in ents])
            sample["imgs"] = [np.ndarray(e[2], np.dtype(e[3]), buffer=self._shm.buf, offset=e[0] * self._slot_bytes)
                              for e in ents]
            self._pins_of[id(sample)] = paths
            handed.append(sample)
            handed = [s for s in handed if id(s) in self._pins_of]
            yield sample
