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
