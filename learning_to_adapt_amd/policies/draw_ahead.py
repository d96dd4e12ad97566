"""Draw the planner's NEXT random numbers while the GPU is still busy with the current plan - without changing
a single bit of what the reference would compute.

The reference draws its candidates from NumPy's global generator at the top of every controller step
(``policies/mpc_controller.py:67-69,85,114``); the stream is a pure function of the generator state, so the draw
of step k + 1 is known as soon as step k has consumed its numbers.  A ``DrawAhead`` chain keeps a PRIVATE copy of
the state (``utils/fast_rng.State``) and a worker thread that produces the upcoming blocks from it (into pinned
staging, uploading them on a side stream) while the caller waits for the GPU.  The global generator is never
touched by the worker.  When the caller comes back it asks ``take(sig)``:

* the chain was built for the same request shape AND the global generator is still exactly in the state the next
  block started from (nobody else drew from ``np.random`` in between - compared word for word): the block is
  adopted and the global state is set to the block's end state, i.e. precisely where the reference's own draw
  would have left it;
* otherwise the chain is dropped, ``take`` returns ``None`` and the caller draws synchronously as before.

Either way the numbers, their order and the generator state seen by everybody else are those of the reference.
"""

import collections
import os
import threading
import time

from ..utils import fast_rng


class _Block(object):
    __slots__ = ("base", "end", "payload", "done", "error", "gen")

    def __init__(self, base, gen):
        self.base, self.end, self.payload, self.done, self.error, self.gen = base, None, None, False, None, gen


class DrawAhead(object):
    def __init__(self, depth=1):
        self.depth = int(depth)
        self.cv = threading.Condition()
        self.blocks = collections.deque()
        self.sig = None
        self.producer = None
        self.tail = None            # state after the last block handed to the worker
        self.gen = 0                # bumped on every flush: in-flight work of older generations is discarded
        self.slot = 0
        self.thread = None
        self.pid = None
        self.misses = 0             # consecutive failed takes (somebody else is drawing from np.random)
        self.hits = 0
        self.cooldown = 0
        self.words_only = False     # compare / store key + pos only (uniform stream; ~1 us instead of ~50)
        self.produce_s = 0.0        # time the worker spent producing blocks (diagnostics)
        self.produced = 0

    @property
    def n_slots(self):
        return self.depth + 2

    # ------------------------------------------------------------------ consumer side
    def _worker_alive(self):
        return self.pid == os.getpid() and self.thread is not None and self.thread.is_alive()

    def take(self, sig):
        """Payload of the next block if it is valid for the current global generator state, else None."""
        if self.pid is not None and self.pid != os.getpid():
            # forked child: the worker thread did not come along and the condition variable may have been held at fork
            # time - start from scratch (the caller draws synchronously and restarts the chain)
            self.cv = threading.Condition()
            self.blocks = collections.deque()
            self.sig = self.producer = self.tail = self.thread = None
            self.pid = None
            self.gen += 1
            return None
        with self.cv:
            if self.sig != sig or (self.producer is None and not self.blocks):
                self._flush()
                return None
            # the worker has been told to produce and is about to post the block; should it have died, fall back to the
            # synchronous draw instead of waiting for ever
            while not self.blocks or not self.blocks[0].done:
                if not self._worker_alive():
                    self._flush()
                    self.misses += 1
                    return None
                self.cv.wait(0.05)
            blk = self.blocks[0]
            # Word-level compare / store (~1 us instead of get_state / set_state) is exact whenever no cached Gaussian
            # is involved: with has_gauss == 0 on both sides of the block the words ARE the state (nobody can set the
            # flag without consuming words).  Uniform chains never touch the flag at all.
            fast = self.words_only or (blk.error is None and not blk.base.has_gauss.value and not blk.end.has_gauss.value)
            same = blk.error is None and (blk.base.same_words_as_global() if fast else blk.base.same_as_global())
            if not same:
                self._flush()
                self.misses += 1
                return None
            self.blocks.popleft()
            if fast:
                blk.end.words_to_global()
            else:
                blk.end.to_global()
            self.misses = 0
            self.hits += 1
            self.cv.notify_all()
            return blk.payload

    def active_for(self, sig):
        """Is the chain already producing blocks for this request shape?"""
        with self.cv:
            return self.sig == sig and self.producer is not None

    def start(self, sig, producer, depth=None, words_only=False):
        """(Re)start the chain at the CURRENT global state (call it right after a synchronous draw).  Backs off
        when takes keep failing (a consumer of ``np.random`` runs between controller steps)."""
        if not fast_rng.available("double"):
            return
        with self.cv:
            if self.sig == sig and self.producer is not None:
                return                              # already running for this request (its tail IS the global state)
            if self.misses >= 2:
                self.cooldown += 1
                if self.cooldown % 16 != 0:
                    return
            base = fast_rng.State.from_global()
            if base is None:
                return
            self._flush()
            if depth is not None:
                self.depth = int(depth)
            self.sig, self.producer, self.tail = sig, producer, base
            self.words_only = bool(words_only)
            self._ensure_thread()
            self.cv.notify_all()

    def stop(self):
        with self.cv:
            self._flush()

    def _flush(self):
        self.gen += 1
        self.blocks.clear()
        self.sig = None
        self.producer = None
        self.tail = None
        self.cv.notify_all()

    # ------------------------------------------------------------------ worker side
    def _ensure_thread(self):
        if self.thread is None or self.pid != os.getpid() or not self.thread.is_alive():
            self.pid = os.getpid()
            self.thread = threading.Thread(target=self._run, name="l2a-draw-ahead", daemon=True)
            self.thread.start()

    def _run(self):
        cv = self.cv                    # the condition this worker belongs to; a chain re-created after a fork has another
        while True:
            with cv:
                while self.cv is cv and (self.producer is None or len(self.blocks) >= self.depth):
                    cv.wait()
                if self.cv is not cv:
                    return
                gen, producer = self.gen, self.producer
                state = self.tail.copy()
                blk = _Block(self.tail, gen)
                self.blocks.append(blk)
                slot = self.slot
                self.slot = (self.slot + 1) % self.n_slots
            payload, error = None, None
            t0 = time.perf_counter()
            try:
                payload = producer(state, slot)
            except BaseException as exc:        # surfaces as a miss on the consumer side
                error = exc
            self.produce_s += time.perf_counter() - t0
            self.produced += 1
            with cv:
                if self.cv is not cv:
                    return
                if gen == self.gen:
                    blk.payload, blk.error, blk.end = payload, error, state
                    self.tail = state.copy()
                    if error is not None:
                        self.producer = None
                blk.done = True
                cv.notify_all()
