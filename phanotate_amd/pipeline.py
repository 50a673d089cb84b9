"""Batches in flight: a stream of batches over the GPUs of one node, from ONE process.

Two things overlap.  On one GPU: the shortest-path kernel runs one wavefront per contig and is bound by that wavefront's instruction
stream — while it runs, most of the device idles.  A second context on its own stream fills it with the next batch's upload and
throughput kernels, so a stream of batches moves ~20 % faster than one batch after the other (1000 x 50 kb contigs: 1.8 ms per batch
instead of 2.3 with the inputs resident, and the H2D copy of the next batch disappears behind the kernels of the current one).
Across GPUs: contigs never interact (phanotate.py:40,56), so batch k simply goes to GPU k mod N — one host thread and `depth` libphx
contexts per GPU, no torchrun, no process group, no collective ("one host thread + one phx_ctx per GPU; results
re-ordered to input order on the host").  Results come back in the order the batches went in; every batch is computed exactly as
Annotator.annotate_flat would compute it.
"""
import queue
import threading
from collections import deque

from .api import Annotator


class _Lane:
    """One GPU: `depth` contexts taking turns, driven by one host thread."""

    def __init__(self, params, device, depth, first=None):
        self.device = device
        self.own = [Annotator(params, device=device) for _ in range(max(1, int(depth)) - (1 if first is not None else 0))]
        self.anns = ([first] if first is not None else []) + self.own
        self.turn = 0

    def close(self):
        for a in self.own:
            a.close()

    def next_ctx(self):
        a = self.anns[self.turn % len(self.anns)]
        self.turn += 1
        return a


def _load(a, batch):
    """batch: a list of sequences (bytes / str), or (ptrs, lens, keep[, trnas]) as Annotator.upload_raw takes them."""
    trnas = None
    if isinstance(batch, tuple):
        a.upload_raw(batch[0], batch[1], batch[2] if len(batch) > 2 else None)
        trnas = batch[3] if len(batch) > 3 else None
        if callable(trnas):  # the tRNA finder of a batch runs when the batch is loaded, beside the GPU work of the batches before
            trnas = trnas()
    else:
        a.upload(batch)
    a.set_trnas(trnas)


class Pipeline:
    def __init__(self, params=None, device=0, depth=2, devices=None, first=None):
        """devices: the GPUs to spread the batches over (default: [device]); an ordinal may repeat (two lanes on one GPU).
        first: an Annotator the caller already holds on the first device (with the same params): it becomes the first context of the
        first lane instead of a new one (the caller keeps ownership)."""
        devs = [int(d) for d in devices] if devices is not None else [int(device)]
        if not devs:
            raise ValueError("Pipeline needs at least one device")
        self.depth = max(1, int(depth))
        self.lanes = []
        try:
            for j, d in enumerate(devs):
                self.lanes.append(_Lane(params, d, self.depth, first if j == 0 else None))
        except BaseException:
            self.close()
            raise
        self.anns = [a for ln in self.lanes for a in ln.anns]

    def close(self):
        for ln in self.lanes:
            ln.close()
        self.lanes = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def run(self, batches):
        """Generator over (status, offsets, genes) of every batch (Annotator.download_flat), in order.  While the caller consumes
        batch k, the batches behind it are already running."""
        if len(self.lanes) == 1:
            yield from self._run_one(self.lanes[0], batches)
            return
        yield from self._run_many(batches)

    def _run_one(self, lane, batches):
        busy = deque()
        for batch in batches:
            if len(busy) == self.depth:
                yield busy.popleft().download_flat()  # waits for the oldest run; its context is the one the new batch takes
            a = lane.next_ctx()
            _load(a, batch)
            a.run_async()
            busy.append(a)
        while busy:
            yield busy.popleft().download_flat()

    def _run_many(self, batches):
        nl = len(self.lanes)
        inq = [queue.Queue(maxsize=self.depth + 1) for _ in range(nl)]
        done, cond, errors = {}, threading.Condition(), []

        def publish(k, res):
            with cond:
                done[k] = res
                cond.notify_all()

        def worker(j):
            lane, busy = self.lanes[j], deque()
            try:
                while True:
                    try:  # (polls: on the consumer's error path the sentinel may not fit a full queue — the stop flag ends the worker then)
                        item = inq[j].get(timeout=0.1)
                    except queue.Empty:
                        if stop.is_set():
                            break
                        continue
                    if item is None or stop.is_set():
                        break
                    k, batch = item
                    if len(busy) == self.depth:
                        kk, a = busy.popleft()
                        publish(kk, a.download_flat())
                    a = lane.next_ctx()
                    _load(a, batch)
                    a.run_async()
                    busy.append((k, a))
                while busy:
                    kk, a = busy.popleft()
                    publish(kk, a.download_flat())
            except BaseException as e:  # handed to the consumer
                with cond:
                    errors.append(e)
                    cond.notify_all()
                while True:  # keep the feeder from blocking on a full queue
                    try:
                        if inq[j].get(timeout=0.05) is None:
                            break
                    except queue.Empty:
                        if stop.is_set():
                            break

        stop = threading.Event()
        threads = [threading.Thread(target=worker, args=(j,), daemon=True) for j in range(nl)]
        for t in threads:
            t.start()
        fed = nxt = 0
        try:
            for batch in batches:
                inq[fed % nl].put((fed, batch))
                fed += 1
                while True:  # hand out what is ready, in order, without waiting
                    with cond:
                        if errors:
                            raise errors[0]
                        if nxt not in done:
                            break
                        res = done.pop(nxt)
                    yield res
                    nxt += 1
            for q in inq:
                q.put(None)
            while nxt < fed:
                with cond:
                    while nxt not in done and not errors:
                        cond.wait()
                    if errors:
                        raise errors[0]
                    res = done.pop(nxt)
                yield res
                nxt += 1
        finally:
            stop.set()
            for q in inq:
                try:
                    q.put_nowait(None)
                except queue.Full:
                    pass
            for t in threads:  # a lane's contexts may only be closed once its thread has really left them
                t.join()

    def annotate_flat(self, batches):
        return list(self.run(batches))
