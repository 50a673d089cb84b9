"""Batches in flight: a stream of batches over two (or more) contexts of one GPU.

The shortest-path kernel runs one wavefront per contig and is bound by that wavefront's instruction stream: while it runs, most
of the device idles.  A second context on its own stream fills it with the next batch's upload and throughput kernels, so a
stream of batches moves ~20 % faster than one batch after the other (1000 x 50 kb contigs: 1.8 ms per batch instead of 2.3 with
the inputs resident, and the H2D copy of the next batch disappears behind the kernels of the current one).  Results come back in
the order the batches went in; every batch is computed exactly as Annotator.annotate_flat would compute it.
"""
from collections import deque

from .api import Annotator


class Pipeline:
    def __init__(self, params=None, device=0, depth=2):
        self.anns = [Annotator(params, device=device) for _ in range(max(1, int(depth)))]

    def close(self):
        for a in self.anns:
            a.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _load(self, a, batch):
        """batch: a list of sequences (bytes / str), or (ptrs, lens, keep[, trnas]) as Annotator.upload_raw takes them."""
        trnas = None
        if isinstance(batch, tuple):
            a.upload_raw(batch[0], batch[1], batch[2] if len(batch) > 2 else None)
            trnas = batch[3] if len(batch) > 3 else None
        else:
            a.upload(batch)
        a.set_trnas(trnas)

    def run(self, batches):
        """Generator over (status, offsets, genes) of every batch (Annotator.download_flat), in order.  While the caller consumes
        batch k, batch k+1 is already running."""
        depth = len(self.anns)
        busy = deque()
        for k, batch in enumerate(batches):
            if len(busy) == depth:
                yield busy.popleft().download_flat()  # waits for the oldest run; its context is the one batch k takes
            a = self.anns[k % depth]
            self._load(a, batch)
            a.run_async()
            busy.append(a)
        while busy:
            yield busy.popleft().download_flat()

    def annotate_flat(self, batches):
        return list(self.run(batches))
