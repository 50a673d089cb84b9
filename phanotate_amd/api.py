"""Host-side front end of libphx: parameter handling, batching, result unpacking.

Mirrors the way phanotate.py drives the path (phanotate.py:40-76): per contig
get_orfs -> get_graph -> shortest path -> features; here a whole batch of contigs goes
through the HIP kernels at once.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PhxError


def make_params(start_codons="atg:0.85,gtg:0.10,ttg:0.05", stop_codons="tag,tga,taa", minlen=90):
    """Same flag syntax and normalisation as file_handling.get_args (file_handling.py:51-66): phx_params_from_flags.  The weights also
    travel as the texts the user wrote (phx_params.start_w_text): the reference holds Decimal(text) / max."""
    p = _lib.Params()
    rc = _lib.lib().phx_params_from_flags(start_codons.encode(), stop_codons.encode(), int(minlen), C.byref(p))
    if rc:
        raise ValueError("start / stop codons %r / %r, minlen %r: libphx takes codon:weight pairs and codons of exactly 3 letters out of acgt, "
                         "at most %d of each, minlen >= 6 (%s)" % (start_codons, stop_codons, minlen, _lib.MAXC, _lib.lib().phx_strerror(rc).decode()))
    p.start_codons_text = start_codons  # the flag as given (dump.py's Decimal replay reads it)
    return p


def synth_contig(seed, L=50000):
    """Deterministic synthetic phage-like contig (host utility of libphx)."""
    buf = C.create_string_buffer(int(L))
    rc = _lib.lib().phx_synth_contig(int(seed), int(L), buf)
    if rc:
        raise PhxError(rc, "phx_synth_contig")
    return buf.raw


class Pool:
    """phx_pool: ONE call annotates a list of contigs over several GPUs from this process — a host thread and two contexts per device
    inside the library, batches round the devices, results in input order.  devices: GPU ordinals (one may repeat)."""

    def __init__(self, params=None, devices=(0,), flags=()):
        self.L = _lib.lib()
        self.params = params or make_params()
        fl = 0
        for f in flags:
            fl |= Annotator.FLAGS[f]
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self.L.phx_pool_create(C.byref(self.params), len(devices), devs, fl, C.byref(h))
        if rc:
            raise PhxError(rc, "%s (%s)" % (self.L.phx_strerror(rc).decode(), self.L.phx_last_error(None).decode()))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.phx_pool_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def annotate(self, seqs, trnas=None, batch_bases=0):
        """[(status, genes structured array)] per contig, in input order (phx_pool_annotate)."""
        seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(seqs)
        arr = (C.c_char_p * max(n, 1))(*seqs)
        lens = (C.c_int64 * max(n, 1))(*[len(s) for s in seqs])
        res = (_lib.Result * max(n, 1))()
        to = ta = tz = None
        if trnas is not None:
            offs = np.zeros(n + 1, np.int64)
            np.cumsum([len(t) for t in trnas], out=offs[1:])
            a = np.ascontiguousarray([h[0] for t in trnas for h in t] or [0], np.int32)
            z = np.ascontiguousarray([h[1] for t in trnas for h in t] or [0], np.int32)
            to, ta, tz = (x.ctypes.data_as(C.c_void_p) for x in (offs, a, z))
        rc = self.L.phx_pool_annotate(self.h, n, arr, lens, int(batch_bases), to, ta, tz, res)
        if rc:
            raise PhxError(rc, "phx_pool_annotate: %s (%s)" % (self.L.phx_strerror(rc).decode(), self.L.phx_pool_last_error(self.h).decode()))
        out = []
        for i in range(n):
            g = np.zeros(res[i].n_genes, _lib.GENE_DT)
            if res[i].n_genes:
                C.memmove(g.ctypes.data, res[i].genes, res[i].n_genes * _lib.GENE_DT.itemsize)
            out.append((int(res[i].status), g))
        self.L.phx_free_results(res, n)
        return out


class Annotator:
    """One libphx context = one (host thread, GPU).

    `stream`: None lets the context create its own (non-blocking) stream; an int is a raw hipStream_t used as given —
    0 is HIP's null stream, which is what `torch.cuda.current_stream().cuda_stream` returns by default, so that the
    context's work is ordered after the caller's on that stream (phx_create_ex, PHX_CREATE_USE_STREAM)."""

    FLAGS = {"no_graph": 2, "size_every_run": 4, "solver_global": 8, "solver_no_wave": 16, "no_certify": 32, "cert_tight": 64, "cert_wide": 128, "poison": 256, "one_stream": 512, "no_exact": 1024, "no_fuse": 2048, "no_duo": 4096, "no_seg": 8192}  # PHX_CREATE_* development / test switches

    def __init__(self, params=None, device=0, stream=None, flags=()):
        self.L = _lib.lib()
        self.params = params or make_params()
        h = C.c_void_p()
        fl = 0
        for f in flags:
            fl |= self.FLAGS[f]
        if stream is None:
            rc = self.L.phx_create_ex(C.byref(self.params), int(device), None, fl, C.byref(h))
        else:
            rc = self.L.phx_create_ex(C.byref(self.params), int(device), C.c_void_p(int(stream)), 1 | fl, C.byref(h))
        if rc:
            raise PhxError(rc, "%s (%s)" % (self.L.phx_strerror(rc).decode(), self.L.phx_last_error(None).decode()))
        self.h = h
        self.n = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.phx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise PhxError(rc, "%s: %s (%s)" % (what, self.L.phx_strerror(rc).decode(), self.L.phx_last_error(self.h).decode()))

    # ---- the path ----
    def upload(self, seqs):
        seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(seqs)
        arr = (C.c_char_p * n)(*seqs)
        lens = (C.c_int64 * n)(*[len(s) for s in seqs])
        self._keep = (seqs, arr, lens)
        self._chk(self.L.phx_upload(self.h, n, arr, lens), "phx_upload")
        self.n = n

    def upload_raw(self, ptrs, lens, keep=None):
        """The same from raw addresses: ptrs uint64[n] (host memory that stays valid for the call), lens int64[n]."""
        ptrs = np.ascontiguousarray(ptrs, np.uint64)
        lens = np.ascontiguousarray(lens, np.int64)
        self._keep = (ptrs, lens, keep)
        self._chk(self.L.phx_upload(self.h, len(lens), C.cast(ptrs.ctypes.data, C.POINTER(C.c_char_p)), C.cast(lens.ctypes.data, C.POINTER(C.c_int64))), "phx_upload")
        self.n = len(lens)

    def attach(self, dev_ptr, offsets):
        """Concatenated ASCII already in HBM (e.g. a torch uint8 tensor's data_ptr()); offsets has n+1 entries."""
        offs = (C.c_int64 * len(offsets))(*[int(x) for x in offsets])
        self._keep = (offs,)
        self._chk(self.L.phx_attach(self.h, len(offsets) - 1, C.c_void_p(int(dev_ptr)), offs), "phx_attach")
        self.n = len(offsets) - 1

    def set_trnas(self, trnas):
        """tRNA hits for the batch just uploaded: one list of (start, stop) per contig, as functions.add_trnas holds them
        (start > stop for a complement hit); None = no tRNA finder installed (functions.py:493-495)."""
        if trnas is None:
            self._chk(self.L.phx_set_trnas(self.h, None, None, None), "phx_set_trnas")
            return
        if len(trnas) != self.n:
            raise ValueError("one hit list per contig of the batch")
        offs = np.zeros(self.n + 1, np.int64)
        np.cumsum([len(t) for t in trnas], out=offs[1:])
        a = np.ascontiguousarray([h[0] for t in trnas for h in t] or [0], np.int32)
        z = np.ascontiguousarray([h[1] for t in trnas for h in t] or [0], np.int32)
        vp = lambda x: x.ctypes.data_as(C.c_void_p)
        self._chk(self.L.phx_set_trnas(self.h, vp(offs), vp(a), vp(z)), "phx_set_trnas")

    def run(self):
        self._chk(self.L.phx_run(self.h), "phx_run")

    def run_async(self):
        """Enqueue the run and return (phx_run_async); wait() — or any other call on this context — collects it.  With two
        contexts alternating, one batch's upload and kernels overlap the other's shortest-path kernel: see pipeline.Pipeline."""
        self._chk(self.L.phx_run_async(self.h), "phx_run_async")

    def wait(self):
        self._chk(self.L.phx_wait(self.h), "phx_wait")

    def set_exact(self, on):
        """phx_set_exact: off — the downloads hand out the device's own lists (no certificate, no host re-solve) and run_async does not
        put the certificate kernels behind the run."""
        self._chk(self.L.phx_set_exact(self.h, 1 if on else 0), "phx_set_exact")

    def download_flat(self, exact=True):
        """(status int32[n], offsets int64[n+1], genes structured array[total]): genes of contig i are genes[offsets[i]:offsets[i+1]]
        in path order (phx_download_flat: no per-contig allocation).  The library delivers the reference's genes: a contig the device
        could not certify against the reference's Decimal-derived integers (phx_certified; none is expected) is solved again on those
        integers inside phx_download_flat (csrc/phx_exact.inc); its indices are then in self.resolved.  exact=False: the device's
        own lists for every contig (phx_set_exact; tests and measurements)."""
        if not exact:
            self._chk(self.L.phx_set_exact(self.h, 0), "phx_set_exact")
            try:
                res = self._download_flat()
            finally:
                self._chk(self.L.phx_set_exact(self.h, 1), "phx_set_exact")
            self.resolved = []
            return res
        res = self._download_flat()
        self.resolved = [int(i) for i in np.nonzero(self.certified() == 2)[0]] if self.n else []
        return res

    def _download_flat(self):
        n = self.n
        offs = np.zeros(n + 1, np.int64)
        status = np.zeros(max(n, 1), np.int32)
        total = C.c_int64(0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        # one call when the gene array of the size the last batch needed (plus a margin) is large enough: phx_download_flat reports the
        # total and copies nothing if it is not
        guess = max(int(getattr(self, "_genes_seen", 0) * 1.25) + 64, 1)
        genes = np.empty(guess, _lib.GENE_DT)
        rc = self.L.phx_download_flat(self.h, vp(genes), len(genes), vp(offs), vp(status), C.byref(total))
        if rc == -1 and int(total.value) > len(genes):  # too small: the total is known now
            genes = np.empty(int(total.value), _lib.GENE_DT)
            rc = self.L.phx_download_flat(self.h, vp(genes), len(genes), vp(offs), vp(status), C.byref(total))
        self._chk(rc, "phx_download_flat")
        self._genes_seen = int(total.value)
        return status[:n], offs, genes[: int(total.value)]

    def certified(self):
        """int8[n]: 1 the contig's genes are proven on the device to be what the reference's Decimal-derived integers give
        (phx_certified), 2 not proven there and solved again on those integers on the host (inside the library), 0 neither,
        -1 the context runs without the certificate."""
        c = np.zeros(max(self.n, 1), np.int8)
        self._chk(self.L.phx_certified(self.h, c.ctypes.data_as(C.c_void_p)), "phx_certified")
        return c[: self.n]

    def download(self):
        """[(status, genes structured array)] per contig; the arrays are views into one flat buffer (phx_download_flat)."""
        status, offs, genes = self.download_flat()
        o = offs.tolist()
        st = status.tolist()
        return [(st[i], genes[o[i]:o[i + 1]]) for i in range(self.n)]

    def annotate(self, seqs, trnas=None):
        """[(status, genes structured array)] for every contig, in input order.  trnas: see set_trnas."""
        self.upload(seqs)
        if trnas is not None:
            self.set_trnas(trnas)
        self.run_async()  # (the certificate the download asks for goes behind the run on the stream; the download waits for both)
        return self.download()

    def annotate_flat(self, seqs):
        """The same as three flat arrays, see download_flat."""
        self.upload(seqs)
        self.run_async()
        return self.download_flat()

    def annotate_flat_raw(self, ptrs, lens, keep=None):
        """annotate_flat for a caller that holds the C-ABI's own arguments: the contigs' addresses (uint64[n]) and lengths (int64[n])."""
        self.upload_raw(ptrs, lens, keep)
        self.run_async()
        return self.download_flat()

    def dump_text(self, i):
        """-d/--dump of the reference for contig i of the batch last run (phx_dump_text): bytes, one line per edge."""
        text, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.phx_dump_text(self.h, int(i), C.byref(text), C.byref(n)), "phx_dump_text")
        out = C.string_at(text.value, n.value)
        self.L.phx_free_text(text)
        return out

    # ---- stage taps (parity tests) ----
    def globals(self, i):
        g = _lib.Globals()
        self._chk(self.L.phx_tap_globals(self.h, i, C.byref(g)), "phx_tap_globals")
        return g

    def positions(self, i):
        L = int(self.globals(i).L)
        a = [np.zeros(L, np.uint8) for _ in range(4)]
        self._chk(self.L.phx_tap_positions(self.h, i, *[x.ctypes.data_as(C.c_void_p) for x in a]), "phx_tap_positions")
        return dict(cls=a[0], gcc=a[1], binF=a[2], binR=a[3])

    def orfs(self, i):
        g = self.globals(i)
        a = np.zeros(max(g.n_orf, 0), _lib.ORF_DT)
        self._chk(self.L.phx_tap_orfs(self.h, i, a.ctypes.data_as(C.c_void_p)), "phx_tap_orfs")
        return a

    def nodes(self, i):
        g = self.globals(i)
        a = np.zeros(max(g.n_node, 0), _lib.NODE_DT)
        self._chk(self.L.phx_tap_nodes(self.h, i, a.ctypes.data_as(C.c_void_p)), "phx_tap_nodes")
        return a

    def edges(self, i):
        g = self.globals(i)
        a = np.zeros(max(g.n_edge, 0), _lib.EDGE_DT)
        self._chk(self.L.phx_tap_edges(self.h, i, a.ctypes.data_as(C.c_void_p)), "phx_tap_edges")
        return a

    def path(self, i):
        g = self.globals(i)
        p = np.zeros(max(g.n_node, 1), np.int32)
        n = C.c_int32()
        limbs = np.zeros(32, np.uint64)
        self._chk(self.L.phx_tap_path(self.h, i, p.ctypes.data_as(C.c_void_p), len(p), C.byref(n), limbs.ctypes.data_as(C.c_void_p), 32), "phx_tap_path")
        nl = max(g.n_limbs, 1)
        v = 0
        for k in range(nl):
            v |= int(limbs[k]) << (64 * k)
        if v >> (64 * nl - 1):
            v -= 1 << (64 * nl)
        return p[: n.value].copy(), v

    def dist(self, i):
        """Exact distance of every node from the source as python ints (None: unreached), device node order."""
        g = self.globals(i)
        nl, V = max(g.n_limbs, 1), max(g.n_node, 0)
        a = np.zeros((max(V, 1), nl), np.uint64)
        self._chk(self.L.phx_tap_dist(self.h, i, a.ctypes.data_as(C.c_void_p), a.size), "phx_tap_dist")
        out = []
        for v in range(V):
            x = 0
            for k in range(nl):
                x |= int(a[v, k]) << (64 * k)
            if x >> (64 * nl - 1):
                x -= 1 << (64 * nl)
            out.append(None if x >= 1 << (64 * nl - 3) else x)
        return out

    # ---- solver alone (fastpathz boundary) ----
    def solve(self, V, src, dst, weights, source, target, n_limbs=None):
        """Exact shortest path over integer weights (python ints).  Returns (path node ids, distance) or ([], None)."""
        E = len(src)
        mx = max([abs(int(w)) for w in weights] + [1])
        bits = mx.bit_length() + max(V, 2).bit_length() + 3
        if n_limbs is None:
            n_limbs = 2 if bits <= 128 else 4 if bits <= 256 else 8 if bits <= 512 else 17
        wl = np.zeros((max(E, 1), n_limbs), np.uint64)
        mask = (1 << 64) - 1
        for e, w in enumerate(weights):
            w = int(w) & ((1 << (64 * n_limbs)) - 1)
            for k in range(n_limbs):
                wl[e, k] = (w >> (64 * k)) & mask
        s = np.ascontiguousarray(src, np.int32)
        d = np.ascontiguousarray(dst, np.int32)
        path = np.zeros(V, np.int32)
        n = C.c_int32()
        dl = np.zeros(n_limbs, np.uint64)
        self._chk(self.L.phx_solve(self.h, V, E, s.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), wl.ctypes.data_as(C.c_void_p),
                                   n_limbs, source, target, path.ctypes.data_as(C.c_void_p), V, C.byref(n), dl.ctypes.data_as(C.c_void_p)), "phx_solve")
        if n.value == 0:
            return [], None
        v = 0
        for k in range(n_limbs):
            v |= int(dl[k]) << (64 * k)
        if v >> (64 * n_limbs - 1):
            v -= 1 << (64 * n_limbs)
        return path[: n.value].tolist(), v

    # ---- measurement ----
    def set_profiling(self, on=True):
        self._chk(self.L.phx_set_profiling(self.h, 1 if on else 0), "phx_set_profiling")

    def set_profiling_stages(self, names):
        """Bracket only the named stages with events (two events per run for one stage); [] switches profiling off."""
        idx = {self.L.phx_stage_name(k).decode(): k for k in range(_lib.N_STAGES)}
        mask = 0
        for nm in names:
            mask |= 1 << idx[nm]
        self._chk(self.L.phx_set_profiling_stages(self.h, mask), "phx_set_profiling_stages")

    def stage_ms(self, reset=True):
        ms = (C.c_float * _lib.N_STAGES)()
        nl = (C.c_int32 * _lib.N_STAGES)()
        self._chk(self.L.phx_get_stage_ms(self.h, ms, nl, 1 if reset else 0), "phx_get_stage_ms")
        return {self.L.phx_stage_name(k).decode(): (float(ms[k]), int(nl[k])) for k in range(_lib.N_STAGES)}

    def front_runs(self):
        """Runs whose front end was the fused launch of small batches (phx_front_runs; negative: it was switched off after a stall)."""
        return int(self.L.phx_front_runs(self.h))

    def seg_runs(self):
        """Runs whose 128-bit contigs were solved in segments side by side (phx_seg_runs)."""
        return int(self.L.phx_seg_runs(self.h))

    def seg_fallbacks(self):
        """Contigs, over the life of the context, whose segments could not be joined or proven and that one sweep solved in the same run."""
        return int(self.L.phx_seg_fallbacks(self.h))

    def seg_stats(self, i):
        """Per-segment records of contig i in the last run (phx_seg_stats): rows of [windows | done, status, first node, end node, ticks of 10 ns, phases, packs, step-backs]."""
        import numpy as np

        out = np.zeros((64, 8), np.int32)
        n = self.L.phx_seg_stats(self.h, int(i), out.ctypes.data_as(C.c_void_p), 64)
        return out[:n]

    def plan_timeouts(self):
        """Contigs, over the life of the context, whose shortest-path wavefront gave up waiting for the planner it was launched beside
        (include/phx.h: phx_plan_timeouts): 0 unless other contexts / processes kept the planner's wavefronts off the device."""
        return int(self.L.phx_plan_timeouts(self.h))

    def batch_sizes(self):
        v = [C.c_int64() for _ in range(4)]
        self._chk(self.L.phx_batch_sizes(self.h, *[C.byref(x) for x in v]), "phx_batch_sizes")
        return dict(L=v[0].value, n_orf=v[1].value, n_node=v[2].value, n_edge=v[3].value)
