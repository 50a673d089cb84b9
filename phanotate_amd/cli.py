"""phanotate.py-compatible command line (same flags as file_handling.get_args, file_handling.py:42-68).

    phanotate.py [-o OUT] [-f FORMAT] [-s atg:0.85,gtg:0.10,ttg:0.05] [-e tag,tga,taa] [-l 90] [-d] infile

All contigs of the input go through the GPU in batches (phanotate.py:40 loops over them one by one).
Inputs of more than one batch stream through phanotate_amd.pipeline.Pipeline: two batches in flight per GPU, and with --gpus N
the batches go round the first N GPUs of the node — one process, one host thread and two libphx contexts per GPU, no process group
(contigs never interact, phanotate.py:40,56).  `python -m torch.distributed.run --nproc-per-node N phanotate.py ...` still works:
contigs are then sharded across ranks (phanotate_amd.shard) and rank 0 writes the output in input order.
"""
import argparse
import os
import sys

from . import __version__
from .api import Annotator, make_params
from .writers import FORMATS, write

STATUS_TEXT = {-2: "letter outside the nucleotide alphabet (the reference raises KeyError)", -3: "contig shorter than 6 bases", -4: "a tRNA hit lies outside the contig",
               -6: "parallel edges are forbidden (graphs.py:74)", -7: "integer overflow in path sums", -8: "an open reading frame of more than 65535 codons", -9: "negative cycle"}


def is_valid_file(x):
    if not os.path.exists(x):
        raise argparse.ArgumentTypeError("{0} does not exist".format(x))
    return x


def get_args(argv=None):
    usage = "phanotate.py [-opt1, [-opt2, ...]] infile"
    p = argparse.ArgumentParser(description="PHANOTATE: A phage genome annotator (MI355X-native path)", formatter_class=argparse.RawTextHelpFormatter, usage=usage)
    p.add_argument("infile", type=is_valid_file, help="input file in fasta format")
    p.add_argument("-o", "--outfile", action="store", default=sys.stdout, type=argparse.FileType("w"), help="where to write the output [stdout]")
    p.add_argument("-f", "--format", help="Output the features in the specified format [tabular]", type=str, default="tabular", choices=FORMATS)
    p.add_argument("-s", "--start_codons", action="store", default="atg:0.85,gtg:0.10,ttg:0.05", dest="start_codons", help="comma separated list of start codons and frequency [atg:0.85,gtg:0.10,ttg:0.05]")
    p.add_argument("-e", "--stop_codons", action="store", default="tag,tga,taa", dest="stop_codons", help="comma separated list of stop codons [tag,tga,taa]")
    p.add_argument("-l", "--minlen", action="store", type=int, default=90, dest="min_orf_len", help="to store a variable")
    p.add_argument("-d", "--dump", action="store_true")
    p.add_argument("-V", "--version", action="version", version=__version__)
    p.add_argument("--device", type=int, default=None, help="GPU ordinal [LOCAL_RANK or 0]")
    p.add_argument("--gpus", type=int, default=1, help="spread the batches over the first N GPUs of the node, from this one process [1]")
    p.add_argument("--batch-bases", type=int, default=400_000_000, help="bases per GPU batch [4e8]")
    p.add_argument("--single-device-ranks", action="store_true", help=argparse.SUPPRESS)  # tests: every rank of a sharded launch on GPU `--device` (gloo-only group)
    return p.parse_args(argv)


def dump_edges(out, ann, i, seq=None, start_codons="atg:0.85,gtg:0.10,ttg:0.05"):
    """-d/--dump (phanotate.py:58,61): one line per edge, repr(src) TAB repr(dst) TAB str(weight*1000), in the reference's
    Graph.iteredges order, the weights as the reference's 28-digit Decimal values (phx_dump_text)."""
    out.write(ann.dump_text(i).decode())


def format_tabular(names, status, offsets, genes):
    """locus.py:39-56 for a run of contigs, as bytes (libphx's phx_format_tabular)."""
    import ctypes as C

    import numpy as np

    from . import _lib

    L = _lib.lib()
    n = len(names)
    enc = [x.encode() for x in names]
    arr = (C.c_char_p * max(n, 1))(*enc)
    status = np.ascontiguousarray(status, np.int32)
    offsets = np.ascontiguousarray(offsets, np.int64)
    genes = np.ascontiguousarray(genes)
    text, tlen = C.c_void_p(), C.c_int64()
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    rc = L.phx_format_tabular(n, arr, vp(genes), vp(offsets), vp(status), C.byref(text), C.byref(tlen))
    if rc:
        raise _lib.PhxError(rc, "phx_format_tabular")
    out = C.string_at(text.value, tlen.value)
    L.phx_free_text(text)
    return out


def main(argv=None):
    import json
    import time

    import numpy as np

    from .fasta import Fasta
    from .shard import partition, run_sharded_flat

    t_start = time.perf_counter()
    args = get_args(argv)
    device = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    # the HIP runtime and the context come up (~0.2 s) on a worker thread while this one reads the FASTA file
    import threading

    ctx_box = {}

    def make_context():
        try:
            ctx_box["ann"] = Annotator(make_params(args.start_codons, args.stop_codons, args.min_orf_len), device=device)
        except BaseException as e:  # re-raised on the main thread
            ctx_box["err"] = e

    ctx_thread = threading.Thread(target=make_context, daemon=True)
    threaded = int(os.environ.get("WORLD_SIZE", "1")) == 1  # (sharded runs bring torch.distributed up first, on this thread)
    if threaded:
        ctx_thread.start()

    def drop_context():  # every early exit: never leave the interpreter while the worker is still inside hipInit / phx_create
        if threaded:
            ctx_thread.join()
        if "ann" in ctx_box:
            ctx_box.pop("ann").close()

    try:
        fa = Fasta(args.infile)
    except BaseException:
        drop_context()
        raise
    if not len(fa) or not int(fa.lens.sum()):
        drop_context()
        sys.stdout.write("Error: no sequences found in infile\n")  # phanotate.py:33-35
        return 0
    t_parsed = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        from .shard import init_group

        dist, _ = init_group(rank, world, device=device, single_device=args.single_device_ranks)
    if world == 1:
        ctx_thread.join()
    else:
        make_context()
    if "err" in ctx_box:
        raise ctx_box["err"]
    ann = ctx_box["ann"]
    t_ctx = time.perf_counter()
    t_parts = {"upload_s": 0.0, "run_s": 0.0, "download_s": 0.0, "batches": 0}
    import shutil

    from .trna import find_trnas_many

    have_finder = bool(shutil.which("aragorn") or shutil.which("tRNAscan-SE"))

    def trnas_of(idx):  # functions.add_trnas per contig (functions.py:457-495); None: neither tool is installed
        if not have_finder:
            for _ in idx:
                sys.stderr.write("Warning: tRNAscan or Aragorn were not found, proceding without tRNA masking.\n")
            return None
        return find_trnas_many(fa.seq(int(i)) for i in idx)  # the finder processes of a batch run side by side

    if args.dump:  # the reference dumps the first contig's edges and exits (phanotate.py:58-61)
        ann.upload_raw(fa.ptrs[:1], fa.lens[:1], fa)
        ann.set_trnas(trnas_of([0]))
        ann.run()
        dump_edges(args.outfile, ann, 0, fa.seq(0), args.start_codons)
        ann.close()
        return 0

    n_total = len(fa)
    mine = list(range(n_total)) if world == 1 else partition(fa.lens.tolist(), world)[rank]

    def annotate_flat(idx):  # this rank's contigs, in batches of --batch-bases, straight from the C buffer of the FASTA reader
        idx = np.asarray(idx, np.int64)
        n_gpu = max(1, int(args.gpus)) if world == 1 else 1
        # batches: at most --batch-bases each; with several GPUs at least two per GPU, so that every lane has two in flight
        limit = args.batch_bases
        if n_gpu > 1 and len(idx):
            limit = max(1, min(limit, -(-int(fa.lens[idx].sum()) // (2 * n_gpu))))
        cuts, lo = [], 0
        while lo < len(idx) or not cuts:
            hi, size = lo, 0
            while hi < len(idx) and (hi == lo or size + int(fa.lens[idx[hi]]) <= limit):
                size += int(fa.lens[idx[hi]])
                hi += 1
            cuts.append((lo, hi))
            lo = hi
            if lo >= len(idx):
                break
        t_parts["batches"] = len(cuts)
        if len(cuts) == 1 and n_gpu == 1:
            lo, hi = cuts[0]
            t0 = time.perf_counter()
            ann.upload_raw(fa.ptrs[idx[lo:hi]], fa.lens[idx[lo:hi]], fa)
            ann.set_trnas(trnas_of(idx[lo:hi]))
            t1 = time.perf_counter()
            ann.run()
            t2 = time.perf_counter()
            parts = [ann.download_flat()]
            t3 = time.perf_counter()
            t_parts["upload_s"] += t1 - t0; t_parts["run_s"] += t2 - t1; t_parts["download_s"] += t3 - t2
        else:  # a stream of batches: two in flight per GPU, the batches round the GPUs (pipeline.Pipeline)
            from .pipeline import Pipeline

            t0 = time.perf_counter()
            pipe = Pipeline(ann.params, device=device, depth=2, devices=([device] + [d for d in range(n_gpu) if d != device][: n_gpu - 1]) if n_gpu > 1 else None, first=ann)
            t1 = time.perf_counter()
            gen = ((fa.ptrs[idx[lo:hi]], fa.lens[idx[lo:hi]], fa, (lambda a=lo, b=hi: trnas_of(idx[a:b]))) for lo, hi in cuts)
            try:
                parts = list(pipe.run(gen))
                t2 = time.perf_counter()
            finally:
                pipe.close()
            t_parts["contexts_s"] = t1 - t0; t_parts["pipeline_s"] = t2 - t1; t_parts["gpus"] = n_gpu
        if len(parts) == 1:
            return parts[0]
        st = np.concatenate([p[0] for p in parts])
        genes = np.concatenate([p[2] for p in parts])
        counts = np.concatenate([np.diff(p[1]) for p in parts])
        return st, np.concatenate([[0], np.cumsum(counts)]).astype(np.int64), genes

    merged = run_sharded_flat(mine, annotate_flat, rank, world, dist, mine=(mine, n_total))
    t_gpu = time.perf_counter()
    rc = 0
    t_fmt = t_gpu
    if rank == 0:
        status, offsets, genes = merged
        for i in np.nonzero(status < 0)[0]:
            sys.stderr.write("Error: contig %s: %s\n" % (fa.names[i], STATUS_TEXT.get(int(status[i]), "status %d" % int(status[i]))))
            rc = 1
        if args.format == "tabular":
            text = format_tabular(fa.names, status, offsets, genes)
            t_fmt = time.perf_counter()
            args.outfile.flush()
            if hasattr(args.outfile, "buffer"):
                args.outfile.buffer.write(text)
            else:
                args.outfile.write(text.decode())
        else:
            for i in range(n_total):
                if status[i] >= 0:
                    write(args.outfile, args.format, fa.names[i], fa.seq(i).decode(), genes[offsets[i] : offsets[i + 1]])
            t_fmt = time.perf_counter()
        args.outfile.flush()
    t_end = time.perf_counter()
    if os.environ.get("PHX_CLI_TIMING") and rank == 0:
        sys.stderr.write("PHX_CLI_TIMING " + json.dumps({"parse_s": round(t_parsed - t_start, 4), "gpu_s": round(t_gpu - t_parsed, 4), "format_s": round(t_fmt - t_gpu, 4),
                                                         "write_s": round(t_end - t_fmt, 4), "total_s": round(t_end - t_start, 4), "bases": int(fa.lens.sum()), "genes": int(len(merged[2])),
                                                         "gpu_parts": dict({k: round(v, 4) for k, v in t_parts.items()}, context_s=round(t_ctx - t_parsed, 4))}) + "\n")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
