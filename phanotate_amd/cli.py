"""phanotate.py-compatible command line (same flags as file_handling.get_args, file_handling.py:42-68).

    phanotate.py [-o OUT] [-f FORMAT] [-s atg:0.85,gtg:0.10,ttg:0.05] [-e tag,tga,taa] [-l 90] [-d] infile

All contigs of the input go through the GPU in batches (phanotate.py:40 loops over them one by one).
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N phanotate.py ...`; contigs
are sharded across ranks (phanotate_amd.shard) and rank 0 writes the output in input order.
"""
import argparse
import os
import sys

from . import __version__
from .api import Annotator, make_params
from .fasta import read_fasta
from .shard import run_sharded
from .writers import FORMATS, write

STATUS_TEXT = {-2: "letter outside the nucleotide alphabet (the reference raises KeyError)", -3: "contig shorter than 6 bases",
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
    p.add_argument("--batch-bases", type=int, default=400_000_000, help="bases per GPU batch [4e8]")
    return p.parse_args(argv)


def dump_edges(out, ann, i):
    """-d/--dump (phanotate.py:58,61): one line per edge, repr(src) TAB repr(dst) TAB weight*1000, in the
    reference's Graph.iteredges order.  Weights are fp64 here, Decimal (28 digits) in the reference."""
    nd = ann.nodes(i)
    ed = ann.edges(i)
    tname = {0: "start", 1: "stop", 2: "source", 3: "target"}

    def rep(v):
        n = nd[v]
        gene = "CDS" if n["type"] < 2 else tname[int(n["type"])]
        return "Node(%r,%r,%r,%r)" % (gene, tname[int(n["type"])], int(n["frame"]), int(n["pos"]))

    ref = nd["refidx"]
    V = len(nd)
    keyed = []
    for e in ed:
        s, d = int(e["src"]), int(e["dst"])
        ts, td = int(nd[s]["type"]), int(nd[d]["type"])
        orf_edge = ts < 2 and td < 2 and nd[s]["frame"] == nd[d]["frame"] and ((nd[s]["frame"] > 0 and ts == 0 and td == 1) or (nd[s]["frame"] < 0 and ts == 1 and td == 0))
        if orf_edge:
            k = (0, int(ref[d]), 0)
        elif ts == 2 or td == 3:
            k = (3, int(ref[d]), 0)
        else:  # connector: the reference loops right node outer, left node inner (functions.py:360-366)
            l, r = (s, d) if nd[s]["pos"] < nd[d]["pos"] else (d, s)
            bridge = abs(int(nd[s]["pos"]) - int(nd[d]["pos"])) >= 500
            k = (1 if bridge else 2, int(ref[r]), int(ref[l]))
        keyed.append((int(ref[s]), k, s, d, float(e["w"])))
    keyed.sort(key=lambda t: (t[0], t[1]))
    for _, _, s, d, w in keyed:
        out.write("%s\t%s\t%s\n" % (rep(s), rep(d), repr(w * 1000)))


def main(argv=None):
    args = get_args(argv)
    records = read_fasta(args.infile)
    if not records or not any(s for _, s in records):
        sys.stdout.write("Error: no sequences found in infile\n")  # phanotate.py:33-35
        return 0
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(device)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    params = make_params(args.start_codons, args.stop_codons, args.min_orf_len)
    ann = Annotator(params, device=device)
    seqs = [s for _, s in records]
    if args.dump:  # the reference dumps the first contig's edges and exits (phanotate.py:58-61)
        ann.annotate(seqs[:1])
        dump_edges(args.outfile, ann, 0)
        return 0

    def annotate(batch):
        out, cur, size = [], [], 0
        for s in batch:
            if cur and size + len(s) > args.batch_bases:
                out.extend(ann.annotate(cur))
                cur, size = [], 0
            cur.append(s)
            size += len(s)
        if cur:
            out.extend(ann.annotate(cur))
        return out

    results = run_sharded(seqs, annotate, rank, world, dist)
    rc = 0
    if rank == 0:
        for (name, seq), (status, genes) in zip(records, results):
            if status < 0:
                sys.stderr.write("Error: contig %s: %s\n" % (name, STATUS_TEXT.get(status, "status %d" % status)))
                rc = 1
                continue
            write(args.outfile, args.format, name, seq, genes)
        args.outfile.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
