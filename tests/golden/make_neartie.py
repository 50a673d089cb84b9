#!/usr/bin/env python3
"""A contig on which fp64-derived and Decimal-derived integers give DIFFERENT paths (VERDICT r3 #1c), found by IMPORTING the reference
(read-only, in this container only; what is committed is data: the FASTA, the -s flag and the reference's results).

The weight of a start codon is a user flag (-s atg:..,gtg:..,ttg:.., file_handling.py:51-62) that the reference keeps as a 28-digit
Decimal, while a double carries 16 digits of it.  The cost of a path is linear in that weight, so between two values of the gtg weight
that give different paths there is a value x* where the two paths tie; with x within ~1e-24 of x* the reference's integers
(trunc(Decimal(w) * 1000), edges.py:17-23) still decide by a few units, while the same flag read as a double is off by ~1e-16 x, i.e.
by ~1e7 units on an ORF edge of 1e23: the fp64-derived integers cannot know which side they are on.  This script bisects x on the
reference itself (functions.get_orfs once, Orf.score + functions.get_graph + the exact-integer Bellman-Ford of make_golden.py per
step) and writes the fixtures neartie_lo / neartie_hi: the same contig with the two adjacent 28-digit weights, whose reference paths
differ.  At least one of them is a contig whose fp64 path differs from its Decimal path.

Run:  python tests/golden/make_neartie.py
"""
import os
import sys
from decimal import Decimal as D

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference in place)

functions = mg.functions


def path_for(orfs, x):
    """The reference's graph and exact-integer path with the gtg weight set to the Decimal x (already divided by the maximum)."""
    orfs.start_codons["gtg"] = x
    for o in orfs.iter_orfs():
        o.start_codons["gtg"] = x
        o.score()  # orfs.py:122-127
    g = functions.get_graph(orfs)
    nodes = list(g.iternodes())
    edges = list(g.iteredges())
    path, dist, rounds, E = mg.bellman_ford(nodes, edges, nodes[-2], nodes[-1])
    return [repr(nodes[i]) for i in path], nodes, edges, E, path


def main():
    os.system("gcc -O2 -shared -fPIC -o %s/_synth.so %s/phanotate_amd/csrc/phx_synth.c -lm" % (HERE, mg.REPO))
    t0 = D("0.05") / D("0.85")
    for seed in range(500, 600):
        seq = mg.synth(seed, 24000)
        lo0, hi0 = D("0.10") / D("0.85"), D("0.985")
        locus = mg.StubLocus(seq, start_codons="atg:1,gtg:%s,ttg:%s" % (lo0, t0))
        orfs, graph, cap, _ = mg.run_reference(locus)
        big = [o for o in orfs.iter_orfs() if o.start_codon() == "gtg" and abs(o.weight) > D("1e20")]
        if not big:
            continue
        big.sort(key=lambda o: o.weight)

        def on_path(o, x):
            names, nodes, edges, E, path = path_for(orfs, x)
            a, b = (o.start, o.stop) if o.frame > 0 else (o.stop, o.start)  # the ORF edge runs start -> stop / stop -> start (functions.py:311-318)
            pos = [nodes[i].position for i in path]
            fr = [nodes[i].frame for i in path]
            return any(pos[k] == a and pos[k + 1] == b and fr[k] == o.frame and fr[k + 1] == o.frame for k in range(len(path) - 1)), names

        for o in big[:3]:
            lo, hi = lo0, hi0
            in_lo, p_lo = on_path(o, lo)
            in_hi, p_hi = on_path(o, hi)
            print("seed %d: gtg ORF %d..%d frame %d weight %.3g: on the path at the default weight %s, at %s %s" % (seed, o.start, o.stop, o.frame, float(o.weight), in_lo, hi, in_hi), flush=True)
            if in_lo or not in_hi:
                continue
            for it in range(140):
                mid = (lo + hi) / 2
                if mid == lo or mid == hi:
                    break
                if on_path(o, mid)[0]:
                    hi = mid
                else:
                    lo = mid
            p_lo, p_hi = on_path(o, lo)[1], on_path(o, hi)[1]
            assert p_lo != p_hi
            print("  breakpoint between gtg = %s and %s" % (lo, hi), flush=True)
            for tag, x in (("lo", lo), ("hi", hi)):
                name = "neartie_%s" % tag
                mg.save_fasta_gz(os.path.join(HERE, name + ".fasta.gz"), name, seq)
                mg.make_case(name, name, seq, HERE, start_codons="atg:1,gtg:%s,ttg:%s" % (x, t0))
            print("written: neartie_lo / neartie_hi from synth seed %d, L %d: the reference's paths differ in the gtg ORF %d..%d (weight %.3g at the breakpoint)" % (seed, len(seq), o.start, o.stop, float(o.weight)))
            return
    raise SystemExit("no contig found")


if __name__ == "__main__":
    main()
