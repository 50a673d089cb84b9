"""Output formats of phanotate.py.  `tabular` is the in-tree writer of the reference
(phanotate_modules/locus.py:39-56) and is reproduced byte for byte; genbank / fna / faa follow the
excerpts pinned in the reference README (README.md:45-54, 60-61, 67-68) — the full writers live in the
external `genbank` package, which is absent."""

import numpy as np

_COMP = str.maketrans("acgtrykmbvdhswnACGTRYKMBVDHSWN", "tgcayrmkvbhdswnTGCAYRMKVBHDSWN")
_CODON = {}
for _i, _a in enumerate("tcag"):
    for _j, _b in enumerate("tcag"):
        for _k, _c in enumerate("tcag"):
            _CODON[_a + _b + _c] = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"[_i * 16 + _j * 4 + _k]


def gene_seq(seq, g):
    s = seq[g["left"] - 1 : g["right"]].lower()
    if g["strand"] < 0:
        s = s.translate(_COMP)[::-1]
    return s


def translate(nt):
    return "".join(_CODON.get(nt[i : i + 3], "X") for i in range(0, len(nt) - 2, 3))


def score_text(x):
    return "%E" % x  # phanotate.py:75-76


def write_tabular(out, name, genes):
    """locus.py:39-56."""
    out.write("#id:\t" + name + "\n")
    out.write("#START\tSTOP\tFRAME\tCONTIG\tSCORE\n")
    for g in genes:
        if abs(int(g["frame"])) == 4:  # a tRNA feature (functions.py:497-509): features(include=['CDS']) leaves it out (locus.py:42)
            continue
        left, right = int(g["left"]), int(g["right"])
        if g["strand"] < 0:
            left, right = right, left  # locus.py:44-46
        out.write("%d\t%d\t%s\t%s\t%s\n" % (left, right, chr(44 - int(g["strand"])), name, score_text(g["score"])))


def _location(g):
    loc = "%d..%d" % (g["left"], g["right"])
    return loc if g["strand"] > 0 else "complement(%s)" % loc


def write_genbank(out, name, seq, genes):
    out.write("LOCUS       %s %s bp \n" % (name.ljust(20), str(len(seq)).rjust(7)))
    out.write("FEATURES             Location/Qualifiers\n")
    for g in genes:
        out.write("     %s%s\n" % (("tRNA" if abs(int(g["frame"])) == 4 else "CDS").ljust(16), _location(g)))
        out.write("                     /note=score:%s\n" % score_text(g["score"]))
    out.write("ORIGIN\n")
    s = seq.lower()
    for i in range(0, len(s), 60):
        out.write("%9d %s\n" % (i + 1, " ".join(s[j : j + 10] for j in range(i, min(i + 60, len(s)), 10))))
    out.write("//\n")


def write_fna(out, name, seq, genes):
    for g in genes[np.abs(genes["frame"]) != 4] if len(genes) else genes:
        out.write(">%s_CDS_[%s] [note=score:%s]\n%s\n" % (name, _location(g), score_text(g["score"]), gene_seq(seq, g)))


def write_faa(out, name, seq, genes):
    for g in genes[np.abs(genes["frame"]) != 4] if len(genes) else genes:
        out.write(">%s_CDS_[%s] [note=score:%s]\n%s\n" % (name, _location(g), score_text(g["score"]), translate(gene_seq(seq, g))))


def write_gff(out, name, seq, genes):
    out.write("##gff-version 3\n##sequence-region %s 1 %d\n" % (name, len(seq)))
    for g in genes:
        out.write("%s\tPHANOTATE\t" % name + ("tRNA" if abs(int(g["frame"])) == 4 else "CDS") + "\t%d\t%d\t%s\t%s\t0\tnote=score:%s\n" % (g["left"], g["right"], score_text(g["score"]), chr(44 - int(g["strand"])), score_text(g["score"])))


FORMATS = ["tabular", "genbank", "fasta", "fna", "faa", "gff", "gff3"]


def write(out, fmt, name, seq, genes):
    if fmt == "tabular":
        write_tabular(out, name, genes)
    elif fmt == "genbank":
        write_genbank(out, name, seq, genes)
    elif fmt in ("fna", "fasta"):  # phanotate.py:25-26
        write_fna(out, name, seq, genes)
    elif fmt == "faa":
        write_faa(out, name, seq, genes)
    elif fmt in ("gff", "gff3"):
        write_gff(out, name, seq, genes)
    else:
        raise ValueError("unknown format " + fmt)
