"""FASTA(.gz) reader for the CLI.  The reference delegates this to the external `genbank` package
(phanotate_modules/file.py:1-5); only what phanotate.py needs of it is provided: every record's name
(first token of the header, README.md:45 `LOCUS phiX174`) and sequence."""
import gzip


def read_fasta(path):
    """-> list of (name, sequence str); sequence case is preserved (the kernels lower-case)."""
    op = gzip.open if str(path).endswith(".gz") else open
    out, name, chunks = [], None, []
    with op(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(chunks)))
                tok = line[1:].split()
                name = tok[0] if tok else ""
                chunks = []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        out.append((name, "".join(chunks)))
    return out
