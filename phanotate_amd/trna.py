"""tRNA finders for the masking step (functions.add_trnas, functions.py:457-495): the reference writes the contig to a
temporary FASTA, runs `aragorn -t -w` and `tRNAscan-SE -B -q --brief` if they are installed, and turns their output into a list
of [start, stop] pairs (reversed for hits on the complement strand).  libphx takes that list (phx_set_trnas) and adds the
tRNA nodes and edges on the GPU.  Same commands, same parsing, same warning when neither tool exists."""
import ast
import sys
import tempfile
from subprocess import PIPE, Popen


def parse_aragorn(text, trnas, seen):
    """`aragorn -t -w` batch output: every line that neither starts with '>' nor ends with 'found' is a hit; its third column is
    [begin,end] or c[begin,end] (functions.py:471-480)."""
    for line in text.splitlines():
        if not line.startswith(">") and not line.endswith("found"):
            column = line.split()
            pair = list(ast.literal_eval(column[2].replace("c", "")))
            trnas.append(pair if "c" not in column[2] else pair[::-1])
            seen.extend(range(*pair))


def parse_trnascan(text, trnas, seen):
    """`tRNAscan-SE -B -q --brief`: tab-separated, begin and end in columns 3-4; a hit that overlaps an aragorn hit is
    dropped (functions.py:484-491)."""
    for line in text.splitlines():
        column = line.split("\t")
        a, b = map(int, column[2:4])
        if a < b and not set(seen) & set(range(a, b)):
            trnas.append([a, b])
        elif a > b and not set(seen) & set(range(b, a)):
            trnas.append([a, b])


def find_trnas(seq, warn=True):
    """-> list of [start, stop] as add_trnas holds it, or None when neither tool could be started (the reference then skips
    the masking with a warning, functions.py:493-495)."""
    if isinstance(seq, (bytes, bytearray)):
        seq = seq.decode()
    trnas, seen = [], []
    out1 = out2 = None
    with tempfile.NamedTemporaryFile(mode="wt", suffix=".fasta") as f:
        f.write(">temp\n")
        f.write(seq.lower())
        f.flush()
        try:  # a parse error ends this tool's contribution and keeps what was read so far, as the reference's bare except does
            out2 = Popen(["aragorn", "-t", "-w", f.name], stdout=PIPE, stdin=PIPE, stderr=PIPE).stdout.read()
            parse_aragorn(out2.decode(), trnas, seen)
        except Exception:
            pass
        try:
            out1 = Popen(["tRNAscan-SE", "-B", "-q", "--brief", f.name], stdout=PIPE, stdin=PIPE, stderr=PIPE).stdout.read()
            parse_trnascan(out1.decode(), trnas, seen)
        except Exception:
            pass
    if out1 is None and out2 is None:
        if warn:
            sys.stderr.write("Warning: tRNAscan or Aragorn were not found, proceding without tRNA masking.\n")
        return None
    return trnas
