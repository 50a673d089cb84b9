"""tRNA finders for the masking step (functions.add_trnas, functions.py:457-495): the reference writes the contig to a
temporary FASTA, runs `aragorn -t -w` and `tRNAscan-SE -B -q --brief` if they are installed, and turns their output into a list
of [start, stop] pairs (reversed for hits on the complement strand).  libphx takes that list (phx_set_trnas) and adds the
tRNA nodes and edges on the GPU.  Same commands, same parsing, same warning when neither tool exists."""
import ast
import subprocess
import sys
import tempfile


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


def _tool_output(cmd):
    """stdout of a finder.  The child is waited for and its stderr is discarded (the reference reads stdout of a Popen with stderr on
    an unread pipe: a tool that writes more than a pipe buffer of diagnostics would block forever, and nothing reaps the process)."""
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL).stdout


def find_trnas_many(seqs, workers=8):
    """find_trnas for the contigs of a batch, the finder processes of several contigs side by side (the reference runs them one
    contig at a time; per contig the calls are what find_trnas makes, in the same order).  -> list (one entry per contig) or None when
    no tool exists."""
    from concurrent.futures import ThreadPoolExecutor

    seqs = list(seqs)
    if not seqs:
        return []
    with ThreadPoolExecutor(max_workers=max(1, min(workers, len(seqs)))) as ex:
        hits = list(ex.map(lambda s: find_trnas(s, warn=False), seqs))
    if any(h is None for h in hits):
        return None
    return hits


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
            out2 = _tool_output(["aragorn", "-t", "-w", f.name])
            parse_aragorn(out2.decode(), trnas, seen)
        except Exception:
            pass
        try:
            out1 = _tool_output(["tRNAscan-SE", "-B", "-q", "--brief", f.name])
            parse_trnascan(out1.decode(), trnas, seen)
        except Exception:
            pass
    if out1 is None and out2 is None:
        if warn:
            sys.stderr.write("Warning: tRNAscan or Aragorn were not found, proceding without tRNA masking.\n")
        return None
    return trnas
