"""FASTA(.gz) reader for the CLI.  The reference delegates this to the external `genbank` package
(phanotate_modules/file.py:1-5); only what phanotate.py needs of it is provided: every record's name
(first token of the header, README.md:45 `LOCUS phiX174`) and sequence.  The parsing is libphx's
(phx_fasta_read, csrc/phx_host.c): the sequences stay in one C buffer and go to phx_upload by pointer."""
import ctypes as C

import numpy as np

from . import _lib


class Fasta:
    """All records of a FASTA file.  names: list of str; ptrs / lens: numpy arrays (addresses into the C buffer, lengths)."""

    def __init__(self, path):
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = self.L.phx_fasta_read(str(path).encode(), C.byref(h))
        if rc:
            raise _lib.PhxError(rc, "phx_fasta_read(%s): %s" % (path, self.L.phx_strerror(rc).decode()))
        self.h = h
        n = self.n = int(self.L.phx_fasta_count(h))
        self.name_ptrs = np.zeros(max(n, 1), np.uint64)
        self.ptrs = np.zeros(max(n, 1), np.uint64)
        self.lens = np.zeros(max(n, 1), np.int64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self.L.phx_fasta_arrays(h, vp(self.name_ptrs), vp(self.ptrs), vp(self.lens))
        self.name_ptrs, self.ptrs, self.lens = self.name_ptrs[:n], self.ptrs[:n], self.lens[:n]
        self.names = [C.string_at(int(p)).decode() for p in self.name_ptrs]

    def __len__(self):
        return self.n

    def seq(self, i):
        """Sequence i as bytes (a copy)."""
        return C.string_at(int(self.ptrs[i]), int(self.lens[i]))

    def close(self):
        if getattr(self, "h", None):
            self.L.phx_fasta_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_fasta(path):
    """-> list of (name, sequence str); sequence case is preserved (the kernels lower-case)."""
    f = Fasta(path)
    out = [(f.names[i], f.seq(i).decode()) for i in range(len(f))]
    f.close()
    return out
