#!/usr/bin/env python3
"""Drop-in for the reference's phanotate.py (same flags, same output); the per-contig path runs in libphx on the GPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from phanotate_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
