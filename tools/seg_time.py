"""Lone contigs and small batches: resident step with segments (phx_sssp_seg.inc) and with one sweep per contig (PHX_CREATE_NO_SEG),
the solver stage by HIP events, for several margins.  python tools/seg_time.py [lambda|t4|synthN ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import phanotate_amd as pa
from conftest import load_golden


def batch(name):
    if name == "lambda":
        return [load_golden("NC_001416.1")[2]]
    if name == "t4":
        return [load_golden("NC_000866.1")[2]]
    if name == "phix":
        return [load_golden("phiX174")[2]]
    if name.startswith("short"):
        n = int(name[5:])
        return [pa.synth_contig(7 + i, 6000 + 500 * i) for i in range(n)]
    if name.startswith("len"):
        return [pa.synth_contig(5, int(name[3:]))]
    if name.startswith("synth"):
        n = int(name[5:])
        return [pa.synth_contig(i, 50000) for i in range(n)]
    raise SystemExit(name)


def timed(seqs, flags, env):
    for k, v in env.items():
        os.environ[k] = v
    a = pa.Annotator(flags=flags)
    for k in env:
        os.environ.pop(k)
    a.annotate_flat(seqs)
    for _ in range(5):
        a.run()
    import torch

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        a.run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    a.set_profiling(True)
    a.run(); a.stage_ms(reset=True)
    for _ in range(5):
        a.run()
    st = a.stage_ms(reset=True)
    a.set_profiling(False)
    segs = a.seg_runs()
    a.close()
    return ms, {k: v[0] / max(1, v[1]) * (v[1] / 5.0) for k, v in st.items() if v[1]}, segs


for name in sys.argv[1:] or ["lambda", "t4", "synth8", "synth64"]:
    seqs = batch(name)
    for label, flags, env in (("one sweep", ("no_seg",), {}), ("segments 6 kb", (), {}), ("seg, staged", ("no_fuse",), {}), ("segments 4 kb", (), {"PHX_SEG_MARGIN_BP": "4000"}), ("segments 9 kb", (), {"PHX_SEG_MARGIN_BP": "9000"})):
        ms, st, segs = timed(seqs, flags, env)
        print("%-8s %-14s step %.4f ms  sssp %.4f  wave_plan %.4f  edges_fill %.4f  inorder %.4f  seg_runs %d" % (name, label, ms, st.get("sssp", 0), st.get("wave_plan", 0), st.get("edges_fill", 0), st.get("inorder", 0), segs), flush=True)
