"""The CPU oracle on every host core: `python oracle/cpu_rate.py --contigs N --length L [--procs P]` prints one JSON line.

TEST INFRASTRUCTURE (bench.py's cpu_baseline leg): the C restatement of the reference (oracle/phx_oracle.c, orc_run: all three
stages) on the first N synthetic contigs of the benchmark series, one contig at a time per worker PROCESS — processes, not threads:
orc_run allocates and frees tens of MB per contig, and threads of one process serialise on the address-space lock in mmap / munmap /
page faults (256 threads scaled 10x in round 3).  The workers are forked before any GPU runtime is touched (this script never
initialises HIP: the contigs come from libphx's host-only generator).
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

SEQS = []
LIB = None
PAR = None


def _init():
    global LIB, PAR
    from oracle import oracle

    LIB, PAR = oracle.lib(), oracle.make_params()


def _one(i):
    from oracle import oracle

    seq = SEQS[i]
    r = oracle.Result()
    LIB.orc_run(seq, len(seq), C.byref(PAR), 3, C.byref(r))
    st = r.status
    LIB.orc_free(C.byref(r))
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=1024)
    ap.add_argument("--length", type=int, default=50000)
    ap.add_argument("--procs", type=int, default=0)
    args = ap.parse_args()
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None  # the container's CPU allowance (cgroup v2 cpu.max, v1 cfs quota): worker processes beyond it only take turns
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) if q > 0 else None
        except Exception:
            pass
    procs = args.procs or max(1, min(visible, int(quota + 0.5) if quota else visible))
    so = C.CDLL(os.path.join(ROOT, "phanotate_amd", "libphx.so"))
    so.phx_synth_contig.argtypes = [C.c_uint64, C.c_int64, C.c_char_p]
    for s in range(args.contigs):
        buf = C.create_string_buffer(args.length)
        assert so.phx_synth_contig(s, args.length, buf) == 0
        SEQS.append(buf.raw)
    _init()
    t0 = time.perf_counter()
    for i in range(min(8, args.contigs)):
        assert _one(i) == 0
    t_one = (time.perf_counter() - t0) / min(8, args.contigs)
    ctx = mp.get_context("fork")
    with ctx.Pool(procs, initializer=_init) as pool:
        pool.map(_one, range(min(procs, args.contigs)), chunksize=1)  # every worker up and warm
        t0 = time.perf_counter()
        st = pool.map(_one, range(args.contigs), chunksize=1)
        dt = time.perf_counter() - t0
    assert all(x == 0 for x in st)
    rate = args.contigs * args.length / dt / 1e6
    one = args.length / t_one / 1e6
    print(json.dumps({"value": round(rate, 4), "unit": "Mbp/s", "cores": procs, "kind": "port", "seconds": round(dt, 3), "contigs": args.contigs,
                      "one_core_Mbp_s": round(one, 4), "scaling_over_one_core": round(rate / one, 2), "cpus_visible": visible, "cgroup_cpu_quota": quota}))


if __name__ == "__main__":
    main()
