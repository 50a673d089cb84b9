#!/usr/bin/env python3
"""bench.py — Mbp/s annotated on synthetic 50 kb phage contigs (BASELINE.json metric).

A "step" is one pass of the whole hot path (libphx phx_run: features -> ORF scan -> scoring -> graph -> exact shortest
path -> genes) over one batch of contigs that is already resident in HBM: that is `value`.  Which number is the metric is
decided by the task contract, section (4) "Measurement": "`value` is whole-job throughput with inputs already resident in HBM
when the timed region starts (if the boundary hands over host buffers, note the PCIe-inclusive rate in DESIGN.md — it is never
`value`)".  The quantity SURVEY.md §8(d) defines — host ASCII contigs -> host gene lists, i.e. H2D + every kernel + the
certificate + D2H (+ the gather to rank 0 when N > 1) — is the PCIe-inclusive rate of that sentence: timed the same way (K steps,
barrier + synchronize on both sides, max over ranks) and reported beside it as `host_to_host`, for the batch and for config 5's
10 000 contigs (`strong_scaling_base.host_to_host`: the N = 1 point of the §8(d) curve).

Workloads (BASELINE.json configs):
  N = 1 (default)         config 4: 1000 synthetic 50 kb contigs (seeds 0..999) on one GPU
  N > 1                   config 5: 10 000 synthetic 50 kb contigs (seeds 0..9999) sharded per contig over the N ranks by
                          phanotate_amd.shard.partition (1250 per GPU at N = 8); no collective in the data path, rank 0
                          gathers the gene lists (shard.run_sharded_flat)
  --workload lambda | t4  configs 2-3: one contig (tests/golden/NC_001416.1 48.5 kb / NC_000866.1 169 kb) on one GPU

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import glob
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
GOLDEN = os.path.join(ROOT, "tests", "golden")
SINGLE = {"lambda": "NC_001416.1", "t4": "NC_000866.1"}


def algorithmic_bytes(sz):
    """B_algo = L/4 + 4 L + 64 N_orf + 32 E + 64 V for the whole batch."""
    return sz["L"] / 4.0 + 4.0 * sz["L"] + 64.0 * sz["n_orf"] + 32.0 * sz["n_edge"] + 64.0 * sz["n_node"]


STAGE_KERNEL = {"sssp": ("k_sssp_duo<0,", "k_sssp_wave<2, 0,"), "features": "k_features", "edges_fill": ("k_edges<true, false, true>", "k_edges<true, false, false>"), "edges_count": "k_edges<false, false, false>",
                "orf_stats": "k_orf_stats", "orf_emit": "k_orf<true,", "orf_count": "k_orf<false,", "nodes": "k_node_build", "score": "k_score",
                "inorder": "k_inorder<2,"}  # substrings of the kernel names as rocprofv3 prints them (several: the first that occurs — k_sssp_duo, or k_sssp_wave<2> under PHX_NO_DUO)


def stage_matches(stage, kernel_name):
    pats = STAGE_KERNEL.get(stage, ())
    return any(p in kernel_name for p in ((pats,) if isinstance(pats, str) else pats))


def pmc_traffic_committed(stage, contigs, length):
    """Fallback: HBM bytes per launch of the stage's kernel from the committed rocprofv3 PMC run (profiles/traffic.json: separate
    FETCH_SIZE / WRITE_SIZE passes of `bench.py`, gfx950 correction applied by tools/pmc_summary.py).  Only meaningful for the
    workload it was collected on."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if contigs != 1000 or length != 50000 or stage not in STAGE_KERNEL or not os.path.exists(path):
        return None
    try:
        d = json.load(open(path))
    except Exception:
        return None
    for k, v in d.items():
        if stage_matches(stage, k) and "hbm_bytes" in v:
            return int(v["hbm_bytes"])
    return None


def pmc_traffic_live(contigs, length, device):
    """HBM bytes per launch of every kernel of one step, measured now: two rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc
    WRITE_SIZE`, kernel trace only — the counters do not fit one pass) over a one-step run of this script on the same workload.
    FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes (MI355X_MICROARCH.md, HBM
    section), so bytes = (2 FETCH + WRITE) * 1024 — calibrated for streaming reads, applied to scattered ones as well.
    Returns ({kernel name: bytes per launch}, total per step) or None (no rocprofv3, or this process is itself being profiled)."""
    import csv
    import shutil

    exe = shutil.which("rocprofv3")
    if not exe or os.environ.get("ROCP_TOOL_LIBRARIES") or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None
    per = {}
    try:
        with tempfile.TemporaryDirectory() as td:
            env = dict(os.environ, TMPDIR=td, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(device)))
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(td, ctr)
                cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                       "--steps", "1", "--warmup", "0", "--no-extras", "--no-pipeline", "--contigs", str(contigs), "--length", str(length)]
                r = subprocess.run(cmd, cwd=td, env=env, capture_output=True, text=True, timeout=900)
                files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
                if r.returncode != 0 or not files:
                    return None
                full = {}
                for row in csv.DictReader(open(files[0])):
                    if row["Counter_Name"] == ctr and "rocclr" not in row["Kernel_Name"]:
                        # the longest dispatch of every kernel, among equals the last: the full-batch launch of a steady-state run
                        # (phx_upload also launches k_features piece by piece behind its copies)
                        dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                        if row["Kernel_Name"] not in full or dur >= full[row["Kernel_Name"]][0]:
                            full[row["Kernel_Name"]] = (dur, float(row["Counter_Value"]))
                for k, v in full.items():
                    per.setdefault(k, {})[ctr] = v[1]
    except Exception:
        return None
    out = {k: int((2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0) for k, v in per.items() if k.startswith("k_") or "k_" in k}
    return (out, sum(out.values())) if out else None


def read_golden_fasta(case):
    with gzip.open(os.path.join(GOLDEN, case + ".fasta.gz"), "rt") as f:
        lines = f.read().split("\n")
    return "".join(lines[1:]).encode()


def reference_python_rate():
    """The reference's own Python (get_orfs + get_graph), timed in the build container when the golden vectors were made
    (tests/golden/make_golden.py stores the seconds in every fixture): kbp/s per core on the synthetic 50 kb fixtures.
    Cross-machine: never a ratio against this box."""
    import numpy as np

    rates = []
    for fn in sorted(glob.glob(os.path.join(GOLDEN, "synth50k_*.npz"))):
        g = np.load(fn)
        if "ref_seconds" in g:
            rates.append(float(g["L"]) / float(np.sum(g["ref_seconds"])) / 1e3)
    if not rates:
        return None
    return {"kbp_s_per_core": round(sum(rates) / len(rates), 3), "min": round(min(rates), 3), "max": round(max(rates), 3), "contigs": len(rates),
            "what": "phanotate_modules.functions get_orfs + get_graph (no solver, no I/O), 1 core, build container (Xeon 2.1 GHz), from tests/golden/synth50k_*.npz"}


def cpu_baselines(seqs, L_, n_one, per_core):
    """The C oracle (oracle/phx_oracle.c: orc_run, all three stages) on a bounded sample of the same batch: one thread here, then one
    worker PROCESS per host core (oracle/cpu_rate.py, a subprocess that never touches the GPU runtime and forks its workers: threads of
    one process serialise on the address-space lock under orc_run's allocations — 256 threads scaled 10x in round 3)."""
    import ctypes as C

    from oracle import oracle

    lib, P = oracle.lib(), oracle.make_params()

    def one_contig(seq):
        r = oracle.Result()
        lib.orc_run(seq, len(seq), C.byref(P), 3, C.byref(r))
        st = r.status
        lib.orc_free(C.byref(r))
        return st

    n1 = max(1, min(n_one, len(seqs)))
    t0 = time.perf_counter()
    for i in range(n1):
        assert one_contig(seqs[i]) == 0
    t1 = time.perf_counter() - t0
    one = {"value": round(n1 * L_ / t1 / 1e6, 4), "unit": "Mbp/s", "cores": 1, "kind": "port",
           "sample": "first %d of the %d contigs of this batch, C oracle (oracle/phx_oracle.c), %.1f s" % (n1, len(seqs), t1)}
    cores = os.cpu_count() or 1
    nall = max(64, min(len(seqs), per_core * cores))
    allc = {"error": "oracle/cpu_rate.py did not run"}
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_rate.py"), "--contigs", str(nall), "--length", str(L_)],
                           capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and line:
            d = json.loads(line[-1])
            allc = {"value": d["value"], "unit": "Mbp/s", "cores": d["cores"], "kind": "port",
                    "scaling_over_one_core": round(d["value"] / one["value"], 1) if one["value"] else None,
                    "cpus_visible": d.get("cpus_visible"), "cgroup_cpu_quota": d.get("cgroup_cpu_quota"),
                    "sample": "first %d contigs of the series, one contig at a time per worker process, %d processes = the CPUs this container may use (%s visible, cgroup quota %s: more processes only take turns), %.2f s; one core inside that run: %.3f Mbp/s" % (d["contigs"], d["cores"], d.get("cpus_visible"), d.get("cgroup_cpu_quota"), d["seconds"], d["one_core_Mbp_s"])}
        else:
            allc = {"error": (r.stderr or r.stdout)[-300:]}
    except Exception as e:  # the baseline is a reported figure, never a reason to lose the line
        allc = {"error": repr(e)[:300]}
    return one, allc


def cli_end_to_end(seqs, device):
    """phanotate.py (the drop-in CLI) on a FASTA of these contigs: seconds for parsing, the GPU path and formatting."""
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "in.fasta")
        with open(fa, "wb") as f:
            for i, s in enumerate(seqs):
                f.write(b">contig%05d synthetic\n" % i)
                for k in range(0, len(s), 70):
                    f.write(s[k : k + 70] + b"\n")
        out = os.path.join(td, "out.tsv")
        env = dict(os.environ, PHX_CLI_TIMING="1")
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "phanotate.py"), "--device", str(device), "-o", out, fa], capture_output=True, text=True, env=env, timeout=1800)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": r.stderr[-400:]}
        timing = {}
        for line in r.stderr.splitlines():
            if line.startswith("PHX_CLI_TIMING "):
                timing = json.loads(line[len("PHX_CLI_TIMING "):])
        bases = sum(len(s) for s in seqs)
        timing.update({"contigs": len(seqs), "process_wall_s": round(wall, 3), "output_bytes": os.path.getsize(out),
                       "Mbp_s_process_wall": round(bases / wall / 1e6, 2)})
        return timing


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["synthetic", "lambda", "t4"], default="synthetic")
    ap.add_argument("--contigs", type=int, default=None, help="total synthetic contigs [1000 at N=1 (config 4), 10000 at N>1 (config 5)]")
    ap.add_argument("--length", type=int, default=50000)
    ap.add_argument("--cpu-contigs", type=int, default=192, help="bounded sample for the 1-core CPU baseline (rank 0, N=1)")
    ap.add_argument("--cpu-per-core", type=int, default=4, help="contigs per host core in the all-core CPU baseline")
    ap.add_argument("--cli-contigs", type=int, default=1000, help="contigs of the end-to-end phanotate.py run (0: skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-contig lines, the CPU baselines and the CLI run")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the two_batches_in_flight lines (profiling passes: every launch then belongs to the one-batch-at-a-time regions)")
    ap.add_argument("--smoke-single-device", action="store_true",
                    help="N > 1 ranks on ONE GPU: the sharded code path (partition, per-rank shards, gather of the flat arrays over the group's gloo half) where no multi-GPU node is at hand; only the barrier / reductions differ from a multi-GPU launch (gloo instead of RCCL); its numbers mean nothing")
    ap.add_argument("--force-dist", action="store_true", help="bring the process group up even with one rank (exercises the RCCL + gloo group of a multi-GPU launch on a 1-GPU box)")
    ap.add_argument("--dump-merged", default=None, help="rank 0 writes the merged (status, offsets, genes) of the host-to-host region to this .npz (tests)")
    ap.add_argument("--no-pipeline-host", action="store_true", help="skip the host-to-host region of two_batches_in_flight (a kernel trace then ends with the resident steps of the two contexts)")
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes for roofline.traffic (use profiles/traffic.json)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import phanotate_amd as pa
    from phanotate_amd.shard import partition, run_sharded_flat

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: WORLD_SIZE=%d but --gpus %d; launch with torch.distributed.run for N>1\n" % (world, args.gpus))
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible; libphx has no CPU path\n")
        sys.exit(3)
    if args.smoke_single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    red_dev = "cuda"
    if world > 1 or args.force_dist:
        # one group, two backends: barrier / reductions of CUDA tensors over RCCL, the result gather (host arrays) over gloo
        from phanotate_amd.shard import init_group

        os.environ.setdefault("MASTER_PORT", "29500")
        dist, red_dev = init_group(rank, world, device=local_rank, single_device=args.smoke_single_device)

    L_ = args.length
    if args.workload == "synthetic":
        n_total = args.contigs if args.contigs else (1000 if world == 1 else 10000)
        # config 5: the whole job is seeds 0..n_total-1; this rank's shard by the product's partitioner (equal lengths: the
        # partition only needs the lengths, so every rank derives it without generating the other ranks' contigs)
        mine = list(range(n_total)) if world == 1 else partition([L_] * n_total, world)[rank]
        seqs = [pa.synth_contig(i, L_) for i in mine]
        wl = ("batch of %d synthetic %d bp phage contigs on one GPU, resident in HBM (%s)" % (n_total, L_, "BASELINE config 4" if n_total == 1000 else "BASELINE config 5's job on one GPU: the N = 1 base of the strong-scaling curve" if n_total == 10000 else "non-standard size")) if world == 1 else (
            "%d synthetic %d bp phage contigs (seeds 0..%d) sharded per contig over %d GPUs by shard.partition, %d on rank 0 (BASELINE config 5)" % (n_total, L_, n_total - 1, world, len(mine)))
    else:
        if world != 1:
            sys.stderr.write("bench.py: a single contig is never split across GPUs (replicas only): run --workload %s with --gpus 1\n" % args.workload)
            sys.exit(2)
        n_total, mine = 1, [0]
        seqs = [read_golden_fasta(SINGLE[args.workload])]
        L_ = len(seqs[0])
        wl = "%s, one %d bp contig on one GPU (BASELINE config %d)" % (SINGLE[args.workload], L_, 2 if args.workload == "lambda" else 3)
    # two contexts on this GPU (their own streams; torch is only here for the process group and the barrier): the first one does all
    # the one-batch-at-a-time measurements, both take turns for the `two_batches_in_flight` lines.  (No third context: with more
    # streams than hardware queues the streams of two contexts alias and their kernels stop overlapping.)
    pipe = pa.Pipeline(device=local_rank, depth=2)
    ann = pipe.anns[0]

    def barrier():
        if dist is not None:
            if red_dev == "cuda":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()
        # (the context's stream is idle between calls: phx_run / phx_download block until their results are on the host)

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    status, offs, genes = ann.annotate_flat(seqs)  # sizes the context's buffers
    for _ in range(args.warmup):
        ann.run()
    # untimed pass with every stage bracketed by HIP events: the stage table, and which kernel dominates
    ann.set_profiling(True)
    ann.stage_ms(reset=True)
    for _ in range(3):
        ann.run()
    stages_all = ann.stage_ms(reset=True)
    kern = {k: v for k, v in stages_all.items() if k not in ("copies", "memset", "wave_plan") and v[1] > 0}  # (wave_plan: on a side stream beside edges_fill)
    ranked = sorted(kern, key=lambda k: -kern[k][0])
    dom, second = ranked[0], (ranked[1] if len(ranked) > 1 else ranked[0])  # (features and the shortest path are within a few per cent of each other)
    kernel_ms_per_step = sum(v[0] for v in kern.values()) / 3
    # ---- timed region A (`value`): K steps on the resident batch; only the dominant stage keeps its two HIP events (on the
    #      launch stream), the rest of the run is enqueued without any (a full set of stage events costs 3 % of the step)
    # (a lone contig's front end is ONE kernel, k_front, unless one of its stages is being timed: configs 2-3 time the dominant stage only)
    ann.set_profiling_stages([dom, second] if args.workload == "synthetic" else [dom])
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ann.run()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    timed = ann.stage_ms(reset=True)
    dom_total, dom_n = timed[dom]
    sec_total, sec_n = timed[second] if args.workload == "synthetic" else (stages_all[second][0], 3)
    ann.set_profiling(False)
    # the certificate (phx_certified: the proof that every gene list is what the reference's Decimal-derived integers give) is computed
    # on demand from the state a run leaves on the device: its cost on top of a run, and how many contigs it covers
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ann.run_async()  # (the certificate kernels behind the run on the stream, as every caller that goes on to download has them: no host round trip between the two)
        cert = ann.certified()
    dt_cert = max_over_ranks(time.perf_counter() - t0)
    n_uncert = int(sum_over_ranks(float((cert != 1).sum())))   # not proven on the device
    n_again = int(sum_over_ranks(float((cert == 2).sum())))     # ... and therefore solved again on the host, inside the library
    sz = ann.batch_sizes()
    bp_total = sum_over_ranks(float(sum(len(s) for s in seqs)))
    value = bp_total * args.steps / dt / 1e6

    # ---- timed region B: host ASCII -> host gene lists: upload, every kernel, download, gather to rank 0 ----
    # (the caller's side of the C-ABI — the contigs' addresses and lengths, phx_upload's arguments — is built once: a C caller has them)
    import ctypes as C_
    seqs_b = [s if isinstance(s, bytes) else bytes(s, "ascii") for s in seqs]
    h_ptrs = np.array([C_.cast(C_.c_char_p(s), C_.c_void_p).value for s in seqs_b], np.uint64)
    h_lens = np.array([len(s) for s in seqs_b], np.int64)

    def host_step():
        return run_sharded_flat(seqs, lambda _: ann.annotate_flat_raw(h_ptrs, h_lens, seqs_b), rank, world, dist, mine=(mine, n_total))

    host_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        merged = host_step()
    barrier()
    dt_host = max_over_ranks(time.perf_counter() - t0)

    # ---- the same two regions with two batches in flight (pipeline.Pipeline: two contexts alternating on this GPU): what a
    #      stream of batches — a job of many batches, the CLI on a large FASTA — moves at.  Extra lines: `value` stays the
    #      one-batch-at-a-time figure, the one the roofline of the dominant kernel is measured in ----
    dt_pipe = dt_pipe_cert = dt_pipe_host = None
    if not args.no_pipeline:
        for a2 in pipe.anns:
            a2.annotate_flat(seqs)
            for _ in range(max(3, args.warmup)):  # (the third run on a batch layout captures the graph the later ones launch)
                a2.run()
        # phx_run_async puts the certificate kernels behind the run (round 6) unless exactness is off: first the runs alone — the
        # region of `value` —, then with the certificate — the region of `value_with_certificate`
        for exact in (False, True):
            for a2 in pipe.anns:
                a2.set_exact(exact)
            barrier()
            t0 = time.perf_counter()
            for k in range(args.steps):
                pipe.anns[k % 2].run_async()  # (a context with a run in flight collects it before it starts the next)
            for a2 in pipe.anns:
                a2.wait()
            barrier()
            if exact:
                dt_pipe_cert = max_over_ranks(time.perf_counter() - t0)
            else:
                dt_pipe = max_over_ranks(time.perf_counter() - t0)
    if world == 1 and not args.no_pipeline and not args.no_pipeline_host:
        raw = (h_ptrs, h_lens, seqs_b)  # (the C-ABI's own arguments, as in the one-context region above: a C caller has them)
        for _ in pipe.run([raw, raw]):
            pass
        barrier()
        t0 = time.perf_counter()
        for _ in pipe.run([raw] * args.steps):
            pass
        barrier()
        dt_pipe_host = time.perf_counter() - t0

    if rank == 0:
        st_all, offs_all, genes_all = merged
        if args.dump_merged:
            np.savez(args.dump_merged, status=st_all, offsets=offs_all, genes=genes_all)
        # roofline.traffic: HBM bytes per launch from the PMC counters, measured now when rocprofv3 is at hand (two extra one-step
        # runs of this script under `rocprofv3 --pmc`, kernel trace only), else from the committed passes of the same command
        live = None
        if world == 1 and args.workload == "synthetic" and not args.no_extras and not args.no_traffic:
            live = pmc_traffic_live(len(seqs), L_, local_rank)

        def traffic_of(stage):
            if live is not None:
                hits = [v for k, v in live[0].items() if stage_matches(stage, k)]
                return hits[0] if hits else None
            return pmc_traffic_committed(stage, len(seqs), L_) if args.workload == "synthetic" else None

        dom_ms = dom_total / max(dom_n, 1)
        dom_traffic = traffic_of(dom)
        if live is not None and dom_traffic is None:
            # (BENCH_r04 lost this number to a renamed kernel: the stage -> kernel-name map no longer matched the trace)
            raise RuntimeError("bench.py: the live PMC pass ran but no kernel name contains %r (stage %s): STAGE_KERNEL is stale; kernels seen: %s"
                               % (STAGE_KERNEL.get(dom), dom, sorted(live[0])))
        balgo = algorithmic_bytes(sz)
        achieved = balgo / (dom_ms * 1e-3) / 1e9
        step_achieved = balgo / (kernel_ms_per_step * 1e-3) / 1e9
        kern_of = [int(ann.globals(i).sssp_kernel) for i in range(len(seqs))]
        solver_kernels = {"wavefront_tight": kern_of.count(2), "wavefront_roomy": kern_of.count(3), "workgroup": kern_of.count(1), "global": kern_of.count(0)}
        out = {
            "metric": "Mbp/s annotated (whole node) on 50 kb synthetic phage contigs" if args.workload == "synthetic" else "Mbp/s annotated, single contig",
            "value": round(value, 3),
            "unit": "Mbp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak" if world == 1 and n_total != 10000 else "strong",
            "vs_baseline": None,
            "dtype": "u8 bases / fp64 scores / int64 edge weights / int128 distances",
            "data": ("synthetic" if args.workload == "synthetic" else "reference test genome (tests/golden)") + (" [SMOKE: all ranks on one GPU, gloo]" if args.smoke_single_device else ""),
            "value_with_certificate": round(bp_total * args.steps / dt_cert / 1e6, 3),
            "value_with_certificate_is": "the same resident step followed by phx_certified (k_refine + k_certify: the proof that every gene list is the one the reference's Decimal-derived integers give): the rate that carries the bit-identity guarantee; `value` is phx_run alone",
            "value_is": "inputs resident in HBM when the timed region starts (task contract); host ASCII -> host gene lists is `host_to_host`",
            "config": {
                "workload": wl,
                "contigs_total": n_total,
                "contigs_rank0": len(seqs),
                "contig_length": L_,
                "sharding": "per-contig across %d GPU(s) (phanotate_amd.shard.partition), no collective in the data path" % world,
                "genes_called_total": int(len(genes_all)),
                "contigs_with_error_status": int((st_all < 0).sum()),
                "int_limbs": int(ann.globals(0).n_limbs),
                "solver_kernel_contig0": int(ann.globals(0).sssp_kernel),
                "solver_kernels": solver_kernels,
            },
            "certificate": {"ms_per_step_with_run": round(dt_cert / args.steps * 1e3, 4), "ms_on_top_of_run": round((dt_cert - dt) / args.steps * 1e3, 4), "contigs_not_certified_on_device": n_uncert,
                            "contigs_solved_again_on_host": n_again,
                            "what": "phx_run_async + phx_certified per step: k_refine evaluates the edges fp64 could not place between two integers once more in double-double (bounds on the reference's integers from the error of its own 28-digit chain), k_certify proves per contig, in exact integers, that the gene list is the one the reference's Decimal-derived integers give (phx_refine.inc, phx_certify.inc); a contig it cannot prove is solved again on the host in the reference's own arithmetic, inside the library (phx_exact.inc).  Computed on demand, so `value` does not contain it; `host_to_host` (phx_download_flat asks for it) does"},
            "host_to_host": {
                "value": round(bp_total * args.steps / dt_host / 1e6, 3),
                "unit": "Mbp/s",
                "ms_per_step": round(dt_host / args.steps * 1e3, 4),
                "what": "SURVEY.md §8(d): host ASCII contigs -> host gene lists (phx_upload H2D + phx_run + phx_download_flat D2H%s), %d timed steps, barrier + synchronize around them, max over ranks" % ("" if world == 1 else " + gather of the flat gene arrays to rank 0", args.steps),
            },
            "two_batches_in_flight": None if dt_pipe is None else dict({
                "value": round(bp_total * args.steps / dt_pipe / 1e6, 3),
                "unit": "Mbp/s",
                "ms_per_step": round(dt_pipe / args.steps * 1e3, 4),
                "what": "the timed region of `value` with two contexts on their own streams taking the steps in turn (phx_run_async / phx_wait): the shortest-path kernel of one step runs beside the throughput kernels of the next; every step is still one whole pass over the resident batch",
                "with_certificate": {"value": round(bp_total * args.steps / dt_pipe_cert / 1e6, 3), "unit": "Mbp/s", "ms_per_step": round(dt_pipe_cert / args.steps * 1e3, 4),
                                     "what": "the same with k_refine + k_certify behind every run on its context's stream (what phx_run_async enqueues when the downloads are to deliver the reference's genes)"},
            }, **({} if dt_pipe_host is None else {"host_to_host": {"value": round(bp_total * args.steps / dt_pipe_host / 1e6, 3), "unit": "Mbp/s", "ms_per_step": round(dt_pipe_host / args.steps * 1e3, 4),
                                                   "what": "host ASCII -> host gene lists for a stream of batches through pipeline.Pipeline: the upload of a batch overlaps the kernels of the one before"}})),
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": round(achieved, 3),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": dom_traffic,
                "frac_by_traffic": None if not dom_traffic else round(dom_traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                "frac_by_traffic_is": "the dominant kernel's HBM bytes from the PMC counters / its average duration / peak: what the kernel really moves, beside `frac` (algorithmic bytes of the whole step / the same duration)",
                "traffic_source": ("measured in this run: two `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, WRITE_SIZE) of a one-step run of this command; bytes = (2 FETCH + WRITE) KiB: the guide's gfx950 correction, which is calibrated on streaming reads and is applied here to scattered reads as well (uncalibrated for them)" if live is not None
                                   else "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; rocprofv3 not available to this process or --no-traffic)"),
                "traffic_step_total": None if live is None else live[1],
                "algorithmic_bytes_per_launch": int(balgo),
                "avg_launch_ms": round(dom_ms, 4),
                "step_frac": round(step_achieved / HBM_PEAK_GBS, 6),
                "step_achieved": round(step_achieved, 3),
                "step_kernel_ms": round(kernel_ms_per_step, 4),
                "step_frac_is": "SURVEY.md §8(d): sum of algorithmic bytes / sum of kernel time of one step (all stages, HIP events), / peak",
                "runner_up": {"kernel": second, "avg_launch_ms": round(sec_total / max(sec_n, 1), 4), "frac": round(balgo / (sec_total / max(sec_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                              "traffic": traffic_of(second),
                              "why": "the exact shortest path (k_sssp_duo; k_sssp_wave<2> beyond 1024 contigs per GPU) and the edge fill (k_edges<true>) take about as long as each other (0.40-0.43 ms per 1000 contigs): which of the two is longer changes from run to run; both are measured with HIP events in the timed region"},
            },
            "stage_ms_per_step": {k: round(v[0] / 3, 4) for k, v in stages_all.items() if v[1] > 0},
        }
        ref = reference_python_rate()
        if ref:
            out["reference_python"] = ref
        if world == 1 and not args.no_extras:
            if args.workload == "synthetic":
                single = {}
                for key, case in SINGLE.items():  # configs 2 and 3: one contig, resident, 100 runs
                    s1 = read_golden_fasta(case)
                    a1 = pa.Annotator(device=local_rank)
                    (st1, g1), = a1.annotate([s1])
                    # a run is under 2 ms and the GPU has been idle behind the rocprofv3 passes above: 60 untimed runs bring the clocks back up
                    # (BENCH_r03's Lambda figure, 0.87 ms against 0.69 here, was taken 3 runs after that idle stretch), 100 timed ones
                    for _ in range(60):
                        a1.run()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(100):
                        a1.run()
                    t1 = (time.perf_counter() - t0) / 100
                    gl = a1.globals(0)
                    single[key] = {"contig": case, "bp": len(s1), "ms_per_contig": round(t1 * 1e3, 4), "Mbp_s": round(len(s1) / t1 / 1e6, 2), "genes": int(len(g1)),
                                   "int_limbs": int(gl.n_limbs), "solver_kernel": int(gl.sssp_kernel), "status": int(st1),
                                   "runs_solved_in_segments": int(a1.seg_runs()), "contigs_solved_by_one_sweep_behind_them": int(a1.seg_fallbacks())}
                    a1.close()
                    # the same contig by ONE sweep (PHX_CREATE_NO_SEG): what the segments buy (phx_sssp_seg.inc: up to 32 wavefront pairs side by
                    # side in frames of their own, joined and proven; the results are bit-equal, tests/test_seg_gpu.py)
                    a1 = pa.Annotator(device=local_rank, flags=("no_seg",))
                    a1.annotate([s1])
                    for _ in range(60):
                        a1.run()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(100):
                        a1.run()
                    single[key]["ms_per_contig_one_sweep"] = round((time.perf_counter() - t0) / 100 * 1e3, 4)
                    a1.close()
                out["single_contig"] = single
            if args.workload == "synthetic" and n_total == 1000 and L_ == 50000:
                # BASELINE config 5's whole job (seeds 0..9999) on this one GPU: the N = 1 point of the strong-scaling curve
                # whose N > 1 points `bench.py --gpus N` measures (the same contigs, sharded by shard.partition)
                big = [pa.synth_contig(i, L_) for i in range(10000)]
                a5 = pa.Annotator(device=local_rank)
                st5, offs5, g5 = a5.annotate_flat(big)
                for _ in range(2):
                    a5.run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    a5.run()
                t5 = (time.perf_counter() - t0) / 3
                p5 = np.array([C_.cast(C_.c_char_p(s_), C_.c_void_p).value for s_ in big], np.uint64)  # the C caller's arguments of phx_upload, built once
                l5 = np.array([len(s_) for s_ in big], np.int64)
                a5.annotate_flat_raw(p5, l5, big)
                t0 = time.perf_counter()
                for _ in range(3):
                    a5.annotate_flat_raw(p5, l5, big)
                t5h = (time.perf_counter() - t0) / 3
                c5 = np.asarray(a5.certified())
                n_again = int((c5 == 2).sum())
                out["strong_scaling_base"] = {"contigs": 10000, "n_gpus": 1, "value": round(len(big) * L_ / t5 / 1e6, 3), "unit": "Mbp/s", "ms_per_step": round(t5 * 1e3, 3),
                                              "host_to_host": {"value": round(len(big) * L_ / t5h / 1e6, 3), "unit": "Mbp/s", "ms_per_step": round(t5h * 1e3, 3), "contigs_solved_again_on_host": n_again, "contigs_not_certified_on_device": int((c5 != 1).sum()),
                                                               "what": "SURVEY.md §8(d) at N = 1: upload + run + certificate + download of config 5's 10 000 contigs; a contig the certificate does not cover is solved again on the host in the reference's Decimal-derived integers inside phx_download_flat (C, worker threads)"},
                                              "genes_called_total": int(len(g5)), "contigs_with_error_status": int((st5 < 0).sum()),
                                              "what": "config 5's 10 000 contigs as one batch resident on one GPU (3 timed runs; = `bench.py --gpus 1 --contigs 10000`): divide the N > 1 lines' value by this for strong-scaling efficiency"}
                a5.close()
                del big
            if not args.no_cpu:
                one, allc = cpu_baselines(seqs, L_, args.cpu_contigs if args.workload == "synthetic" else 1, args.cpu_per_core)
                out["cpu_baseline"] = one
                out["cpu_baseline_all_cores"] = allc
            if args.cli_contigs > 0 and args.workload == "synthetic":
                out["cli_end_to_end"] = cli_end_to_end(seqs[: args.cli_contigs], local_rank)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    pipe.close()


if __name__ == "__main__":
    main()
