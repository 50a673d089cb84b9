"""BASELINE config 5 at full size on the production code path, on the one GPU a test box has: 8 ranks x 1250 contigs
(shard.partition of seeds 0..9999), every rank its own libphx contexts, the flat gene arrays gathered to rank 0 as fixed-dtype
CPU tensors (shard.gather_flat over the group's gloo half) — byte-equal to one process annotating the 10 000 contigs, and a
sample of them equal to the oracle.  Only the barrier / scalar reductions differ from an 8-GPU launch (gloo instead of RCCL,
because RCCL refuses two ranks on one device); the RCCL + gloo group itself is brought up once with a single rank."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu


def _torchrun(nproc, port, script, *args, timeout=1500):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(args)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def test_config5_eight_ranks_on_one_gpu_equal_one_process_and_the_oracle(tmp_path, oracle):
    import phanotate_amd as pa

    out = tmp_path / "merged.npz"
    r = _torchrun(8, 29551, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--smoke-single-device", "--no-extras", "--no-pipeline",
                  "--dump-merged", str(out))
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["contigs_total"] == 10000 and d["config"]["contigs_rank0"] == 1250
    assert d["config"]["contigs_with_error_status"] == 0
    m = np.load(out)
    st8, offs8, g8 = m["status"], m["offsets"], m["genes"]
    assert len(st8) == 10000 and len(offs8) == 10001 and d["config"]["genes_called_total"] == len(g8)
    # the same 10 000 contigs in one process, one batch
    seqs = [pa.synth_contig(i, 50000) for i in range(10000)]
    ann = pa.Annotator(device=0)
    st1, offs1, g1 = ann.annotate_flat(seqs)
    ann.close()
    assert st8.tobytes() == st1.tobytes()
    assert offs8.tobytes() == offs1.tobytes()
    assert g8.tobytes() == g1.tobytes()
    assert (st1 == 0).all() and (np.diff(offs1) > 20).all()
    for i in range(0, 10000, 97):  # ... and the oracle's genes for every 97th contig
        o = oracle.run(seqs[i])
        g = g8[offs8[i] : offs8[i + 1]]
        assert o["status"] == 0
        assert np.array_equal(g["left"], o["gene_left"]) and np.array_equal(g["right"], o["gene_right"]), i
        assert np.array_equal(g["strand"], o["gene_strand"].astype(np.int32)) and np.array_equal(g["frame"], o["gene_frame"].astype(np.int32)), i
        np.testing.assert_allclose(g["score"], o["gene_score"], rtol=1e-9)


FIRST_RUN_WORKER = """
import os, sys
sys.path.insert(0, %r)
import numpy as np
import phanotate_amd as pa
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
seqs = [pa.synth_contig(i, 50000) for i in range(rank, 8000, world)]
ann = pa.Annotator(device=0)
ann.upload(seqs); ann.run()
first = ann.download_flat(exact=False)
cert = ann.certified()
for i in range(0, len(seqs), 40):
    ann.edges(i)  # the tap checks the solver's integer edge records against the recomputed fp64 weights
ann.run()
second = ann.download_flat(exact=False)
assert all(a.tobytes() == b.tobytes() for a, b in zip(first, second)), "rank %%d: the first run of the context differs from its second" %% rank
assert (first[0] == 0).all() and (cert == 1).sum() >= len(seqs) - 2
print("FIRST_RUN_OK", rank, flush=True)
"""


def test_first_run_of_a_context_with_eight_processes_on_one_gpu(tmp_path):
    """A context's first run sizes its buffers between the kernels.  Until round 3 it cleared a device flag from the host in
    between (hipMemsetAsync of 4 bytes), which later kernels did not always see once several processes shared the GPU: about one
    process in ten came back with unreachable targets and shifted edge rows on its first batch — found by k_certify and the edge
    tap, invisible on a box of its own.  Eight processes, 1000 contigs each, twice."""
    script = tmp_path / "first_run.py"
    script.write_text(FIRST_RUN_WORKER % ROOT)
    for trial in range(2):
        r = _torchrun(8, 29556 + trial, str(script), timeout=900)
        assert r.stdout.count("FIRST_RUN_OK") == 8, r.stdout[-2000:] + r.stderr[-2000:]


def test_rccl_plus_gloo_group_comes_up_with_one_rank():
    """shard.init_group's production branch ("cpu:gloo,cuda:nccl", device bound): barrier and all_reduce on the GPU over RCCL, the
    gather over gloo — with the single rank a 1-GPU box allows."""
    r = _torchrun(1, 29552, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "1", "--warmup", "0", "--contigs", "40", "--length", "20000",
                  "--no-extras", "--no-pipeline", timeout=600)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["contigs_total"] == 40 and d["config"]["contigs_with_error_status"] == 0 and d["value"] > 0


def test_cli_sharded_over_two_ranks_equals_one_rank(tmp_path):
    """phanotate.py under torch.distributed.run (cli.py's N > 1 path: partition by length, per-rank batches, gather_flat, rank 0
    writes in input order): same bytes as the single-process run."""
    import phanotate_amd as pa

    fa = tmp_path / "in.fasta"
    with open(fa, "w") as f:
        for c in ("phiX174", "edge_L200", "synth6k_100", "edge_bridge", "NC_001416.1"):
            g, name, seq = load_golden(c)
            f.write(">%s\n%s\n" % (name, seq))
        for i in range(37):
            f.write(">s%d\n%s\n" % (i, pa.synth_contig(7000 + i, 3000 + 911 * i).decode()))
    one = subprocess.run([sys.executable, os.path.join(ROOT, "phanotate.py"), str(fa)], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr
    out = tmp_path / "two.tsv"
    _torchrun(2, 29553, os.path.join(ROOT, "phanotate.py"), "--single-device-ranks", "--device", "0", "-o", str(out), str(fa), timeout=600)
    assert out.read_text() == one.stdout and one.stdout.count("#id:") == 42


def test_one_process_spreads_batches_over_lanes(tmp_path):
    """SURVEY.md §8(e) as written: ONE process, one host thread + libphx contexts per GPU, results placed by index — no torchrun, no
    process group.  pipeline.Pipeline(devices=[0, 0]) (two lanes; on this box both on GPU 0) over a stream of different batches gives
    byte for byte what one Annotator gives batch by batch; and the CLI on a FASTA that splits into three batches (--batch-bases), with
    and without --gpus, writes what the single-batch run writes."""
    import numpy as np

    import phanotate_amd as pa

    rng = np.random.RandomState(5)
    batches = [[pa.synth_contig(100 * k + j, int(rng.randint(800, 30000))) for j in range(int(rng.randint(1, 40)))] for k in range(9)] + [[], [b"acgtnnacgx" * 30]]
    one = pa.Annotator()
    want = [one.annotate_flat(b) for b in batches]
    one.close()
    with pa.Pipeline(devices=[0, 0], depth=2) as pipe:
        assert len(pipe.lanes) == 2 and len(pipe.anns) == 4
        for rnd in range(2):  # the second round runs on warm contexts: everything asynchronous
            got = list(pipe.run(iter(batches)))
            assert len(got) == len(want)
            for (ws, wo, wg), (gs, go, gg) in zip(want, got):
                assert np.array_equal(ws, gs) and np.array_equal(wo, go) and wg.tobytes() == gg.tobytes()
    # the same below the C-ABI: phx_pool_annotate (a host thread + two contexts per lane inside the library), with tRNA hits
    flat = [s for b in batches for s in b]
    one = pa.Annotator()
    hits = [[(100, 180)] if len(s) > 400 else [] for s in flat]
    want_flat = one.annotate(flat, trnas=hits)
    one.close()
    with pa.Pool(devices=[0, 0]) as pool:
        for bb in (0, 150000):  # one batch per lane pair, and many small batches
            got = pool.annotate(flat, trnas=hits, batch_bases=bb)
            assert len(got) == len(want_flat)
            for (ws, wg), (gs, gg) in zip(want_flat, got):
                assert ws == gs and wg.tobytes() == gg.tobytes()
        assert pool.annotate([]) == []
    fa = tmp_path / "in.fasta"
    total = 0
    with open(fa, "w") as f:
        for i in range(45):
            s = pa.synth_contig(8000 + i, 2000 + 700 * i).decode()
            total += len(s)
            f.write(">c%d\n%s\n" % (i, s))
    single = _cli(fa)
    assert single.count("#id:") == 45
    env = dict(os.environ, PHX_CLI_TIMING="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "phanotate.py"), "--batch-bases", str(total // 3 + 1), str(fa)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout == single, r.stderr
    import json

    t = json.loads([l for l in r.stderr.splitlines() if l.startswith("PHX_CLI_TIMING ")][-1][len("PHX_CLI_TIMING "):])
    assert t["gpu_parts"]["batches"] >= 3 and "pipeline_s" in t["gpu_parts"]
    assert _cli("--gpus", "1", "--batch-bases", str(total // 5), fa) == single


def test_pool_deals_a_skewed_input_longest_first():
    """the contigs go to the lanes greedily, longest first, by the bases a lane already holds (phx_pool_annotate used to deal
    consecutive batches round the lanes: fine for equal lengths, lopsided for one T4 among short contigs).  T4 + 200 short contigs —
    some with tRNA hits, one with a bad letter, one too short — over two lanes on this box's one GPU: byte for byte what one context
    gives, in input order, for several batch sizes."""
    import phanotate_amd as pa

    _, _, t4 = load_golden("NC_000866.1")
    rng = np.random.RandomState(3)
    seqs = [pa.synth_contig(5000 + i, int(rng.randint(400, 6000))) for i in range(60)] + [t4.encode()] + \
           [pa.synth_contig(6000 + i, int(rng.randint(400, 6000))) for i in range(140)] + [b"acgtnnacgx" * 40, b"acg"]
    hits = [[(100, 180)] if (i % 7 == 0 and len(s) > 400) else [] for i, s in enumerate(seqs)]
    one = pa.Annotator()
    want = one.annotate(seqs, trnas=hits)
    one.close()
    assert want[60][0] == 0 and len(want[60][1]) > 250 and want[-2][0] < 0 and want[-1][0] < 0
    with pa.Pool(devices=[0, 0]) as pool:
        for bb in (0, 40000, 1):
            got = pool.annotate(seqs, trnas=hits, batch_bases=bb)
            assert len(got) == len(want)
            for k, ((ws, wg), (gs, gg)) in enumerate(zip(want, got)):
                assert ws == gs and wg.tobytes() == gg.tobytes(), (bb, k)


# ---- f-4: the other output formats, through the GPU path (README.md:45-54, 60-61, 67-68) ----
def _cli(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "phanotate.py")] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_cli_genbank_fna_faa_on_phix174_match_the_readme():
    fa = os.path.join(ROOT, "tests", "golden", "phiX174.fasta.gz")
    gb = _cli("-f", "genbank", fa).split("\n")
    # README.md:45-54
    assert gb[:10] == ["LOCUS       phiX174                 5386 bp ",
                       "FEATURES             Location/Qualifiers",
                       "     CDS             100..627",
                       "                     /note=score:-4.827981E+02",
                       "     CDS             687..1622",
                       "                     /note=score:-4.857517E+06",
                       "     CDS             1686..3227",
                       "                     /note=score:-3.785434E+10",
                       "     CDS             3224..3484",
                       "                     /note=score:-3.779878E+02"]
    assert "ORIGIN" in gb and gb[-2] == "//"
    g, name, seq = load_golden("phiX174")
    fna = _cli("-f", "fna", fa).split("\n")
    assert fna[0] == ">phiX174_CDS_[100..627] [note=score:-4.827981E+02]"  # README.md:60
    assert fna[1] == seq[99:627].lower() and fna[1].startswith("atgtttcagacttttatttctcgccataattcaaactttttttctgataag")  # README.md:61
    assert len(fna) == 2 * len(g["gene_left"]) + 1
    assert _cli("-f", "fasta", fa).split("\n") == fna  # phanotate.py:25-26
    faa = _cli("-f", "faa", fa).split("\n")
    assert faa[0] == ">phiX174_CDS_[100..627] [note=score:-4.827981E+02]"  # README.md:67
    assert faa[1] == "MFQTFISRHNSNFFSDKLVLTSVTPASSAPVLQTPKATSSTLYFDSLTVNAGNGGFLHCIQMDTSVNAANQVVSVGADIAFDADPKFFACLVRFESSSVPTTLPTAYDVYPLNGRHDGGYYTVKDCVTIDVLPRTPGNNVYVGFMVWSNFTATKCRGLVSLNQVIKEIICLQPLK*"  # README.md:68
    # every record is the gene of the golden path, reverse-strand ones reverse-complemented
    comp = str.maketrans("acgt", "tgca")
    for k in range(len(g["gene_left"])):
        s = seq[int(g["gene_left"][k]) - 1 : int(g["gene_right"][k])].lower()
        if g["gene_strand"][k] < 0:
            s = s.translate(comp)[::-1]
            assert fna[2 * k].startswith(">phiX174_CDS_[complement(%d..%d)]" % (g["gene_left"][k], g["gene_right"][k]))
        assert fna[2 * k + 1] == s
        assert len(faa[2 * k + 1]) == len(s) // 3 and faa[2 * k + 1].endswith("*")


def test_cli_formats_with_trna_features(tmp_path):
    """A contig whose path runs through a tRNA (fixture generated through the reference with a fake aragorn): the tRNA shows as a
    `tRNA` key in genbank (phanotate.py:75 passes left.gene as the feature type), and is left out of the tabular text
    (locus.py:42) and of fna / faa, which hold CDS records."""
    from conftest import golden_cases

    case = None
    for c in golden_cases():
        if c.startswith("trna_"):
            g, name, seq = load_golden(c)
            if not str(g["error"]) and "gene_frame" in g and (np.abs(g["gene_frame"]) == 4).any():
                case = c
                break
    assert case, "no tRNA fixture has a tRNA feature on its path"
    # a fake aragorn on PATH that prints the text the fixture was generated with (`aragorn -t -w` batch form, tests/golden/make_golden.py)
    bind = tmp_path / "bin"
    bind.mkdir()
    (tmp_path / "hits.txt").write_text(str(g["aragorn_text"]))
    (bind / "aragorn").write_text("#!/bin/sh\ncat %s\n" % (tmp_path / "hits.txt"))
    os.chmod(bind / "aragorn", 0o755)
    fa = tmp_path / "t.fa"
    fa.write_text(">%s\n%s\n" % (name, seq))
    env = dict(os.environ, PATH=str(bind) + os.pathsep + os.environ["PATH"])

    def cli(fmt):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "phanotate.py"), "-f", fmt, str(fa)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        return r.stdout

    assert cli("tabular") == str(g["tabular"])
    n_trna = int((np.abs(g["gene_frame"]) == 4).sum())
    n_cds = len(g["gene_left"]) - n_trna
    gb = cli("genbank")
    assert gb.count("\n     tRNA            ") == n_trna and gb.count("\n     CDS             ") == n_cds
    k = int(np.nonzero(np.abs(g["gene_frame"]) == 4)[0][0])
    loc = "%d..%d" % (g["gene_left"][k], g["gene_right"][k])
    assert ("     tRNA            " + (loc if g["gene_strand"][k] > 0 else "complement(%s)" % loc) + "\n") in gb
    for fmt in ("fna", "faa"):
        txt = cli(fmt)
        assert txt.count(">") == n_cds and "tRNA" not in txt and "[" + loc + "]" not in txt
