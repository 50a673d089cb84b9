"""The N>1 path on CPU: world_size 2 over gloo.  The GPU compute is replaced by a fake so that the
sharding / gather-in-input-order plumbing of phanotate_amd.shard (init_group, partition, gather_flat, merge_flat) is what is under test."""
import hashlib
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent(
    """
    import hashlib, os, sys
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from phanotate_amd.shard import init_group, partition
    import phanotate_amd as pa
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist, red_dev = init_group(rank, world, device=None)  # no GPU here: the gloo half of the production group
    assert red_dev == "cpu"
    seqs = [pa.synth_contig(s, 500 + 137 * s) for s in range(23)]
    parts = partition([len(s) for s in seqs], world)
    # the flat protocol (fixed-dtype CPU tensors, point to point: shard.gather_flat): bench.py's and the CLI's N > 1 path
    import numpy as np
    from phanotate_amd.shard import run_sharded_flat
    from phanotate_amd import _lib
    def fake_flat(batch):
        cnt = [len(s) %% 5 for s in batch]
        g = np.zeros(sum(cnt), _lib.GENE_DT)
        g["left"] = np.concatenate([np.full(c, len(s)) + np.arange(c) for s, c in zip(batch, cnt)]) if sum(cnt) else []
        return np.array([len(s) %% 3 - 1 for s in batch], np.int32), np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), g
    flat = run_sharded_flat(seqs, fake_flat, rank, world, dist)
    mine = parts[rank]
    flat2 = run_sharded_flat([seqs[i] for i in mine], fake_flat, rank, world, dist, mine=(mine, len(seqs)))
    if rank == 0:
        for st, offs, g in (flat, flat2):
            assert st.tolist() == [len(s) %% 3 - 1 for s in seqs]
            assert np.diff(offs).tolist() == [len(s) %% 5 for s in seqs]
            for i, s in enumerate(seqs):
                assert g["left"][offs[i]:offs[i + 1]].tolist() == [len(s) + k for k in range(len(s) %% 5)]
    else:
        assert flat is None and flat2 is None
    # ragged corner cases of the gather: a rank without contigs, contigs without genes
    few = seqs[:1]
    def fake_none(batch):
        return np.zeros(len(batch), np.int32), np.zeros(len(batch) + 1, np.int64), np.zeros(0, _lib.GENE_DT)
    e1 = run_sharded_flat(few, fake_flat, rank, world, dist)
    e2 = run_sharded_flat(seqs, fake_none, rank, world, dist)
    if rank == 0:
        assert len(e1[0]) == 1 and np.diff(e1[1]).tolist() == [len(few[0]) %% 5]
        assert len(e2[0]) == len(seqs) and int(e2[1][-1]) == 0 and len(e2[2]) == 0
        print("SHARD_OK", [len(p) for p in parts])
    dist.barrier()
    dist.destroy_process_group()
    """
)


def test_two_rank_sharding_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "SHARD_OK" in r.stdout
