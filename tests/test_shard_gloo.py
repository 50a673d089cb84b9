"""The N>1 path on CPU: world_size 2 over gloo.  The GPU compute is replaced by a checksum so that the
sharding / gather-in-input-order plumbing of phanotate_amd.shard is what is under test."""
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
    from phanotate_amd.shard import run_sharded, partition
    import phanotate_amd as pa
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    seqs = [pa.synth_contig(s, 500 + 137 * s) for s in range(23)]
    calls = []
    def fake_annotate(batch):
        calls.append(len(batch))
        return [(0, hashlib.md5(s).hexdigest()) for s in batch]
    out = run_sharded(seqs, fake_annotate, rank, world, dist)
    parts = partition([len(s) for s in seqs], world)
    assert calls == [len(parts[rank])]
    if rank == 0:
        assert out == [(0, hashlib.md5(s).hexdigest()) for s in seqs]
        print("SHARD_OK", [len(p) for p in parts])
    else:
        assert out is None
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
