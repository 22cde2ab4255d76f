"""CPU rehearsal of how the driver starts the multi-GPU bench: `python bench.py --gpus N` as a plain process.
bench.py must start its own `torch.distributed.run` child (never touching the GPU in the parent), the ranks must
meet, and the shard layout must be BASELINE.json configs[3]'s: ONE SF10 lineitem table cut by chunk into 8 canonical
octants, rank r owning octants [r·8/N, (r+1)·8/N) (strong scaling); `--scaling weak` gives each rank an SF10 shard."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _dry_run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-layout", *extra], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 alone prints, one JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4])
def test_plain_start_spawns_the_ranks_and_shards_one_sf10_table(world):
    doc = _dry_run("--gpus", str(world))
    assert doc["n_gpus"] == world and doc["scaling"] == "strong" and doc["workload"] == "q1_sf10"
    assert doc["total_rows"] == 59986052 and doc["n_chunks"] == 458 and doc["rows_covered"] == doc["total_rows"]
    ranks = sorted(doc["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == list(range(world))
    nxt_chunk, nxt_row = 0, 0
    for r in ranks:  # contiguous chunk ranges in rank order, octants [r·8/N, (r+1)·8/N)
        assert r["first_chunk"] == nxt_chunk and r["first_row"] == nxt_row
        assert r["octants"] == list(range(r["rank"] * 8 // world, (r["rank"] + 1) * 8 // world))
        assert r["first_chunk"] == doc["octant_chunk_begin"][r["octants"][0]]
        nxt_chunk += r["n_chunks"]
        nxt_row += r["local_rows"]
    assert nxt_chunk == 458


def test_weak_scaling_gives_every_rank_an_sf10_shard():
    doc = _dry_run("--gpus", "2", "--scaling", "weak")
    assert doc["scaling"] == "weak" and doc["total_rows"] == 2 * 59986052
    assert [r["local_rows"] for r in sorted(doc["ranks"], key=lambda r: r["rank"])] == [59986052, 59986052]


def test_one_rank_needs_no_launcher():
    doc = _dry_run()
    assert doc["n_gpus"] == 1 and doc["ranks"][0]["local_rows"] == 59986052 and doc["ranks"][0]["octants"] == list(range(8))


def test_a_rank_without_the_communicator_ends_every_rank_with_a_nonzero_code():
    """No fallback transport in a measurement: if ncclCommInitRank fails on ANY rank, all ranks agree (an all-reduce of the
    outcome) and leave with exit code 3 — rehearsed over gloo with rank 1 pretending; nothing is printed as a result."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-layout", "--gpus", "2", "--dry-run-comm-failure", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "rank 0: the library's RCCL communicator could not be created on another rank" in p.stderr
    assert "rank 1: the library's RCCL communicator could not be created (rehearsed failure)" in p.stderr
