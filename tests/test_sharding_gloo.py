"""CPU suite: the N > 1 path of bench.py -- files dealt round-robin over ranks (file i -> rank i mod N),
no data-path collective, aggregation with all_reduce/all_gather -- on 2 gloo ranks.  Each rank also drives the product
library on its own shard as far as a machine without a GPU allows: the host-side frame walk (mzd_content_size) sizes every
output, and the decode entry points refuse loudly (MZD_E_DEVICE) instead of falling back to a CPU decoder."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["MZD_ROOT"])
import numpy as np
import torch, torch.distributed as dist
import bench, corpus, oracle
import fuse_zstd_amd as mzd
mzd.build()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
nfiles = 6
for workload in ("cfg2", "cfg4lu"):
    kind, cfg_id, kind_mod, desc = bench.WORKLOADS[workload]
    sizes = [min(s, 40000) for s in bench.file_sizes(workload, nfiles, rank, world)]
    cp = corpus.build_corpus(kind, cfg_id, sizes, first_index=rank, stride=world, level=3, kind_mod=kind_mod, nthreads=2)
    # this rank's files are exactly the global files rank, rank+world, ...
    for i in range(nfiles):
        g = rank + i * world
        assert cp.raw_file(i).tobytes() == corpus.gen(kind, cfg_id, g, sizes[i]), (workload, rank, i)
        rc, out = oracle.decode(cp.comp_file(i).tobytes(), cap=sizes[i])
        assert rc == 0 and out == cp.raw_file(i).tobytes()
        assert mzd.content_size(cp.comp_file(i).tobytes()) == sizes[i]   # product: what the caller sizes dst with
    if torch.cuda.device_count() == 0:                                   # product: no GPU -> every rank's batch is refused, loudly
        try:
            mzd.init([0])
            raise SystemExit("mzd.init succeeded without a GPU")
        except mzd.MzdError as e:
            assert e.code == mzd.E_DEVICE
        jobs = mzd.api.make_jobs([cp.comp.ctypes.data + int(o) for o in cp.comp_offs], cp.comp_sizes, [0] * nfiles, [0] * nfiles)
        assert mzd.api.lib().mzd_decode_batch(jobs, nfiles) == mzd.E_DEVICE
    mine = torch.tensor([rank + i * world for i in range(nfiles)], dtype=torch.int64)
    allidx = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allidx, mine)
    flat = sorted(int(x) for t in allidx for x in t)
    assert flat == list(range(nfiles * world)), flat          # disjoint and complete
    tot = torch.tensor([float(cp.raw_sizes.sum())], dtype=torch.float64)
    dist.all_reduce(tot)                                        # the only aggregation bench.py needs
    elapsed = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)             # MAX over ranks, as the bench contract says
    assert float(elapsed) == 0.5 + world - 1
    if rank == 0:
        print(json.dumps({"workload": workload, "total_bytes": float(tot)}))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.timeout(300)
def test_round_robin_sharding_two_gloo_ranks(tmp_path):
    import corpus
    if not corpus.have_zstd():
        pytest.skip("no libzstd shared object to compress a corpus with")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MZD_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]
    outs = [p.communicate(timeout=280) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se.decode()[-2000:]
    lines = [l for l in outs[0][0].decode().splitlines() if l.startswith("{")]
    assert len(lines) == 2


def test_recorded_two_rank_bench_line_carries_per_rank():
    """profiles/r03_rehearsal.json is the rank-0 line of `bench.py --gpus 2` run as two ranks on a one-GPU box (gloo
    collectives, MZD_BENCH_REHEARSAL=1).  Its value means nothing; the N > 1 line's shape is what the driver's scaling
    runs rely on: whole-job value, one entry per rank, weak scaling, the bench contract's keys."""
    import json
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_rehearsal.json")
    d = json.load(open(p))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["unit"] == "GiB/s" and d["verified_byte_exact"] is True
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1]
    assert all(r["value"] > 0 and r["kernel_ms"] > 0 and 0 < r["roofline_frac"] < 1 for r in pr)
    assert d["config"]["files_total"] == 2 * d["config"]["files_per_gpu"]
    # the whole-job value is total bytes over the slowest rank's time: at least the slower rank's own rate, at most the sum
    assert min(r["value"] for r in pr) <= d["value"] <= sum(r["value"] for r in pr) * 1.001
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d
