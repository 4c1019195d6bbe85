"""CPU, world_size 2 over gloo: the N>1 path of batch mode — sharding of pair ids and the gather of
64-byte result records — with the per-pair results produced by the CPU emulation of the kernels
(tests/hostcheck), so the gathered batch can be checked pair by pair."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

RESULT_DTYPE = np.dtype([("pose", np.float64, 7), ("termination", np.uint32), ("iterations", np.uint32)])
H, W, SEED, TOTAL = 8, 128, 5, 5  # 5 pairs over 2 ranks: uneven shards (3 + 2)


def _one_pair(pair):
    import hostcheck_lib as Hc
    import oracle_lib as O
    A = Hc.synth_scan(SEED, pair, 0, H, W, 0.01)
    B = Hc.synth_scan(SEED, pair, 1, H, W, 0.01)
    fe = O.FeParams()
    ea, pa = O.extract_features(A, H, W, 1.0, 120.0, fe)
    eb, pb = O.extract_features(B, H, W, 1.0, 120.0, fe)
    prm = Hc.reg_params()
    prm.min_associations = 20
    pose, term, iters = Hc.register(B[eb], B[pb], A[ea], A[pa], prm=prm)
    return pose, term, iters


def _worker(rank, world, port, q):
    sys.path[:0] = [HERE, ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from loam_amd import distributed as D
    lo, hi = D.shard_range(TOTAL, world, rank)
    rec = np.zeros(hi - lo, dtype=RESULT_DTYPE)
    for i, pair in enumerate(range(lo, hi)):
        rec[i]["pose"], rec[i]["termination"], rec[i]["iterations"] = _one_pair(pair)
    local = torch.from_numpy(rec.view(np.uint8).copy())
    gathered = D.gather_results(local, TOTAL)
    gathered2 = D.gather_results(local)  # sizes discovered by an extra all_gather
    assert torch.equal(gathered, gathered2)
    if rank == 0:
        q.put(gathered.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from loam_amd import distributed as D
    for total in (0, 1, 5, 8, 8192):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(s[1] - s[0] for s in spans) - min(s[1] - s[0] for s in spans) <= 1


def test_two_rank_gather_matches_single_process(oracle):
    import hostcheck_lib as Hc
    Hc.lib()  # build before forking
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rec = got.view(RESULT_DTYPE)
    assert len(rec) == TOTAL
    for pair in range(TOTAL):
        pose, term, iters = _one_pair(pair)
        assert np.array_equal(rec[pair]["pose"], pose) and rec[pair]["termination"] == term and rec[pair]["iterations"] == iters
