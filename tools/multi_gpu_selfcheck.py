#!/usr/bin/env python3
"""Pre-flight of the multi-GPU batch mode on any box with >= 2 GPUs: ONE command, no arguments needed.

    python tools/multi_gpu_selfcheck.py [--gpus N] [--pairs-per-gpu P] [--steps K]
    python tools/multi_gpu_selfcheck.py --plan --gpus 8 --total-pairs 8191     # the shard plan only, no GPU needed

One process per GPU (started here as `python -m torch.distributed.run` child ranks, before this process touches HIP).
Every rank runs the single-GPU pipeline (loamx_register_scan_pairs_dev) on its contiguous block of pair ids and the
result records are gathered by RCCL behind the C ABI (loamx_gather_results_dev). Checked, and reported as one JSON line:

  * RCCL's own view: ncclCommCount == N on every rank, rank ids and devices as gathered through the collective
  * equal shards  (total = N * P, ncclAllGather): every rank's gathered array is the same on all ranks, and the block of
    rank (r + 1) % N equals what rank r computes for those pair ids by itself — bit for bit
  * uneven shards (total = N * P - 1, grouped ncclBroadcast): the same two checks
  * weak-scaling efficiency: pairs/s of N ranks together over N x pairs/s of rank 0 alone (same P per rank)
  * with --total-pairs T: the fixed-total (strong-scaling) curve point for this N

torch.distributed (gloo) is only the rendezvous for the 128-byte RCCL id and the pass / fail exchange; the data path
and the barrier around the timed region are the library's (loamx_comm_barrier).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]

H, W, SEED, SIGMA = 64, 1024, 20240311, 0.01


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks (default: every visible GPU)")
    ap.add_argument("--pairs-per-gpu", type=int, default=256)
    ap.add_argument("--total-pairs", type=int, default=0, help="also time a fixed total (strong scaling point)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--plan", action="store_true", help="print the shard plan and exit (no GPU, no ranks)")
    return ap.parse_args()


def plan(world, total):
    from loam_amd import capi
    shards = [capi.shard_range(total, world, r) for r in range(world)]
    sizes = [n for _, n in shards]
    return {"world_size": world, "total_pairs": total, "shards": [{"rank": r, "first": f, "count": n} for r, (f, n) in enumerate(shards)],
            "collective": "ncclAllGather" if len(set(sizes)) == 1 else "grouped ncclBroadcast (uneven shards)",
            "covers_everything": shards[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(shards, shards[1:])) and
                                 shards[-1][0] + shards[-1][1] == total}


def fan_out(args, world):
    from loam_amd import build as B
    B.build()  # hipcc only: the ranks find the library up to date
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(world), "--pairs-per-gpu", str(args.pairs_per_gpu),
           "--total-pairs", str(args.total_pairs), "--steps", str(args.steps)]
    return subprocess.run(cmd, env=env).returncode


def rank_main(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from loam_amd import capi

    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.device_count() < world:
        raise SystemExit(f"multi_gpu_selfcheck: {world} ranks but {torch.cuda.device_count()} GPU(s) visible (one rank per GPU)")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    ctx = capi.Context(local)
    uid = [capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    comm = capi.Comm(ctx, uid[0], world, rank)
    if world == 1:  # one rank: no shortcut — the collectives themselves run on the one-rank communicator (option FORCE_RCCL)
        ctx.set_option("FORCE_RCCL", 1)
    lidar, fe, reg = capi.LidarParams(H, W, 1.0, 120.0), capi.FeatureExtractionParams(), capi.RegistrationParams()
    N = H * W
    report = {"world_size": world, "rccl_comm_nranks": comm.info()["world_size"], "failures": []}

    def run_shard(first, count):
        """the single-GPU pipeline on pair ids [first, first + count): device buffer of count records"""
        d_xyz, d_res = ctx.alloc(max(count, 1) * 2 * N * 24), ctx.alloc(max(count, 1) * 64)
        if count:
            ctx.synth_scan_pairs_dev(SEED, first, count, H, W, SIGMA, d_xyz.ptr)
            ctx.register_scan_pairs_dev(d_xyz.ptr, count, lidar, fe, reg, d_res.ptr)
        ctx.synchronize()
        return d_xyz, d_res

    def all_ok(flag):
        t = torch.tensor([1 if flag else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def check_total(total, label):
        first, count = capi.shard_range(total, world, rank)
        d_xyz, d_res = run_shard(first, count)
        d_all = ctx.alloc(total * 64)
        before = comm.stats()
        comm.gather_results_dev(d_res.ptr, count, total, d_all.ptr)
        ctx.synchronize()
        after = comm.stats()
        enqueued = {k: after[k] - before[k] for k in after if after[k] != before[k]}  # what this call REALLY enqueued
        got = d_all.download(np.uint8, total * 64).copy()
        # (1) every rank holds the same array
        ref = [got.tobytes() if rank == 0 else None]
        dist.broadcast_object_list(ref, src=0)
        same = ref[0] == got.tobytes()
        # (2) my neighbour's block is what I compute for its pair ids myself
        nf, nc = capi.shard_range(total, world, (rank + 1) % world)
        n_xyz, n_res = run_shard(nf, nc)
        mine = n_res.download(np.uint8, max(nc, 1) * 64)[: nc * 64]
        neighbour = bool(np.array_equal(got[nf * 64:(nf + nc) * 64], mine))
        own = bool(np.array_equal(got[first * 64:(first + count) * 64], d_res.download(np.uint8, max(count, 1) * 64)[: count * 64]))
        rec = got.view(capi.RESULT_DTYPE)
        converged = int((rec["termination"] == capi.CONVERGED).sum())
        for b in (d_xyz, d_res, d_all, n_xyz, n_res):
            b.free()
        ok = all_ok(same and neighbour and own)
        if rank == 0:
            report[label] = {"total_pairs": total, "planned_collective": plan(world, total)["collective"], "enqueued": enqueued,
                             "identical_on_every_rank": ok, "converged": converged}
            if not ok:
                report["failures"].append(label)
        return ok

    # RCCL's own view of the ranks: one record {rank, device} per rank through the same entry point
    proof = np.zeros(1, dtype=capi.RESULT_DTYPE)
    proof["termination"], proof["iterations"] = rank, local
    d_p, d_pa = ctx.alloc(64).upload(proof.view(np.uint8)), ctx.alloc(world * 64)
    comm.gather_results_dev(d_p.ptr, 1, world, d_pa.ptr)
    ctx.synchronize()
    ids = d_pa.download(np.uint8, world * 64).view(capi.RESULT_DTYPE)
    report["gathered_rank_ids"], report["gathered_devices"] = [int(x) for x in ids["termination"]], [int(x) for x in ids["iterations"]]
    if report["rccl_comm_nranks"] != world or report["gathered_rank_ids"] != list(range(world)):
        report["failures"].append("communicator")

    P = args.pairs_per_gpu
    check_total(world * P, "equal_shards")
    check_total(world * P - 1, "uneven_shards")

    # ---- timing: rank 0 alone, then all ranks (weak scaling), then the fixed total (strong scaling)
    def timed(count, total, participate):
        first, _ = capi.shard_range(total, world, rank) if total else (rank * count, count)
        d_xyz, d_res = ctx.alloc(max(count, 1) * 2 * N * 24), ctx.alloc(max(count, 1) * 64)
        d_all = ctx.alloc(max(total, 1) * 64)
        if count and participate:
            ctx.synth_scan_pairs_dev(SEED, first, count, H, W, SIGMA, d_xyz.ptr)

        def step():
            if participate and count:
                ctx.register_scan_pairs_dev(d_xyz.ptr, count, lidar, fe, reg, d_res.ptr)
            if total:
                comm.gather_results_dev(d_res.ptr, count, total, d_all.ptr)

        step()
        comm.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ctx.synchronize()
        dt = comm.barrier(time.perf_counter() - t0)  # max over ranks
        for b in (d_xyz, d_res, d_all):
            b.free()
        return dt

    t1 = timed(P, 0, rank == 0)             # rank 0 alone, no collective
    tn = timed(P, world * P, True)          # every rank, P pairs each + the gather
    alone, together = P * args.steps / t1, world * P * args.steps / tn
    report["weak_scaling"] = {"pairs_per_gpu": P, "pairs_per_s_rank0_alone": round(alone, 1), "pairs_per_s_all_ranks": round(together, 1),
                              "efficiency": round(together / (world * alone), 4)}
    if args.total_pairs:
        _, cnt = capi.shard_range(args.total_pairs, world, rank)
        ts = timed(cnt, args.total_pairs, True)
        report["strong_scaling_point"] = {"total_pairs": args.total_pairs, "n_gpus": world, "pairs_per_s": round(args.total_pairs * args.steps / ts, 1)}
    report["enqueued_total"] = comm.stats()
    if world == 1 and not all(report["enqueued_total"][k] for k in ("ncclAllGather", "ncclBroadcast", "ncclAllReduce")):
        report["failures"].append("FORCE_RCCL: a collective was not enqueued")
    report["ok"] = not report["failures"]
    if rank == 0:
        print(json.dumps(report), flush=True)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()
    return 0 if report["ok"] or rank else 1


def main():
    args = parse()
    if args.plan:
        world = args.gpus or 8
        print(json.dumps(plan(world, args.total_pairs or world * args.pairs_per_gpu)))
        return 0
    if "WORLD_SIZE" in os.environ:
        return rank_main(args)
    import torch  # (device_count does not initialise the GPU)
    world = args.gpus or torch.cuda.device_count()
    if world < 1:
        raise SystemExit("multi_gpu_selfcheck: no GPU visible (the HIP path has no CPU fallback)")
    return fan_out(args, world)


if __name__ == "__main__":
    sys.exit(main())
