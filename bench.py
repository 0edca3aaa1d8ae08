#!/usr/bin/env python3
"""Headline benchmark: self-play games/sec + MCTS node-expansions/sec at 800 sims/move.

Workload (BASELINE.json configs[1]): 4096 parallel self-play games per GPU, 800 simulations per
move, random-init residual policy/value net (32 filters, 3 residual blocks), self-play settings
(Dirichlet alpha 0.3, fraction 0.25, 6 sampled opening moves).  Synthetic data: games start from the
empty board, weights are seeded random-init.

A "step" = one rollout step of the hot path over the whole batch: the HIP tree kernel (apply the
previous leaf evaluations + backup, PUCT descent, expansion, move choice / re-rooting, leaf emission)
followed by the leaf-batch network forward.  Every live game performs >= 1 simulation per step.

    python bench.py --gpus N --steps K --warmup W

N>1: launched by torch.distributed.run, one rank per GPU; games shard across ranks (disjoint RNG
streams seed+rank), there is NO collective inside the rollout path (weak scaling).  Rank 0 prints ONE
JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_MATRIX_PEAK_TF = 157.3  # fp32-in MFMA / vector peak
BF16_MFMA_PEAK_TF = 2500.0   # dense
NET_MFLOP_PER_POSITION = 4.74  # BASELINE.md section 4 (32 filters, 3 residual blocks)


def tree_bytes_per_sim(mean_depth):
    # BASELINE.md section 4 / SURVEY.md 8d: algorithmic bytes per simulation = 136*D + 332
    return 136.0 * mean_depth + 332.0


def cpu_baseline(state_dict, sims, seconds, n_games):
    """The oracle's lock-step many-game self-play (game_pool.py + inference_server.py shape) with
    the same net evaluated by PyTorch on the host cores.  Checker timed as a baseline -- never the
    product path."""
    import numpy as np
    import torch
    from connect4_amd.net import PolicyValueNet
    from oracle import c4oracle as oc

    # a one-GPU box share is 16 host threads; more threads than that oversubscribes the tiny batch
    cores = min(os.cpu_count() or 1, 16)
    oc.set_threads(cores)
    torch.set_num_threads(cores)
    net = PolicyValueNet(PolicyValueNet.config_from_state_dict(state_dict))
    net.load_state_dict(state_dict)
    net.eval()
    cfg = oc.make_config(sims, root_dirichlet_alpha=0.3, root_exploration_fraction=0.25, num_sampling_moves=6)
    pool = oc.Pool(cfg, n_games, seed=0)
    t0 = time.time()
    steps = 0
    with torch.no_grad():
        while time.time() - t0 < seconds:
            planes = pool.collect()
            v, p = net(torch.from_numpy(planes.astype(np.float32)))
            pool.apply(v.numpy(), p.numpy())
            steps += 1
    dt = time.time() - t0
    st = pool.stats()
    pool.close()
    return dict(value=st["expansions"] / dt, unit="node-expansions/s", cores=cores, kind="port",
                sims_per_s=st["sims"] / dt, games_per_s=st["games"] / dt,
                sample="%d lock-step games x %d steps (%.1f s) of the same workload, oracle C MCTS (OpenMP) + "
                       "PyTorch CPU net on %d threads" % (n_games, steps, dt, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48000)
    ap.add_argument("--warmup", type=int, default=24000,
                    help="default covers one full game length so games/sec is a steady-state rate")
    ap.add_argument("--slots", type=int, default=4096, help="parallel games per GPU")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--net", default="fused", choices=["fused", "torch"],
                    help="fused: hand-written gfx950 MFMA kernel (fp16 storage, fp32 accumulate); torch: PyTorch-ROCm/MIOpen")
    ap.add_argument("--net-dtype", default=None, choices=["f32", "f16", "bf16"], help="torch net only (default f32)")
    ap.add_argument("--steps-per-graph", type=int, default=8)
    ap.add_argument("--max-inner", type=int, default=8, help="evaluator-free simulations a slot may run per tree call (0 = engine default)")
    ap.add_argument("--eval-cache", type=int, default=0, help="log2 entries of the evaluation cache (0 auto, -1 off)")
    ap.add_argument("--level-budget", type=int, default=0, help="descent levels per slot per launch (0 unlimited)")
    ap.add_argument("--time-budget", type=int, default=80000,
                    help="shader cycles of one step: the fused kernel runs every wave for steps x this many cycles per launch")
    ap.add_argument("--pipeline", type=int, default=1, choices=[1, 2],
                    help="2: two half-batches on two streams, tree kernel of one half under the net of the other")
    ap.add_argument("--fused-loop", type=int, default=1, help="1: tree step + net in one persistent kernel")
    ap.add_argument("--steps-per-launch", type=int, default=128)
    ap.add_argument("--pmc-mode", action="store_true",
                    help="for rocprofv3 --pmc passes: warm up with the fused kernel, then run the timed steps as "
                         "separate eager launches (no HIP graph: PMC collection crashes inside graph replay)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=200, help="event-timed eager steps for the roofline")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        entry.build()
    # rehearsal knobs (one-GPU box): C4_BENCH_BACKEND=gloo C4_BENCH_DEVICE=0 run several ranks on one card
    backend = os.environ.get("C4_BENCH_BACKEND", "nccl")
    if "C4_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["C4_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dist.barrier()   # rank 0 has finished building
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from connect4_amd.config import MCTSConfig
    from connect4_amd.net import InferenceNet, NetConfig, random_init_state_dict
    from connect4_amd.selfplay import SelfPlay

    if args.net == "fused":
        args.net_dtype = "f16"
    elif args.net_dtype is None:
        args.net_dtype = "f32"
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.net_dtype]
    sd = random_init_state_dict(NetConfig(), seed=0)
    if args.net == "fused":
        from connect4_amd.fused_net import FusedNet
        net = FusedNet(sd, device=local_rank)
        tdt = torch.float32   # planes are not materialised on this path
    else:
        net = InferenceNet(sd, device="cuda:%d" % local_rank, dtype=tdt)
    sp = SelfPlay(net, args.slots, MCTSConfig.self_play(args.sims), seed=rank, device=local_rank,
                  games_target=-1, record_capacity_games=2 * args.slots, planes_dtype=tdt,
                  use_graph=not args.no_graph, steps_per_graph=args.steps_per_graph, max_inner_iters=args.max_inner,
                  eval_cache_log2_entries=args.eval_cache, level_budget=args.level_budget, time_budget_cycles=args.time_budget, pipeline=args.pipeline,
                  fused_loop=bool(args.fused_loop), steps_per_launch=args.steps_per_launch)

    def barrier():
        if world > 1:
            dist.barrier()

    if args.pmc_mode:
        sp._fused_loop, sp._use_graph = True, False
    sp.run_steps(args.warmup)
    sp.synchronize()
    if args.pmc_mode:
        sp._fused_loop = False
    s0 = sp.stats()
    barrier()
    torch.cuda.synchronize()
    # HIP events on the stream the kernels are launched on: the fused kernel's average launch duration
    kstream = torch.cuda.current_stream()
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record(kstream)
    sp.run_steps(args.steps)
    ev_b.record(kstream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed_gpu_ms = ev_a.elapsed_time(ev_b)
    barrier()
    s1 = sp.stats()
    delta = {k: s1[k] - s0[k] for k in s1}

    # max elapsed over ranks, sum of units over ranks
    tens = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    units = torch.tensor([delta["expansions"], delta["simulations"], delta["games_finished"], delta["moves"],
                          delta["leaf_evals"], delta["terminal_sims"], delta["depth_sum"], delta["children_created"]],
                         dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tens, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    elapsed = float(tens.item())
    exps, sims, games, moves, evals, term, depth_sum, children = [float(x) for x in units.tolist()]

    # ---- event-timed eager segment: per-kernel averages for the roofline (same stream as the kernels)
    prof = None
    if rank == 0 and args.profile_steps > 0:
        p0 = sp.stats()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.profile_steps)]
        stream = torch.cuda.current_stream()
        sp.engine.set_stream(stream.cuda_stream)
        for a, b, c in ev:
            a.record(stream)
            if args.net == "fused":
                sp.engine.step(sp.values, sp.priors, None)
                b.record(stream)
                sp.net.forward_bitboards(sp._leaf_c0, sp._leaf_c1, sp.n_slots, sp.values, sp.priors, stream.cuda_stream)
            else:
                sp.engine.step(sp.values, sp.priors, sp.planes)
                b.record(stream)
                v, p = sp.net(sp.planes)
                sp.values.copy_(v)
                sp.priors.copy_(p)
            c.record(stream)
            sp.steps_done += 1
        torch.cuda.synchronize()
        p1 = sp.stats()
        # an event pair around ONE short kernel also times the record/launch gap: calibrate it with
        # empty pairs on the same stream and subtract (rocprofv3's kernel-trace average is the check)
        cal = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(200)]
        for a, b in cal:
            a.record(stream)
            b.record(stream)
        torch.cuda.synchronize()
        ev_overhead_ms = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2]
        tree_ms = sum(a.elapsed_time(b) for a, b, _ in ev) / len(ev) - ev_overhead_ms
        net_ms = sum(b.elapsed_time(c) for _, b, c in ev) / len(ev) - ev_overhead_ms
        if args.net == "fused":
            # the per-step pair also times the dispatch gap in front of this 160-KB-LDS kernel (it cannot
            # start before the previous kernel has drained); its own duration is measured back to back
            # on the leaves of the last step (same inputs every launch, outputs unchanged)
            reps = 100
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record(stream)
            for _ in range(reps):
                sp.net.forward_bitboards(sp._leaf_c0, sp._leaf_c1, sp.n_slots, sp.values, sp.priors, stream.cuda_stream)
            eb.record(stream)
            torch.cuda.synchronize()
            net_step_ms = net_ms
            net_ms = (ea.elapsed_time(eb) - ev_overhead_ms) / reps
        else:
            net_step_ms = net_ms
        psims = p1["simulations"] - p0["simulations"]
        pdepth = (p1["depth_sum"] - p0["depth_sum"]) / max(1, psims)
        sims_per_launch = psims / len(ev)
        tree_bytes = tree_bytes_per_sim(pdepth) * sims_per_launch
        prof = dict(tree_ms=tree_ms, net_ms=net_ms, net_step_ms=net_step_ms, ev_overhead_ms=ev_overhead_ms, sims_per_launch=sims_per_launch, mean_depth=pdepth,
                    tree_bytes_per_launch=tree_bytes)

    if rank == 0:
        mean_depth = depth_sum / max(1.0, sims)
        out = {
            "metric": "mcts_node_expansions_per_sec",
            "value": exps / elapsed,
            "unit": "node-expansions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f16": "f16", "bf16": "bf16"}[args.net_dtype],
            "data": "synthetic",
            "games_per_sec": games / elapsed,
            "sims_per_sec": sims / elapsed,
            "leaf_evals_per_sec": evals / elapsed,
            "children_created_per_sec": children / elapsed,
            "moves_per_sec": moves / elapsed,
            "terminal_sim_fraction": term / max(1.0, sims),
            "bad_evals": delta["bad_evals"],
            "eval_cache_hit_rate": (delta["eval_cache_hits"] / max(1, delta["eval_cache_probes"])),
            "mean_leaf_depth": mean_depth,
            "config": {
                "workload": "%d parallel self-play games per GPU, %d sims/move, random-init resnet "
                            "(32 filters, 3 residual blocks), %dxMI355X" % (args.slots, args.sims, world),
                "slots_per_gpu": args.slots, "simulations": args.sims, "net": "32f-3res-4fc",
                "net_impl": args.net, "net_dtype": args.net_dtype, "tree_dtype": "u64 bitboards, u32 visits, f64 value sums/priors",
                "parallelism": "games sharded over %d GPU(s), no collective in the rollout path" % world,
                "max_inner_iters": args.max_inner, "eval_cache_log2_entries": args.eval_cache, "level_budget": args.level_budget, "time_budget_cycles": args.time_budget, "pipeline_halves": args.pipeline, "fused_loop": bool(args.fused_loop), "steps_per_launch": args.steps_per_launch, "hip_graph": (not args.no_graph), "steps_per_graph": args.steps_per_graph,
                "dirichlet_alpha": 0.3, "exploration_fraction": 0.25, "num_sampling_moves": 6,
            },
        }
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
        except OSError:
            pass
        fused = None
        if args.fused_loop and args.net == "fused" and not args.pmc_mode:
            # the timed region is back-to-back launches of ONE kernel; per launch of steps_per_launch steps:
            n_launch = (args.steps + args.steps_per_launch - 1) // args.steps_per_launch
            launch_ms = timed_gpu_ms / n_launch
            r_sims, r_evals = delta["simulations"] / n_launch, (delta["leaf_evals"] - delta["eval_cache_hits"]) / n_launch
            r_depth = delta["depth_sum"] / max(1, delta["simulations"])
            tree_b = tree_bytes_per_sim(r_depth) * r_sims
            ach = tree_b / (launch_ms * 1e-3) / 1e9
            mfma_tf = NET_MFLOP_PER_POSITION * 1e6 * r_evals / (launch_ms * 1e-3) / 1e12
            pmc_ok = args.slots == pmc.get("slots", 4096) and args.sims == 800 and args.max_inner == pmc.get("max_inner", -1) \
                and args.steps_per_launch == pmc.get("steps_per_launch", 128) and pmc.get("kernel", "") == "c4_selfplay_wave_kernel"
            fused = {
                "kernel": "c4_selfplay_wave_kernel (per wave: PUCT tree walk of its slots + policy/value net on their leaves; the only kernel of the timed region)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "traffic": ((2.0 * pmc["FETCH_SIZE_fused"] + pmc["WRITE_SIZE_fused"]) * 1024.0 if pmc_ok and "FETCH_SIZE_fused" in pmc else None),
                "avg_launch_ms": launch_ms, "launches": n_launch, "steps_per_launch": args.steps_per_launch,
                "sims_per_launch": r_sims, "mean_depth": r_depth, "algorithmic_bytes_per_launch": tree_b,
                "net_positions_per_launch": r_evals, "mfma_achieved_tflops": mfma_tf, "mfma_frac_of_dense_f16_peak": mfma_tf / BF16_MFMA_PEAK_TF,
                "note": "tree walk = dependent-load (latency) bound pointer chase, bytes = (136*D+332) per simulation (node records, "
                        "path, cache line); the network part of the same kernel is counted in mfma_achieved_tflops "
                        "(4.74 MFLOP per evaluated leaf); duration = HIP events around the timed region / launches",
            }
        if prof:
            ach = prof["tree_bytes_per_launch"] / (prof["tree_ms"] * 1e-3) / 1e9
            pmc_ok = args.slots == 4096 and args.sims == 800 and args.max_inner == pmc.get("max_inner", -1)
            tree = {
                "kernel": "c4_step_kernel<EXTERNAL_F32> (tree walk: apply+backup, PUCT descent, expand, move choice, emit)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS,
                # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
                # command (profiles/r01_pmc_traffic.json), KB -> bytes with the guide's gfx950 correction
                # (FETCH_SIZE x 2 for 16-byte-per-lane loads, WRITE_SIZE as is; MI355X_MICROARCH.md section
                # HBM); not collectable inside this process
                "traffic": ((2.0 * pmc["FETCH_SIZE_tree"] + pmc["WRITE_SIZE_tree"]) * 1024.0 if pmc_ok and "FETCH_SIZE_tree" in pmc else None),
                "avg_launch_ms": prof["tree_ms"], "event_overhead_ms_subtracted": prof["ev_overhead_ms"],
                "sims_per_launch": prof["sims_per_launch"], "mean_depth": prof["mean_depth"],
                "algorithmic_bytes_per_launch": prof["tree_bytes_per_launch"],
                "note": "dependent-load (latency) bound pointer chase; bytes = (136*D+332) per simulation",
            }
            tf = NET_MFLOP_PER_POSITION * 1e6 * args.slots / (prof["net_ms"] * 1e-3) / 1e12
            peak = FP32_MATRIX_PEAK_TF if args.net_dtype == "f32" else BF16_MFMA_PEAK_TF
            net_leaves = (p1["leaf_evals"] - p0["leaf_evals"] - (p1["eval_cache_hits"] - p0["eval_cache_hits"])) / len(ev)
            netr = {
                "kernel": ("c4_net_kernel (fused stem+tower+heads, v_mfma_f32_32x32x16_f16)" if args.net == "fused"
                           else "leaf-batch policy/value net forward (PyTorch-ROCm / MIOpen convs)"),
                "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                "avg_forward_ms": prof["net_ms"], "avg_ms_in_step_incl_dispatch_gap": prof["net_step_ms"],
                "traffic": ((2.0 * pmc["FETCH_SIZE_net"] + pmc["WRITE_SIZE_net"]) * 1024.0
                            if pmc_ok and args.net == "fused" and "FETCH_SIZE_net" in pmc else None),
                "positions_per_launch": args.slots, "leaves_needing_the_net_per_launch": net_leaves,
                "note": "achieved counts every row the kernel computes (static batch); slots whose simulation "
                        "ended on a terminal or cached leaf still occupy a row",
            }
            out["roofline_tree"] = tree
            out["roofline_net"] = netr
            if fused is None:
                out["roofline"] = dict(tree if prof["tree_ms"] >= prof["net_ms"] else netr)   # the dominant kernel by time
                out["roofline"]["share_of_step"] = max(prof["tree_ms"], prof["net_ms"]) / (prof["tree_ms"] + prof["net_ms"])
        if fused is not None:
            out["roofline"] = fused   # the dominant (only) kernel of the timed region
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, args.sims, args.cpu_seconds, 256)
        print(json.dumps(out))
        sys.stdout.flush()
    sp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
