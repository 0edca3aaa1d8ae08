#!/usr/bin/env python3
"""Headline benchmark: self-play games/sec + MCTS node-expansions/sec at 800 sims/move.

Workload (BASELINE.json configs[1]): 4096 parallel self-play games per GPU, 800 simulations per
move, random-init residual policy/value net (32 filters, 3 residual blocks), self-play settings
(Dirichlet alpha 0.3, fraction 0.25, 6 sampled opening moves).  Synthetic data: games start from the
empty board, weights are seeded random-init.  Leaves are evaluated in the REFERENCE's precision (the reference's net is
float32, model.py:252-282): `dtype: "f32x3"` = every fp32 operand as an fp16 hi + lo pair, three MFMAs per product, fp32
accumulation.  The opt-in fp16-storage mode (`--net-precision f16`) is timed in a short second run and reported as the
secondary block `f16_storage_mode` of the same line -- it is NOT the headline.

    python bench.py --gpus N --steps K --warmup W

A "step" = ONE launch of the persistent self-play kernel (c4_selfplay_split_kernel: tree waves + network waves): `--quanta-per-step`
(256) time quanta of `--time-budget` (80,000) shader cycles for every one of the 4096 games per GPU,
about 9 ms, in which every game completes several hundred simulations (tree walk + network on the
leaves).  K of them are timed exactly, after W untimed ones.

Before W and K a declared, UNTIMED pre-roll brings the engine to its steady state (the timed region
is meaningless on 4096 identical empty boards with a cold evaluation cache): launches run until
`--preroll-games-per-slot` x slots games have finished, i.e. the slots are spread over all plies of
a game and the evaluation cache (the reference's memo table, evaluators.py:9-25) holds the openings.
It is reported as preroll_s / preroll_steps / preroll_games.

N>1 (scaling runs use the default `--slots 4096` games PER GPU, so N = 1 is exactly this line; BASELINE configs[2]'s 65,536 games
on 8 GPUs are `--slots 8192`): `python bench.py --gpus N` starts N rank processes itself (torch.distributed.run, one per GPU,
before anything touches the GPU) and relays rank 0's JSON line; under an external torchrun it uses
the RANK/LOCAL_RANK/WORLD_SIZE it finds.  Games shard across ranks (disjoint RNG streams seed+rank),
there is NO collective inside the rollout path (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the fused kernel c4_selfplay_steps launches (C4_FUSED_MODE selects the older variants for A/B runs)
FUSED_KERNEL = {"wave": "c4_selfplay_wave_kernel", "block": "c4_selfplay_kernel"}.get(os.environ.get("C4_FUSED_MODE", ""), "c4_selfplay_split_kernel")
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_MATRIX_PEAK_TF = 157.3  # fp32-in MFMA / vector peak
F16_MFMA_PEAK_TF = 2500.0    # dense


def net_mflop_per_position(filters=32, residuals=3):
    """4.74 MFLOP for 32 filters / 3 residual blocks (stem + tower + 1x1 heads + MLPs)."""
    macs = 42 * 9 * 3 * filters + 2 * residuals * 42 * 9 * filters * filters + 42 * filters * 3 + 42 * 42 + 42 + 84 * 7
    return 2.0 * macs / 1e6


PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")


def tree_bytes_per_sim(mean_depth):
    # BASELINE.md section 4 / SURVEY.md 8d: algorithmic bytes per simulation = 136*D + 332
    return 136.0 * mean_depth + 332.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (one persistent-kernel launch each)")
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps after the pre-roll")
    ap.add_argument("--slots", type=int, default=4096,
                    help="parallel games PER GPU (weak scaling: --gpus N plays N x this many; the default is BASELINE configs[1] and what a "
                         "scaling run should keep so that N = 1 equals the single-GPU line; configs[2] = --gpus 8 --slots 8192)")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--filters", type=int, default=32, help="net width (32 = the reference's default / BASELINE config; 64 = its example_config)")
    ap.add_argument("--residuals", type=int, default=3)
    ap.add_argument("--fc-layers", type=int, default=4)
    ap.add_argument("--quanta-per-step", type=int, default=256,
                    help="time quanta (rollout steps) per bench step = per launch of the fused kernel")
    ap.add_argument("--preroll-games-per-slot", type=float, default=4.0,
                    help="untimed pre-roll until this many games per slot have finished (0 = none)")
    ap.add_argument("--preroll-max-s", type=float, default=60.0)
    ap.add_argument("--net", default="fused", choices=["fused", "torch"],
                    help="fused: hand-written gfx950 MFMA kernel; torch: PyTorch-ROCm/MIOpen")
    ap.add_argument("--net-precision", default="f32x3", choices=["f16", "f32x3"],
                    help="fused net arithmetic: f32x3 = reference precision (the reference evaluates leaves in float32, model.py:252-282: "
                         "every fp32 operand as fp16 hi + lo, three MFMAs per k-step, fp32 accumulate; reproduces the reference's visit "
                         "counts) -- the headline; f16 = fp16 storage / fp32 accumulate, reduced precision, reported as a secondary block")
    ap.add_argument("--net-dtype", default=None, choices=["f32", "f16", "bf16"], help="torch net only (default f32)")
    ap.add_argument("--steps-per-graph", type=int, default=8)
    ap.add_argument("--max-inner", type=int, default=32, help="evaluator-free simulations a slot may run per tree call (0 = engine default)")
    ap.add_argument("--eval-cache", type=int, default=0, help="log2 entries of the evaluation cache (0 auto, -1 off)")
    ap.add_argument("--level-budget", type=int, default=0, help="descent levels per slot per launch (0 unlimited)")
    ap.add_argument("--time-budget", type=int, default=80000,
                    help="shader cycles of one quantum: the fused kernel runs every wave for quanta x this many cycles per launch")
    ap.add_argument("--pipeline", type=int, default=1, choices=[1, 2],
                    help="2: two half-batches on two streams, tree kernel of one half under the net of the other")
    ap.add_argument("--fused-loop", type=int, default=1, help="1: tree step + net in one persistent kernel")
    ap.add_argument("--pmc-mode", action="store_true",
                    help="for rocprofv3 --pmc passes of the standalone kernels: pre-roll with the fused kernel, then run "
                         "the timed steps as separate eager launches (no HIP graph: PMC collection crashes inside graph replay)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="event-timed eager rollout steps of the STANDALONE kernels (c4_step_kernel, c4_net_kernel) for secondary per-kernel "
                         "rooflines; 0 = none (the timed region launches neither of them)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction plumbing only (no GPU, gloo): what the CPU test of --gpus N runs; "
                         "the line it prints is flagged dry_run and carries no measurement")
    ap.add_argument("--precise-compare", type=int, default=1,
                    help="1: at N=1 also time a short run of the OTHER fused-net precision (f16 storage next to the f32x3 headline, "
                         "or the reverse) and report its throughput as a secondary block")
    return ap.parse_args(argv)


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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """--gpus N without an external launcher: start N rank processes (one per GPU) as a CHILD process
    tree before this process has touched the GPU, relay their output and exit code.  Shape replaced:
    the reference's Pool x Pipe grid of game processes (oinkoink/neural/training.py:113-131)."""
    if not args.dry_run:
        import __graft_entry__ as entry
        entry.build()     # once, before the ranks start (they find the library up to date)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    # rank 0's JSON line is the only thing this process prints on stdout; whatever else the ranks or their
    # libraries write there (e.g. gloo's connection notes) is passed on to stderr
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def make_selfplay(args, sd, rank, local_rank, precision):
    import torch
    from connect4_amd.config import MCTSConfig
    from connect4_amd.net import InferenceNet
    from connect4_amd.selfplay import SelfPlay
    if args.net == "fused":
        from connect4_amd.fused_net import FusedNet
        net = FusedNet(sd, device=local_rank, precision=precision)
        tdt = torch.float32   # planes are not materialised on this path
    else:
        tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.net_dtype]
        net = InferenceNet(sd, device="cuda:%d" % local_rank, dtype=tdt)
    sp = SelfPlay(net, args.slots, MCTSConfig.self_play(args.sims), seed=rank, device=local_rank,
                  games_target=-1, record_capacity_games=0, planes_dtype=tdt,
                  use_graph=not args.no_graph, steps_per_graph=args.steps_per_graph, max_inner_iters=args.max_inner,
                  eval_cache_log2_entries=args.eval_cache, level_budget=args.level_budget, time_budget_cycles=args.time_budget, pipeline=args.pipeline,
                  fused_loop=bool(args.fused_loop), steps_per_launch=args.quanta_per_step)
    return sp, net


def consume_games(sp):
    """The finished games of the launches so far leave the engine as packed training records, on the device and
    on the launch stream (c4_export_games_dev: three small kernels, no host synchronisation) -- part of every
    step, so the timed region is self-play that actually delivers its games."""
    if not hasattr(sp, "_bench_export"):
        sp._bench_export = sp.engine.export_buffers()
        import torch
        sp._bench_exported = torch.zeros(2, dtype=torch.int64, device=sp.device)
    t, counts = sp._bench_export
    sp.engine.export_games_async(t, counts)
    sp._bench_exported += counts


def preroll(sp, args):
    """Untimed, declared: launches until `preroll_games_per_slot` x slots games have finished."""
    t0 = time.perf_counter()
    want = int(args.preroll_games_per_slot * args.slots)
    steps = 0
    st = sp.stats()
    while st["games_finished"] < want and time.perf_counter() - t0 < args.preroll_max_s:
        for _ in range(8):
            sp.run_steps(args.quanta_per_step)
            consume_games(sp)
        steps += 8
        st = sp.stats()     # synchronises
    sp.synchronize()
    return dict(preroll_s=time.perf_counter() - t0, preroll_steps=steps, preroll_games=st["games_finished"],
                preroll_target_games=want)


def timed_region(sp, args, barrier):
    """W untimed + exactly K timed steps, barrier + synchronize on both sides; HIP events on the launch stream."""
    import torch
    for _ in range(args.warmup):
        sp.run_steps(args.quanta_per_step)
        consume_games(sp)
    sp.synchronize()
    s0 = sp.stats()
    barrier()
    torch.cuda.synchronize()
    kstream = torch.cuda.current_stream()
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record(kstream)
    for _ in range(args.steps):
        sp.run_steps(args.quanta_per_step)
        consume_games(sp)
    ev_b.record(kstream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gpu_ms = ev_a.elapsed_time(ev_b)
    barrier()
    s1 = sp.stats()
    d = {k: s1[k] - s0[k] for k in s1}
    d["games_exported_total"] = int(sp._bench_exported[0].item())
    d["positions_exported_total"] = int(sp._bench_exported[1].item())
    d["games_finished_total"] = s1["games_finished"]
    d["dropped_games_total"] = s1["dropped_games"]
    return elapsed, gpu_ms, d


def run_rank(args):
    import torch
    import torch.distributed as dist

    import __graft_entry__ as entry
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0 and not args.dry_run:
        entry.build()
    # rehearsal knobs (one-GPU box): C4_BENCH_BACKEND=gloo C4_BENCH_DEVICE=0 run several ranks on one card
    backend = "gloo" if args.dry_run else os.environ.get("C4_BENCH_BACKEND", "nccl")
    if "C4_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["C4_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dist.barrier()   # rank 0 has finished building
    if args.dry_run:
        # the timing contract's reductions on synthetic numbers: MAX of (1 + rank) seconds, SUM of (rank + 1) units
        tens = torch.tensor([1.0 + rank], dtype=torch.float64)
        units = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tens, op=dist.ReduceOp.MAX)
            dist.all_reduce(units, op=dist.ReduceOp.SUM)
        if rank == 0:
            print(json.dumps({"metric": "dry_run", "dry_run": True, "value": None, "n_gpus": world,
                              "world_size_observed": (dist.get_world_size() if world > 1 else 1),
                              "collective_backend": (backend if world > 1 else None),
                              "max_check": float(tens.item()), "sum_check": float(units.item())}))
            sys.stdout.flush()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    observed_world = dist.get_world_size() if world > 1 else 1

    from connect4_amd.net import NetConfig, random_init_state_dict

    if args.net == "fused":
        if args.filters != 32 and args.net_precision == "f32x3":
            args.net_precision = "f16"     # the 64-filter forward exists in fp16 storage only (declared in dtype / config)
        # what the leaf evaluation computes in: "f32x3" = float32 operands carried as fp16 hi + lo parts, products exact,
        # fp32 accumulation (the reference's float32, model.py:252-282); "f16" = fp16 storage, fp32 accumulation
        args.net_dtype = args.net_precision
    elif args.net_dtype is None:
        args.net_dtype = "f32"
    sd = random_init_state_dict(NetConfig(filters=args.filters, n_residuals=args.residuals, n_fc_layers=args.fc_layers), seed=0)
    args.net_mflop = net_mflop_per_position(args.filters, args.residuals)
    sp, net = make_selfplay(args, sd, rank, local_rank, args.net_precision)

    def barrier():
        if world > 1:
            dist.barrier()

    fused_timed = bool(args.fused_loop) and args.net == "fused" and not args.pmc_mode
    if args.pmc_mode:
        sp._fused_loop, sp._use_graph = True, False
    pre = preroll(sp, args) if args.preroll_games_per_slot > 0 else dict(preroll_s=0.0, preroll_steps=0, preroll_games=0, preroll_target_games=0)
    if args.pmc_mode:
        sp._fused_loop = False
    elapsed, timed_gpu_ms, delta = timed_region(sp, args, barrier)

    # max elapsed over ranks, sum of units over ranks
    tens = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    keys = ["expansions", "simulations", "games_finished", "moves", "leaf_evals", "terminal_sims", "depth_sum",
            "children_created", "eval_cache_hits", "eval_cache_probes", "bad_evals", "speculative_evals"]
    units = torch.tensor([delta[k] for k in keys], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tens, op=dist.ReduceOp.MAX)
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
    elapsed = float(tens.item())
    tot = dict(zip(keys, [float(x) for x in units.tolist()]))

    # ---- event-timed eager segment: per-kernel averages for the secondary rooflines (rank 0, N=1 shape)
    prof = None
    if rank == 0 and args.profile_steps > 0:
        prof = profile_eager(sp, args)

    out = None
    if rank == 0:
        sims = tot["simulations"]
        out = {
            "metric": "mcts_node_expansions_per_sec",
            "value": tot["expansions"] / elapsed,
            "unit": "node-expansions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.net_dtype,
            "data": "synthetic",
            "games_per_sec": tot["games_finished"] / elapsed,
            "sims_per_sec": sims / elapsed,
            "leaf_evals_per_sec": tot["leaf_evals"] / elapsed,
            "children_created_per_sec": tot["children_created"] / elapsed,
            "moves_per_sec": tot["moves"] / elapsed,
            "terminal_sim_fraction": tot["terminal_sims"] / max(1.0, sims),
            "bad_evals": tot["bad_evals"],
            "net_evals_per_sec": (tot["leaf_evals"] - tot["eval_cache_hits"] + tot.get("speculative_evals", 0)) / elapsed,
            "speculative_evals_per_sec": tot.get("speculative_evals", 0) / elapsed,
            "eval_cache_hit_rate": tot["eval_cache_hits"] / max(1.0, tot["eval_cache_probes"]),
            "mean_leaf_depth": tot["depth_sum"] / max(1.0, sims),
            "timed_region_s": elapsed,
            "games_exported_on_device": {"exported": delta["games_exported_total"], "finished": delta["games_finished_total"],
                                         "dropped": delta["dropped_games_total"], "positions": delta["positions_exported_total"]},
            "world_size_observed": observed_world,
            "collective_backend": (backend if world > 1 else None),
            "config": {
                "workload": "%d parallel self-play games per GPU, %d sims/move, random-init resnet "
                            "(%d filters, %d residual blocks), %dxMI355X" % (args.slots, args.sims, args.filters, args.residuals, world),
                "step": "one launch of the persistent self-play kernel = %d quanta of %d shader cycles for every game"
                        % (args.quanta_per_step, args.time_budget),
                "slots_per_gpu": args.slots, "simulations": args.sims, "net": "%df-%dres-%dfc" % (args.filters, args.residuals, args.fc_layers),
                "net_impl": args.net, "net_precision": (args.net_precision if args.net == "fused" else args.net_dtype),
                "tree_dtype": "u64 bitboards, u32 visits, f64 value sums/priors",
                "parallelism": "games sharded over %d GPU(s), no collective in the rollout path" % world,
                "max_inner_iters": args.max_inner, "eval_cache_log2_entries": args.eval_cache, "level_budget": args.level_budget,
                "time_budget_cycles": args.time_budget, "quanta_per_step": args.quanta_per_step, "pipeline_halves": args.pipeline,
                "fused_loop": bool(args.fused_loop), "hip_graph": (not args.no_graph and not args.fused_loop),
                "dirichlet_alpha": 0.3, "exploration_fraction": 0.25, "num_sampling_moves": 6,
            },
        }
        out.update(pre)
        pmc = {}
        try:
            with open(PMC_FILE) as f:
                pmc = json.load(f)
        except OSError:
            pass
        fused = None
        if fused_timed:
            # the timed region is back-to-back launches of ONE kernel, one launch per step (rank 0's own counters)
            n_launch = args.steps
            launch_ms = timed_gpu_ms / n_launch
            r_sims = delta["simulations"] / n_launch
            r_evals = (delta["leaf_evals"] - delta["eval_cache_hits"] + delta.get("speculative_evals", 0)) / n_launch
            r_depth = delta["depth_sum"] / max(1, delta["simulations"])
            tree_b = tree_bytes_per_sim(r_depth) * r_sims
            ach = tree_b / (launch_ms * 1e-3) / 1e9
            # algorithmic network FLOPs (4.74 MFLOP per position); the f32x3 mode issues three fp16 MFMAs per algorithmic product,
            # reported separately as mfma_issued_tflops
            mfma_tf = args.net_mflop * 1e6 * r_evals / (launch_ms * 1e-3) / 1e12
            mfma_issue_factor = 3.0 if args.net_precision == "f32x3" else 1.0
            # PMC traffic only from a record of launches of exactly this shape
            pmc_ok = (pmc.get("kernel", "") == FUSED_KERNEL and args.slots == pmc.get("slots") and args.sims == pmc.get("sims")
                      and args.max_inner == pmc.get("max_inner") and args.quanta_per_step == pmc.get("quanta_per_launch")
                      and args.time_budget == pmc.get("time_budget_cycles") and args.net_precision == pmc.get("net_precision", "f16")
                      and args.filters == pmc.get("filters", 32) and args.residuals == pmc.get("residuals", 3)
                      and "FETCH_SIZE_fused" in pmc)
            fused = {
                "kernel": FUSED_KERNEL + " (tree waves: PUCT tree walk of their slots; network waves: policy/value net on the posted leaves; the only kernel of the timed region)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "traffic": ((2.0 * pmc["FETCH_SIZE_fused"] + pmc["WRITE_SIZE_fused"]) * 1024.0 if pmc_ok else None),
                "traffic_source": (os.path.relpath(PMC_FILE, ROOT) if pmc_ok else None),
                "avg_launch_ms": launch_ms, "launches": n_launch, "quanta_per_launch": args.quanta_per_step,
                "sims_per_launch": r_sims, "mean_depth": r_depth, "algorithmic_bytes_per_launch": tree_b,
                "net_positions_per_launch": r_evals, "mfma_achieved_tflops": mfma_tf,
                "mfma_frac_of_dense_f16_peak": mfma_tf / F16_MFMA_PEAK_TF,
                "mfma_issued_tflops": mfma_tf * mfma_issue_factor,
                "mfma_issued_frac_of_dense_f16_peak": mfma_tf * mfma_issue_factor / F16_MFMA_PEAK_TF,
                "note": "tree walk = dependent-load (latency) bound pointer chase, bytes = (136*D+332) per simulation (node records, "
                        "path, cache line); the network part of the same kernel is counted in mfma_achieved_tflops "
                        "(%.2f MFLOP per evaluated leaf); duration = HIP events around the timed region / launches" % args.net_mflop,
            }
        if prof:
            out.update(secondary_rooflines(prof, args, pmc))
            if fused is None:
                dom = "roofline_tree" if prof["tree_ms"] >= prof["net_ms"] else "roofline_net"
                out["roofline"] = dict(out[dom])   # the dominant kernel by time
                out["roofline"]["share_of_step"] = max(prof["tree_ms"], prof["net_ms"]) / (prof["tree_ms"] + prof["net_ms"])
        if fused is not None:
            out["roofline"] = fused   # the dominant (only) kernel of the timed region
    sp.close()
    if hasattr(net, "close"):
        net.close()
    if rank == 0 and world == 1:
        if args.precise_compare and fused_timed and args.filters == 32:
            if args.net_precision == "f16":
                out["reference_precision_mode"] = precise_compare(args, sd, out, "f32x3")
            else:
                out["f16_storage_mode"] = precise_compare(args, sd, out, "f16")
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, args.sims, args.cpu_seconds, 256)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def precise_compare(args, sd, main_line, precision):
    """The same workload with the fused net's other arithmetic (f32x3: f16 hi/lo split, 3 MFMAs per k-step = the reference's
    float32; f16: fp16 storage, reduced precision): a short separate run, reported next to the headline."""
    import copy
    a = copy.copy(args)
    a.net_precision = precision
    a.steps = max(10, min(args.steps, 40))
    try:
        sp, net = make_selfplay(a, sd, 0, 0, precision)
    except Exception as e:   # noqa: BLE001 -- report, never hide
        return {"error": "%s: %s" % (type(e).__name__, e)}
    try:
        pre = preroll(sp, a)
        elapsed, _, d = timed_region(sp, a, lambda: None)
    finally:
        sp.close()
        net.close()
    v = d["expansions"] / elapsed
    label = ("f32x3 (fp16 hi+lo split, 3 MFMAs per k-step, fp32 accumulate; visit counts equal to the reference's fp32 net on the golden searches)"
             if precision == "f32x3" else
             "f16 (fp16 storage, fp32 accumulate: REDUCED precision against the reference's float32 net -- answers within 2e-2, visit "
             "distributions within 0.08 total variation on the golden searches; opt-in, not the headline)")
    return {"net_precision": label,
            "value": v, "unit": "node-expansions/s", "games_per_sec": d["games_finished"] / elapsed,
            "sims_per_sec": d["simulations"] / elapsed, "steps": a.steps, "timed_region_s": elapsed,
            "eval_cache_hit_rate": d["eval_cache_hits"] / max(1, d["eval_cache_probes"]),
            "throughput_vs_headline": v / main_line["value"], "preroll_s": pre["preroll_s"]}


def profile_eager(sp, args):
    """Event-timed eager rollout steps (standalone tree kernel + standalone net kernel) for the
    secondary per-kernel rooflines; runs after the timed region."""
    import torch
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
    net_step_ms = net_ms
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
        net_ms = (ea.elapsed_time(eb) - ev_overhead_ms) / reps
    psims = p1["simulations"] - p0["simulations"]
    pdepth = (p1["depth_sum"] - p0["depth_sum"]) / max(1, psims)
    sims_per_launch = psims / len(ev)
    net_leaves = (p1["leaf_evals"] - p0["leaf_evals"] - (p1["eval_cache_hits"] - p0["eval_cache_hits"])) / len(ev)
    return dict(tree_ms=tree_ms, net_ms=net_ms, net_step_ms=net_step_ms, ev_overhead_ms=ev_overhead_ms, sims_per_launch=sims_per_launch,
                mean_depth=pdepth, tree_bytes_per_launch=tree_bytes_per_sim(pdepth) * sims_per_launch, net_leaves=net_leaves)


def secondary_rooflines(prof, args, pmc):
    ach = prof["tree_bytes_per_launch"] / (prof["tree_ms"] * 1e-3) / 1e9
    pmc_ok = args.slots == pmc.get("slots") and args.sims == pmc.get("sims") and args.max_inner == pmc.get("max_inner")
    tree = {
        "kernel": "c4_step_kernel<EXTERNAL_F32> (tree walk: apply+backup, PUCT descent, expand, move choice, emit)",
        "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": ach / HBM_PEAK_GBPS,
        # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --pmc-mode`
        # (profiles/r02_pmc_traffic.json), KB -> bytes with the guide's gfx950 correction (FETCH_SIZE x 2 for
        # 16-byte-per-lane loads, WRITE_SIZE as is; MI355X_MICROARCH.md section HBM); not collectable inside this process
        "traffic": ((2.0 * pmc["FETCH_SIZE_tree"] + pmc["WRITE_SIZE_tree"]) * 1024.0 if pmc_ok and "FETCH_SIZE_tree" in pmc else None),
        "avg_launch_ms": prof["tree_ms"], "event_overhead_ms_subtracted": prof["ev_overhead_ms"],
        "sims_per_launch": prof["sims_per_launch"], "mean_depth": prof["mean_depth"],
        "algorithmic_bytes_per_launch": prof["tree_bytes_per_launch"],
        "note": "dependent-load (latency) bound pointer chase; bytes = (136*D+332) per simulation",
    }
    tf = args.net_mflop * 1e6 * args.slots / (prof["net_ms"] * 1e-3) / 1e12
    peak = FP32_MATRIX_PEAK_TF if (args.net != "fused" and args.net_dtype == "f32") else F16_MFMA_PEAK_TF
    netr = {
        "kernel": ("c4_net_kernel (fused stem+tower+heads, v_mfma_f32_32x32x16_f16, precision %s)" % args.net_precision if args.net == "fused"
                   else "leaf-batch policy/value net forward (PyTorch-ROCm / MIOpen convs)"),
        "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
        "avg_forward_ms": prof["net_ms"], "avg_ms_in_step_incl_dispatch_gap": prof["net_step_ms"],
        "traffic": ((2.0 * pmc["FETCH_SIZE_net"] + pmc["WRITE_SIZE_net"]) * 1024.0
                    if pmc_ok and args.net == "fused" and args.net_precision == pmc.get("net_precision", "f16") and "FETCH_SIZE_net" in pmc else None),
        "positions_per_launch": args.slots, "leaves_needing_the_net_per_launch": prof["net_leaves"],
        "note": "achieved counts every row the kernel computes (static batch); slots whose simulation "
                "ended on a terminal or cached leaf still occupy a row",
    }
    return {"roofline_tree": tree, "roofline_net": netr}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become the launcher.  Nothing in this process has touched the GPU
        # (no torch import, no HIP call), and the ranks are children -- this process is never replaced.
        return launch_ranks(args, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
