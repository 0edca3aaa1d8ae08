#!/usr/bin/env python3
"""Host side of tools/final_measure.sh: copy / condense gpurun_out/final/* into profiles/r03_*."""
import glob
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "final")
P = os.path.join(R, "profiles")
s = json.load(open(os.path.join(O, "pmc", "summary.json")))
b = json.load(open(os.path.join(O, "bench_final.json")))
cfg = b["config"]
json.dump({"kernel": "c4_selfplay_split_kernel", "slots": cfg["slots_per_gpu"], "sims": cfg["simulations"], "filters": 32, "residuals": 3,
           "max_inner": cfg["max_inner_iters"], "quanta_per_launch": cfg["quanta_per_step"], "time_budget_cycles": cfg["time_budget_cycles"],
           "net_precision": cfg["net_precision"],
           "note": "KB per launch as reported by rocprofv3 --pmc (separate FETCH_SIZE and WRITE_SIZE passes, tools/pmc_passes.sh: `rocprofv3 --kernel-trace "
                   "--pmc <counter> -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --precise-compare 0 --profile-steps 0`, every launch = 256 "
                   "quanta of 80,000 cycles), mean over the second half of the launches (steady state: pre-roll launches are the first half). bench.py reports "
                   "`traffic` = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes: the guide's gfx950 correction for 16-byte-per-lane loads.",
           "FETCH_SIZE_fused": s["FETCH_SIZE"]["mean_per_launch"], "WRITE_SIZE_fused": s["WRITE_SIZE"]["mean_per_launch"],
           "launches_averaged": s["FETCH_SIZE"]["launches_averaged"]}, open(os.path.join(P, "r03_pmc_traffic.json"), "w"), indent=1)
sq = {k: v["mean_per_launch"] for k, v in s.items() if k.startswith("SQ_")}
wc = sq["SQ_WAVE_CYCLES"]
json.dump({"kernel": "c4_selfplay_split_kernel<16, %s, 4>" % cfg["net_precision"],
           "per_launch": "256 quanta x 80,000 shader cycles x 2048 waves (4096 games: 1024 tree waves of 4 slots, 1024 network waves = 512 pairs in the reference-precision mode); SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_INST_* count quad-cycles",
           "command": "bash tools/pmc_passes.sh <dir> (three SQ passes of 8 counters each); python tools/pmc_summary.py <dir>",
           "counters": sq,
           "derived": {"wave_time_waiting_on_waitcnt (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": sq["SQ_WAIT_ANY"] / wc,
                       "wave_time_issue_stalled (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": sq["SQ_WAIT_INST_ANY"] / wc,
                       "wave_time_issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)": sq["SQ_ACTIVE_INST_ANY"] / wc,
                       "  of which VALU": sq["SQ_ACTIVE_INST_VALU"] / wc, "  of which scalar": sq["SQ_ACTIVE_INST_SCA"] / wc,
                       "  of which LDS": sq["SQ_ACTIVE_INST_LDS"] / wc,
                       "simd_issue_occupancy (2 waves per SIMD x issuing share)": 2 * sq["SQ_ACTIVE_INST_ANY"] / wc,
                       "valu_pipe_busy (2 waves x VALU share)": 2 * sq["SQ_ACTIVE_INST_VALU"] / wc,
                       "mfma_pipe_busy (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x 256 x 80000 cycles))": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * 256 * 80000.0),
                       "valu_instructions_per_simulation (incl. MFMA)": sq["SQ_INSTS_VALU"] / b["roofline"]["sims_per_launch"],
                       "f64_share_of_valu_instructions": (sq["SQ_INSTS_VALU_FMA_F64"] + sq["SQ_INSTS_VALU_MUL_F64"] + sq["SQ_INSTS_VALU_ADD_F64"]) / sq["SQ_INSTS_VALU"],
                       "lds_bank_conflict_share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)": sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]}},
          open(os.path.join(P, "r03_pmc_sq_fused.json"), "w"), indent=1)
for src, dst in (("bench_final", "r03_bench_final"), ("bench_driver_args", "r03_bench_driver_args"), ("bench_8192", "r03_bench_8192_slots"),
                 ("bench_8192x3200", "r03_bench_8192x3200"), ("bench_64f", "r03_bench_64f_6res"), ("bench_f16", "r03_bench_f16_storage"),
                 ("generation", "r03_generation_1gpu_share"), ("generation_1200", "r03_generation_1200_games")):
    if os.path.exists(os.path.join(O, src + ".json")):
        shutil.copy(os.path.join(O, src + ".json"), os.path.join(P, dst + ".json"))
shutil.copy(max(glob.glob(os.path.join(O, "prof", "*", "*kernel_stats.csv")), key=os.path.getmtime), os.path.join(P, "r03_kernel_stats_final.csv"))   # the newest run
t = json.load(open(os.path.join(P, "r03_pmc_traffic.json")))
for name in ("r03_bench_final", "r03_bench_driver_args"):
    fn = os.path.join(P, name + ".json")
    dd = json.load(open(fn))
    dd["roofline"]["traffic"] = (2.0 * t["FETCH_SIZE_fused"] + t["WRITE_SIZE_fused"]) * 1024.0
    dd["roofline"]["traffic_source"] = "profiles/r03_pmc_traffic.json (PMC passes of the same tools/final_measure.sh run, collected after this line was printed)"
    json.dump(dd, open(fn, "w"))
b = json.load(open(os.path.join(P, "r03_bench_final.json")))
r = b["roofline"]
print("bench_final: %.1f M exp/s, %.1f M sims/s, %.0f games/s, hit %.3f" % (b["value"] / 1e6, b["sims_per_sec"] / 1e6, b["games_per_sec"], b["eval_cache_hit_rate"]))
print("roofline:", {k: r[k] for k in ("achieved", "frac", "traffic", "avg_launch_ms", "sims_per_launch", "algorithmic_bytes_per_launch", "mfma_achieved_tflops")})
print("f16 storage mode:", b.get("f16_storage_mode", {}).get("value"), "cpu:", b["cpu_baseline"]["value"], "preroll:", b["preroll_s"], b["preroll_games"])
d = json.load(open(os.path.join(P, "r03_pmc_sq_fused.json")))["derived"]
print(json.dumps(d, indent=1))
print(open(os.path.join(P, "r03_kernel_stats_final.csv")).read().splitlines()[1][:200])
