#!/bin/bash
# On the GPU box, from the repo root: every measurement profiles/ is built from (tools/refresh_profiles.py turns
# the outputs under gpurun_out/final into profiles/r03_*).  About 4 minutes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final
mkdir -p $O
python3 bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args.json 2>/dev/null
python3 bench.py --slots 8192 --no-cpu-baseline --precise-compare 0 --profile-steps 0 --steps 100 > $O/bench_8192.json 2>/dev/null
python3 bench.py --slots 8192 --sims 3200 --no-cpu-baseline --precise-compare 0 --profile-steps 0 --steps 100 > $O/bench_8192x3200.json 2>/dev/null
python3 bench.py --filters 64 --residuals 6 --fc-layers 6 --no-cpu-baseline --precise-compare 0 --profile-steps 0 --steps 60 > $O/bench_64f.json 2>/dev/null
python3 bench.py --net-precision f16 --no-cpu-baseline --precise-compare 0 --profile-steps 0 --steps 100 > $O/bench_f16.json 2>/dev/null
for f in bench_final bench_driver_args bench_8192 bench_8192x3200 bench_64f bench_f16; do python3 -c "
import json;d=json.load(open('$O/$f.json'));r=d['roofline'];print('$f %s %.1fM exp/s %.1fM sims/s %.0f games/s hit %.3f frac %.4f mfma %.3f launch %.3f ms' % (d['dtype'],d['value']/1e6,d['sims_per_sec']/1e6,d['games_per_sec'],d['eval_cache_hit_rate'],r['frac'],r['mfma_frac_of_dense_f16_peak'],r['avg_launch_ms']), d.get('f16_storage_mode',{}).get('value'))"; done
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --precise-compare 0 > $O/bench_prof.json 2> $O/bench_prof.err; echo "rocprof rc=$?"
bash tools/pmc_passes.sh $O/pmc > /dev/null && python3 tools/pmc_summary.py $O/pmc --json $O/pmc/summary.json | grep -E "FETCH|WRITE|SQ_INSTS_VALU |SQ_WAIT_ANY"
python3 tools/bench_generation.py > $O/generation.json 2>/dev/null; cat $O/generation.json | cut -c1-400
python3 tools/bench_generation.py --games 1200 --slots 1200 > $O/generation_1200.json 2>/dev/null; cat $O/generation_1200.json | cut -c1-400
