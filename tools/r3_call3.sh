#!/bin/bash
# Round-3 call 3: GPU suite with the full-size production-kernel parity tests, finite-batch time lines, headline bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c3
mkdir -p $O
python3 -m pytest tests -m gpu -x -q -s --durations=12 2>&1 | grep -E "passed|failed|Error|error|replayed|GPU train|slowest|s call|s setup" | tee $O/pytest.txt
for cfg in "1200 1200" "1200 600" "1200 300" "8192 4096" "8192 8192" "4096 4096"; do
  set -- $cfg
  python3 tools/gen_profile.py --games $1 --slots $2 2>/dev/null | head -1 | tee -a $O/gen.txt
done
python3 tools/gen_profile.py --games 8192 --slots 4096 --series 1 --poll 32 2>/dev/null > $O/gen_series_8192.txt
python3 tools/gen_profile.py --games 1200 --slots 1200 --series 1 --poll 32 2>/dev/null > $O/gen_series_1200.txt
C4_FUSED_PACK=dense python3 tools/gen_profile.py --games 1200 --slots 1200 2>/dev/null | head -1 | tee -a $O/gen_dense.txt
python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value']/1e6,d['dtype'],d['games_per_sec'],d['eval_cache_hit_rate'],d['roofline']['frac'],d.get('f16_storage_mode',{}).get('value'))"
