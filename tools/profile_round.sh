#!/bin/bash
# One GPU-box pass that produces everything profiles/ carries for a round (run from the repo root through gpurun):
#   bash tools/profile_round.sh r04a          ->  gpurun_out/<tag>/ : bench JSON (plain and under rocprofv3), kernel stats CSV, PMC passes,
#                                                  pmc.json + <tag>_pmc.txt, OSD workload counts (diagnostic build), issue-rate table
# Copy what is to be judged into profiles/ afterwards (tools/cp them by hand: see profiles/README.md).
TAG=${1:-r04f}
OUT=gpurun_out/$TAG
mkdir -p "$OUT/pmc"
export TMPDIR=/tmp
set -o pipefail
echo "[1] issue rates" && timeout -k 10 300 ./tools/microbench/build/issue_rate > "$OUT/issue_rate.txt" 2>&1
timeout -k 10 120 ./tools/microbench/build/chain_latency > "$OUT/chain_latency.txt" 2>&1
echo "[2] OSD workload counts (timers build)" && timeout -k 10 300 python3 tools/kbench_circuit.py --timers --serial --trials 32768 --counts-out "$OUT/pmc/osd_counts.json" > "$OUT/kbench_circuit_timers.txt" 2>&1
echo "[3] PMC passes" && timeout -k 10 900 bash tools/pmc_passes.sh "$OUT/pmc" > "$OUT/pmc_passes.log" 2>&1
python3 tools/pmc_summarise.py "$OUT/pmc" "$TAG" > "$OUT/pmc_summary.txt" 2>&1
cp profiles/pmc.json "$OUT/pmc.json"; cp "profiles/${TAG}_pmc.txt" "$OUT/${TAG}_pmc.txt"
echo "[4] bench (plain)" && timeout -k 10 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "[5] bench under rocprofv3 --kernel-trace --stats" && (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -- python3 "$OLDPWD/bench.py" --no-cpu-baseline > "$OLDPWD/$OUT/bench_under_rocprof.json" 2> "$OLDPWD/$OUT/bench_under_rocprof.err")
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
echo "[6] other configs" && (timeout -k 10 200 python3 bench.py --code bb72 --circuit none --no-cpu-baseline; timeout -k 10 200 python3 bench.py --code bb72 --batch 4096 --steps 200 --circuit none --no-cpu-baseline; timeout -k 10 300 python3 bench.py --code bb288 --p-sweep 0.004,0.005,0.006 --circuit none --no-cpu-baseline;
  echo "# config 2 as BASELINE quotes it: ONE run() call over 2 097 152 shots with batch = 4096 (tools/kbench_batch.py; granule = min_launch of the plan, 0 = the batch taken literally)";
  timeout -k 10 200 python3 tools/kbench_batch.py --batches 4096,8192,32768 --granule=0,32768,262144) > "$OUT/other_configs.txt" 2>&1
rm -rf "$OUT/prof" "$OUT/pmc/sq_a" "$OUT/pmc/sq_b" "$OUT/pmc/sq_c" "$OUT/pmc/fetch" "$OUT/pmc/write" "$OUT/pmc/grbm"
head -c 1500 "$OUT/bench.json"; echo; tail -5 "$OUT/pmc_summary.txt" | cut -c1-400
