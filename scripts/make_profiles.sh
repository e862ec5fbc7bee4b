#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box into gpurun_out/profiles/ (copy them into profiles/r05_* afterwards):
#   bench_n1.json               python3 bench.py (default command: f16x3c, calibrated gate, two streams x 50 176 queries for the
#                               headline + a single-stream pass of batch 100 000 for the roofline object)
#   bench_steps20.json          the driver's form of the command (--steps 20 --warmup 5)
#   bench_under_rocprof.json    the default workload on ONE stream (--streams 1: per-launch durations describe one kernel) under
#                               rocprofv3 --kernel-trace --stats
#   bench_kernel_stats.csv      its per-kernel summary (average duration must agree with roofline.avg_launch_ms)
#   pmc_traffic.json            HBM bytes from separate --pmc FETCH_SIZE / WRITE_SIZE passes of ONE step of the same
#                               configuration (same dtype, batch, calibrated routing), gfx950-corrected
#   pmc_mfma.txt                SQ counters of the gate tower's launches (f16 and f16x3), scripts/pmc_gate.sh
#   per_launch_f16x3c.txt       every dispatch of one pass in launch order
# rocprofv3 is always followed directly by `-- python3 <script>` (no env / shell hop).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>> $O/bench_n1.err
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --streams 1 --no-cpu-baseline --no-parity --no-secondary > $O/bench_under_rocprof.json 2> /tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc/pmc_$c && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc/pmc_$c -- python3 $R/bench.py --streams 1 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity --no-secondary > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc/pmc_$c -name "*counter_collection.csv" | head -1); mkdir -p /tmp/pmc/pmc_$c; cp $f /tmp/pmc/pmc_$c/p_counter_collection.csv
done
python3 $R/scripts/summarize_pmc.py /tmp/pmc 100000 $O/pmc_traffic.json f16x3c 100000
(echo "== f16 (the cascade's filter pass)"; bash $R/scripts/pmc_gate.sh 4096 "" f16; echo "== f16x3 (pair K loop: recheck pass and experts)"; bash $R/scripts/pmc_gate.sh 4096 "" f16x3) > $O/pmc_mfma.txt 2>&1
bash $R/scripts/per_launch_trace.sh f16x3c > /dev/null 2>&1; cp $R/gpurun_out/per_launch_f16x3c.txt $O/ 2>/dev/null
head -c 400 $O/bench_n1.json; echo; head -6 $O/bench_kernel_stats.csv; tail -3 $O/pmc_mfma.txt | cut -c1-200
