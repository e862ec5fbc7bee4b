#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box into gpurun_out/profiles/ (copy them into profiles/r06_* afterwards):
#   bench_n1.json               python3 bench.py (default command: f16x8c, calibrated gate, two streams x 50 176 queries for the
#                               headline + a single-stream pass of batch 100 000 for the roofline object)
#   bench_steps20.json          the driver's form of the command (--steps 20 --warmup 5)
#   bench_under_rocprof.json    the default workload on ONE stream (--streams 1: per-launch durations describe one kernel) under
#                               rocprofv3 --kernel-trace --stats
#   bench_kernel_stats.csv      its per-kernel summary (average duration must agree with roofline.avg_launch_ms)
#   pmc_traffic.json            HBM bytes from separate --pmc FETCH_SIZE / WRITE_SIZE passes of ONE step of the same
#                               configuration (same dtype, batch, calibrated routing), gfx950-corrected
#   pmc_mfma.txt                SQ counters of what the PRODUCT runs: the gate's filter pass (f16x3c with tau = 0:
#                               plain-f16 tap layers + the X2 one-tap loop) at the bench batch's launch shapes, its f16x3 recheck pass, and the
#                               expert towers (one single-scale expert + Expert_6) in f16x8 (FP8 cross-term loop) and f16x3, scripts/pmc_gate.sh
#   per_launch_f16x8c.txt       every dispatch of one pass in launch order
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
python3 $R/scripts/summarize_pmc.py /tmp/pmc 100000 $O/pmc_traffic.json f16x8c 100000
(echo "== f16x3c with tau = 0: the cascade's FILTER pass as the product runs it (plain-f16 tap layers, X2 loop conv_igemm_kernel<..., true> in the one-tap layers), batch 8192"; PROF_ROWS=24 bash $R/scripts/pmc_gate.sh 8192 "" f16x3c
 echo "== f16x3: the gate's RECHECK pass (pair K loop), batch 4096"; PROF_ROWS=24 bash $R/scripts/pmc_gate.sh 4096 "" f16x3
 echo "== expert towers, f16x8 (the 5^3 tap layers in the FP8 cross-term loop conv8n_kernel<2, 5, 3, .> since the FP6 form is the default), 4096 queries: half Expert_0, half Expert_6"; PROF_DRIVER=prof_expert.py PROF_ROWS=40 bash $R/scripts/pmc_gate.sh 4096 "" f16x8
 echo "== expert towers, f16x3 (pair K loop everywhere), the same queries"; PROF_DRIVER=prof_expert.py PROF_ROWS=40 bash $R/scripts/pmc_gate.sh 4096 "" f16x3) > $O/pmc_mfma.txt 2>&1
bash $R/scripts/per_launch_trace.sh f16x8c > /dev/null 2>&1; cp $R/gpurun_out/per_launch_f16x8c.txt $O/ 2>/dev/null
head -c 400 $O/bench_n1.json; echo; head -6 $O/bench_kernel_stats.csv; tail -3 $O/pmc_mfma.txt | cut -c1-200
