#!/bin/bash
# Regenerates the judged profile artefacts on the GPU box into gpurun_out/profiles/ (copy them into profiles/ afterwards):
#   bench_n1.json               python3 bench.py (default command)
#   bench_under_rocprof.json    the same command under rocprofv3 --kernel-trace --stats
#   bench_kernel_stats.csv      its per-kernel summary (average duration must agree with roofline.avg_launch_ms)
#   pmc_traffic.json            HBM bytes from separate --pmc FETCH_SIZE / WRITE_SIZE passes (gfx950-corrected)
#   bench_mups_only.json        BASELINE config 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python3 $R/bench.py --mups-only > $O/bench_mups_only.json 2>> $O/bench_n1.err
rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc/pmc_$c && rocprofv3 --pmc $c --output-format csv -d /tmp/pmc/pmc_$c -- python3 $R/bench.py --points 25000 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --uncalibrated-gate > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc/pmc_$c -name "*counter_collection.csv" | head -1); mkdir -p /tmp/pmc/pmc_$c; cp $f /tmp/pmc/pmc_$c/p_counter_collection.csv
done
python3 $R/scripts/summarize_pmc.py /tmp/pmc 25000 $O/pmc_traffic.json
head -c 600 $O/bench_n1.json; echo; head -5 $O/bench_kernel_stats.csv
