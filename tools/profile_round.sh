#!/bin/bash
# Reproduces the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r04
# Separate passes: kernel-trace stats, then one --pmc pass per counter group (tools/pmc.py / tools/pmc_all.py; never combined with
# sys/hip/hsa traces).  Everything lands in gpurun_out/<tag>/; tools/collect_profiles.py copies the summaries to profiles/.
TAG=${1:-r05}
PART=${2:-all}      # A: profiler passes + bench records, B: tools (one gpurun call holds 20 minutes: run "A" and "B" in two calls)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT/stats $OUT/stats_C5 $OUT/stats_trials
if [ $PART != B ]; then
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-pipeline --no-pmc --no-c5 --no-hits"
# kernel durations of the bench's own command (Cm), of the C5 workload, and of a 64-trial batch of the ycb frame
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 20 --warmup 5 > $OUT/stats/log 2>&1; echo "stats rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C5 -- $B --workload C5 --steps 8 --warmup 3 > $OUT/stats_C5/log 2>&1; echo "stats C5 rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_trials -- python3 $R/tools/trials.py --example ycb_024_bowl --trials 64 --batch 64 > $OUT/stats_trials/log 2>&1; echo "stats trials rc $?"
# counters of every kernel of a Cm trial (single-trial calls) and of a 64-trial batch
timeout -k 10 500 python3 $R/tools/pmc_all.py $OUT/pmc_pipe -- python3 $R/tools/pipeline_time.py Cm 1234 4 > $OUT/pmc_pipeline_Cm.json 2> $OUT/pmc_pipeline_Cm.err; echo "pmc_all rc $?"
timeout -k 10 500 python3 $R/tools/pmc_all.py $OUT/pmc_trials -- python3 $R/tools/trials.py --example synth:Cm --trials 16 --batch 16 > $OUT/pmc_trials_Cm16.json 2> $OUT/pmc_trials_Cm16.err; echo "pmc_all trials rc $?"
timeout -k 10 300 python3 $R/tools/pmc.py "lcp_coopq_kernel<false" $OUT/pmc_lcp -- $B --steps 3 --warmup 1 > $OUT/lcp_pmc.json 2> $OUT/lcp_pmc.err; echo "pmc lcp rc $?"
cd $R
timeout -k 10 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
cp bench_details.json $OUT/bench_details.json 2>/dev/null
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench2.err; echo "bench (the driver's command) rc $?"
cp bench_details.json $OUT/bench_driver_command_details.json 2>/dev/null
timeout -k 10 900 python3 bench.py --workload C5 --steps 40 --warmup 3 --no-pipeline --cpu-seconds 6 > $OUT/bench_C5.json 2> $OUT/bench_C5.err; echo "bench C5 rc $?"
cp bench_details.json $OUT/bench_C5_details.json 2>/dev/null
fi
if [ $PART = A ]; then echo "part A written under $OUT"; exit 0; fi
cd $R
timeout -k 10 300 python3 tools/frame_latency.py 8 --cpu-reference > $OUT/frame_latency.json 2> $OUT/frame.err; echo "frame rc $?"
timeout -k 10 600 python3 tools/sweep.py > $OUT/sweep.json 2> $OUT/sweep.err; echo "sweep rc $?"
for ex in packed_dove ycb_024_bowl linemod_obj_06 synth:Cm; do
  n=${ex#synth:}
  cpu=100; if [ $ex = synth:Cm ]; then cpu=8; fi     # the reference's CPU path beside every whole-path number (oracle, one core; Cm: 8 of the 100 attempts, scaled)
  timeout -k 10 300 python3 tools/trials.py --example $ex --trials 64 --seed 3 > $OUT/trials64_${n}_single.json 2>> $OUT/trials.err
  timeout -k 10 300 python3 tools/trials.py --example $ex --trials 64 --seed 3 --streams 8 > $OUT/trials64_${n}_streams8.json 2>> $OUT/trials.err
  timeout -k 10 400 python3 tools/trials.py --example $ex --trials 64 --seed 3 --batch 64 --cpu-reference $cpu > $OUT/trials64_${n}_batch64.json 2>> $OUT/trials.err
done; echo "trials done"
# round 5: pruned against full candidate lists, the library's sort against rocPRIM, the frame stream, the rocPRIM-sorted pipeline for A/B
timeout -k 10 300 python3 tools/prune_ab.py Cm C5 dense small > $OUT/prune_ab.json 2> $OUT/prune_ab.err; echo "prune A/B rc $?"
timeout -k 10 300 python3 tools/sort_bench.py > $OUT/sort_bench.jsonl 2> $OUT/sort.err; echo "sort bench rc $?"
timeout -k 10 300 python3 tools/frame_latency.py --stream 3 240 1 > $OUT/frame_stream_1_trial.json 2> $OUT/stream.err; echo "frame stream rc $?"
timeout -k 10 300 python3 tools/frame_latency.py --stream 3 120 64 > $OUT/frame_stream_64_trials.json 2>> $OUT/stream.err; echo "frame stream 64 rc $?"
STOCS_SORT=rocprim timeout -k 10 300 python3 tools/trials.py --example synth:Cm --trials 64 --seed 3 --batch 64 > $OUT/trials64_Cm_batch64_rocprim_sort.json 2>> $OUT/trials.err
STOCS_CLASS_FULL_KERNEL=1 timeout -k 10 300 python3 tools/trials.py --example ycb_024_bowl --trials 64 --seed 3 --batch 64 > $OUT/trials64_ycb_024_bowl_batch64_full_class_kernel.json 2>> $OUT/trials.err
timeout -k 10 600 python3 tools/stall_watch.py 3000 5000 0 > $OUT/soak_3000_trials.json 2> $OUT/soak.err; echo "soak rc $?"
timeout -k 10 300 python3 tools/pipeline_time.py Cm 1234 12 > $OUT/pipeline_Cm.json 2> $OUT/pipe.err; echo "pipeline rc $?"
timeout -k 10 300 python3 tools/lcp_cold_warm.py > $OUT/lcp_cold_warm.json 2> $OUT/cw.err; echo "cold/warm rc $?"
timeout -k 10 300 python3 tools/percall_time.py > $OUT/percall.json 2> $OUT/percall.err; echo "percall rc $?"
echo "profiles written under $OUT"
