#!/bin/bash
# Developer script (GPU box): the round's closing measurements in one call -- bench lines of every configuration, the rocprofv3 profiles
# (kernel stats + PMC traffic passes), alone-times, the evaluator.  usage: bash tools/dev/final_round.sh <tag>     (e.g. r04)
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$TAG; mkdir -p "$O"
cd $R
python bench.py --steps 200 --warmup 5 > $O/bench_line_c1.json 2> $O/bench_c1.err; tail -c 300 $O/bench_line_c1.json; echo
python bench.py --steps 20 --warmup 5 > $O/bench_line_driver_shape.json 2>> $O/bench_c1.err
for c in c0 c2 c3; do python bench.py --config $c --steps 200 --warmup 5 --no-llh-eval > $O/bench_line_$c.json 2> $O/bench_$c.err; done
python bench.py --precision fp32 --steps 60 --warmup 5 --no-llh-eval > $O/bench_line_c1_fp32.json 2> $O/bench_fp32.err
python bench.py --config c0 --precision fp32 --steps 100 --warmup 5 --no-llh-eval --no-cpu-baseline > $O/bench_line_c0_fp32.json 2>> $O/bench_fp32.err
for c in c2 c3; do python bench.py --config $c --precision fp32 --steps 60 --warmup 5 --no-llh-eval --no-cpu-baseline > $O/bench_line_${c}_fp32.json 2>> $O/bench_fp32.err; done
python bench.py --force-dist --steps 200 --warmup 5 --no-llh-eval --no-cpu-baseline > $O/bench_line_c1_forcedist.json 2> $O/bench_forcedist.err
for f in $O/bench_line_*.json; do python3 - "$f" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-40s ms/step %.4f  value %.0f %s  dominant %s %.1f us frac %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], d["unit"], d["roofline"].get("timed_as"), d["roofline"].get("avg_launch_us", 0), d["roofline"].get("frac")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
bash tools/profile_bench.sh $TAG c1 > $O/profile_c1.txt 2>&1; tail -25 $O/profile_c1.txt
bash tools/profile_bench.sh ${TAG}_c2 c2 quick > $O/profile_c2.txt 2>&1; tail -25 $O/profile_c2.txt
bash tools/dev/alone.sh "bern_pipe|dec_bwd|wgradws|wgrad_rows|reduce_grads|block_fwd|block_bwd|latent|eps_gen" wg16=128 > $O/alone_times.txt 2>&1; cat $O/alone_times.txt
BENCH_ARGS="--config c2" bash tools/dev/alone.sh "chain2|gblock|bern_pipe|dec_bwd|wgradws|wgradp|wgrad_rows|reduce_grads|block_fwd|block_bwd|latent" > $O/alone_times_c2.txt 2>&1; cat $O/alone_times_c2.txt
for p in fp32 bf16; do python tools/dev/eval_only.py $p 4190 >> $O/eval.txt 2>&1; python tools/dev/eval_only.py $p 10000 >> $O/eval.txt 2>&1; done; cat $O/eval.txt
bash tools/dev/trace_f32.sh $TAG > $O/f32_step_timeline.txt 2>&1; tail -3 $O/f32_step_timeline.txt
