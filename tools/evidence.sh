#!/bin/bash
# everything profiles/ is refreshed from, in one call:  tools/evidence.sh <tag>
TAG=$1
cd $GRAFT_REPO_ROOT
bash tools/prof_bench_1lane.sh $TAG > gpurun_out/evidence_1lane_$TAG.txt 2>&1 && echo "1-lane stats done" &&
bash tools/prof_bench.sh $TAG > gpurun_out/evidence_default_$TAG.txt 2>&1 && echo "default stats + PMC done" &&
timeout -k 10 300 python3 bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err && echo "default bench done" && tail -1 gpurun_out/bench_default_$TAG.json | cut -c1-200 &&
: > gpurun_out/bench_configs_$TAG.jsonl &&
for c in disney_bsdf_test/disney_bsdf.xml:256 veach_mi/mi.xml:512 sponza/sponza.xml:1024; do
  timeout -k 10 300 python3 bench.py --scene scenes/${c%%:*} --spp ${c##*:} --steps 3 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_configs_$TAG.jsonl && echo "config ${c%%:*} done" || exit 1
done
cut -c1-160 gpurun_out/bench_configs_$TAG.jsonl
