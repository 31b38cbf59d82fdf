#!/bin/bash
# A/B of library builds on the GPU box:  tools/ab.sh <scene.xml|-> <spp> <variant> [<variant> ...]
# A variant is a library built beforehand (here, cross-compiled) with
#   LJ_VARIANT=<name> LJ_EXTRA_HIPCC_FLAGS="..." python -m lajolla_public_amd.build
# ("default" = the shipped build).  Prints ms per render (bench.py, 5 steps, no CPU baseline) per variant, two rounds interleaved.
SCENE=$1; SPP=$2; shift 2
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --spp $SPP"
[ "$SCENE" != "-" ] && ARGS="$ARGS --scene $SCENE"
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "default" ]; then unset LJ_VARIANT; else export LJ_VARIANT=$v; fi
    timeout -k 10 300 python3 bench.py $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
k = d['roofline']['all_kernels']
print('$v round $round: %.2f ms/render  %.0f Msamples/s | kernels alone: %s' % (d['ms_per_step'], d['value'], ', '.join('%s %.2f ms' % (n, x['total_ms']) for n, x in k.items())))" || echo "$v FAILED"
  done
done
