#!/bin/bash
# Developer A/B on the GPU box: tests/quick_async.py at the given sizes for each library (interleaved, two rounds each)
# usage (inside gpurun): tools/ab.sh "32" libA.so libB.so ...
sizes="$1"; shift
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (rep $rep)"
    MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/$lib python tests/quick_async.py $sizes 2>&1 | grep "B="
  done
done
