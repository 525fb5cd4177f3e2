#!/bin/bash
# A/B of two builds of libcvmi355.so inside ONE gpurun call (boxes differ by ~10 %): tools/ab/prev.so vs the in-tree library.
# usage: tools/ab/ab.sh [bench args]      (tools/ab/prev.so = the build to compare against; it is not tracked)
set -e
L=circuitvision_amd/libcvmi355.so
[ -f tools/ab/prev.so ] || { echo "tools/ab/prev.so missing: copy the previous build there first" >&2; exit 2; }
cp $L /tmp/new.so
trap 'cp /tmp/new.so $L' EXIT            # whatever happens, the in-tree library is the NEW build again afterwards
one() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d['roofline'] or {}).get('kernel_ms_per_step'))"; }
for i in 1 2; do
  cp tools/ab/prev.so $L; echo -n "prev: "; one "$@"
  cp /tmp/new.so $L;      echo -n "new:  "; one "$@"
done
