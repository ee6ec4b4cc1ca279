#!/bin/bash
# usage: tools/ab.sh "bench args" lib1.so lib2.so ...   -- the same bench.py run against several builds of libminipath_hip.so
# (MINIPATH_HIP_SO), default build first.  Prints ms per step and the kernel time of each.
ARGS=$1; shift
for so in "" "$@"; do
  MINIPATH_HIP_SO=$so python bench.py --no-cpu-baseline --no-extension $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('${so:-default}'.ljust(34), 'Mrays/s %9.1f  ms/step %8.3f  kernel_ms %8.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
