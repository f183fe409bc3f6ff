#!/bin/bash
# Developer run: cycle shares of k_regions per region size class (RTK_TRACE_CLASS makes the kernel skip the other classes: output invalid, timing only).
for c in 0 2 3 4; do
  echo "== class $c"
  RTK_TRACE=1 RTK_TRACE_CLASS=$c timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-legs --serial 2>&1 >/dev/null | grep "size class\|fine shares\|DFS book\|shares of\|k_regions attempt" | tail -5
done
