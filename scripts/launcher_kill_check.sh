#!/bin/bash
# Dev tool (GPU box): the self-launching parent of `bench.py --gpus N` must take its ranks with it when the driver stops it --
# SIGTERM (forwarded to the ranks' process group) and SIGKILL (PR_SET_PDEATHSIG on the launcher, which sets the same for its ranks).
cd "$GRAFT_REPO_ROOT" || exit 1
for SIG in TERM KILL; do
  python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 > /dev/null 2> gpurun_out/kill_$SIG.err &
  P=$!
  sleep 12
  kill -$SIG $P
  sleep 6
  echo "after SIG$SIG to the parent (pid $P), python processes of the run still alive:"
  ps -eo pid,ppid,stat,etimes,cmd | grep "[p]ython.*bench.py" | cut -c1-150
  echo "(end of list)"
done
