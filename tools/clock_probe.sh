#!/bin/bash
# Samples the GPU's shader clock and socket power (rocm-smi) while a command runs: bash tools/clock_probe.sh <out file> <command...>
out=$1; shift
"$@" > $out.cmd.log 2>&1 &
pid=$!
: > $out
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $out
  echo >> $out
  sleep 0.2
done
wait $pid
