#!/bin/bash
# Runs GPU steps one after another on the gpurun box; a step that times out or is killed ends the sequence
# (no further GPU step after a hang).  Usage: tools/gpu_steps.sh OUTDIR "name|timeout_s|command" ...
out="$1"; shift
mkdir -p "$out"
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "== $name (limit ${tmo}s)"
  timeout -k 10 "$tmo" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"
  tail -4 "$out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name hit its limit: stopping"; exit $rc; fi
done
exit 0
