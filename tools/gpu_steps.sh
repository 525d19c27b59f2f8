#!/bin/bash
# Runs the given steps ("seconds::name::command" arguments) one after the other on the GPU box, each under its own
# `timeout -k 10`, logging to gpurun_out/<name>.log.  A step that fails is reported and the next one still runs; a step
# that TIMES OUT (or is killed) ends the session: nothing further is started on a GPU that may be stuck.
mkdir -p gpurun_out
overall=0
for spec in "$@"; do
  secs="${spec%%::*}"; rest="${spec#*::}"; name="${rest%%::*}"; cmd="${rest#*::}"
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name: exit $rc after $(( $(date +%s) - start )) s"
  tail -n 6 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name timed out: stopping the session"; exit 124; fi
  [ $rc -ne 0 ] && overall=1
done
exit $overall
