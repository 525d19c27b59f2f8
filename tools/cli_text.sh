#!/bin/bash
# The driver with every sample row as text, stdout to /dev/null: formatted on the host from the rows (default) and on the
# GPU through the text sink (--stream-text).  usage: tools/cli_text.sh   (writes profiles-style lines to stdout)
cd "$(dirname "$0")/.."
make -C mcpar_amd/drivers > /dev/null 2>&1 || { echo "driver build failed"; exit 1; }
echo "# mcpar-run --func rosen1 --np 16 --nc 65536 --nburn 500, stdout to /dev/null (one MI355X box, 16 host cores); last line of stderr"
for NS in 100 1000; do
  for mode in "" "--stream-text"; do
    t0=$(date +%s%N)
    mcpar_amd/drivers/mcpar-run --func rosen1 --np 16 --nc 65536 --nburn 500 --nsamp $NS $mode > /dev/null 2> /tmp/mcx_cli.err
    t1=$(date +%s%N)
    echo "nsamp=$NS mode=[${mode:-rows, host threads format}] wall=$(( (t1 - t0) / 1000000 )) ms (process start to exit) | $(grep chain-steps /tmp/mcx_cli.err | tail -1)"
  done
done
