#!/bin/bash
# Two MPI ranks on the box's one GPU (host-staged exchange), C3's shape per rank, every sample row as text: the funnel
# through rank 0 against MCout::text_file (every rank writes its own share at an MPI_Exscan'd offset), into files on /tmp.
# usage: tools/cli_text_mpi.sh [nsamp]   (writes profiles-style lines to stdout)
cd "$(dirname "$0")/.."
make -C mcpar_amd/drivers mpi > /dev/null 2>&1 || { echo "MPI build failed"; exit 1; }
NS=${1:-100}
MPIEXEC=/opt/conda/bin/mpiexec
RUN="mcpar_amd/drivers/mcpar-run-mpi --func rosen1 --np 16 --nc 65536 --nburn 500 --nsamp $NS --stream-text"
echo "# mpiexec -n 2 mcpar-run-mpi --func rosen1 --np 16 --nc 65536 --nburn 500 --nsamp $NS --stream-text (two ranks on one MI355X, 16 host cores); wall = process start to exit"
for mode in funnel side-by-side; do
  rm -f /tmp/mcx_text_$mode.txt
  t0=$(date +%s%N)
  if [ $mode = funnel ]; then $MPIEXEC -n 2 $RUN > /tmp/mcx_text_$mode.txt 2> /tmp/mcx_text_$mode.err
  else $MPIEXEC -n 2 $RUN --out /tmp/mcx_text_$mode.txt > /dev/null 2> /tmp/mcx_text_$mode.err; fi
  t1=$(date +%s%N)
  bytes=$(stat -c %s /tmp/mcx_text_$mode.txt)
  echo "nsamp=$NS ranks=2 mode=[$mode] wall=$(( (t1 - t0) / 1000000 )) ms, $bytes bytes of text | $(grep chain-steps /tmp/mcx_text_$mode.err | tail -1)"
done
cmp /tmp/mcx_text_funnel.txt /tmp/mcx_text_side-by-side.txt && echo "the two files are byte-identical"
rm -f /tmp/mcx_text_funnel.txt /tmp/mcx_text_side-by-side.txt
