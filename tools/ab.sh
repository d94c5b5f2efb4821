#!/bin/bash
# GPU-box helper: A/B of two builds in ONE call (box-to-box variation is larger than most kernel changes):
# the library in the tree against tools/ab_head/libfreddy_gpu.so.  usage: tools/ab.sh [N] [bench args]
N=${1:-2}; shift
L=postgres-word2vec_amd/libfreddy_gpu.so
cp $L /tmp/new.so
for r in $(seq $N); do
  echo "== new"; cp /tmp/new.so $L; tools/rep.sh 1 "$@"
  echo "== base"; cp tools/ab_head/libfreddy_gpu.so $L; tools/rep.sh 1 "$@"
done
cp /tmp/new.so $L
