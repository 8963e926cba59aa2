#!/bin/bash
# Usage: tools/ab_lib.sh <old lib> <out file> <frames> case...   (on the GPU box, repo root): tools/sweep.py on each
# case with the old library and the tree's, alternating, default knobs
OLD=$1; OUT=$2; FR=$3; shift 3
: > $OUT
for c in "$@"; do
  for rep in 1 2; do
    TTS_LIB=$OLD timeout -k 10 120 python tools/sweep.py --case $c --frames $FR --segs 0 --ns 8 2>/dev/null | grep SEG_ROWS | sed "s/^/old $c /" >> $OUT || exit 1
    timeout -k 10 120 python tools/sweep.py --case $c --frames $FR --segs 0 --ns 8 2>/dev/null | grep SEG_ROWS | sed "s/^/new $c /" >> $OUT || exit 1
  done
done
