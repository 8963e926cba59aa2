#!/bin/bash
# Usage (GPU box, repo root): bash tools/final_evidence.sh <prefix>   -> gpurun_out/<prefix>_*
# The round's closing evidence: bench.py default and --frames 300, the default command under rocprofv3
# --kernel-trace --stats, smoke(), and tools/sweep.py over every case at 300 and 1,200 frames per launch.
set -u
P=${1:-final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
# (a line a minute: the harness takes three silent minutes for a hang, and the profiled bench writes nothing until it ends)
( while sleep 45; do echo "[final_evidence] $(date +%T) running"; done ) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
timeout -k 10 300 python bench.py > $O/${P}_bench.json 2> $O/${P}_bench.err || { echo "bench failed"; tail -n 5 $O/${P}_bench.err; exit 1; }
timeout -k 10 300 python bench.py --frames 300 > $O/${P}_bench300.json 2> $O/${P}_bench300.err || { echo "bench 300 failed"; exit 1; }
(cd /tmp && TMPDIR=/tmp timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_benchprof -- python3 $R/bench.py --no-cpu-baseline > $O/${P}_bench_under_rocprof.json 2> $O/${P}_benchprof.err) || { echo "rocprofv3 bench failed"; exit 1; }
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
: > $O/${P}_sweep.txt
for c in portrait 4k 720p web stretch thumb thumbref thumbref_portrait thumbref_4k half third portrait_rgb24 portrait_rgba thumbref_rgb24 web_rgb24 cvt_rgb24 cvt_rgba cvt_rgbin portrait_rgbin stretch_rgbin web_rgba 4k_rgb24 half_rgb24 half_rgbin web_rgbin thumb_rgbin cvt_bgrin thumbref_yuvj thumbref_yuvj_inscale half_yuvj_inscale portrait_yuvj_inscale thumbref_rgbin thumbref_rgbin_yuvj thumbref_rgbin_inctx thumbref_rgbain thumbref_portrait_rgbin portrait_rgbin_inctx portrait_rgbin_yuvj cvt_rgbin_yuvj cvt_rgbain; do
  for fr in 300 1200; do
    timeout -k 10 100 python tools/sweep.py --case $c --frames $fr --segs 0 --ns 8 2>&1 | grep SEG_ROWS | sed "s/^/$c $fr /" >> $O/${P}_sweep.txt || exit 1
  done
done
cut -c1-240 $O/${P}_bench.json
cat $O/${P}_sweep.txt
