# round 5, GPU pass 6: the whole -m gpu suite + smoke on the final tree
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | tail -30 > $OUT/t6.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
tail -6 $OUT/t6.log; tail -2 $OUT/smoke.log
