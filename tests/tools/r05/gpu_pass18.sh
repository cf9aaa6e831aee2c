set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
python -m pytest tests -m gpu -q -k "pick or n1_cube or rlpick" 2>&1 | tail -40 > $OUT/t18.log
python -m pytest tests/test_gpu_fence.py -m gpu -q -s -k "pick_32768_free" 2>&1 | grep -E "env-steps|assert|^E" | cut -c1-1500 > $OUT/t18_fence.log
tail -6 $OUT/t18.log
