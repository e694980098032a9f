timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "streaming or stream" > gpurun_out/$1_t.log 2>&1; tail -4 gpurun_out/$1_t.log
for c in 0 1; do for ch in 1 16; do for pr in f32 bf16; do
EAB_ST_CHAIN=$c timeout -k 10 120 python tools/diag_stream.py $ch $pr 2>&1 | grep "ms per step" | sed "s/^/chain=$c /" || exit 1
done; done; done
