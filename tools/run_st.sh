timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "every_op or e2e or streaming_equals or c1_full or shortest or constructor_variants_vs or random_config or causality" > gpurun_out/$1_t.log 2>&1; tail -4 gpurun_out/$1_t.log
timeout -k 10 200 python tools/diag_st_stamps.py > gpurun_out/$1_stamps.txt 2>&1
BENCH_ARGS="--pipeline 1 --no-next --no-alt --no-c1" bash tools/prof.sh $1_prof > /dev/null 2>&1; tail -1 gpurun_out/$1_prof/log.txt
