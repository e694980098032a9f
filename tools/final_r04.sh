# final measurements of the round (one gpurun call): kernel trace + PMC passes of the inference program, kernel traces of the
# training steps, streaming timeline
set -o pipefail
G="SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAIT_INST_ANY,SQ_WAVE_CYCLES FETCH_SIZE WRITE_SIZE TCC_HIT_sum,TCC_MISS_sum"
bash tools/prof.sh r04_final "$G" > /dev/null 2>&1; tail -3 gpurun_out/r04_final/log.txt
BENCH_ARGS="--train" bash tools/prof.sh r04_train "SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_INSTS_MFMA" > /dev/null 2>&1; tail -2 gpurun_out/r04_train/log.txt
BENCH_ARGS="--train --precision bf16" bash tools/prof.sh r04_train_bf16 > /dev/null 2>&1; tail -1 gpurun_out/r04_train_bf16/log.txt
bash tools/trace_stream.sh r04_stream 1 f32 > /dev/null 2>&1; tail -1 gpurun_out/r04_stream_timeline.txt
python3 $GRAFT_REPO_ROOT/tools/bench_hbm_kernels.py > gpurun_out/r04_hbm_kernels.txt 2>&1; tail -4 gpurun_out/r04_hbm_kernels.txt
