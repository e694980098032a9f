for bm in 16 32; do
  EAB_ST_BM_LR=$bm timeout -k 10 200 python bench.py --steps 10 --warmup 3 --pipeline 1 --no-alt --no-c1 --no-next --no-cpu-baseline --per-op gpurun_out/r03_perop_lr$bm.txt > gpurun_out/r03_lr$bm.json 2> gpurun_out/r03_lr.err
done
