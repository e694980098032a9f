for th in 512 256 128 64; do
EAB_ST_GLU_TILES=$th timeout -k 10 300 python bench.py --no-train --no-cpu-baseline --no-alt --no-next --no-roofline > gpurun_out/glu2_$th.json 2> gpurun_out/glu2_$th.err || exit 1
python - $th <<'PY'
import json,sys
j=json.loads(open("gpurun_out/glu2_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], j["value"], j["ms_per_step"], j.get("single_utterance_c1"))
PY
done
