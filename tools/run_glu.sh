timeout -k 10 700 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "every_op or e2e or streaming or c1_full or shortest or constructor_variants_vs or random_config or causality or long_utt or golden or fixture" > gpurun_out/$1_t.log 2>&1; tail -4 gpurun_out/$1_t.log
EAB_ST_GLU=0 timeout -k 10 300 python bench.py --no-train --no-cpu-baseline --no-alt > gpurun_out/$1_b0.json 2> gpurun_out/$1_b0.err && \
EAB_ST_GLU=1 timeout -k 10 300 python bench.py --no-train --no-cpu-baseline --no-alt > gpurun_out/$1_b1.json 2> gpurun_out/$1_b1.err
python - $1 <<'PY'
import json,sys
for t in ("b0","b1"):
    try:
        j=json.loads(open(f"gpurun_out/%s_%s.json" % (sys.argv[1] if len(sys.argv)>1 else "glu", t)).read().strip().splitlines()[-1])
    except Exception as e:
        print(t, "fail", e); continue
    nr=j.get("next_rows",{})
    print(t, j["value"], j["ms_per_step"], {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if "ms" in kk}) for k,v in nr.items() if k!="training"}, j.get("latency_b1"), j.get("c1"))
PY
