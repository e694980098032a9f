# kernel timeline of the last streaming step: start offsets, durations and gaps (gpurun_out/<tag>_timeline.txt)
TAG=${1:-stream}; CH=${2:-1}; PREC=${3:-f32}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/tools/diag_c1.py > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last step = the kernels after the last gap > 200 us ... steps are back to back; take the last N kernels where N = kernels per step
names = [r["Kernel_Name"] for r in rows]
# find period: the first kernel name of a step is the program's first kernel; use the last occurrence of the most common first name
n = len(rows)
per = None
for p in range(20, 400):
    if n > 3 * p and names[n - p:] == names[n - 2 * p:n - p] == names[n - 3 * p:n - 2 * p]:
        per = p; break
assert per, "no period found"
last = rows[n - per:]
t0 = int(last[0]["Start_Timestamp"]); prev_end = t0
tot = 0
with open(out + "_timeline.txt", "w") as g:
    g.write(f"{per} kernels per step\n")
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        nm = re.sub(r"\(.*", "", r["Kernel_Name"])[:90]
        g.write(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  grid {r.get('Grid_Size_X', '?'):>7} wg {r.get('Workgroup_Size_X', '?'):>4}  {nm}\n")
        tot += e - s; prev_end = e
    g.write(f"span {(prev_end - t0) / 1e3:.2f} us, kernel time {tot / 1e3:.2f} us\n")
print(open(out + "_timeline.txt").read()[-300:])
PY
tail -2 $OUT/log.txt
