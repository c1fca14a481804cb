#!/bin/bash
# GPU box: per-dispatch durations of the kernels that PREPARE a C4 registration (upload expansion, grid build, source
# order) -- rocprofv3 kernel trace of tools/e2e_probe.py, the dispatches of the last C4 upload + run listed in order.
#   bash tools/aux_trace.sh <out-dir under gpurun_out/>
out=${1:-gpurun_out/aux_trace}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$out/raw" -- python "$OLDPWD/tools/e2e_probe.py" > "$OLDPWD/$out/probe.log" 2>&1
cd "$OLDPWD"
python - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/raw/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last launch of the C4-size search kernel's FIRST pass marks a C4 run; print what precedes it back to the previous search launch
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, r in enumerate(rows) if "nn_grid_reduce_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) >= 262144]
last = idx[-1]
j = last - 1
while j >= 0 and "nn_" not in names[j]:
    j -= 1
t0 = int(rows[j + 1]["Start_Timestamp"])
tot = 0
with open(out + "/c4_prepare_dispatches.csv", "w") as o:
    o.write("kernel,start_us,duration_us\n")
    for r in rows[j + 1:last + 1]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        o.write("%s,%.1f,%.1f\n" % (r["Kernel_Name"].split("(")[0][:70].replace(",", ";"), (int(r["Start_Timestamp"]) - t0) / 1e3, d))
print(open(out + "/c4_prepare_dispatches.csv").read())
print("sum of kernel durations %.1f us, span %.1f us" % (tot, (int(rows[last]["End_Timestamp"]) - t0) / 1e3))
PY
