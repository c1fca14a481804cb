mkdir -p gpurun_out/r05
tools/ubench/_build/relay_latency > gpurun_out/r05/relay_latency.txt 2>&1; cat gpurun_out/r05/relay_latency.txt
for cfg in "base:" "cert:VISMA_ICP_COOP_KERNEL=cert" "fold:VISMA_ICP_SOLVE_IN_FOLD=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r05/c3_$name.json 2>gpurun_out/r05/c3_$name.err < /dev/null
  env $envs python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05/c5_$name.json 2>gpurun_out/r05/c5_$name.err < /dev/null
done
python - <<'PY'
import json
for w in ("c3","c5"):
    for n in ("base","cert","fold"):
        try:
            d=json.load(open("gpurun_out/r05/%s_%s.json"%(w,n))); print(w,n,round(d["value"]),round(d["ms_per_step"],3),d.get("registrations_per_sec"))
        except Exception as e: print(w,n,"ERR",e)
PY
