mkdir -p gpurun_out/r05
tools/profile_round.sh r05/prof100 --proteomes 100 > gpurun_out/r05/prof100.log 2>&1
head -14 gpurun_out/r05/prof100/pmc_traffic.txt | cut -c1-55,70-150
tail -2 gpurun_out/r05/prof100/stream_gaps.txt
python - <<PY
import json
d=json.load(open("gpurun_out/r05/prof100/pmc_traffic.json"))
tot=sum(v["bytes_per_launch"]*v["launches"] for k,v in d.items() if k.startswith("prefilter"))
print("prefilter PMC bytes of the --steps 1 run:", tot/1e9, "GB")
for k,v in sorted(d.items(), key=lambda kv:-kv[1]["bytes_per_launch"]*kv[1]["launches"])[:12]: print(k, v["launches"], round(v["bytes_per_launch"]/1e9,2))
PY
