#!/bin/bash
# Re-take the round-end evidence on the GPU and copy it into profiles/ ONLY if the call succeeded and the stamp matches the sources:
#   bash tools/refresh_final.sh r04
TAG=${1:-r04}; O=gpurun_out/final_$TAG
rm -rf $O
/usr/local/graft/bin/gpurun --timeout 2400 -- "bash tools/final_profile.sh $TAG" > /tmp/final_$TAG.log 2>&1
grep "gpurun\]" /tmp/final_$TAG.log | cut -c1-160
[ -s $O/pmc_traffic.json ] || { echo "no outputs (call refused or failed): profiles/ left untouched"; exit 1; }
python3 - "$O" "$TAG" <<'PY'
import json, sys, shutil, re
sys.path.insert(0, "."); import bench
O, TAG = sys.argv[1], sys.argv[2]
sha = bench.csrc_sha(); got = json.load(open(O + "/pmc_traffic.json"))["_meta"]["csrc_sha"]
assert sha == got, (sha, got)
for a, b in (("bench_line.json", "bench_line.json"), ("bench_line_selective.json", "bench_line_selective.json"), ("bench_line_rocprof_run.json", "bench_line_rocprof_run.json"),
             ("kernel_stats.txt", "kernel_stats.txt"), ("pmc_traffic.json", "pmc_traffic.json"), ("pmc_sq_attn_gemm.txt", "pmc_sq_attn_gemm.txt")):
    shutil.copy(O + "/" + a, "profiles/%s_final_%s" % (TAG, b))
shutil.copy(O + "/pmc_traffic.json", "profiles/pmc_traffic_current.json")
s = open("DESIGN.md").read()
# only the paragraph of THIS round carries the current hash (earlier rounds keep theirs)
s = re.sub(r"(profiles/%s_final_\*`; PMC file stamped with the final `csrc/` hash )`[0-9a-f]{16}`" % TAG, r"\g<1>`%s`" % sha, s)
open("DESIGN.md", "w").write(s)
d = json.loads(open(O + "/bench_line.json").read().strip().splitlines()[-1])
print("copied; csrc", sha, "|", d["value"], "img/s", d["ms_per_step"], "ms", d["roofline"]["frac"], (d.get("power") or {}).get("watts_avg"), "W e2e", d["e2e"]["extract_images_per_s"])
PY
