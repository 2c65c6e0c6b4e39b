# round 6: counter evidence for the bf16 inference lines.  For c3 (BASELINE configs[3]) and c1 in bf16 mode:
# tools/bf16_maps.py alone (HIP-event time per launch), then under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes,
# MI355X_MICROARCH.md), joined per map size by tools/bf16_maps_join.py -> gpurun_out/pmc_bf16_<w>.json
# gpurun --timeout 900 -- 'bash tools/_pmc_path_bf16.sh'
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in ${1:-c3 c1}; do
  O=$R/gpurun_out/pmc_bf16_$w
  rm -rf $O; mkdir -p $O
  python3 $R/tools/bf16_maps.py --workload $w --out $R/gpurun_out/bf16_maps_$w.json
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/tools/bf16_maps.py --workload $w --out $O/calls_fetch.json > $O/fetch.log 2>&1
  echo $w fetch done
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/tools/bf16_maps.py --workload $w --out $O/calls_write.json > $O/write.log 2>&1
  echo $w write done
  python3 $R/tools/bf16_maps_join.py $R/gpurun_out/bf16_maps_$w.json $O > $R/gpurun_out/pmc_bf16_$w.json
  python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/pmc_bf16_$w.json"))
print("$w: %d launches, %.3f ms alone, blended frac %.3f, traffic / algorithmic %s" % (d["launches_per_iteration"], d["conv_ms_alone_per_iteration"], d["frac_blended"], d["traffic_over_algorithmic"]))
for m in d["maps"]:
    print("  %-9s B%d %3d->%3d s%d p%d io%d%d res%d x%3d  %7.1f us  alg %7.2f MB  hbm %s MB  %6.0f GB/s  frac %.3f  share %.3f  %s wgs %s" % (
        m["map"], m["B"], m["cin"], m["cout"], m["stride"], m["pool"], m["x_bf16"], m["y_bf16"], m["residual"], m["launches"], m["us_alone"], m["algorithmic_mb"],
        m["hbm_mb"], m["gbs"], m["frac_hbm"], m["share_of_conv_time"], (m["kernel"] or "")[:44], m["workgroups"]))
PY
done
