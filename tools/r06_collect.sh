#!/bin/bash
# round 6: final artefacts, part 2 (in the container, after `gpurun -- bash tools/r06_final.sh` merged its files into gpurun_out/): copy
# them to profiles/r06_* and write the freshness sidecars bench.py checks (tools/profile_meta.py: content hashes of the sources each
# profile depends on).  Then part 3: gpurun -- 'python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err'.
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
for t in replay_only register_only; do
  cp $G/prof_r06/kernel_stats_$t.csv $P/r06_kernel_stats_$t.csv
  cp $G/prof_r06/bench_${t}_under_rocprof.json $P/r06_bench_${t}_under_rocprof.json
done
cp $G/prof_r06/kernel_stats_train.csv $P/r06_train_kernel_stats.csv
cp $G/prof_r06/kernel_stats_train_geo.csv $P/r06_train_geo_kernel_stats.csv
cp $G/prof_r06/kernel_stats_train_geo_c5.csv $P/r06_train_geo_c5_kernel_stats.csv
for t in train train_geo train_geo_c5; do cp $G/prof_r06/bench_${t}_under_rocprof.json $P/r06_bench_${t}_under_rocprof.json; done
cp $G/r06_phases_f32.txt $P/
for f in r06_pmc_path.json r06_pmc_train.json r06_pmc_train_geo.json r06_pmc_train_geo_c5.json; do cp $G/$f $P/$f; done
cp $G/pmc_bf16_c3.json $P/r06_pmc_bf16_c3.json
cp $G/pmc_bf16_c1.json $P/r06_pmc_bf16_c1.json
C=cmr_agent_amd/csrc
python3 tools/profile_meta.py $P/r06_kernel_stats_replay_only.csv $C/conv_wino.hip bench.py cmr_agent_amd/runtime.py cmr_agent_amd/ops.py
python3 tools/profile_meta.py $P/r06_pmc_path.json $C/conv_wino.hip bench.py
python3 tools/profile_meta.py $P/r06_pmc_train.json $C/bn_linear.hip $C/train.hip $C/wgrad.hip $C/conv_bf16.hip cmr_agent_amd/train/agent_update.py bench.py
python3 tools/profile_meta.py $P/r06_pmc_train_geo.json $C/conv_wino.hip $C/wgrad_wino.hip $C/bn_linear.hip $C/train_geo.hip cmr_agent_amd/train/geo_update.py bench.py
python3 tools/profile_meta.py $P/r06_pmc_train_geo_c5.json $C/conv_wino.hip $C/wgrad_wino.hip $C/bn_linear.hip $C/train_geo.hip cmr_agent_amd/train/geo_update.py bench.py
python3 tools/profile_meta.py $P/r06_pmc_bf16_c3.json $C/conv_bf16.hip bench.py
python3 tools/profile_meta.py $P/r06_pmc_bf16_c1.json $C/conv_bf16.hip bench.py
ls $P | grep r06 | wc -l
