# PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, MI355X_MICROARCH.md) over one eager step of the geometric update (160x512 / 65 536
# points) and of the bf16 agent update, side streams off so that a dispatch's counters are its own.
#   -> gpurun_out/pmc_train/{geo,agent}.json  (committed as profiles/r05_pmc_train_geo.json / r05_pmc_train.json / r05_pmc_train_geo_c5.json; bench.py reads them)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_train
rm -rf $O; mkdir -p $O/geo $O/agent $O/c5
export CMR_STREAMS=0 CMR_AGENT_UPDATE_FORK=0
for c in FETCH_SIZE:fetch WRITE_SIZE:write; do
  rocprofv3 --kernel-trace --pmc ${c%%:*} -d $O/geo/${c##*:} --output-format csv -- python3 $R/bench.py --mode train-geo --num-pt 65536 --steps 1 --warmup 1 --eager --no-cpu-baseline > $O/geo/${c##*:}.log 2>&1
  rocprofv3 --kernel-trace --pmc ${c%%:*} -d $O/agent/${c##*:} --output-format csv -- python3 $R/bench.py --mode train --dtype bf16 --steps 1 --warmup 1 --eager --no-cpu-baseline > $O/agent/${c##*:}.log 2>&1
  rocprofv3 --kernel-trace --pmc ${c%%:*} -d $O/c5/${c##*:} --output-format csv -- python3 $R/bench.py --mode train-geo --num-pt 65536 --img 352x1216 --prologue --steps 1 --warmup 1 --eager --no-cpu-baseline > $O/c5/${c##*:}.log 2>&1
  echo ${c%%:*} done
done
python3 $R/tools/pmc_train.py $O/c5 > $O/c5.json
python3 $R/tools/pmc_train.py $O/geo > $O/geo.json
python3 $R/tools/pmc_train.py $O/agent > $O/agent.json
cat $O/geo.json $O/agent.json $O/c5.json
