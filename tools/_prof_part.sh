set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_part_$1
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $R/tools/prof_part.py $1 > $O/log.txt 2>&1
ls $O/*/ | head -3
