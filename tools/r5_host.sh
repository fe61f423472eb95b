#!/bin/bash
# tools/r5_host.sh -- host side of the drop-in calls: api_time, pipeline_time with a host profile, kernel stats of the api loop
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r5h; mkdir -p $OUT; cd $ROOT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
python tools/api_time.py 2>$OUT/api.err | grep "^{" | tail -1 | tee $OUT/api.json
BFA_PROFILE_HOST=1 python tools/pipeline_time.py 4096 > $OUT/pipeline_profile.txt 2>&1
grep "^{" $OUT/pipeline_profile.txt | cut -c 100-330
sed -n '/cumulative/,$p' $OUT/pipeline_profile.txt | head -40
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_api -o t -- python $ROOT/tools/api_time.py > $OUT/st_api.log 2>&1
cd $ROOT
cp $(find $OUT/st_api -name "*kernel_stats.csv" | head -1) $OUT/api_kernel_stats.csv 2>/dev/null; rm -rf $OUT/st_api
cut -d, -f1-4 $OUT/api_kernel_stats.csv | head -14
