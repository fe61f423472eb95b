cd $GRAFT_REPO_ROOT
BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_post_times.so python tools/post_stamps.py 2>&1 | tail -12
BFA_HIP_LIBRARY=$PWD/bournemouth-forced-aligner_amd/variants/libbfa_post_times.so python tools/post_stamps.py --ragged 2>&1 | tail -12
bash tools/r6_mixclass.sh 2>&1 | tail -60
