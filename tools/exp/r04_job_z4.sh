# round 4, last session: where does the device wait inside top-k queries run one at a time? (kernel trace, all streams)
mkdir -p gpurun_out
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
HIP_FORCE_DEV_KERNARG=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/gt_topk -- python3 $root/tools/prof_topk_single.py > $root/gpurun_out/s3_gap_topk.log 2>&1
cd $root
python3 tools/exp/gap_timeline.py /tmp/gt_topk >> gpurun_out/s3_gap_topk.log 2>&1
grep -v "^W2\|^E2\|simple_timer" gpurun_out/s3_gap_topk.log | tail -34
