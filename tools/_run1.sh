cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in 1408 7552; do
export FFM_PANEL_MASK=$m
bash tools/prof.sh ks_$m > /dev/null 2>&1
python tools/trace_timeline.py gpurun_out/ks_${m}_kernel_trace.csv > gpurun_out/ks_${m}_timeline.txt 2>&1
rm -f gpurun_out/ks_${m}_kernel_trace.csv
echo "=== mask $m"; head -28 gpurun_out/ks_${m}_timeline.txt | cut -c1-130
done
