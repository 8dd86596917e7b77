O=gpurun_out/r6ae; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do python tools/lk_ab.py 2>&1 | tail -1; SSX_LK_UNFUSED=1 python tools/lk_ab.py 2>&1 | tail -1; done | tee $O/lk_ab.txt
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lkprof -- python $GRAFT_REPO_ROOT/tools/lk_ab.py > /dev/null 2>&1; cd - > /dev/null
F=$(ls /tmp/lkprof/*/*kernel_stats.csv | tail -1); grep -i "k_lk" $F | tee $O/lk_kernel_stats_fused.csv
cd /tmp; SSX_LK_UNFUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lkprof2 -- python $GRAFT_REPO_ROOT/tools/lk_ab.py > /dev/null 2>&1; cd - > /dev/null
F=$(ls /tmp/lkprof2/*/*kernel_stats.csv | tail -1); grep -i "k_lk" $F | tee $O/lk_kernel_stats_unfused.csv
python tools/kernel_resources.py 2>/dev/null | grep -i "k_lk" | head
