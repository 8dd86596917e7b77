O=gpurun_out/r6w; mkdir -p $O
export TMPDIR=/tmp
python tools/c1_stages.py run default 2>&1 | tee $O/stages_default.txt
python tools/c1_stages.py run hard 2>&1 | tee $O/stages_hard.txt
CMD=$(python tools/c1_stages.py cmd hard | tail -1)
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/c1trace -- $CMD > /dev/null 2>&1; cd - > /dev/null
F=$(ls /tmp/c1trace/*/*kernel_trace.csv | tail -1)
python tools/c1_stages.py trace $F 2>&1 | tee $O/trace_hard.txt
