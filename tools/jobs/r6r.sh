O=$PWD/gpurun_out/r6r; mkdir -p $O
export TMPDIR=/tmp
python tools/c5_time.py 200 8 > /dev/null 2>&1   # renders the drive, builds
D=/tmp/ssx_c1_corridor_200
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $OLDPWD/ssvio_amd/host/ssx_run_kitti --config_yaml_path=$D/cfg_c5.yaml --kitti_dataset_path=$D --streams=64 --preload=1 --batched=1 > $O/run.log 2>&1
cd $OLDPWD
python tools/prof_stats.py 2>/dev/null | head -0
f=$(ls $O/trace/*/*kernel_stats.csv | tail -1); head -25 $f | cut -d, -f1-7 | cut -c1-200
tail -4 $O/run.log | cut -c1-300
rm -rf $O/trace
