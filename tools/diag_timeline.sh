R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/diag1
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for b in 32 8; do
timeout 600 rocprofv3 --kernel-trace -d $O/prof_$b -o t --output-format csv -- python $R/bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-c3 --no-c5 --no-points --no-roofline > $O/bench_$b.log 2>&1
f=$(find $O/prof_$b -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f $O/timeline_b$b.txt
rm -rf $O/prof_$b
done
cd $R
python tools/bench_conv.py 1920 > $O/bench_conv_1920.txt 2>&1
python tools/bench_conv.py 480 > $O/bench_conv_480.txt 2>&1
tail -3 $O/timeline_b32.txt
