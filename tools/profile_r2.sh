# round-2 profiles: rocprofv3 kernel stats of the bench variants + PMC of the cfg2 feature kernels
set -e
R=$PWD
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "cfg2:" "cfg2_bf16:--bf16" "cfg2_two_kernel:--two-kernel" "cfg5:--config cfg5"; do
  name=${v%%:*}; flags=${v#*:}
  python3 $R/bench.py --no-cpu-baseline $flags > $O/${name}_line.json 2> $O/${name}.err || true     # the line: without the profiler
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py --no-cpu-baseline $flags > $O/${name}_line_profiled.json 2>> $O/${name}.err || true
  f=$(ls $O/$name/*/*kernel_stats.csv | tail -1); cp $f $O/${name}_kernel_stats.csv
done
cd $R
bash tools/pmc_r2.sh cfg2 > $O/pmc_cfg2.txt 2>&1 || true
python tools/bench_fast.py > $O/bench_fast.txt 2>&1 || true
python tools/bench_small_batch.py > $O/bench_small_batch.txt 2>&1 || true
find $O -name "*kernel_trace.csv" -size +5M -delete
ls $O
