# round-4 profiles: the driver-style bench line (cfg2 fp32 + "also" cfg3 / cfg5), rocprofv3 kernel stats of the bench
# variants, PMC of the cfg2 feature kernels (serial and lane-scan extractor), the plain two-rank gloo rehearsal line
set -e
R=$PWD
O=$R/gpurun_out/prof_r4
mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_default.json 2> $O/default.err || true
cd /tmp && export TMPDIR=/tmp
for v in "cfg2:" "cfg2_bf16:--bf16" "cfg5:--config cfg5"; do
  name=${v%%:*}; flags=${v#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py --no-cpu-baseline --no-also $flags > $O/${name}_line_profiled.json 2>> $O/${name}.err || true
  f=$(ls $O/$name/*/*kernel_stats.csv | tail -1); cp $f $O/${name}_kernel_stats.csv
done
cd $R
bash tools/pmc_serial.sh pmc_serial_r4 > $O/pmc_cfg2_fused.txt 2>&1 || true
bash tools/pmc_r2.sh cfg2 > $O/pmc_cfg2.txt 2>&1 || true
cp gpurun_out/pmc_r2_cfg2/summary.txt $O/pmc_cfg2_summary.txt || true
python tools/bench_features.py > $O/bench_features.txt 2>&1 || true
ISD_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_line_gloo2_plain.json 2> $O/gloo2.err || true
find $O -name "*kernel_trace.csv" -size +5M -delete
find $O -name "*.db" -delete
ls $O
