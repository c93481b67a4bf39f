# round-3 profiles: the driver-style bench line (cfg2 fp32 + "also" cfg3 / cfg5), rocprofv3 kernel stats of the bench
# variants, PMC of the cfg2 and cfg5 feature kernels, the FAST tables, the gloo two-rank rehearsal line
set -e
R=$PWD
O=$R/gpurun_out/prof_r3
mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_default.json 2> $O/default.err || true
cd /tmp && export TMPDIR=/tmp
for v in "cfg2:" "cfg2_bf16:--bf16" "cfg5:--config cfg5"; do
  name=${v%%:*}; flags=${v#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py --no-cpu-baseline --no-also $flags > $O/${name}_line_profiled.json 2>> $O/${name}.err || true
  f=$(ls $O/$name/*/*kernel_stats.csv | tail -1); cp $f $O/${name}_kernel_stats.csv
done
cd $R
bash tools/pmc_r2.sh cfg2 > $O/pmc_cfg2.txt 2>&1 || true
cp gpurun_out/pmc_r2_cfg2/summary.txt $O/pmc_cfg2_summary.txt || true
bash tools/pmc_r2.sh cfg5 > $O/pmc_cfg5.txt 2>&1 || true
cp gpurun_out/pmc_r2_cfg5/summary.txt $O/pmc_cfg5_summary.txt || true
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast_replay -- python3 $R/tools/bench_fast.py --replay f32 > $O/fast_replay.log 2>&1; cp $(ls $O/fast_replay/*/*kernel_stats.csv | tail -1) $O/fast_replay_b64_kernel_stats.csv) || true
python tools/bench_fast.py > $O/bench_fast.txt 2>&1 || true
python tools/bench_fast.py --heads > $O/bench_fast_heads.txt 2>&1 || true
python tools/bench_features.py > $O/bench_features.txt 2>&1 || true
ISD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_line_gloo2.json 2> $O/gloo2.err || true
find $O -name "*kernel_trace.csv" -size +5M -delete
ls $O
