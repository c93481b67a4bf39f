# FAST (raw-EEG model) profiles of round 2: full GPU suite, tools/bench_fast.py tables, rocprofv3 kernel table of train_head bf16 B=4096
set -e
R=$PWD
O=$R/gpurun_out/prof_r2b
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
tail -2 $O/gpu_tests.log
python tools/bench_fast.py > $O/bench_fast.txt 2>&1 || true
python tools/bench_fast.py --heads > $O/bench_fast_heads.txt 2>&1 || true
cd /tmp && export TMPDIR=/tmp
export ISD_PROF_ACT=bf16 ISD_PROF_B=4096
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast_bf16 -- python3 $R/tools/prof_fast.py 6 > $O/fast_bf16.log 2>&1 || true
f=$(ls $O/fast_bf16/*/*kernel_stats.csv | tail -1); cp $f $O/fast_train_head_bf16_kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +5M -delete
cd $R
ls $O
