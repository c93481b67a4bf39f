set -e
R=$PWD
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof/bench_prof_line.json 2>$R/gpurun_out/prof/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/fir -- python3 $R/tools/bench_fir.py > $R/gpurun_out/prof/fir.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stress -- python3 $R/tools/bench_stress.py 128 > $R/gpurun_out/prof/stress.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/fast -- python3 $R/tools/bench_fast.py > $R/gpurun_out/prof/fast.txt 2>&1
cd $R
python bench.py > gpurun_out/prof/bench_line.json 2>gpurun_out/prof/bench.err
python bench.py --two-kernel --no-cpu-baseline > gpurun_out/prof/bench_line_two_kernel.json 2>>gpurun_out/prof/bench.err
python bench.py --bf16 --no-cpu-baseline > gpurun_out/prof/bench_line_bf16.json 2>>gpurun_out/prof/bench.err
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
ls -R gpurun_out/prof | head -40
