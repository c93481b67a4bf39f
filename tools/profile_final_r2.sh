# end-of-round refresh: tools/profile_r2.sh + the FAST head table + the EEGNet-head kernel table at B = 64
set -e
R=$PWD
mkdir -p $R/gpurun_out/prof_r2 $R/gpurun_out/prof_r2b
bash tools/profile_r2.sh > $R/gpurun_out/prof_r2/log.txt 2>&1 || true
python tools/bench_fast.py --heads > $R/gpurun_out/prof_r2b/bench_fast_heads.txt 2>&1 || true
cd /tmp && export TMPDIR=/tmp
export ISD_PROF_HEAD=EEGNet_Encoder
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2b/eeg64 -- python3 $R/tools/prof_default64.py > /dev/null 2>&1 || true
f=$(ls $R/gpurun_out/prof_r2b/eeg64/*/*kernel_stats.csv | tail -1); cp $f $R/gpurun_out/prof_r2b/eegnet_head_b64_kernel_stats.csv
find $R/gpurun_out -name "*kernel_trace.csv" -size +5M -delete
ls $R/gpurun_out/prof_r2 | head -30
