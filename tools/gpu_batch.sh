# scratch driver for one gpurun call
set -e
R=$PWD
O=$R/gpurun_out/r3c
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1
python bench.py --overlap --no-cpu-baseline --no-also --no-hbm-roofline > $O/bench_overlap.json 2> $O/bench.err
python bench.py --overlap --bf16 --no-cpu-baseline --no-also --no-hbm-roofline > $O/bench_overlap_bf16.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --no-also --no-hbm-roofline > $O/bench_plain.json 2>> $O/bench.err
python tools/bench_fast.py --replay f32 > $O/fast_replay.txt 2>&1
python tools/bench_fast.py --replay f32 --torch-adamw > $O/fast_replay_torch_adamw.txt 2>&1
python tools/bench_fast.py --replay bf16 > $O/fast_replay_bf16.txt 2>&1
