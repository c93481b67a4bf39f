# PMC passes (one counter group per run, --kernel-trace only) over tools/prof_stress.py
set -e
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/fetch -- python3 $R/tools/prof_stress.py > $R/gpurun_out/pmc/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/write -- python3 $R/tools/prof_stress.py > $R/gpurun_out/pmc/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc/sq -- python3 $R/tools/prof_stress.py > $R/gpurun_out/pmc/sq.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for d in ("fetch", "write", "sq"):
    for f in glob.glob(f"gpurun_out/pmc/{d}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "fused_long" in k or "fir_kernel" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                out[k][c] = sum(v) / len(v)
for k, v in out.items():
    print(k + ": " + json.dumps(v, sort_keys=True))
PY
