# PMC passes (one counter group per run) over tools/prof_fast.py; ISD_PROF_ACT / ISD_PROF_B select the variant
set -e
R=$PWD
O=$R/gpurun_out/pmc_fast_${ISD_PROF_ACT:-f32}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/prof_fast.py 3 > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/prof_fast.py 3 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/prof_fast.py 3 > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 $R/tools/prof_fast.py 3 > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/tools/prof_fast.py 3 > $O/sq2.log 2>&1
cd $R
python - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
out = collections.defaultdict(dict)
for d in ("fetch", "write", "sq", "sq2"):
    for f in glob.glob(f"{O}/{d}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "conv4_fused" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                out[k][c] = sum(v) / len(v)
for f in glob.glob(f"{O}/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0]
        if "conv4_fused" in k:
            out[k]["avg_ns"] = float(r["AverageNs"]); out[k]["calls"] = int(r["Calls"])
with open(f"{O}/summary.txt", "w") as fh:
    for k, v in sorted(out.items()):
        line = k + ": " + json.dumps(v, sort_keys=True)
        print(line); fh.write(line + "\n")
PY
find $O -name "*kernel_trace.csv" -size +5M -delete
