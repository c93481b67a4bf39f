# PMC passes (one counter group per run, --kernel-trace only) over tools/prof_cfg5.py <cfg>; prints per-kernel means
set -e
R=$PWD
CFG=${1:-cfg5}
O=$R/gpurun_out/pmc_r2_$CFG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/prof_cfg5.py $CFG > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/prof_cfg5.py $CFG > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/prof_cfg5.py $CFG > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/sq -- python3 $R/tools/prof_cfg5.py $CFG > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/tools/prof_cfg5.py $CFG > $O/sq2.log 2>&1
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
            if "isd::" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                out[k][c] = sum(v) / len(v)
for f in glob.glob(f"{O}/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0]
        if "isd::" in k:
            out[k]["avg_ns"] = float(r["AverageNs"]); out[k]["calls"] = int(r["Calls"])
with open(f"{O}/summary.txt", "w") as fh:
    for k, v in sorted(out.items()):
        line = k + ": " + json.dumps(v, sort_keys=True)
        print(line); fh.write(line + "\n")
PY
find $O -name "*kernel_trace.csv" -size +5M -delete
