# PMC passes over the fused extractor at the headline shape, old (16 lanes per row) and new (one row per lane) kernel
set -e
R=$PWD
O=$R/gpurun_out/${1:-pmc_serial}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export ISD_FUSED_SERIAL=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$v -- python3 $R/tools/prof_fused.py > $O/trace$v.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d $O/sq$v -- python3 $R/tools/prof_fused.py > $O/sq$v.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/sqb$v -- python3 $R/tools/prof_fused.py > $O/sqb$v.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC --output-format csv -d $O/sqc$v -- python3 $R/tools/prof_fused.py > $O/sqc$v.log 2>&1 || true
done
cd $R
python - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
with open(f"{O}/summary.txt", "w") as fh:
    for v in "01":
        out = collections.defaultdict(dict)
        for d in ("sq", "sqb", "sqc"):
            for f in glob.glob(f"{O}/{d}{v}/*/*counter_collection.csv"):
                acc = collections.defaultdict(lambda: collections.defaultdict(list))
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"].split("(")[0]
                    if "isd::fused" in k:
                        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k, cs in acc.items():
                    for c, vv in cs.items():
                        out[k][c] = sum(vv) / len(vv)
        for f in glob.glob(f"{O}/trace{v}/*/*kernel_stats.csv"):
            for r in csv.DictReader(open(f)):
                k = r["Name"].split("(")[0]
                if "isd::fused" in k:
                    out[k]["avg_ns"] = float(r["AverageNs"]); out[k]["calls"] = int(r["Calls"])
        for k, vv in sorted(out.items()):
            line = f"ISD_FUSED_SERIAL={v} " + k + ": " + json.dumps(vv, sort_keys=True)
            print(line); fh.write(line + "\n")
PY
find $O -name "*kernel_trace.csv" -size +5M -delete
find $O -name "*.db" -delete
