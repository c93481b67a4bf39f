# PMC pass over tools/prof_cfg5.py cfg5: L1 (TCP) / L2 (TCC) request counters per kernel
set -e
R=$PWD
O=$R/gpurun_out/pmc_tcp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/a -- python3 $R/tools/prof_cfg5.py cfg5 > $O/a.log 2>&1 || tail -5 $O/a.log
cd $R
python - $O <<'PY'
import csv, glob, collections, sys
O=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/a/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "isd::" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    print(k, {c: sum(v)/len(v) for c,v in cs.items()})
PY
