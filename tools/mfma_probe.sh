# DESIGN 3.2, "band DFT on the matrix cores": what co-issuing its MFMA load costs fused_kernel<float> at cfg 2.
# Three builds of the library (ISD_MFMA_PROBE = 0 / 32 / 64 fp32 16x16x4 MFMAs per band and wave beside the unchanged
# vector code): times the extractor with HIP events, and collects the matrix / vector busy counters for each build.
set -e
R=$PWD
O=$R/gpurun_out/mfma_probe
mkdir -p $O
# the three libraries are built beforehand (no GPU needed):
#   for N in 32 64; do ISD_HIPCC_FLAGS=-DISD_MFMA_PROBE=$N python -c "from isd_amd import build as b; b.build(force=True)";
#     cp imagined-speech-decoding_amd/libisd_hip.so tools/ubench/_probe/lib_$N.so; done    (lib_0.so = the plain build)
cp imagined-speech-decoding_amd/libisd_hip.so $O/lib_keep.so
for N in 0 32 64; do
  cp tools/ubench/_probe/lib_$N.so imagined-speech-decoding_amd/libisd_hip.so
  python tools/bench_features.py > $O/bench_$N.txt 2>&1 || true
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$N -- python3 $R/tools/prof_cfg5.py cfg2 > $O/pmc_$N.log 2>&1) || true
done
cp $O/lib_keep.so imagined-speech-decoding_amd/libisd_hip.so
rm -f $O/lib_keep.so
python - $O <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
for N in (0, 32, 64):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{N}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "fused_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"probe {N}:", json.dumps({k: sum(v) / len(v) for k, v in sorted(acc.items())}))
    for ln in open(f"{O}/bench_{N}.txt"):
        if "cfg2" in ln or "fused" in ln:
            print("   ", ln.strip())
PY
find $O -name "*kernel_trace.csv" -size +2M -delete
