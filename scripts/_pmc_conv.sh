cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export NO_LIBRARY=1
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY"; do
  rm -rf gpurun_out/pmc_conv; mkdir -p gpurun_out/pmc_conv
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_conv -o pmc -- python scripts/bench_lpips_mc.py > gpurun_out/pmc_conv/log.txt 2>&1
  python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_conv/*counter_collection.csv")
if not f:
    print(open("gpurun_out/pmc_conv/log.txt").read()[-1500:])
else:
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "conv3x3" not in k: continue
        k = k.split("::")[1].split("(")[0] + " grid" + r.get("Grid_Size", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    names = sorted({c for k in tot for c in tot[k]})
    print("kernel".ljust(58), *[c[-22:].rjust(24) for c in names])
    for k in sorted(tot, key=lambda k: -sum(n[k].values()))[:6]:
        print(k[:58].ljust(58), *[("%.4g" % (tot[k][c] / max(1, n[k][c]))).rjust(24) for c in names], "launches", max(n[k].values()))
PY
done
