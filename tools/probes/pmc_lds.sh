set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repository copy on the GPU box)}" || exit 1
OUT=gpurun_out/pmclds
mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_LDS\|SQ_WAIT_INST_LDS\|SQ_ACTIVE_INST_LDS" | sort -u > $OUT/avail.txt
cat $OUT/avail.txt | tr '\n' ' '
i=0
for g in "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/p$i -- python3 bench.py --steps 5 --warmup 2 --profile --no-graph --serial > $OUT/pmc_$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (s, n) in sorted(acc.items()):
    if "project_x6" in kn or "gates_x6" in kn or "aggregate_kernel" in kn:
        print(f"{kn},{cn},{s / n:.1f},{n}")
PY
rm -rf $OUT/p1 $OUT/p2
