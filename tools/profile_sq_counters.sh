export TMPDIR=/tmp
export VGPA_HEAD=${VGPA_HEAD:-$(cat vgpa_amd/_tree.txt 2>/dev/null)}      # the commit of the measured tree (tools/stamp_tree.sh), into every summary header
BATCH=${1:-512}      # usage: bash tools/profile_sq_counters.sh [batch]
echo "# tree $VGPA_HEAD, bench.py --batch $BATCH"
rocprofv3 --list-avail 2>/dev/null | grep -o -E "SQ_(LDS|VALU_MFMA|INSTS_VALU_MFMA|WAIT|ACTIVE_INST|WAVE_CYCLES|BUSY_CY|INST_CYCLES)[A-Z0-9_]*" | sort -u > gpurun_out/pmc_avail.txt
cat gpurun_out/pmc_avail.txt | tr '\n' ' '
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/prof_$tag
  rocprofv3 --pmc $set -d gpurun_out/prof_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-problem --no-config5 --no-config4 --no-config2 --batch $BATCH > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
  DB=$(find gpurun_out/prof_$tag -name "*results.db" | head -1)
  python3 - "$DB" <<'PY'
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for did, kname, cname, val in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
    acc[kname][(did, cname)].append(float(val))
out = collections.defaultdict(lambda: collections.defaultdict(list))
for k, dd in acc.items():
    per = collections.defaultdict(float)
    for (did, c), vals in dd.items():
        per[(did, c)] += sum(vals)
    for (did, c), v in per.items():
        out[k][c].append(v)
for k, cc in out.items():
    if "k_ode" in k or "k_energy" in k or "k_grad" in k:
        print(k[:70], {c: round(sum(v) / len(v)) for c, v in cc.items()})
PY
  rm -rf gpurun_out/prof_$tag
done
