cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r03_dit3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_dit3/prof -o t -- bash -c "cd $R && python profiles/acq_ab.py 128000" > $R/gpurun_out/r03_dit3/run.log 2>&1
find $R/gpurun_out/r03_dit3 -name "*.db" -delete
f=$(find $R/gpurun_out/r03_dit3 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "oc_" in r["Name"]:
        print("%-90s calls=%5s avg=%9.2f us  %5s%%" % (r["Name"].replace("void gsh::(anonymous namespace)::","").split("(")[0][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
tail -3 $R/gpurun_out/r03_dit3/run.log
