#!/bin/bash
# Round 6: effective shader clock of the level-0 wide conv on random vs zero-filled operands = GRBM_GUI_ACTIVE / kernel duration
# (MI355X_MICROARCH.md "DVFS give-back"), from one rocprofv3 --pmc pass per fill (profiled passes clock a little lower than
# un-profiled ones: compare the two fills with each other, not with the un-profiled durations).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-clockpmc}; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for fill in randn zeros; do
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT/pmc_$fill -o p --output-format csv -- python $R/tools/bench_ops.py conv 64 320 256 16 320 12 $fill > $OUT/pmc_$fill.log 2>&1
  python - <<PY
import csv, glob
cc = glob.glob("$OUT/pmc_$fill/*counter_collection.csv")
kt = glob.glob("$OUT/pmc_$fill/*kernel_trace.csv")
rows = list(csv.DictReader(open(cc[0])))
if rows:
    print("counter csv columns:", list(rows[0].keys()))
dur = {}
for r in csv.DictReader(open(kt[0])):
    if "conv3x3_wide" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
vals = []
for r in rows:
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "conv3x3_wide" in r["Kernel_Name"]:
        d = dur.get(r["Dispatch_Id"])
        if d:
            vals.append((float(r["Counter_Value"]), d))
with open("$OUT/clock_summary.txt", "a") as f:
    for c, d in vals[2:]:
        pass
    if vals:
        import statistics
        clk = [c / d for c, d in vals[2:]]          # cycles per ns = GHz
        f.write("fill %-6s: %d launches, median duration %.1f us, median GRBM_GUI_ACTIVE %.0f cycles -> effective clock %.3f GHz\n"
                % ("$fill", len(vals), statistics.median(d for c, d in vals[2:]) / 1e3, statistics.median(c for c, d in vals[2:]), statistics.median(clk)))
PY
  rm -rf $OUT/pmc_$fill
done
cat $OUT/clock_summary.txt
