"""Timeline of the last dispatches of a rocprofv3 --kernel-trace of acq_long.py: per dispatch its kernel (F forward, C cells, R rows), queue, start and end
(us, relative), so that the overlap of successive batches' kernels can be read off."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
def kind(n):
    return "C" if "oc_cell_kernel" in n else "R" if "oc_rows_kernel" in n else "F" if "oc_forward" in n or "forward" in n else None
d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows if kind(r["Kernel_Name"])]
d.sort()
tail = d[-45:-3]
t0 = tail[0][0]
print("kernel queue   start     end   dur (us)")
for s, e, k, q in tail:
    print("  %s   %6s  %8.1f %8.1f %6.1f" % (k, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
cs = [x for x in tail if x[2] == "C"]
print("C starts %.1f us apart on average; C durations %.1f us" % ((cs[-1][0] - cs[0][0]) / 1e3 / (len(cs) - 1), sum(e - s for s, e, _, _ in cs) / 1e3 / len(cs)))
