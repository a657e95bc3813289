import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 26
for r in rows[:n]:
    print("%-64s calls %6d total %8.1f ms avg %8.1f us %5.1f%%" % (r["Name"][:64], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                                    float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("total ms %.1f kernels %d" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
