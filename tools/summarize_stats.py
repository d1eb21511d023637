import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step: %.2f   launches/step: %.0f" % (tot / 1e6 / steps, sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-60s calls/step %7.1f  ms/step %6.2f  avg %8.1f us %5.1f%%" % (r["Name"][:60], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
