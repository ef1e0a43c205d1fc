import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in sys.argv[2:]):
        print(f'{float(r["AverageNs"]) / 1e3:8.1f} us x{r["Calls"]:>5}  min {float(r["MinNs"]) / 1e3:6.1f}  {n[:110]}')
