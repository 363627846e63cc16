import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=float(sys.argv[2]) if len(sys.argv)>2 else 1
tot=0
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 16]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(5), ("%.1f"%(float(r['AverageNs'])/1e3)).rjust(9), r['Percentage'])
print("total ms/step", sum(float(r['TotalDurationNs']) for r in rows)/1e6/steps)
