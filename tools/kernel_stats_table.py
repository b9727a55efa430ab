import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
steps=float(sys.argv[2])
for r in rows[:int(sys.argv[3])]: print(r["Name"][:100].ljust(100), r["Calls"].rjust(5), ("%.1f"%(float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"].rjust(6), "%.3f ms/step"%(float(r["TotalDurationNs"])/1e6/steps))
