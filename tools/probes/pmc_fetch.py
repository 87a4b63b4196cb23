import csv, sys, collections, glob
path = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
for x in csv.DictReader(open(path)):
    k = x["Kernel_Name"][:44]
    acc[k][x["Counter_Name"]] += float(x["Counter_Value"])
    if x["Dispatch_Id"] not in seen:
        seen.add(x["Dispatch_Id"]); n[k]+=1; acc[k]["ns"] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
for k,c in acc.items():
    if "rocclr" in k: continue
    print(k, "calls", n[k], "avg us %.1f"%(c["ns"]/n[k]/1e3), {kk: "%.3g"%(v/n[k]) for kk,v in c.items() if kk!="ns"})
