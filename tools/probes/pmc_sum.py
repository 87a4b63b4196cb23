import csv, sys, collections, glob
path = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=set()
for x in csv.DictReader(open(path)):
    k = x["Kernel_Name"][:40]
    acc[k][x["Counter_Name"]] += float(x["Counter_Value"])
    if (x["Dispatch_Id"]) not in seen:
        seen.add(x["Dispatch_Id"]); n[k]+=1; acc[k]["ns"] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
for k,c in acc.items():
    gui=c["GRBM_GUI_ACTIVE"]; 
    print(k, "calls", n[k], "avg us %.1f"%(c["ns"]/n[k]/1e3), "clock GHz %.2f"%(gui/8/c["ns"]), "mfma busy %% %.1f"%(100*c["SQ_VALU_MFMA_BUSY_CYCLES"]/(128*gui)),
          "wait_any %% %.1f"%(100*c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"]), "wait_inst %% %.1f"%(100*c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"]),
          "active_inst %% %.1f"%(100*c["SQ_ACTIVE_INST_ANY"]/c["SQ_WAVE_CYCLES"]), "lds conflict %% %.1f"%(100*c["SQ_LDS_BANK_CONFLICT"]/max(c["SQ_LDS_IDX_ACTIVE"],1)),
          "lds active/gui*128 %.1f"%(100*c["SQ_LDS_IDX_ACTIVE"]/(32*gui)))
