import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gemm_skinny" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].replace("void gemm_skinny_kernel", "").replace("(GemmArgs)", ""), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_X"]) // 64)
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(((k, statistics.median(v[4:]), len(v)) for k, v in acc.items()), key=lambda t: (t[0][0].split(",")[2:4], t[0][1] * t[0][2], t[1]))
for k, med, n in rows:
    print(f"{k[0]:28s} blocks {k[1]:6d} W {k[2]:3d}  median {med:8.2f} us  n={n}")
