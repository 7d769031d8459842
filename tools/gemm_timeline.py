import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if any(t in r["Kernel_Name"] for t in ("k_gemm_wide", "k_wide_", "k_mlp_chain", "k_chain_pack"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=rows[-16:]
t0=int(seq[0]["Start_Timestamp"])
for r in seq:
    n=r["Kernel_Name"]; mode=n[n.find("<")+1] if "<" in n else "r"
    print(mode, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), "start %7.1f us  dur %6.1f us" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
