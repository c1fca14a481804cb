import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l)
        print("%-12s ns=%7d us/it=%6.1f nn_us=%6.1f cand/q=%5.1f WG=%5.0f parts/wg=%4.2f glob=%5.0f pts/wg=%5.0f rr=%4.0f phases=%s" % (d["cfg"],d["ns"],d["us_per_iter"],d["nn_us"],d["cand_per_q"],d["tile_wg"],d.get("parts_per_wg",0),d["fallback_wg"],d["tile_pts_per_wg"],d["reranks"],d["phase_cyc_per_wg"]))
    elif "MISMATCH" in l or l.startswith("check") or "TOTAL" in l or "Error" in l or "error" in l: print(l.rstrip())
