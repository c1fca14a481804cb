import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import tile_probe, json
for ns in (262144, 131072, 65536, 32768):
    rows = tile_probe.timing(ns, 4194304, 30, [("default", {}, "exact")])
