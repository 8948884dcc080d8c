"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid size): per-layer launch times.  usage: layer_trace.py <dir> [filter]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else "conv"
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if flt not in name:
        continue
    key = (name[:64], (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:64s} blocks {str(k[1]):>16s} vgpr {k[2]:>3s}+{k[3]:>3s} lds {k[4]:>6s}  n {v[0]:3d}  avg {v[1] / v[0]:8.1f} us  total {v[1] / 1e3:7.2f} ms")
