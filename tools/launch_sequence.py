"""GPU box: the ORDERED kernel launches of one steady-state training iteration from a rocprofv3 --kernel-trace CSV
(tools/launch_sequence.py <dir>): start offset, duration and short name per launch, to see which sequences are worth fusing."""
import csv, glob, re, sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
it = [i for i, r in enumerate(rows) if "mlp_fwd" in r["Kernel_Name"] and "<true" in r["Kernel_Name"]]
seg = rows[it[-3]:it[-2]]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::|std::array<char\*, \d+ul>|binary_internal::|<unnamed>::", "", r["Kernel_Name"])
    n = re.sub(r"\s+", " ", n)
    print("%9.1f %7.1f  q%-3s g%-6s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                          r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), n[:150]))
