"""Reads a rocprofv3 kernel_trace.csv of tools/time_qr.py and reports, for the LAST QR in it, how much of the factor /
apply kernel time overlapped (two-stream look-ahead)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "k_qr_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last QR = the trailing run of kernels after the last k_qr_block gap: take the last N launches until a gap > 1 ms
sel = [rows[-1]]
for r in reversed(rows[:-1]):
    if int(sel[-1]["Start_Timestamp"]) - int(r["End_Timestamp"]) > 1_000_000:
        break
    sel.append(r)
sel.reverse()
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
fac = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel if "factor" in r["Kernel_Name"]]
app = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel if "apply" in r["Kernel_Name"]]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
both = union(fac + app)
print(f"launches {len(sel)} (factor {len(fac)}, apply {len(app)}), wall {1e-6*(t1-t0):.2f} ms, "
      f"factor sum {1e-6*sum(e-s for s,e in fac):.2f} ms, apply sum {1e-6*sum(e-s for s,e in app):.2f} ms, "
      f"busy (union) {1e-6*both:.2f} ms, idle {1e-6*(t1-t0-both):.2f} ms, "
      f"overlap {1e-6*(union(fac)+union(app)-both):.2f} ms; queues {sorted(set(r['Queue_Id'] for r in sel))}")
