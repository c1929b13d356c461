cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp && mkdir -p gpurun_out/$1 && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_sr -- python tools/sr_timing.py > gpurun_out/$1/sr_under_rocprof.txt 2>&1; find /tmp/rp_sr -name "*kernel_trace.csv" -exec cp {} /tmp/kt.csv \; ; python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/kt.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tail=rows[-60:]
t0=int(tail[0]["Start_Timestamp"]); prev=None; out=[]
for r in tail:
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    gap=(st-prev)/1000 if prev else 0
    out.append("%9.2f us  dur %7.2f  gap %6.2f  %s grid=%s" % ((st-t0)/1000,(en-st)/1000,gap,r["Kernel_Name"][:56],r.get("Grid_Size_X","")))
    prev=en
open("gpurun_out/$1/sr_kernel_timeline.txt","w").write("\n".join(out)+"\n")
PY
