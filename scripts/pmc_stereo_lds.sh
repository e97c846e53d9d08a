#!/bin/bash
# LDS-side SQ counters of the stereo kernels from one serialised step (GPU box): gpurun -- 'bash scripts/pmc_stereo_lds.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p_lds_st2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/p_lds_st2 -o runc -- python3 bench.py --stereo --stereo-batch 64 --frames 64 --steps 1 --warmup 0 --no-cpu --serial-only > gpurun_out/p_lds_st2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/p_lds_st2/**/*counter_collection.csv",recursive=True)[0]
tr=glob.glob("gpurun_out/p_lds_st2/**/*kernel_trace.csv",recursive=True)[0]
dur={r["Dispatch_Id"]:int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr))}
agg=collections.defaultdict(lambda: collections.defaultdict(float)); ns=collections.defaultdict(float); seen=set()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); ns[k]+=dur.get(r["Dispatch_Id"],0)
for k,c in sorted(agg.items(), key=lambda kv:-ns[kv[0]])[:8]:
    w=c["SQ_WAVES"]; t=ns[k]*1e-9
    if not w or not t: continue
    clk=c["SQ_BUSY_CYCLES"]/32/t
    print(k[:38], "ms %.3f VALU/w %.0f LDS/w %.0f cyc/w(x4) %.0f ipc %.3f bankconf_cyc/w %.0f waitlds/w %.0f activeLDS/w %.0f" % (ns[k]/1e6, c["SQ_INSTS_VALU"]/w, c["SQ_INSTS_LDS"]/w, c["SQ_WAVE_CYCLES"]/w, c["SQ_INSTS_VALU"]/(t*clk*1024), c["SQ_LDS_BANK_CONFLICT"]/w, c["SQ_WAIT_INST_LDS"]/w, c["SQ_ACTIVE_INST_LDS"]/w))
PY
