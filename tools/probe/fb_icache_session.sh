#!/bin/bash
# VERDICT r5 item 2c, second half: does the process-to-process launch time of wps_layer_fb_kernel follow the address-translation
# counters (physical placement / fragment size of the process's allocations)? 6 processes under rocprofv3 --pmc, shipped kernels.
O=$(pwd)/gpurun_out; REPO=$(pwd); export PYTHONPATH=$REPO
export V4L_LIB=${V4L_LIB:-$REPO/vision4leg_amd/libv4l_hip.so}
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d $O/r6t_pmc/p$i -- python $REPO/tools/probe/fb_blocks.py 12 > $O/r6t_pmc_p$i.log 2>&1
  echo "p$i rc=$?"
done
python - $O/r6t_pmc <<'PY' > $O/r6t_fb_icache.txt
import csv, glob, os, sys, collections
for pd in sorted(glob.glob(sys.argv[1] + "/p*")):
    agg = collections.defaultdict(float); n = 0; dur = []
    for path in glob.glob(os.path.join(pd, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "wps_layer_fb" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"].startswith("SQC_ICACHE_REQ"): n += 1
    for path in glob.glob(os.path.join(pd, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "wps_layer_fb" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    n = max(n, 1); dur = sorted(dur)
    print(os.path.basename(pd), "fb launches %d  median %.1f us  min %.1f |" % (len(dur), dur[len(dur) // 2] if dur else 0, dur[0] if dur else 0),
          "  ".join("%s %.0f" % (k.replace("TCP_UTCL1_", "").replace("_sum", ""), v / n) for k, v in sorted(agg.items())))
PY
cat $O/r6t_fb_icache.txt
find $O/r6t_pmc -name "*.csv" -size +1M -delete; find $O/r6t_pmc -name "*.db" -delete
