#!/bin/bash
# VERDICT r5 item 2c, second half: does the process-to-process launch time of wps_layer_fb_kernel follow the address-translation
# counters (physical placement / fragment size of the process's allocations)? 6 processes under rocprofv3 --pmc, shipped kernels.
O=$(pwd)/gpurun_out; REPO=$(pwd); export PYTHONPATH=$REPO
export V4L_LIB=${V4L_LIB:-$REPO/vision4leg_amd/libv4l_hip.so}
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  timeout 200 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum \
    --output-format csv -d $O/r6h_pmc/p$i -- python $REPO/tools/probe/fb_blocks.py 12 > $O/r6h_pmc_p$i.log 2>&1
  echo "p$i rc=$?"
done
python - $O/r6h_pmc <<'PY' > $O/r6h_fb_tlb.txt
import csv, glob, os, sys, collections
for pd in sorted(glob.glob(sys.argv[1] + "/p*")):
    agg = collections.defaultdict(float); n = 0; dur = []
    for path in glob.glob(os.path.join(pd, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "wps_layer_fb" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"].startswith("TCP_UTCL1_REQUEST"): n += 1
    for path in glob.glob(os.path.join(pd, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "wps_layer_fb" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    n = max(n, 1); dur = sorted(dur)
    print(os.path.basename(pd), "fb launches %d  median %.1f us  min %.1f |" % (len(dur), dur[len(dur) // 2] if dur else 0, dur[0] if dur else 0),
          "  ".join("%s %.0f" % (k.replace("TCP_UTCL1_", "").replace("_sum", ""), v / n) for k, v in sorted(agg.items())))
PY
cat $O/r6h_fb_tlb.txt
find $O/r6h_pmc -name "*.csv" -size +1M -delete; find $O/r6h_pmc -name "*.db" -delete
