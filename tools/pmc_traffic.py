"""tools/pmc_pass.sh summary -> profiles/pmc_traffic.json: measured HBM bytes per launch for bench.py's roofline.traffic.

FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md,
HBM section: TCC_EA0_RDREQ tallied at 64 B for 128-B requests), so reads are doubled as the guide prescribes; WRITE_SIZE is
taken as is (uncalibrated). usage: python tools/pmc_traffic.py <pmc outdir>/summary.txt profiles/pmc_traffic.json"""
import json
import re
import sys

NAMES = {  # rocprof kernel symbol fragment -> bench.py / profiler label
    "bwd_conv_kernel": "fused_conv_bwd", "bwd_conv3_wgrad_kernel": "fused_conv3_wgrad", "gemm_tn_wide_kernel": "gemm_tn_wide",
    "gemm_tn_group_kernel": "gemm_tn_group", "wgrad_reduce_kernel": "wgrad_reduce", "infer_encoder_kernel": "fused_encoder",
    "bwd_layer_kernel": "fused_layer_bwd", "infer_layer_kernel": "fused_layer", "train_encoder_kernel": "fused_encoder",
    "rollout_stack_kernel": "rollout_layers_head", "rollout_encoder2_kernel": "rollout_encoder",
    "wps_layer_fwd_kernel": "wps_layer_stack_head", "wps_layer_bwd_kernel": "wps_layer_bwd_stack", "wps_wgrad_kernel": "wps_wgrad",
    "wps_layer_fb_kernel": "wps_layer_fb_stack", "clip_adam_kernel": "clip_adam", "fb_loss_finish_kernel": "fb_loss_finish",
    "begin_pack_kernel": "begin_pack", "pack_kernel": "pack",  # (first match wins: begin_pack before pack)
}
rows = [l.rstrip("\n").split("\t") for l in open(sys.argv[1])]
h = rows[0]
out = {}
for r in rows[1:]:
    d = dict(zip(h, r))
    calls = int(d["calls(fetch pass)"]) or 1
    label = next((v for k, v in NAMES.items() if k in d["kernel"]), None)
    if label is None:
        continue
    rd = 2.0 * float(d.get("FETCH_SIZE", 0) or 0) * 1024 / calls
    wr = float(d.get("WRITE_SIZE", 0) or 0) * 1024 / calls
    prev = out.get(label)
    rec = {"hbm_bytes_per_launch": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr), "launches": calls,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_pass.sh), reads x2 per the gfx950 correction"}
    if prev is None or rec["hbm_bytes_per_launch"] * calls > prev["hbm_bytes_per_launch"] * prev["launches"]:
        out[label] = rec  # template variants of one kernel: keep the heaviest (the training-shape one)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
