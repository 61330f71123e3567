#!/usr/bin/env python3
"""Instruction mix of the fused kernels' gfx950 ISA (no GPU needed): compiles csrc/v4l_hip.hip to assembly and counts
opcodes per kernel. usage: python tools/isa_mix.py [substring ...] > profiles/rN_isa_instruction_mix.txt"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1:] or ["rollout_stack_kernelIDF16b", "rollout_encoder2_kernel", "infer_layer_kernelIDF16bLi4", "infer_layer_kernelIDF16bLi2",
                        "bwd_layer_kernelIDF16bLi4ELb1ELb0", "bwd_layer_kernelIDF16bLi4ELb0ELb1", "bwd_conv_kernelIDF16b",
                        "train_encoder_kernel", "gemm_tn_wide_kernelIDF16b"]
asm = os.path.join(tempfile.gettempdir(), "v4l_isa.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                       os.path.join(ROOT, "vision4leg_amd/csrc/v4l_hip.hip"), "-o", asm], stderr=subprocess.DEVNULL)
cur, body = None, collections.defaultdict(list)
for line in open(asm):
    m = re.match(r"^(_ZN3v4l\w+):", line)
    if m:
        cur = m.group(1)
        continue
    if cur and re.match(r"^\s+s_endpgm", line):
        cur = None
        continue
    if cur:
        t = line.strip()
        if t and not t.startswith((".", ";")) and not t.endswith(":"):
            body[cur].append(t.split()[0])
for k, ops in body.items():
    if not any(w in k for w in want):
        continue
    c = collections.Counter(ops)
    n = len(ops)
    mfma = sum(v for o, v in c.items() if o.startswith("v_mfma"))
    lds = sum(v for o, v in c.items() if o.startswith("ds_"))
    vmem = sum(v for o, v in c.items() if o.startswith(("global_", "buffer_", "flat_", "scratch_")))
    valu = sum(v for o, v in c.items() if o.startswith("v_")) - mfma
    print("== %s: %d instructions: %d MFMA (%.1f %%), %d other VALU, %d LDS, %d VMEM, %d s_waitcnt, %d s_nop" %
          (k, n, mfma, 100.0 * mfma / n, valu, lds, vmem, c["s_waitcnt"], c["s_nop"]))
    print("    " + ";  ".join("%4d %s" % (v, o) for o, v in c.most_common(14)))
