#!/usr/bin/env python3
"""Compile the translation units of libmcx with -save-temps and summarise one kernel: registers and the instruction
mix of its biggest loop.  usage: tools/kernel_asm.py <mangled-name-substring> [--dump]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TMP = "/tmp/mcx_asm"


def main():
    pat = sys.argv[1]
    os.makedirs(TMP, exist_ok=True)
    s = ""
    for tu in ("mcx_k_fast", "mcx_k_fast_full", "mcx_k_fastb", "mcx_k_fastb_full", "mcx_k_pregen", "mcx_k_generic_main", "mcx_k_generic_burn", "mcx_engine"):
        if pat.startswith("k_fused_fast") and tu not in ("mcx_k_fast", "mcx_k_fast_full", "mcx_k_fastb", "mcx_k_fastb_full", "mcx_k_pregen"):
            continue
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
                               "-fno-fast-math", "-fno-gpu-flush-denormals-to-zero", "-I" + ROOT + "/include", "-c",
                               "-save-temps", "-o", "x.o", ROOT + "/mcpar_amd/csrc/%s.hip" % tu], cwd=TMP,
                              stderr=subprocess.DEVNULL)
        s += open(TMP + "/%s-hip-amdgcn-amd-amdhsa-gfx950.s" % tu).read()
    names = sorted(set(re.findall(r"^(_Z\w+):", s, flags=re.M)))
    hits = [n for n in names if pat in n]
    for name in hits:
        i = s.index("\n" + name + ":")
        j = s.index(".Lfunc_end", i)
        body = s[i:j].split("\n")
        meta = s[s.index(".name:           " + name):]
        regs = {k: re.search(k + r":\s+(\d+)", meta).group(1) for k in (".vgpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size")}
        labels = {}
        for k, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = k
        best = None
        for k, l in enumerate(body):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < k:
                a = labels[m.group(1)]
                if best is None or k - a > best[1] - best[0]:
                    best = (a, k)
        print(name, regs)
        if best:
            ops = collections.Counter()
            for l in body[best[0]:best[1] + 1]:
                l = l.strip()
                if not l or l.startswith((".", ";")) or l.endswith(":"):
                    continue
                ops[l.split()[0]] += 1
            tot = sum(ops.values())
            valu = sum(v for k, v in ops.items() if k.startswith("v_"))
            print("  biggest loop: %d instrs, %d VALU, %d branches" % (tot, valu, sum(v for k, v in ops.items() if "branch" in k)))
            print("  " + ", ".join("%s %d" % kv for kv in ops.most_common(40)))
            if "--dump" in sys.argv:
                print("\n".join(body[best[0]:best[1] + 1]))


if __name__ == "__main__":
    main()
