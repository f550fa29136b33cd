#!/usr/bin/env python
"""Build-time guard: no persistent kernel may use scratch memory.

The whole design of the persistent kernels (csrc/taco_decoder_xcd.h, taco_bigru_xcd.h, taco_chain.h) is that everything a step
touches sits in registers or LDS; a value the compiler demotes to scratch turns into a memory round trip inside the dependent chain
(round 2: the scan's prefetch block, `float4 xld[]`, was kept in 48 bytes of scratch per lane and its far load was waited for at
issue).  csrc/build.sh compiles with -Rpass-analysis=kernel-resource-usage and keeps the remarks in csrc/kernel_resources.txt;
this script reads them and fails when

any instantiation of k_bigru_xcd, k_bigru_duo, k_bigru_oct, k_pointwise_chain, k_cbhg_front, k_head_sweep, k_decoder_xcd or k_decoder_bwd_xcd has
ScratchSize > 0 (no allowances since round 4: the last one, the 8-rows-per-group BPTT kernel's 196 bytes, went when the owner rows' tape
offsets became per-step values instead of 22 hoisted pointers).

    python tools/check_kernel_resources.py [remarks file]        # exit status 1 on a violation; prints a table either way"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "multi-speaker-tacotron-tensorflow_amd", "csrc", "kernel_resources.txt")
PERSISTENT = ("k_bigru_xcd", "k_bigru_duo", "k_bigru_oct", "k_decoder_xcd", "k_decoder_bwd_xcd", "k_pointwise_chain", "k_cbhg_front", "k_head_sweep")  # k_bigru_duo also matches k_bigru_duo_bwd
ALLOWED_SCRATCH = {}      # (mangled name -> bytes per lane; empty since round 4)


def parse(path):
    """[(mangled name, {field: int})] from the remark stream"""
    out, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = (m.group(1), {})
            out.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
        if m and cur is not None:
            cur[1][m.group(1).strip()] = int(m.group(2))
    return out


def check(path=DEFAULT):
    kernels = [(n, f) for n, f in parse(path) if any(p in n for p in PERSISTENT)]
    bad = []
    rows = []
    for n, f in kernels:
        scratch = f.get("ScratchSize", 0)
        limit = ALLOWED_SCRATCH.get(n, 0)
        rows.append((n, f.get("VGPRs", -1), f.get("AGPRs", -1), f.get("TotalSGPRs", -1), scratch, f.get("VGPRs Spill", 0), limit))
        if scratch > limit:
            bad.append("%s: %d bytes of scratch per lane (allowed %d)" % (n, scratch, limit))
    return kernels, rows, bad


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else DEFAULT
    if not os.path.exists(path):
        sys.exit("no remarks file at %s: run csrc/build.sh first" % path)
    kernels, rows, bad = check(path)
    if not kernels:
        sys.exit("no persistent kernel found in %s (did the build flags change?)" % path)
    print("%-52s %6s %6s %6s %8s %7s" % ("kernel", "VGPRs", "AGPRs", "SGPRs", "scratch", "spilled"))
    for n, v, a, sg, sc, sp, lim in rows:
        print("%-52s %6d %6d %6d %8d %7d%s" % (n, v, a, sg, sc, sp, "   (allowed: %d)" % lim if lim else ""))
    if bad:
        print("FAIL:\n  " + "\n  ".join(bad))
        sys.exit(1)
    print("ok: %d persistent kernel instantiations, none with scratch beyond its allowance" % len(rows))


if __name__ == "__main__":
    main()
