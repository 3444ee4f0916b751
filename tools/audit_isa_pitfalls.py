#!/usr/bin/env python3
"""ISA audit of every kernel file for two compiler pitfalls found in round 4:

 * FLAT memory instructions.  Nothing here addresses memory through a generic pointer on purpose; a `flat_load` is a volatile or
   atomic access to LDS through one -- it counts on vmcnt as well as lgkmcnt and is followed by `s_waitcnt vmcnt(0)`, a drain of
   every load and store the wave has in flight (hand-counted LDS-DMA requests included).
 * scratch.  A kernel with a private segment has spilled, or has taken the address of a by-value argument / a lambda's closure
   (runtime-indexed captured arrays): both put memory traffic into latency chains.  The known exceptions are listed in ALLOW_SCRATCH.

Usage: audit_isa_pitfalls.py <hipcc> <arch> <include dir> <file.hip> ..."""
import re
import subprocess
import sys
import tempfile

# known spills (bytes), all outside the benchmarked shapes' kernels except the first: the output layer at its 128-register cap (one
# register); the 32-row form of inc_cell_kernel (129 .. 255 hypotheses; 12 registers); the thin backward scan's width-512 form
# with eight row groups (37 registers: a shape the wide scans have taken over since round 2)
ALLOW_SCRATCH = {"logits_ce_ws_kernel": 16, "inc_cell_kernelILi2E": 48, "inc_cell_hx_kernelILi2E": 48, "lstm_scan_bwd_kernelILi16ELi8ELb1E": 148}


def main():
    hipcc, arch, inc = sys.argv[1:4]
    bad = 0
    n_kernels = 0
    for src in sys.argv[4:]:
        with tempfile.NamedTemporaryFile(suffix=".s") as out:
            r = subprocess.run([hipcc, "--offload-arch=" + arch, "-O3", "-std=c++17", "-I" + inc, "-I.", "--cuda-device-only", "-S", src, "-o", out.name],
                               stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                print("audit_isa_pitfalls: %s does not compile\n%s" % (src, r.stderr[-2000:]))
                return 2
            text = open(out.name).read()
        kernel = None
        for line in text.splitlines():
            m = re.match(r"^(_Z\w+):", line)
            if m:
                kernel = m.group(1)
            if re.match(r"\s+flat_(load|store|atomic)", line):
                print("audit_isa_pitfalls: %s: FLAT instruction in %s: %s" % (src, kernel, line.strip()))
                bad += 1
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text):
            n_kernels += 1
            name, size = m.group(1), int(m.group(2))
            allowed = max([v for k, v in ALLOW_SCRATCH.items() if k in name] or [0])
            if size > allowed:
                print("audit_isa_pitfalls: %s: kernel %s uses %d bytes of scratch" % (src, name, size))
                bad += 1
    print("%d kernels checked for FLAT instructions and scratch, %d problems" % (n_kernels, bad))
    return 1 if bad or n_kernels < 50 else 0


if __name__ == "__main__":
    sys.exit(main())
