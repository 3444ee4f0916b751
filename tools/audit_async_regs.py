"""Audit of the inline-asm register loads in lstm_scan2.hip (the backward scan's epilogue inputs): between an asm
`global_load_*` and the next hand-written `s_waitcnt vmcnt`, no instruction may read or write its destination
registers -- the compiler believes they hold data from the end of the asm statement on and is free to copy or
re-use them (cdna_hip_programming.md 5.7).  Usage: python tools/audit_async_regs.py <file.s>
(hipcc ... -save-temps -c ocrd_keraslm_amd/csrc/lstm_scan2.hip writes the .s)"""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def all_regs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs_of(tok)
    return out


def audit(path, min_checked=20):
    lines = open(path).read().split("\n")
    bad = 0
    kernel = None
    in_asm = False
    pending = {}          # register -> line number of the asm load
    checked = 0
    for i, ln in enumerate(lines):
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel = m.group(1)
            pending = {}
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        if in_asm and re.match(r"(global|buffer)_load_", s) and " lds" not in s:      # (dword*, ushort, ubyte ...: every register form)
            dst = s.split()[1].rstrip(",")
            for r in regs_of(dst):
                pending[r] = i + 1
            checked += 1
            continue
        if in_asm and s.startswith("s_waitcnt vmcnt"):
            pending = {}          # (the counted waits cover every asm load issued before them in this kernel's scheme)
            continue
        if s.startswith("s_endpgm") or s.startswith("s_branch"):
            pending = {}          # (what follows an unconditional branch in the text is another path)
            continue
        if re.match(r"v_mov_b32_e32 v\d+, -1$", s):
            for r in all_regs(s):      # (arming a destination before its load: the other arm of an if / else in program text)
                pending.pop(r, None)
            continue
        hit = all_regs(s) & set(pending)
        if hit:
            bad += 1
            print(f"{kernel}: line {i + 1} touches v{sorted(hit)} loaded by asm at line {pending[min(hit)]}: {s}")
            for r in hit:
                pending.pop(r, None)
    print(f"{checked} asm register loads checked, {bad} premature uses")
    if checked < min_checked:
        # (the audit must not pass vacuously when the mnemonics or the build change)
        print(f"audit: expected at least {min_checked} hand-issued register loads, found {checked}")
        return bad + 1
    return bad


def audit_agprs(path, kernel_substr="lstm_scan_bwd_regtile", min_kernels=2):
    """The backward scan's epilogue inputs land in accumulator registers while the compiler's code runs (lstm_scan2.hip,
    KL_BWD_INPUTS_*): NO compiler-generated instruction of those kernels may name an AGPR -- not as an MFMA accumulator,
    not as a spill slot, not as a copy of one of the pinned register variables -- and every one of them must have been
    built with the accumulator registers the hand-written statements use (a0..a5, a8..a39) inside its 256 registers."""
    text = open(path).read()
    if text.startswith("; KL_NO_REGTILE"):      # (tools/build_agpr_tu.sh's fall-back build: the kernels are not in it)
        print(f"*{kernel_substr}* kernels left out of this build (KL_NO_REGTILE): nothing to check")
        return 0
    lines = text.split("\n")
    kernel = None
    in_asm = False
    bad = checked = 0
    seen = set()
    for i, ln in enumerate(lines):
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel = m.group(1)
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not kernel or kernel_substr not in kernel or not s or s[0] in ";.":
            continue
        seen.add(kernel)
        if in_asm:
            continue
        checked += 1
        if re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", s):
            bad += 1
            print(f"{kernel}: line {i + 1} names an accumulator register outside the hand-written statements: {s}")
    # (kernel descriptors: VGPRs up to the accumulator offset + the accumulator registers must fit the 128 registers a wave of
    #  a 1024-thread workgroup gets -- a build without the "amdgpu-agpr-alloc" attribute shows up here, or as a 64 / 64 split)
    for kname, body in re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, flags=re.S):
        if kernel_substr not in kname:
            continue
        nfree = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        acc = int(re.search(r"\.amdhsa_accum_offset (\d+)", body).group(1))
        limit = 256      # (512-thread workgroups: two waves per SIMD)
        if nfree > limit or nfree - acc < 38 or acc < 100:
            bad += 1
            print(f"{kname}: next_free_vgpr {nfree}, accum_offset {acc}: expected <= {limit} registers with >= 38 accumulator registers behind >= 100 VGPRs")
    print(f"{checked} compiler instructions of {len(seen)} *{kernel_substr}* kernels checked, {bad} problems with accumulator registers")
    if len(seen) < min_kernels:
        print(f"audit: expected at least {min_kernels} *{kernel_substr}* kernels, found {len(seen)}")
        bad += 1
    return bad


def audit_reserved(path, kernel_substr="bwd_wide2", reserved=range(124, 128)):
    """(historic) registers kept from the compiler by convention: no compiler-generated instruction of those kernels
    may name them."""
    lines = open(path).read().split("\n")
    kernel = None
    in_asm = False
    bad = checked = 0
    res = set(reserved)
    for i, ln in enumerate(lines):
        s = ln.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kernel = m.group(1)
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not kernel or kernel_substr not in kernel or in_asm or not s or s[0] in ";.":
            continue
        checked += 1
        hit = all_regs(s) & res
        if hit:
            bad += 1
            print(f"{kernel}: line {i + 1} uses reserved v{sorted(hit)}: {s}")
    print(f"{checked} compiler instructions of *{kernel_substr}* kernels checked, {bad} use v{min(res)}..v{max(res)}")
    return bad


if __name__ == "__main__":
    sys.exit(1 if (audit(sys.argv[1]) + audit_agprs(sys.argv[1])) else 0)
