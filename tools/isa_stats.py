#!/usr/bin/env python3
"""Per-kernel resource table of mdc_kernels.hip's gfx950 ISA: VGPRs, SGPRs, scratch (private segment), LDS, and a few
instruction counts of interest.  usage: python tools/isa_stats.py [filter-substring] [--asm out.s]
(tests/test_isa.py asserts on the scratch column: no instantiation may spill.)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(source="mdc_kernels.hip", defines=()):
    src = os.path.join(ROOT, "mono_dataset_code_amd", "csrc", source)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
           "--cuda-device-only", "-S", "-o", "-", "-I" + os.path.join(ROOT, "include")] + ["-D" + d for d in defines] + [src]
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True).stdout


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout
    return out.splitlines()


def kernels(asm):
    """-> list of dicts: name, vgpr, sgpr, scratch, lds, counts{mnemonic prefix: n}"""
    res = []
    # metadata blocks: .amdhsa_kernel <name> ... .end_amdhsa_kernel
    meta = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        body = m.group(2)

        def field(k):
            x = re.search(r"\.amdhsa_%s (\S+)" % k, body)
            return int(x.group(1), 0) if x else 0
        meta[m.group(1)] = dict(vgpr=field("next_free_vgpr"), sgpr=field("next_free_sgpr"), scratch=field("private_segment_fixed_size"),
                                lds=field("group_segment_fixed_size"), accum_offset=field("accum_offset"))
    # code: from "<name>:" to ".Lfunc_end"
    for name, md in meta.items():
        m = re.search(r"^%s:[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(name), asm, re.S | re.M)
        code = m.group(1) if m else ""
        cnt = {}
        for line in code.splitlines():
            t = line.strip().split()
            if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
                continue
            cnt[t[0]] = cnt.get(t[0], 0) + 1
        md.update(name=name, counts=cnt, n_inst=sum(cnt.values()))
        res.append(md)
    names = demangle([r["name"] for r in res])
    for r, n in zip(res, names):
        n = re.sub(r"\(anonymous namespace\)::|mdc::", "", n).replace("void ", "")
        r["pretty"] = n[:n.index(">(") + 1] if ">(" in n else n.split("(")[0]
    return res


if __name__ == "__main__":
    args = sys.argv[1:]
    flt = [a for i, a in enumerate(args) if not a.startswith("--") and not (i and args[i - 1] in ("--asm", "--from"))]
    asm = open(sys.argv[sys.argv.index("--from") + 1]).read() if "--from" in sys.argv else device_asm()
    if "--asm" in sys.argv:
        open(sys.argv[sys.argv.index("--asm") + 1], "w").write(asm)
    for k in kernels(asm):
        if flt and not any(f in k["pretty"] for f in flt):
            continue
        c = k["counts"]
        pick = lambda pre: sum(v for n, v in c.items() if n.startswith(pre))
        print("%-78s vgpr %3d sgpr %3d scratch %3d  inst %5d  ds_read %4d ds_write %3d buffer %3d v_* %5d s_barrier %2d" % (
            k["pretty"], k["vgpr"], k["sgpr"], k["scratch"], k["n_inst"], pick("ds_read"), pick("ds_write"), pick("buffer_"), pick("v_"), pick("s_barrier")))
