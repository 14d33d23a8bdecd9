"""Where the register spills of the hot kernels sit: compiles the library's device code with
-save-temps and, for every kernel named, counts by the depth of the loop they are in
  * the scratch (VGPR spill) instructions, and
  * the SGPR spills: a spilled scalar lives in a lane of a vector register the compiler sets aside (`v_writelane_b32 vN, sM, k`
    to park it, `v_readlane_b32 sM, vN, k` to bring it back, with a literal lane k) -- VALU slots in loops that are short of
    them.  The spill VGPRs are recognised as the registers that v_writelane writes from SGPR sources at eight or more
    different literal lanes; the lane moves the kernels make on purpose (wave-uniform lane indices in registers, lane 0
    / 15 / 63 edges of the sweeps) name their lane in a register or touch a working register, and are listed apart.
The assembler output labels every basic block with its loop depth.  Depth 0 = set-up / epilogue, 1 = once per work item of
the persistent grid, >= 2 = inside an item's loops.  Hand-written asm statements (the pair sweep's diagonal loops) sit inside
the compiler's blocks: their own loops do not show as depth, so the report also says how many instructions of a kernel are
inside `;;#ASMSTART` .. `;;#ASMEND` and that none of those is a spill (the statements name every register they use).
    python tools/spill_report.py [out.txt]
"""
import hashlib, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["k_ksw_pair", "k_kswILi3E", "k_asm_readsILi8E", "k_asm_combine3ILi5ELb0ELi32E", "k_asm_combine3ILi5ELb0ELi64E", "k_asm_combine3ILi6ELb0ELi32E", "k_asm_combine3ILi4ELb0ELi64ELb1E", "k_tallyILi8E", "k_tally_prep", "k_fallback", "k_prepack_fastILi2ELb0E", "k_prepack_fastILi2ELb1E"]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    src = os.path.join(ROOT, "indelope_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-I", src,
               "-Wno-unused-function", "-ffp-contract=off", "-save-temps", "-Rpass-analysis=kernel-resource-usage", "-o", "lib.so", os.path.join(src, "indelope_hip.hip")]
        pr = subprocess.run(cmd, cwd=td, capture_output=True, text=True)
        if pr.returncode:
            sys.exit(pr.stderr[-2000:])
        remarks = pr.stderr
        asm = open(os.path.join(td, "indelope_hip-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    sys.path.insert(0, ROOT)
    import bench
    lines = ["# scratch (VGPR spill) instructions and SGPR spill moves of the hot kernels by loop depth; sources src_sha16 = %s" % bench.src_sha16(),
             "# depth 0: set-up / epilogue; 1: per work item of the persistent grid; >= 2: inside an item's loops", ""]
    for k in KERNELS:
        m = re.search(r"^(_ZN3ihp\d+%s\w*):" % re.escape(k), asm, re.M)
        if not m:
            lines.append("%s: not found" % k)
            continue
        name = m.group(1)
        body = asm[m.start():asm.find(".end_amdhsa_kernel", m.start())]
        depth = 0
        by_depth, total_by_depth = {}, {}
        # pass 1: which VGPRs hold spilled scalars (written lane by lane from SGPRs with literal lanes, many times)
        wl = {}
        for ln in body.split("\n"):
            wm = re.match(r"\s+v_writelane_b32 (v\d+), s\d+, (\d+)\s*$", ln)
            if wm:
                wl.setdefault(wm.group(1), set()).add(int(wm.group(2)))
        spill_regs = {v for v, lanes in wl.items() if len(lanes) >= 8}       # (a working register gets lane 0 / 15 / 63 only)
        sg_w, sg_r, asm_n, asm_spill, in_asm = {}, {}, 0, 0, False
        for ln in body.split("\n"):
            if ";;#ASMSTART" in ln:
                in_asm = True
                continue
            if ";;#ASMEND" in ln:
                in_asm = False
                continue
            lm = re.match(r"^\.LBB\d+_\d+:\s*;(.*)$", ln)
            if lm:
                dm = re.search(r"Depth=(\d+)", lm.group(1))
                depth = int(dm.group(1)) if dm else 0
                continue
            if re.match(r"^\.LBB\d+_\d+:", ln):
                depth = 0
                continue
            if re.match(r"\s+[vsdg]\w+", ln):
                total_by_depth[depth] = total_by_depth.get(depth, 0) + 1
                if in_asm:
                    asm_n += 1
                if "scratch_" in ln:
                    by_depth[depth] = by_depth.get(depth, 0) + 1
                wm = re.match(r"\s+v_writelane_b32 (v\d+), s\d+, \d+\s*$", ln)
                rm = re.match(r"\s+v_readlane_b32 s\d+, (v\d+), \d+\s*$", ln)
                if wm and wm.group(1) in spill_regs:
                    sg_w[depth] = sg_w.get(depth, 0) + 1
                    asm_spill += in_asm
                if rm and rm.group(1) in spill_regs:
                    sg_r[depth] = sg_r.get(depth, 0) + 1
                    asm_spill += in_asm
        res = {}
        blk = remarks[remarks.find("Function Name: " + name):]
        for key in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill"):
            mm = re.search(re.escape(key) + r": (\d+)", blk)
            res[key] = int(mm.group(1)) if mm else None
        lines.append("%s\n    %s" % (name, ", ".join("%s %s" % (a, b) for a, b in res.items())))
        lines.append("    scratch instructions by loop depth: %s   (instructions by depth: %s)" % (
            {d: by_depth[d] for d in sorted(by_depth)} or "none", {d: total_by_depth[d] for d in sorted(total_by_depth)}))
        lines.append("    SGPR spills by loop depth (spill registers %s): parked (v_writelane) %s, brought back (v_readlane) %s" % (
            ", ".join(sorted(spill_regs, key=lambda v: int(v[1:]))) or "none", {d: sg_w[d] for d in sorted(sg_w)} or "none", {d: sg_r[d] for d in sorted(sg_r)} or "none"))
        if asm_n:
            lines.append("    hand-written asm statements: %d instructions, %d of them spill moves" % (asm_n, asm_spill))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
