"""Where the register spills of the hot kernels sit: compiles the library's device code with -save-temps and,
for every kernel named, counts the scratch (VGPR spill) instructions by the depth of the loop they are in -- the assembler
output labels every basic block with its loop depth.  Depth 0 = set-up / epilogue, 1 = once per work item of the persistent
grid, >= 2 = inside an item's loops.
    python tools/spill_report.py [out.txt]
"""
import hashlib, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ["k_ksw_pair", "k_kswILi3E", "k_asm_readsILi8E", "k_asm_combine3ILi5ELb0E", "k_tally", "k_fallback", "k_prepack_fastILi2E", "k_prepackE"]


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
    lines = ["# scratch (VGPR spill) instructions of the hot kernels by loop depth; sources src_sha16 = %s" % bench.src_sha16(),
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
        for ln in body.split("\n"):
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
                if "scratch_" in ln:
                    by_depth[depth] = by_depth.get(depth, 0) + 1
        res = {}
        blk = remarks[remarks.find("Function Name: " + name):]
        for key in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill"):
            mm = re.search(re.escape(key) + r": (\d+)", blk)
            res[key] = int(mm.group(1)) if mm else None
        lines.append("%s\n    %s" % (name, ", ".join("%s %s" % (a, b) for a, b in res.items())))
        lines.append("    scratch instructions by loop depth: %s   (instructions by depth: %s)" % (
            {d: by_depth[d] for d in sorted(by_depth)} or "none", {d: total_by_depth[d] for d in sorted(total_by_depth)}))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
