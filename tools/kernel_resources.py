#!/usr/bin/env python3
"""What the compiler made of every kernel in velesdb_amd/lib/libvelesdb_hip.so, read from the library itself (no GPU needed):

  * the gfx950 code objects are cut out of the library's clang offload bundles (`__CLANG_OFFLOAD_BUNDLE__`, one per .hip file),
  * each object's NT_AMDGPU_METADATA note (msgpack) gives, per kernel: registers (`.vgpr_count` is the UNIFIED total on gfx950 —
    architectural + accumulation registers, 512 per lane per SIMD), `.agpr_count`, scalar registers, spilled registers, scratch bytes
    per lane (`.private_segment_fixed_size`), static LDS bytes, the declared block size,
  * `llvm-objdump --symbolize-operands` gives the instruction stream cut into basic blocks: `loops()` finds the innermost loop bodies
    (block ranges closed by a backward branch: the steady state) and counts what they issue.

    python tools/kernel_resources.py            # one line per kernel family (ranges over the template instances)
    python tools/kernel_resources.py --all      # one line per kernel
    python tools/kernel_resources.py --loops sweep_topk_gemm_bf16_pp   # the innermost loops of the matching kernels (those with MFMAs, if any)

tests/test_kernel_resources_cpu.py pins the properties the measured numbers of DESIGN.md rest on (a register budget IS an occupancy;
scratch in a sweep kernel or a spill in a matrix-core main loop would be a silent slowdown that no parity test sees)."""
import os
import re
import struct
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "velesdb_amd", "lib", "libvelesdb_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
REGS_PER_LANE = 512  # unified vector registers per lane per SIMD on gfx950 (MI355X_MICROARCH.md), allocated in blocks of 8


def code_objects(lib=LIB):
    """the gfx950 ELF images inside the library, in link order"""
    blob = open(lib, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(BUNDLE_MAGIC, pos)
        if i < 0:
            return out
        n, = struct.unpack_from("<Q", blob, i + len(BUNDLE_MAGIC))
        o = i + len(BUNDLE_MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tl].decode()
            o += 24 + tl
            if "gfx950" in triple and size:
                elf = blob[i + off:i + off + size]
                assert elf[:4] == b"\x7fELF", "compressed or foreign code object: " + triple
                out.append(elf)
        pos = i + len(BUNDLE_MAGIC)


def _metadata(elf):
    import msgpack
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    for s in range(shnum):
        _, typ, _, _, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + s * shentsize)
        if typ != 7:  # SHT_NOTE
            continue
        q = off
        while q < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, q)
            q += 12
            name = elf[q:q + namesz]
            q += (namesz + 3) & ~3
            desc = elf[q:q + descsz]
            q += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):  # NT_AMDGPU_METADATA
                yield msgpack.unpackb(desc, raw=False, strict_map_key=False)


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    out = []
    for n in r.stdout.split("\n")[:len(names)]:
        n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "")
        out.append(n[:n.index("(")] if "(" in n else n)
    return out


def kernels(lib=LIB):
    """[{name (demangled, without the argument list), symbol, obj (index of its code object), vgpr, agpr, sgpr, sgpr_spill, vgpr_spill,
    scratch, lds, block, waves_per_simd}]"""
    rows = []
    for oi, elf in enumerate(code_objects(lib)):
        for md in _metadata(elf):
            for k in md["amdhsa.kernels"]:
                assert k[".wavefront_size"] == 64
                v = k[".vgpr_count"]
                rows.append({"symbol": k[".name"], "obj": oi, "vgpr": v, "agpr": k[".agpr_count"], "sgpr": k[".sgpr_count"],
                             "sgpr_spill": k[".sgpr_spill_count"], "vgpr_spill": k[".vgpr_spill_count"],
                             "scratch": k[".private_segment_fixed_size"], "lds": k[".group_segment_fixed_size"],
                             "block": k[".max_flat_workgroup_size"], "dynamic_stack": bool(k.get(".uses_dynamic_stack", False)),
                             "waves_per_simd": min(8, REGS_PER_LANE // max(8, (v + 7) // 8 * 8))})
    for r, n in zip(rows, demangle([r["symbol"] for r in rows])):
        r["name"] = n
    return rows


def family(name):
    return name.split("<")[0].replace("vdb::", "") or name


def disassemble(elf_bytes):
    """{symbol: [(label, [instruction text, ...]), ...]}: every function of the object cut into basic blocks"""
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(elf_bytes)
        f.flush()
        txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", "--symbolize-operands", f.name], capture_output=True, text=True,
                             check=True).stdout
    funcs, cur, blocks = {}, None, None
    for line in txt.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            lab = m.group(1)
            if re.fullmatch(r"L\d+", lab):
                if blocks is not None:
                    blocks.append((lab, []))
            else:
                cur, blocks = lab, [("entry", [])]
                funcs[cur] = blocks
            continue
        ins = line.split("//")[0].strip()
        if ins and blocks is not None and not ins.endswith(":") and not ins.startswith("Disassembly") and "file format" not in ins:
            blocks[-1][1].append(ins)
    return funcs


def loop_regions(blocks):
    """[(first, last)]: block index ranges closed by a backward branch (a loop body in layout order), innermost first"""
    idx = {lab: i for i, (lab, _) in enumerate(blocks)}
    regs = set()
    for i, (_, ins) in enumerate(blocks):
        for x in ins:
            m = re.match(r"s_c?branch\w* (L\d+)$", x)
            if m and idx.get(m.group(1), i + 1) <= i:
                regs.add((idx[m.group(1)], i))
    return sorted(regs, key=lambda r: (r[1] - r[0], r[0]))


def loops(blocks, want=lambda ins: True):
    """the INNERMOST loop bodies whose instructions satisfy `want`: [(label of the first block, instructions in layout order)]"""
    out, taken = [], []
    for a, b in loop_regions(blocks):
        ins = [x for _, blk in blocks[a:b + 1] for x in blk]
        # the closing block may go on behind its backward branch (a conditional one: the fall-through is the loop's exit path)
        back = [i for i, x in enumerate(ins) if re.match(r"s_c?branch\w* %s$" % re.escape(blocks[a][0]), x)]
        if back:
            ins = ins[:back[-1] + 1]
        if want(ins) and not any(a <= ta and tb <= b for ta, tb in taken):
            taken.append((a, b))
            out.append((blocks[a][0], ins))
    return out


def in_loops(blocks):
    """every instruction that sits inside some loop body"""
    mask = [False] * len(blocks)
    for a, b in loop_regions(blocks):
        for j in range(a, b + 1):
            mask[j] = True
    return [x for (_, blk), m in zip(blocks, mask) if m for x in blk]


def count(ins, *prefixes):
    return sum(1 for x in ins if x.startswith(prefixes))


def main():
    rows = kernels()
    cols = ("vgpr", "agpr", "sgpr", "sgpr_spill", "vgpr_spill", "scratch", "lds", "block", "waves_per_simd")
    if "--loops" in sys.argv:
        pat = sys.argv[sys.argv.index("--loops") + 1]
        objs = code_objects()
        by_obj = defaultdict(list)
        for r in rows:
            if pat in r["name"]:
                by_obj[r["obj"]].append(r)
        for oi, rs in by_obj.items():
            funcs = disassemble(objs[oi])
            for r in rs:
                print(r["name"])
                f_ = funcs[r["symbol"]]
                inner = loops(f_, (lambda ins: count(ins, "v_mfma") > 0) if any(count(b_, "v_mfma") for _, b_ in f_) else (lambda ins: True))
                print(f"  {sum(len(b_) for _, b_ in f_)} instructions, {len(in_loops(f_))} of them inside loops (scratch there: "
                      f"{count(in_loops(f_), 'scratch_')}, flat there: {count(in_loops(f_), 'flat_')})")
                for lab, ins in inner:
                    print(f"  loop {lab}: {len(ins)} instructions, mfma {count(ins, 'v_mfma')}, ds_read {count(ins, 'ds_read')}, "
                          f"ds_write {count(ins, 'ds_write')}, lds-dma {sum(1 for x in ins if x.startswith('buffer_load') and x.endswith(' lds'))}, "
                          f"global/buffer loads {count(ins, 'global_load', 'buffer_load')}, barriers {count(ins, 's_barrier')}, "
                          f"flat {count(ins, 'flat_')}, scratch {count(ins, 'scratch_')}, "
                          f"readlane/writelane {count(ins, 'v_readlane', 'v_writelane')}")
        return
    if "--all" in sys.argv:
        print(f"{'kernel':90s} " + " ".join(f"{c:>10s}" for c in cols))
        for r in sorted(rows, key=lambda r: r["name"]):
            print(f"{r['name'][:90]:90s} " + " ".join(f"{r[c]:10d}" for c in cols))
        return
    fam = defaultdict(list)
    for r in rows:
        fam[family(r["name"])].append(r)
    print(f"{len(rows)} kernels in {len(code_objects())} gfx950 code objects of {os.path.relpath(LIB, ROOT)}; ranges over a family's instances;\n"
          f"vgpr = unified total (incl. agpr), scratch = bytes per lane, lds = STATIC bytes (dynamic LDS is a launch argument)\n")
    print(f"{'kernel family':32s} {'n':>3s} " + " ".join(f"{c:>11s}" for c in cols))
    for name in sorted(fam):
        rs = fam[name]
        cells = []
        for c in cols:
            lo, hi = min(r[c] for r in rs), max(r[c] for r in rs)
            cells.append(f"{lo}" if lo == hi else f"{lo}-{hi}")
        print(f"{name:32s} {len(rs):3d} " + " ".join(f"{c:>11s}" for c in cells))


if __name__ == "__main__":
    main()
