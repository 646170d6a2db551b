#!/usr/bin/env python3
"""scripts/check_barriers.py -- does every s_barrier in the gfx950 code of a built library wait for the wave's LDS writes?

__syncthreads() is a workgroup release fence followed by s_barrier.  On gfx950 the barrier instruction itself waits for
nothing; the `s_waitcnt lgkmcnt(0)` in front of it comes from the fence.  hipcc 7.2 was seen to drop that wait at a
barrier sitting at a loop header (round 2, k_minify_onchip: the next ticket, written to LDS at the end of an iteration,
was read by the other waves before the write had been performed -- one wrong run in two at 1 GiB).  This script takes the
device code out of a shared library (the .hip_fatbin section -> clang offload bundles -> gfx950 code objects),
disassembles it, rebuilds the basic blocks of every kernel and reports every s_barrier that some path reaches from an
LDS store (ds_write / ds_or / ...) without an `s_waitcnt lgkmcnt(0)` in between.  Used by tests/test_build.py; run by hand:
    python scripts/check_barriers.py simdjson_amd/lib/libsjgpu.so
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, workdir):
    fat = os.path.join(workdir, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(workdir, "ignored.so")], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for k, at in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(blob)
        piece = os.path.join(workdir, f"bundle{k}.bin")
        open(piece, "wb").write(blob[at:end])
        listing = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={piece}"], capture_output=True, text=True, check=True).stdout
        for target in listing.split():
            if "gfx950" in target:
                co = os.path.join(workdir, f"dev{k}.co")
                subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={piece}", f"--targets={target}", f"--output={co}"], check=True)
                out.append(co)
    return out


LDS_STORE = re.compile(r"ds_(write|or|add|sub|wrxchg|cmpst|max|min|and|xor|inc|dec)")


def unguarded_barriers(code_object):
    """[(kernel, address)] of every s_barrier some path reaches from an LDS store without an lgkmcnt(0) wait in between"""
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", code_object], capture_output=True, text=True, check=True).stdout
    findings = []
    func, insns = None, []  # insns: (address, mnemonic, operands, branch target or None)

    def flush():
        if not func or not insns or not any(LDS_STORE.match(op) for _, op, _, _ in insns):
            return
        base = insns[0][0]
        index_of = {a: k for k, (a, _, _, _) in enumerate(insns)}
        starts = {base}
        for k, (addr, op, args, target) in enumerate(insns):
            if op.startswith("s_cbranch") or op == "s_branch":
                if k + 1 < len(insns):
                    starts.add(insns[k + 1][0])
                if target is not None and base + target in index_of:
                    starts.add(base + target)
        order = sorted(starts)
        block_of = {}
        blocks = []  # (first index, last index)
        for b, a in enumerate(order):
            first = index_of[a]
            last = (index_of[order[b + 1]] if b + 1 < len(order) else len(insns)) - 1
            blocks.append((first, last))
            for k in range(first, last + 1):
                block_of[k] = b
        preds = [[] for _ in blocks]
        for b, (first, last) in enumerate(blocks):
            addr, op, args, target = insns[last]
            if (op.startswith("s_cbranch") or op == "s_branch") and target is not None and base + target in index_of:
                preds[block_of[index_of[base + target]]].append(b)
            if op != "s_branch" and op != "s_endpgm" and b + 1 < len(blocks):
                preds[b + 1].append(b)

        def reaches_store(b, from_index, seen):
            first, _ = blocks[b]
            for k in range(from_index, first - 1, -1):
                _, op, args, _ = insns[k]
                if op == "s_waitcnt" and ("lgkmcnt(0)" in args or args.strip() in ("0", "0x0")):
                    return False
                if LDS_STORE.match(op):
                    return True
            for p in preds[b]:
                if p not in seen:
                    seen.add(p)
                    if reaches_store(p, blocks[p][1], seen):
                        return True
            return False

        for k, (addr, op, args, _) in enumerate(insns):
            if op == "s_barrier" and reaches_store(block_of[k], k - 1, set()):
                findings.append((func, hex(addr)))

    sys.setrecursionlimit(100000)
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-fA-F]+ <(\S+)>:", line)
        if m:
            flush()
            func, insns = m.group(1), []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<[^>]*\+0x([0-9a-fA-F]+)>)?", line)
        if m and func:
            insns.append((int(m.group(3), 16), m.group(1), m.group(2), int(m.group(4), 16) if m.group(4) else None))
    flush()
    return findings


def kernel_resources(lib):
    """{kernel symbol: {"scratch": bytes per lane, "vgpr": count, "sgpr": count, "lds": bytes}} from the code objects' metadata notes
    (.private_segment_fixed_size and friends) -- what the hardware is told to reserve, not what a compiler remark claimed."""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(lib, d):
            text = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
            cur = {}
            for line in text.split("\n"):
                m = re.match(r"^\s*-?\s*\.(\w+):\s*(.*?)\s*$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2)
                if key == "args" or (key == "agpr_count" and cur.get("_done")):
                    cur = {}  # a new kernel record begins with its .args (or, for kernels without arguments, .agpr_count)
                if key == "private_segment_fixed_size":
                    cur["scratch"] = int(val)
                elif key == "vgpr_count":
                    cur["vgpr"] = int(val)
                elif key == "sgpr_count":
                    cur["sgpr"] = int(val)
                elif key == "group_segment_fixed_size":
                    cur["lds"] = int(val)
                elif key == "name":
                    cur["name"] = val.strip("'\"")
                elif key == "wavefront_size":
                    cur["_done"] = True
                    if "name" in cur:
                        out[cur["name"]] = {k: v for k, v in cur.items() if not k.startswith("_") and k != "name"}
    return out


def l2_flushes(lib):
    """{kernel symbol: number of buffer_wbl2 / buffer_inv instructions}: what an agent-scope fence (__threadfence(), a release / acquire at agent scope)
    compiles to on gfx942 / gfx950, where an XCD's L2 is not coherent with the others' -- a write-back and an invalidation of the XCD's WHOLE L2.  Round 5: two
    such fences in the epilogue of the single-pass kernels (one per wave that leaves) cost the headline kernel 0.22 ms of 0.70.  The tile kernels talk to
    each other through agent-scope ATOMICS (performed past the L2) and waits for their acknowledgements; none of them may hold one of these."""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(lib, d):
            text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
            func = None
            for line in text.split("\n"):
                m = re.match(r"^[0-9a-fA-F]+ <([^>]+)>:", line)
                if m:
                    func = m.group(1)
                    out.setdefault(func, 0)
                elif func and re.match(r"^\s+buffer_(wbl2|inv)\b", line):
                    out[func] += 1
    return out


def stream_hints(lib):
    """{kernel symbol: {"nt_loads": n, "plain_loads": n, "nt_stores": n, "plain_stores": n}} over the 16-byte global accesses (global_load_dwordx4 /
    global_store_dwordx4) of every kernel: which of them carry the non-temporal hint.  Round 5: the split kernels and validate_utf8 read their input -- and
    read and write the masks -- with it (load_chunk_stream, sjgpu_device.h); the hint only pays on instructions that cover whole lines, and costs a factor of
    four on lane-strided ones, so WHICH loads carry it is part of the design (tests/test_host_logic.py keeps it in place)."""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(lib, d):
            text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
            func = None
            for line in text.split("\n"):
                m = re.match(r"^[0-9a-fA-F]+ <([^>]+)>:", line)
                if m:
                    func = m.group(1)
                    out.setdefault(func, {"nt_loads": 0, "plain_loads": 0, "nt_stores": 0, "plain_stores": 0})
                    continue
                m = re.match(r"^\s+global_(load|store)_dwordx4\b(.*)", line)
                if func and m:
                    body = m.group(2).split("//")[0]
                    out[func][("nt_" if re.search(r"\bnt\b", body) else "plain_") + m.group(1) + "s"] += 1
    return out


def check(lib):
    with tempfile.TemporaryDirectory() as d:
        found = []
        objects = code_objects(lib, d)
        for co in objects:
            found += unguarded_barriers(co)
        return len(objects), found


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--resources":
        for name, r in sorted(kernel_resources(sys.argv[1]).items()):
            demangled = re.sub(r"\(anonymous namespace\)::", "", subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()).split("(")[0]
            print(f"{r.get('vgpr', '?'):>4} VGPR {r.get('sgpr', '?'):>4} SGPR {r.get('lds', '?'):>6} B LDS {r.get('scratch', '?'):>5} B scratch  {demangled}")
        sys.exit(0)
    n, found = check(sys.argv[1])
    for func, addr in found:
        print(f"s_barrier reachable from an LDS store without an lgkmcnt(0) wait: {func} at {addr}")
    print(f"{n} gfx950 code objects, {len(found)} unguarded barriers")
    sys.exit(1 if found else 0)
